#!/usr/bin/env python3
"""Drop-in for Large_INDEL/extract_reads_signature.py (same flags, same output file), HIP path."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import reads_signature  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--chr_number', '-chr', type=int)
parser.add_argument('--input_path', '-i')
parser.add_argument('--output_dir', '-o')
parser.add_argument('--max_shift', type=int, default=100)
parser.add_argument('--max_shift_ratio', type=float, default=0.1)
parser.add_argument('--min_reads_support', type=int, default=1)
parser.add_argument('--min_siglen', type=int, default=30)
args = parser.parse_args()
reads_signature.run(args.input_path, args.output_dir, args.chr_number)
