#!/usr/bin/env python3
"""Drop-in for Large_INDEL/FP_filter_v1.py (same flags, FP_filter_v1.py:3-21); the support join runs on the GPU."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import fp_filter  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--input_path', '-i')
parser.add_argument('--signature_dir', '-sigd')
parser.add_argument('--output_path', '-o')
parser.add_argument('--max_comp_svlen', '-max_comp_svlen', type=int, default=250)
parser.add_argument('--max_dist', '-max_dist', type=int, default=1000)
parser.add_argument('--max_shift', '-max_shift', type=int, default=500)
parser.add_argument('--min_size_sim', '-min_size_sim', type=float, default=0.5)
parser.add_argument('--delete_temp_file', '-d', action='store_true')
a = parser.parse_args()
fp_filter.run(a.input_path, a.signature_dir, a.output_path, a.max_comp_svlen, a.max_dist, a.max_shift, a.min_size_sim)
