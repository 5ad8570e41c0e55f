#!/usr/bin/env python3
"""Drop-in for Large_INDEL/extract_contig_signature_Hifi.py (same flags, same output file), HIP path."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import contig_signature  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--chr_number', '-chr', type=int)
parser.add_argument('--bam_path', '-bam')
parser.add_argument('--contig_path', '-contig')
parser.add_argument('--header_path', '-header')
parser.add_argument('--ref_path', '-ref')
parser.add_argument('--output_dir', '-o')
# parsed but, as in the reference (hard-coded call sites), not honoured
parser.add_argument('--max_shift', type=int, default=100)
parser.add_argument('--max_shift_ratio', type=float, default=0.1)
parser.add_argument('--min_reads_support', type=int, default=1)
parser.add_argument('--min_siglen', type=int, default=30)
parser.add_argument('--min_cigar_mapq', type=int, default=50)
parser.add_argument('--min_split_mapq', type=int, default=50)
args = parser.parse_args()
contig_signature.run("Hifi", args.bam_path, args.contig_path, args.ref_path, args.output_dir, args.chr_number, args.header_path)
