#!/usr/bin/env python3
"""Drop-in for Large_INDEL/calculate_signature_support.py (same flags, CS:12-22); the coverage sums run on the GPU."""
import argparse
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import signature_support  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information',
                        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('--vcffile', '-v')
parser.add_argument('--cutesv_dir', '-ct')
parser.add_argument('--n_thread', '-t', type=int, default=22, help="number of threads")
parser.add_argument('--flanking', '-f', type=int, default=1000, help='flanking region around breakpoint')
parser.add_argument('--min_size', '-s', type=int, default=30, help="min signature size")
parser.add_argument('--chr_num', '-chr', type=int, choices=list(range(1, 23)), default=None,
                    help="chrmosome number;Optional; if not provided, will assume input_dir contain chr1-chr22 results")
a = parser.parse_args()
signature_support.run(a.vcffile, a.cutesv_dir, a.flanking, a.min_size, a.chr_num)
