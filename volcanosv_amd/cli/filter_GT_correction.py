#!/usr/bin/env python3
"""Drop-in for Large_INDEL/filter_GT_correction.py (same flags, FGT:3-15): every step runs in this process on the GPU."""
import argparse
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import gt_correction  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('--vcffile', '-vcf')
parser.add_argument('--bamfile', '-bam', help="only needed when presig is not provided")
parser.add_argument('--reference', '-ref', help="only needed when presig is not provided")
parser.add_argument('--pre_cutesig', '-presig', help="pre-extracted cutesv signature directory;optional; if not provided, will generate a new one")
parser.add_argument('--dtype', '-dtype', choices=['Hifi', 'CLR', 'ONT'])
parser.add_argument('--chr_num', '-chr', type=int, choices=list(range(1, 23)), default=None)
parser.add_argument('--n_thread', '-t', type=int, default=22)
a = parser.parse_args()
gt_correction.filter_gt_correction(a.vcffile, a.bamfile, a.reference, a.pre_cutesig, a.dtype, a.chr_num)
