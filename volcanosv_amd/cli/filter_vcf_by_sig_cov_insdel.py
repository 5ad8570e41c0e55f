#!/usr/bin/env python3
"""Drop-in for Large_INDEL/filter_vcf_by_sig_cov_insdel.py (same flags, FV:4-8)."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import sig_cov_filter  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--input_path', '-i')
parser.add_argument('--dtype', '-d', help='hifi/ont/clr')
parser.add_argument('--asm', '-a', help='other/volcano')
parser.add_argument('--vtype', '-v', help='apply filter to which variants, INS/DEL/INSDEL, default = INSDEL', choices=['INS', 'DEL', 'INSDEL'],
                    default='INSDEL')
a = parser.parse_args()
sig_cov_filter.run(a.input_path, a.dtype, a.asm, a.vtype)
