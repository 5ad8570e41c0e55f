#!/usr/bin/env python3
"""Drop-in for Large_INDEL/correct_gt_del_real_data.py (same flags; the -eval branch that scores against a truth set is not
part of the pipeline and is not carried over)."""
import argparse
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import gt_correction  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('--input_path', '-i')
parser.add_argument('--eval_dir', '-eval')
parser.add_argument('--output_path', '-o')
parser.add_argument('--bamfile', '-bam')
parser.add_argument('--sigfile', '--sig_file', '-sig', dest='sigfile')
parser.add_argument('--n_thread', '-t', type=int, default=22)
parser.add_argument('--dtype', '-d', choices=['ONT', 'Hifi', 'CLR'])
parser.add_argument('--vtype', '-v', choices=['INS', 'DEL'])
a = parser.parse_args()
with Engine(0) as eng:
    gt_correction.run_del(a.input_path, a.output_path, a.bamfile, a.sigfile, a.dtype, eng)
