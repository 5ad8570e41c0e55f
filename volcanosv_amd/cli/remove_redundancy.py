#!/usr/bin/env python3
"""Drop-in for Large_INDEL/remove_redundancy.py (same flags, RR:6-15); the pair matching runs on the GPU."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import remove_redundancy  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--input_path', '-i')
parser.add_argument('--output_dir', '-o')
parser.add_argument('--dist_thresh', '-r', type=int, default=500)
parser.add_argument('--dist_thresh_del', '-rd', type=int, default=3000)
parser.add_argument('--overlap_thresh', '-O', type=float, default=0)
parser.add_argument('--size_sim_thresh', '-P', type=float, default=0.5)
parser.add_argument('--size_sim_thresh_del', '-Pd', type=float, default=0.1)
parser.add_argument('--seq_sim_thresh', '-p', type=float, default=0.5)
parser.add_argument('--delete_temp_file', '-d', action='store_true')
a = parser.parse_args()
remove_redundancy.run(a.input_path, a.output_dir, a.dist_thresh, a.dist_thresh_del, a.overlap_thresh, a.size_sim_thresh, a.size_sim_thresh_del,
                      a.seq_sim_thresh)
