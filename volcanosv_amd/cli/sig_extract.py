#!/usr/bin/env python3
"""Drop-in for Large_INDEL/sig_extract.py (same positional arguments and options, SE:660-782). The clustering / genotyping
options of the cuteSV command line are accepted and ignored, as the reference script itself ignores them."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import sig_extract  # noqa: E402

parser = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter)
parser.add_argument("input", metavar="[BAM]", type=str, help="Sorted .bam file from NGMLR or Minimap2.")
parser.add_argument("reference", type=str, help="The reference genome in fasta format.")
parser.add_argument('work_dir', type=str, help="Work-directory for distributed jobs")
parser.add_argument('-t', '--threads', default=16, type=int)
parser.add_argument('-b', '--batches', default=10000000, type=int)
parser.add_argument('-S', '--sample', default="NULL", type=str)
parser.add_argument('--report_readid', action="store_true")
parser.add_argument('-p', '--max_split_parts', default=7, type=int)
parser.add_argument('-q', '--min_mapq', default=20, type=int)
parser.add_argument('-r', '--min_read_len', default=500, type=int)
parser.add_argument('-md', '--merge_del_threshold', default=0, type=int)
parser.add_argument('-mi', '--merge_ins_threshold', default=100, type=int)
parser.add_argument('-include_bed', default=None, type=str)
parser.add_argument('-s', '--min_support', default=10, type=int)
parser.add_argument('-l', '--min_size', default=30, type=int)
parser.add_argument('-L', '--max_size', default=100000, type=int)
parser.add_argument('-sl', '--min_siglength', default=10, type=int)
parser.add_argument('--genotype', action="store_true")
parser.add_argument('--gt_round', default=500, type=int)
parser.add_argument('--max_cluster_bias_INS', default=100, type=int)
parser.add_argument('--diff_ratio_merging_INS', default=0.3, type=float)
parser.add_argument('--max_cluster_bias_DEL', default=200, type=int)
parser.add_argument('--diff_ratio_merging_DEL', default=0.5, type=float)
parser.add_argument('--remain_reads_ratio', default=1.0, type=float)
a = parser.parse_args()
sig_extract.run(a.input, a.reference, a.work_dir, a.batches, a.max_split_parts, a.min_mapq, a.min_read_len, a.merge_del_threshold,
                a.merge_ins_threshold, a.include_bed, a.min_size, a.max_size, a.min_siglength)
