#!/usr/bin/env python3
"""Drop-in for Complex_SV/filter_tra.py (-vcf -o -bam): merges breakends within --max_dist into <o>/TRA_final.vcf."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import bnd  # noqa: E402

ap = ArgumentParser()
ap.add_argument("--vcffile", "-vcf")
ap.add_argument("--output_dir", "-o")
ap.add_argument("--bamfile", "-bam")
ap.add_argument("--n_thread", "-t", type=int, default=30)
ap.add_argument("--flanking", "-f", type=int, default=1000)
ap.add_argument("--min_support", "-ms", type=int, default=10)
ap.add_argument("--min_mapq", "-mq", type=int, default=10)
ap.add_argument("--max_dist", "-d", type=int, default=100)
a = ap.parse_args()
os.makedirs(a.output_dir, exist_ok=True)
hdr, body = bnd.merge_bnd_lines(open(a.vcffile).readlines(), a.max_dist)
with open(os.path.join(a.output_dir, "TRA_final.vcf"), "w") as f:
    f.writelines(hdr)
    f.writelines(body)
