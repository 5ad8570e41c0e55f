#!/usr/bin/env python3
"""Row counts of every table of one step of a bench workload (diagnostic): tools/table_counts.py <shape> <records> <dtype>"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
from volcanosv_amd.abi import DTYPE_BY_NAME
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
shape = sys.argv[1] if len(sys.argv) > 1 else "ont"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
dtype = sys.argv[3] if len(sys.argv) > 3 else "ONT"
cfg = {"hifi": 2, "ont": 3, "contig": 6}[shape]
t, nq, nt = synth.generate(n, shape, seed=20250328 + cfg, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
p = default_params(DTYPE_BY_NAME[dtype])
with Engine(0, max_sigs=1 << 24) as e:
    recs = DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0)
    e.run(recs, p)
    print("records", recs.n_records, "ops", recs.n_ops, "ops/record %.1f" % (recs.n_ops / recs.n_records))
    for name in ("raw", "cigar", "split", "cluster1", "merged", "calls"):
        tb = e.table(name)
        line = "%-9s %9d rows" % (name, len(tb))
        if name == "split":
            line += "  live %d" % int(((tb["meta"] & 8) == 0).sum()) if "meta" in tb.dtype.names else ""
        print(line)
