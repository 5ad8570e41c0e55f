#!/usr/bin/env python3
"""Typed pairing stretches of the row-2c merged table (diagnostic for slim_path.hip sl_pair_fused)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
t, nq, nt = synth.generate(n, "contig", seed=20250328 + 6, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
p = default_params(0)
with Engine(0, max_sigs=1 << 24) as e:
    e.run(DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0), p)
    m = e.table("merged")
hap2 = (m["meta"] & 4) != 0
for ty in (0, 1):
    sel = ((m["meta"] & 1) == ty)
    p1 = m["pos"][sel & ~hap2].astype(np.int64)
    p2 = m["pos"][sel & hap2].astype(np.int64)
    a, b = p1[:-1], p1[1:]
    lo = np.searchsorted(p2, b - 200, side="left")
    hi = np.searchsorted(p2, a + 200, side="right")
    cut = hi <= lo
    heads = np.concatenate(([True], cut))
    idx = np.flatnonzero(heads)
    lens = np.diff(np.concatenate((idx, [len(p1)])))
    print("type", ty, "hp1 rows", len(p1), "hp2 rows", len(p2), "stretches", len(lens), "mean %.2f" % lens.mean(), "max", lens.max(),
          "p99", np.percentile(lens, 99), "p99.9", np.percentile(lens, 99.9), ">32:", (lens > 32).sum(), ">64:", (lens > 64).sum())
    k = int(np.argmax(lens))
    s = idx[k]
    print("  longest at pos", p1[s], "..", p1[s + lens[k] - 1], "gaps", np.diff(p1[s:s + lens[k]])[:20])
    # rows per 512-slot window statistics are in the kernel; here: the candidate count per hp1 row
    c = np.searchsorted(p2, p1 + 200, side="right") - np.searchsorted(p2, p1 - 200, side="left")
    print("  candidates per row: mean %.1f max %d" % (c.mean(), c.max()))
