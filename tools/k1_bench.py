#!/usr/bin/env python3
"""Times cigar_scan_emit alone (HIP events inside the library) on the three synthetic shapes and prints GB/s."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from volcanosv_amd import synth  # noqa: E402
from volcanosv_amd.abi import DTYPE_BY_NAME  # noqa: E402
from volcanosv_amd.engine import DeviceRecords, Engine, default_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="hifi:10000000,ont:2000000,contig:20000")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--dtype", default="Hifi")
args = ap.parse_args()
eng = Engine(0, max_sigs=1 << 24)
p = default_params(DTYPE_BY_NAME[args.dtype])
for spec in args.shapes.split(","):
    shape, n = spec.split(":")
    t, nq, nt = synth.generate(int(n), shape, seed=5, device="cuda")
    dr = DeviceRecords(t, nq, nt)
    ms = []
    for _ in range(args.reps + 2):
        eng.cigar_scan(dr, p)
        ms.append(eng.scan_ms())
    ms = sorted(ms[2:])
    n_raw = len(eng.table("raw"))
    b = 24 * dr.n_records + 4 * dr.n_ops + 32 * n_raw
    med = ms[len(ms) // 2]
    print(json.dumps({"shape": shape, "records": dr.n_records, "ops": dr.n_ops, "raw_sigs": n_raw, "ms_med": round(med, 4),
                      "ms_min": round(ms[0], 4), "GBs_med": round(b / med / 1e6, 1), "frac_8TBs": round(b / med / 1e6 / 8000, 4)}))
    del t, dr
    torch.cuda.empty_cache()
eng.close()
