#!/usr/bin/env python3
"""Step time of a rank-7-of-8 shard (tid 7) with and without the vsv_records.tid_lo hint (sort keys carry tid - tid_lo)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
from volcanosv_amd.abi import DTYPE_HIFI
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
t, nq, _ = synth.generate(10_000_000, "hifi", seed=20250330, tid=7, device="cuda")
p = default_params(DTYPE_HIFI)
with Engine(0) as eng:
    for label, dr in (("n_tids=8, no hint", DeviceRecords(t, nq, 8, max_pos=synth.CHR10_LEN + 200000)),
                      ("n_tids=8, tid_lo=7", DeviceRecords(t, nq, 8, max_pos=synth.CHR10_LEN + 200000, tid_lo=7))):
        for _ in range(3):
            eng.run(dr, p)
        n_calls = len(eng.table("calls"))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            eng.run(dr, p)
        torch.cuda.synchronize()
        print("%-22s %.3f ms/step (single engine), %d calls" % (label, (time.perf_counter() - t0) / 20 * 1e3, n_calls))
