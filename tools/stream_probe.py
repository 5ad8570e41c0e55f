#!/usr/bin/env python3
"""Throughput of S engines (one vsv_handle + HIP stream each) working on independent batches round-robin: the
latency-bound signature stages of one batch overlap the bandwidth-bound cigar_scan_emit of the next."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from volcanosv_amd import synth  # noqa: E402
from volcanosv_amd.abi import DTYPE_BY_NAME  # noqa: E402
from volcanosv_amd.engine import DeviceRecords, Engine, default_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t, nq, nt = synth.generate(n, "hifi", seed=5, device="cuda")
dr = DeviceRecords(t, nq, nt)
p = default_params(DTYPE_BY_NAME[sys.argv[3] if len(sys.argv) > 3 else "Hifi"])
for S in (1, 2, 3, 4, 6):
    streams = [torch.cuda.Stream() for _ in range(S)]
    engs = [Engine(0, stream=s.cuda_stream) for s in streams]
    for e in engs:
        e.run(dr, p)
        e.run(dr, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        e = engs[i % S]
        if i >= S:
            e.finish()
        e.run_async(dr, p)
    for e in engs[: min(S, K)]:
        e.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scan = sum(e.scan_ms() for e in engs) / S
    print("streams %d: %.3f ms/step  %.2f G records/s  (last cigar_scan_emit %.3f ms)" % (S, dt / K * 1e3, n * K / dt / 1e9, scan))
    for e in engs:
        e.close()
