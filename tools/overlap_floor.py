#!/usr/bin/env python3
"""What S engines in flight cost per step when the chain behind the scan shrinks (diagnostic): the same 10 M-record shard through
SVIM (the scan + its placement only), READS (scan, split stage, one sort), Hifi (everything) and Hifi without its split stage.  tools/overlap_floor.py [records]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from volcanosv_amd import synth  # noqa: E402
from volcanosv_amd.abi import DTYPE_BY_NAME  # noqa: E402
from volcanosv_amd.engine import DeviceRecords, Engine, default_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = 60
t, nq, nt = synth.generate(n, "hifi", seed=20250330, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
dr = DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0)
for name in ("SVIM", "READS", "Hifi", "Hifi-nosplit"):
    p = default_params(DTYPE_BY_NAME[name.split("-")[0]])
    if name.endswith("nosplit"):
        p.enable_split = 0          # everything but the split stage (name repeats, candidates, their two sorts, the pair rules)
    for S in (1, 2, 4):
        p.split_overlap = 1 if S >= 3 else 0
        streams = [torch.cuda.Stream() for _ in range(S)]
        engs = [Engine(0, stream=s.cuda_stream) for s in streams]
        for e in engs:
            e.run(dr, p)
            e.run(dr, p)
        best = 1e9
        for rep in range(3):
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
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        print("%-5s engines %d: %.3f ms/step (scan alone %.3f ms)" % (name, S, best, sum(e.scan_ms() for e in engs) / S))
        for e in engs:
            e.close()
