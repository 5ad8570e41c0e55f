#!/usr/bin/env python3
"""Times the synthetic generators on the GPU (how long bench.py spends before it can measure anything)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
for shape, n in (("hifi", 10_000_000), ("hifi", 20_000_000), ("contig", 200_000), ("ont", 50_000_000)):
    torch.cuda.synchronize(); t0 = time.time()
    t, nq, _ = synth.generate(n, shape, seed=1, device="cuda")
    torch.cuda.synchronize()
    print(shape, n, "ops", int(t["cigar"].numel()), "gen_s %.1f" % (time.time() - t0), "peak_GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del t
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
