#!/usr/bin/env python3
"""Host time of a step (diagnostic): how long run_async / finish / the table copy take on the CPU with n engines in flight.
tools/host_cost.py <shape> <records> <dtype> <engines>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
from volcanosv_amd.abi import DTYPE_BY_NAME
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
shape = sys.argv[1] if len(sys.argv) > 1 else "hifi"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dtype = sys.argv[3] if len(sys.argv) > 3 else "Hifi"
ne = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cfg = {"hifi": 2, "ont": 3, "contig": 6}[shape]
t, nq, nt = synth.generate(n, shape, seed=20250328 + cfg, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
p = default_params(DTYPE_BY_NAME[dtype])
p.split_overlap = 1 if ne >= 3 else 0
dev = torch.device("cuda", 0)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(ne - 1)]
engs = [Engine(0, stream=s.cuda_stream, max_sigs=1 << 22) for s in streams]
recs = DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0)
for rep in range(3):
    k = 200
    t_run = t_fin = t_tab = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        e = engs[i % ne]
        if i >= ne:
            a = time.perf_counter(); e.finish(); b = time.perf_counter(); rows = e.table_torch("calls", dev); c = time.perf_counter()
            t_fin += b - a; t_tab += c - b
        a = time.perf_counter(); e.run_async(recs, p); t_run += time.perf_counter() - a
    for e in engs:
        e.finish()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("engines %d: wall %.1f us/step; host: run_async %.1f, finish (waits for the GPU) %.1f, table copy %.1f us/step" %
          (ne, wall / k * 1e6, t_run / k * 1e6, t_fin / k * 1e6, t_tab / k * 1e6))
