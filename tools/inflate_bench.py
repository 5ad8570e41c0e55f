#!/usr/bin/env python3
"""GPU BGZF inflate (vsv_bgzf_inflate, one lane per member) against zlib on the host: BAM-like members (structured records
with 4-bit sequence and quality-like bytes), compressed with zlib level 6 like samtools' default."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from volcanosv_amd.engine import Engine  # noqa: E402

n_members = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(1)
distinct = []
for k in range(64):                                   # 64 distinct 64 KiB blocks, reused round-robin (compression takes the time)
    recs = []
    while sum(len(r) for r in recs) < 65280:
        l_seq = int(rng.integers(8000, 20000))
        hdr = rng.integers(0, 256, 36, dtype=np.uint8).tobytes() + b"m64011_190830_220126/%d/ccs\0" % rng.integers(1, 10**7)
        cigar = rng.integers(0, 1 << 20, int(rng.integers(5, 80)), dtype=np.uint32).tobytes()
        seq = rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes()            # 2 bases per byte: ~incompressible
        qual = rng.choice(np.array([30, 40, 50, 60, 70, 80, 93], dtype=np.uint8), l_seq, p=[.02, .03, .05, .1, .2, .3, .3]).tobytes()
        recs.append(hdr + cigar + seq + qual)
    distinct.append(b"".join(recs)[:65280])
comp = [zlib.compressobj(6, zlib.DEFLATED, -15).compress(d) for d in distinct]
comp = []
for d in distinct:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp.append(c.compress(d) + c.flush())
payloads = [comp[i % 64] for i in range(n_members)]
sizes = [len(distinct[i % 64]) for i in range(n_members)]
raw_mb = sum(sizes) / 1e6
print("members %d, %.1f MB inflated, %.1f MB compressed (ratio %.2f)" % (n_members, raw_mb, sum(len(p) for p in payloads) / 1e6, sum(len(p) for p in payloads) / sum(sizes)))
t0 = time.perf_counter()
for p in payloads[:512]:
    zlib.decompressobj(-15).decompress(p)
dt = time.perf_counter() - t0
print("zlib, 1 host core: %.0f MB/s inflated" % (sum(sizes[:512]) / 1e6 / dt))
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
out = eng.bgzf_inflate(payloads, sizes)
assert all(out[i] == distinct[i % 64] for i in range(0, n_members, 97))
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    eng.bgzf_inflate(payloads, sizes)
    best = min(best, time.perf_counter() - t0)
print("GPU, C-ABI call incl. H2D + D2H + python packing: %.0f MB/s inflated" % (raw_mb / best))
