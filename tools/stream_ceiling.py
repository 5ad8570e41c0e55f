#!/usr/bin/env python3
"""Streaming ceiling of the part with the library's own read-stream / copy kernels (vsv_stream_ceiling) over a 1.3 GB buffer — the
size of config 2's CIGAR array."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402
buf = torch.randint(0, 1 << 30, (327_000_000,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
with Engine(0) as eng:
    r, c = eng.stream_ceiling(buf, reps=10)
print("read stream %.0f GB/s, copy (read + written) %.0f GB/s" % (r, c))
