#!/usr/bin/env python3
"""File to call table, end to end: a coordinate-sorted BAM of HiFi-like reads -> (device or host ingest) -> the whole hot path ->
call table on the host. Usage: e2e_bench.py [records]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from volcanosv_amd import bam, synth  # noqa: E402
from volcanosv_amd.abi import DTYPE_HIFI  # noqa: E402
from volcanosv_amd.engine import Engine, default_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t, nq, _ = synth.generate(n, "hifi", seed=20250330)
soa = synth.to_soa(t, nq)
recs = []
for i in range(soa.n_records):
    a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
    recs.append(dict(tid=0, pos=int(soa.pos[i]), qname="PS%d_hp%d_r" % (int(soa.qid[i]), 1 + (int(soa.flag[i]) >> 3 & 1)), mapq=int(soa.mapq[i]),
                     flag=16 if soa.flag[i] & 1 else 0, cigar=[(int(w) & 15, int(w) >> 4) for w in soa.cigar[a:b]]))
path = os.path.join(tempfile.mkdtemp(), "reads.bam")
bam.write_bam(path, [("chr10", synth.CHR10_LEN)], recs)
print("%d records, %d CIGAR ops, %.1f MB BAM" % (soa.n_records, soa.n_ops, os.path.getsize(path) / 1e6))
p = default_params(DTYPE_HIFI)
eng = Engine(0)
for mode in ("device", "host"):
    best, calls = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        with bam.BamFile(path) as bf:
            if mode == "device":
                view = bf.fetch_device(eng, "chr10")
                view.max_pos = synth.CHR10_LEN + 200000
                eng.run(view, p)
            else:
                s = bf.fetch_soa("chr10")
                s.max_pos = synth.CHR10_LEN + 200000
                eng.run(s, p)
        calls = eng.table("calls")
        best = min(best, time.perf_counter() - t0)
    print("%s ingest: file -> %d calls in %.3f s = %.2f M records/s end to end" % (mode, len(calls), best, soa.n_records / best / 1e6))
eng.close()
