#!/usr/bin/env python3
"""BAM/BGZF -> record SoA ingest rate (the step left of the hot path): writes a synthetic HiFi-like BAM with
volcanosv_amd.bam.write_bam, then times vsv_bam_load with 1..N host inflate threads and, when a GPU is present (and the
second argument is a sequence length), with the GPU inflate. Usage: ingest_bench.py [records] [seq_len] [--device-only]; VSV_BAM_TIMING=1 prints the device reader's phases"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volcanosv_amd import bam, synth  # noqa: E402

device_only = "--device-only" in sys.argv
argv = [a for a in sys.argv if not a.startswith("--")]
n = int(argv[1]) if len(argv) > 1 else 300000
seq_len = int(argv[2]) if len(argv) > 2 else 0
import numpy as np  # noqa: E402
rng = np.random.default_rng(4)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)
t, nq, _ = synth.generate(n, "hifi", seed=9)
soa = synth.to_soa(t, nq)
recs = []
for i in range(soa.n_records):
    a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
    recs.append(dict(tid=0, pos=int(soa.pos[i]), qname="PS%d_hp%d_r" % (int(soa.qid[i]), 1 + (int(soa.flag[i]) >> 3 & 1)), mapq=int(soa.mapq[i]),
                     flag=16 if soa.flag[i] & 1 else 0, cigar=[(int(w) & 15, int(w) >> 4) for w in soa.cigar[a:b]],
                     seq=(bases[rng.integers(0, 4, seq_len)].tobytes().decode() if seq_len else None)))
path = os.path.join(tempfile.mkdtemp(), "reads.bam")
bam.write_bam(path, [("chr10", synth.CHR10_LEN)], recs)
size = os.path.getsize(path)
for th in (() if device_only else (1, 2, 4, 8, 16)):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        with bam.BamFile(path, threads=th) as bf:
            s2 = bf.fetch_soa("chr10")
        best = min(best, time.perf_counter() - t0)
    assert s2.n_records == soa.n_records and s2.n_ops == soa.n_ops
    print("threads %2d: %.3f s  %.2f M records/s  %.0f MB/s compressed (%d records, %.1f MB BAM, SEQ length %d)" % (th, best, soa.n_records / best / 1e6, size / best / 1e6, soa.n_records, size / 1e6, seq_len))

try:
    import torch
    if torch.cuda.is_available():
        from volcanosv_amd.engine import Engine
        eng = Engine(0)
        best = 1e9
        for _ in range(0 if device_only else 3):
            t0 = time.perf_counter()
            with bam.BamFile(path) as bf:
                bf.use_gpu_inflate(eng)
                s3 = bf.fetch_soa("chr10")
            best = min(best, time.perf_counter() - t0)
        if not device_only:
            assert s3.n_records == soa.n_records and s3.n_ops == soa.n_ops
            print("GPU inflate: %.3f s  %.2f M records/s  %.0f MB/s compressed" % (best, soa.n_records / best / 1e6, size / best / 1e6))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            bf = bam.BamFile(path)
            t1 = time.perf_counter()
            view = bf.fetch_device(eng, "chr10")
            t2 = time.perf_counter()
            bf.close()
            t3 = time.perf_counter()
            if os.environ.get("VSV_BAM_TIMING"):
                print("  open %.1f ms, fetch_device %.1f ms, close %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
                import ctypes as C
                from volcanosv_amd.abi import Records
                bf2 = bam.BamFile(path)
                r = Records()
                ta = time.perf_counter()
                bf2.lib.vsv_bam_load_device(bf2.h, eng.h, 0, C.byref(r))
                tb = time.perf_counter()
                ln = C.c_int64()
                pp = bf2.lib.vsv_bam_qnames(bf2.h, C.byref(ln))
                blob = C.string_at(pp, ln.value)
                tc = time.perf_counter()
                ll = bam.LazyLines(blob, int(r.n_qids))
                td = time.perf_counter()
                print("  vsv_bam_load_device %.1f ms, names copy (%d bytes) %.1f ms, LazyLines %.1f ms" % ((tb - ta) * 1e3, ln.value, (tc - tb) * 1e3, (td - tc) * 1e3))
                bf2.close()
            best = min(best, t3 - t0)
        assert view.n_records == soa.n_records and view.n_ops == soa.n_ops
        print("GPU inflate + GPU parse (device-resident SoA, names to host): %.3f s  %.2f M records/s  %.0f MB/s compressed" % (best, soa.n_records / best / 1e6, size / best / 1e6))
        eng.close()
except ImportError:
    pass
