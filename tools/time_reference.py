#!/usr/bin/env python3
"""Times the REFERENCE's own Python functions (AST-extracted, build container only) on a config-2-shaped stream, so the
CPU-oracle numbers bench.py prints can be related to the real Python path. Writes nothing into the repo."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402
from volcanosv_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t, nq, _ = synth.generate(n, "hifi", seed=20250330)
soa = synth.to_soa(t, nq)
ns = mg.load_functions(mg.SCRIPTS["Hifi"], False, [])
reads = []
t0 = time.perf_counter()
cig = soa.cigar
off = soa.cigar_off
for i in range(soa.n_records):
    c = [(int(w) & 15, int(w) >> 4) for w in cig[int(off[i]):int(off[i + 1])]]
    reads.append(mg.FakeRead("chr1", int(soa.pos[i]), "PS%d_hp%d_r" % (int(soa.qid[i]), 1 + (int(soa.flag[i]) >> 3 & 1)), int(soa.mapq[i]), bool(soa.flag[i] & 1), c))
t1 = time.perf_counter()
dels, inss = [], []
for r in reads:
    if r.mapq >= 50:
        d, i_, _, _ = ns["extract_sig_from_cigar"](r, 30)
        dels += d
        inss += i_
t2 = time.perf_counter()
ds, is_ = ns["sort_sig"](dels), ns["sort_sig"](inss)
t3 = time.perf_counter()
cd, ci = ns["cluster_del"](ds), ns["cluster_ins"](is_)
t4 = time.perf_counter()
print("records %d ops %d | build python records %.1fs (not counted) | extract_sig_from_cigar %.2fs = %.0f rec/s, %.2f M ops/s | sort %.3fs | cluster %d+%d sigs %.2fs"
      % (soa.n_records, soa.n_ops, t1 - t0, t2 - t1, soa.n_records / (t2 - t1), soa.n_ops / (t2 - t1) / 1e6, t3 - t2, len(ds), len(is_), t4 - t3))
