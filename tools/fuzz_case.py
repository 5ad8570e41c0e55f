#!/usr/bin/env python3
"""Replays case K of tests/test_gpu_parity.py::test_random_small_inputs_statuses_and_tables and prints, table by table, where the
HIP path and the oracle part ways. Usage: fuzz_case.py {basic|defects} K"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import fuzz_cases  # noqa: E402
from oracle import oracle  # noqa: E402
from volcanosv_amd.abi import VsvError  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402

kind, K = sys.argv[1], int(sys.argv[2])
for case, soa, dtype, p in fuzz_cases(kind, upto=K):
    pass
print("case", K, "records", soa.n_records, "ops", soa.n_ops, "dtype", dtype, {f: getattr(p, f) for f, _ in p._fields_ if f != "reserved"})
st, want = oracle.run(soa, params=p, dtype=dtype)
print("oracle status", st, {k: len(v) for k, v in want.items()} if st == 0 else "")
import torch  # noqa: E402
if torch.cuda.is_available():
    with Engine(0) as eng:
        try:
            eng.run(soa, p)
        except VsvError as e:
            print("hip status", e.status, e)
            sys.exit(0)
        got = eng.tables(dtype)
    print("hip   ", {k: len(v) for k, v in got.items()})
    for k in got:
        a, b = got[k], want[k]
        m = min(len(a), len(b))
        bad = np.flatnonzero(a[:m] != b[:m])
        if len(a) != len(b) or len(bad):
            i = int(bad[0]) if len(bad) else m
            print("table", k, "first difference at row", i)
            for j in range(max(0, i - 2), min(max(len(a), len(b)), i + 4)):
                print("   ", j, "hip", a[j] if j < len(a) else None, "| oracle", b[j] if j < len(b) else None)
