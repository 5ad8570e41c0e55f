#!/usr/bin/env python3
"""Re-creates case K of tests/test_gpu_parity.py::test_random_small_inputs_statuses_and_tables and prints, table by table, where
the HIP path and the oracle part ways. Usage: fuzz_case.py K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from volcanosv_amd import synth
from volcanosv_amd.abi import *
from volcanosv_amd.engine import Engine, default_params
from oracle import oracle


def make_case(K):
    rng = np.random.default_rng(20250403)
    for case in range(K + 1):
        shape = ("hifi", "ont")[int(rng.integers(0, 2))]
        n = int(rng.integers(1, 401)) if shape == "hifi" else int(rng.integers(1, 81))
        t, nq, _ = synth.generate(n, shape, seed=5000 + case, chrom_len=int(rng.integers(60_000, 300_000)) + 40_000,
                                  events_per_record=float(rng.choice([0.0, 0.05, 0.5, 2.0])), site_step=int(rng.choice([200, 1000, 5000])))
        soa = synth.to_soa(t, nq)
        k = soa.n_records
        soa.mapq = np.where(rng.random(k) < 0.2, rng.integers(0, 61, k), soa.mapq).astype(np.uint8)
        fl = soa.flag.copy()
        flip = rng.random(k) < 0.15
        fl[flip] = (fl[flip] & ~np.uint8(F_HP1 | F_HP2)) | rng.choice(np.array([0, F_HP1, F_HP2, F_HP1 | F_HP2], np.uint8), int(flip.sum()))
        fl ^= (rng.random(k) < 0.1).astype(np.uint8)
        soa.flag = fl
        if rng.random() < 0.2 and soa.n_ops:
            cig = soa.cigar.copy()
            ms = np.flatnonzero((cig & 15) == 0)
            if len(ms):
                j = int(ms[rng.integers(0, len(ms))])
                cig[j] = (cig[j] & ~np.uint32(15)) | np.uint32(rng.choice([7, 8, 3]))
                soa.cigar = cig
        dtype = (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS)[int(rng.integers(0, 4))]
        p = default_params(dtype)
        p.min_svlen = int(rng.choice([30, 30, 10, 50])); p.cluster_shift = int(rng.choice([100, 100, 10, 1000]))
        p.pair_shift = int(rng.choice([200, 200, 0, 2000])); p.min_cigar_mapq = int(rng.choice([50, 50, 0, 60]))
    return soa, dtype, p, shape


if __name__ == "__main__":
    K = int(sys.argv[1])
    soa, dtype, p, shape = make_case(K)
    print("case", K, "shape", shape, "records", soa.n_records, "ops", soa.n_ops, "dtype", dtype,
          "min_svlen", p.min_svlen, "cluster_shift", p.cluster_shift, "pair_shift", p.pair_shift, "min_cigar_mapq", p.min_cigar_mapq, "min_split_mapq", p.min_split_mapq)
    st, want = oracle.run(soa, params=p, dtype=dtype)
    print("oracle status", st, {k: len(v) for k, v in want.items()})
    import torch
    if torch.cuda.is_available():
        with Engine(0) as eng:
            eng.run(soa, p)
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
