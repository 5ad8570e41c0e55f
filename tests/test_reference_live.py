"""Oracle vs the reference's OWN functions on fresh random inputs — runs only where the reference is mounted (the build container;
skipped anywhere else). The eight committed fixtures of tests/golden pin the oracle on fixed inputs; this test re-runs the same
machinery (tests/golden/make_golden.py: AST-extracted FunctionDefs behind a fake pysam, stable-argsort shim for ties) on dozens
of new seeds and record shapes, so the pin does not rest on those eight inputs alone. Nothing of the reference is stored."""
import os
import sys

import numpy as np
import pytest

from helpers import compare_contig_tables, rows, sel
from oracle import oracle
from volcanosv_amd.abi import DTYPE_BY_NAME, DTYPE_READS
from volcanosv_amd.soa import RecordSoA

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")), reason="reference not mounted")


def _mg():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    return make_golden


def _soa(recs, chroms):
    tid = {c: i for i, c in enumerate(chroms)}
    return RecordSoA.from_tuples([(tid[r[0]], r[1], r[2], r[3], r[4], r[5]) for r in recs], tid_names=chroms)


def _mutate(recs, rng):
    """mapq around the threshold, haplotype tags dropped / doubled, strands flipped (names stay consistent per contig)."""
    rename = {}
    out = []
    for r in recs:
        chrom, pos, name, mapq, rev, cig = r
        if name not in rename:
            u = rng.random()
            rename[name] = name.replace("_hp1_", "_h_").replace("_hp2_", "_h_") if u < 0.05 else name + "_hp2x" if u < 0.08 and "hp1" in name else name
        if rng.random() < 0.15:
            mapq = int(rng.integers(40, 61))
        if rng.random() < 0.1:
            rev = not rev
        out.append((chrom, pos, rename[name], mapq, bool(rev), cig))
    return out


@pytest.mark.parametrize("style", ["Hifi", "ONT", "CLR"])
def test_contig_path_on_fresh_random_inputs(style):
    mg = _mg()
    rng = np.random.default_rng({"Hifi": 1, "ONT": 2, "CLR": 3}[style])
    n_calls = 0
    for seed in range(3000, 3005):
        recs = mg.make_contig_records(seed, n_sites=int(rng.integers(5, 60)), n_chrom=int(rng.integers(1, 3)), contigs_per_hap=int(rng.integers(1, 5)),
                                      split_pairs=int(rng.integers(0, 20)), style=style, tie_rich=bool(rng.random() < 0.5))
        recs = _mutate(recs, rng)
        exp = mg.run_contig(style, recs, stable=True)
        soa = _soa(recs, exp["chroms"])
        st, tabs = oracle.run(soa, dtype=DTYPE_BY_NAME[style])
        assert st == 0, (style, seed)
        compare_contig_tables({"expected": exp}, soa, tabs)
        n_calls += len(tabs["calls"])
    assert n_calls > 100


def test_reads_path_on_fresh_random_inputs():
    mg = _mg()
    rng = np.random.default_rng(4)
    n_rows = 0
    for seed in range(4000, 4006):
        recs = mg.make_read_records(seed, n=int(rng.integers(50, 500)), n_chrom=int(rng.integers(1, 3)), tie_rich=bool(rng.random() < 0.5))
        exp = mg.run_reads(recs, stable=True)
        soa = _soa(recs, exp["chroms"])
        st, tabs = oracle.run(soa, dtype=DTYPE_READS)
        assert st == 0
        for t, chrom in enumerate(exp["chroms"]):
            got = rows(soa, tabs["reads"], dtype=DTYPE_READS, where=sel(t))
            assert got == exp["per_chrom"][chrom]["merged"], (seed, chrom)
            n_rows += len(got)
    assert n_rows > 100
