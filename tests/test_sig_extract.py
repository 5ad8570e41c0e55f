"""sig_extract.py (cuteSV-derived read-signature collector).

CPU: the oracle (CUTESV op table + generate_combine_sigs restatement) against tests/golden/sig_extract.json.gz — per-read
candidate lists returned by the reference's own parse_read (tests/golden/make_golden_sigextract.py). GPU: the same through
the C-ABI (cigar_scan_emit<3> + combine_kernel), bit-exact tables vs the oracle and text-level candidates vs the reference."""
import gzip
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sig_extract.json.gz")


@pytest.fixture(scope="module")
def doc():
    with gzip.open(GOLDEN, "rt") as f:
        return json.load(f)


def build_soa(reads, min_read_len=500):
    from volcanosv_amd import sig_extract
    from volcanosv_amd.soa import RecordSoA
    recs = [(0, r["pos"], r["name"], r["mapq"], bool(r["flag"] & 16), [tuple(c) for c in r["cigar"]],
             sig_extract.flag_bits(r["flag"], len(r["seq"]), min_read_len)) for r in reads]
    soa = RecordSoA.from_tuples(recs, tid_names=["chr1"])
    soa.n_tids = 1
    return soa


def cigar_expected(expected):
    """The CIGAR-derived candidates of every read: 6-field INS / 5-field DEL rows that are not split-derived. Split-derived
    INS/DEL have a float or different position source; in the `plain*` cases there are none (no SA tags)."""
    return [[c for c in e if (len(c) == 6 and c[4] == "INS") or (len(c) == 5 and c[3] == "DEL")] for e in expected]


def check_case(doc, case, tabs_fn):
    from volcanosv_amd import sig_extract
    reads = doc["cases"]["plain"]["reads"]
    prm = dict(min_siglength=10, merge_del_threshold=0, merge_ins_threshold=100)
    prm.update(doc["cases"][case].get("params", {}))
    soa = build_soa(reads)
    tabs = tabs_fn(soa, sig_extract.params(min_mapq=20, **prm))
    got = sig_extract.cigar_candidates(soa, tabs["raw"], tabs["cigar"], lambda rec: reads[rec]["seq"], "chr1")
    want = cigar_expected(doc["cases"][case]["expected"])
    n = 0
    for i, w in enumerate(want):
        assert got.get(i, []) == w, (case, i, reads[i]["name"])
        n += len(w)
    assert n > 200
    return tabs


def test_oracle_matches_reference_parse_read(doc):
    from oracle import oracle

    def tabs_fn(soa, p):
        st, tabs = oracle.run(soa, params=p)
        assert st == 0
        return tabs

    check_case(doc, "plain", tabs_fn)
    check_case(doc, "plain_params2", tabs_fn)


@pytest.mark.gpu
def test_gpu_matches_reference_and_oracle(doc):
    from oracle import oracle
    from volcanosv_amd.abi import DTYPE_CUTESV
    from volcanosv_amd.engine import Engine
    with Engine(0) as eng:
        for case in ("plain", "plain_params2"):
            def tabs_fn(soa, p):
                eng.run(soa, p)
                got = eng.tables(DTYPE_CUTESV)
                st, want = oracle.run(soa, params=p)
                assert st == 0
                for k in ("raw", "cigar"):
                    assert np.array_equal(got[k], want[k]), k
                return got
            check_case(doc, case, tabs_fn)


@pytest.mark.gpu
def test_gpu_synthetic_reads_vs_oracle():
    """ONT-like synthetic reads (many >= 10 bp indels per read) through the collector tables, bit-exact vs the oracle."""
    from oracle import oracle
    from volcanosv_amd import sig_extract, synth
    from volcanosv_amd.abi import DTYPE_CUTESV
    from volcanosv_amd.engine import Engine
    t, nq, nt = synth.generate(30000, "ont", seed=23, chrom_len=3_000_000)
    soa = synth.to_soa(t, nq)
    with Engine(0, max_sigs=1 << 21) as eng:
        for prm in (dict(), dict(min_siglength=5, merge_del_threshold=40, merge_ins_threshold=30)):
            p = sig_extract.params(**prm)
            eng.run(soa, p)
            got = eng.tables(DTYPE_CUTESV)
            st, want = oracle.run(soa, params=p)
            assert st == 0 and len(want["raw"]) > 3000 and len(want["cigar"]) < len(want["raw"])
            for k in ("raw", "cigar"):
                assert np.array_equal(got[k], want[k]), k


# ---- split-read branch ----------------------------------------------------------------------------------------------
def split_inputs(doc):
    from volcanosv_amd import sig_extract
    reads = doc["cases"]["split"]["reads"]
    soa = build_soa(reads)
    ids = {"chr1": 0}
    names = ["chr1"]

    def chrom_id(n):
        if n not in ids:
            ids[n] = len(names)
            names.append(n)
        return ids[n]

    sreads = sig_extract.split_reads(soa, [r["flag"] for r in reads], [len(r["seq"]) for r in reads], [r["sa"] or "" for r in reads], chrom_id, 20)
    return reads, soa, sig_extract.SplitSegments(sreads), names


def split_expected(doc):
    """INS/DEL candidates of the split fixture (its CIGARs are clip-match-clip, so all of them are split-derived); the float
    positions (a+b)/2 are compared as the %d the script writes (SE:554)."""
    exp = {}
    for i, e in enumerate(doc["cases"]["split"]["expected"]):
        c = [[int(x[0])] + x[1:] for x in e if (len(x) == 6 and x[4] == "INS") or (len(x) == 5 and x[3] == "DEL")]
        if c:
            exp[i] = c
    return exp


def test_oracle_split_branch_matches_reference(doc):
    from oracle import oracle
    from volcanosv_amd import sig_extract
    reads, soa, seg, names = split_inputs(doc)
    rows = oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, 30, 100000, 7)
    got = sig_extract.split_candidates(soa, rows, lambda rec: reads[rec]["seq"], lambda t: names[t])
    want = split_expected(doc)
    assert got == want
    assert sum(len(v) for v in want.values()) > 50 and any(len(c) == 6 for v in want.values() for c in v)


@pytest.mark.gpu
def test_gpu_split_branch_matches_reference_and_oracle(doc):
    from oracle import oracle
    from volcanosv_amd import sig_extract
    from volcanosv_amd.engine import Engine
    reads, soa, seg, names = split_inputs(doc)
    with Engine(0) as eng:
        for parts, size in ((7, 100000), (-1, -1), (3, 2000)):
            rows = eng.cutesv_split(seg, seg.read_len, seg.read_rec, 30, size, parts)
            assert np.array_equal(rows, oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, 30, size, parts))
        rows = eng.cutesv_split(seg, seg.read_len, seg.read_rec, 30, 100000, 7)
        got = sig_extract.split_candidates(soa, rows, lambda rec: reads[rec]["seq"], lambda t: names[t])
        assert got == split_expected(doc)
