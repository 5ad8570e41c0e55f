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
    rows, tra = oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, 30, 100000, 7, tra=True)
    got = sig_extract.split_candidates(soa, rows, lambda rec: reads[rec]["seq"], lambda t: names[t])
    want = split_expected(doc)
    assert got == want
    assert sum(len(v) for v in want.values()) > 50 and any(len(c) == 6 for v in want.values() for c in v)
    # the translocation candidates of analysis_bnd (7 fields, "TRA"): which reads yield one decides whether a task without any
    # INS/DEL candidate still writes its reads (SE:533-535)
    want_tra = {i for i, e in enumerate(doc["cases"]["split"]["expected"]) if any(len(x) == 7 and x[5] == "TRA" for x in e)}
    got_tra = {int(seg.read_rec[r]) for r in np.flatnonzero(tra)}
    assert got_tra == want_tra and len(want_tra) > 10


@pytest.mark.gpu
def test_gpu_split_branch_matches_reference_and_oracle(doc):
    from oracle import oracle
    from volcanosv_amd import sig_extract
    from volcanosv_amd.engine import Engine
    reads, soa, seg, names = split_inputs(doc)
    with Engine(0) as eng:
        for parts, size in ((7, 100000), (-1, -1), (3, 2000)):
            rows = eng.cutesv_split(seg, seg.read_len, seg.read_rec, 30, size, parts)
            orows, otra = oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, 30, size, parts, tra=True)
            assert np.array_equal(rows, orows) and np.array_equal(eng.cutesv_split_tra(), otra)
        rows = eng.cutesv_split(seg, seg.read_len, seg.read_rec, 30, 100000, 7)
        got = sig_extract.split_candidates(soa, rows, lambda rec: reads[rec]["seq"], lambda t: names[t])
        assert got == split_expected(doc)


# ---- the script body: BAM in, INS.sigs / DEL.sigs / reads.sigs out ---------------------------------------------------------
def test_sort_sigs_equals_coreutils_sort(tmp_path):
    """sort_sigs = `sort -u | sort -k 2,2 -k 3,3n` under LC_ALL=C (the declared locale; SE:637-638)."""
    import subprocess
    from volcanosv_amd import sig_extract
    rng = np.random.default_rng(3)
    lines = []
    for _ in range(3000):
        c = "chr%s" % rng.choice(["1", "10", "2", "X", "1_random"])
        lines.append("INS\t%s\t%d\t%d\tread%d\t%s\n" % (c, rng.integers(1, 5000), rng.integers(10, 99), rng.integers(0, 50), "".join(rng.choice(list("ACGTacgt_-"), 5))))
    lines += lines[:500]
    p = tmp_path / "in.txt"
    p.write_text("".join(lines))
    env = dict(os.environ, LC_ALL="C")
    want = subprocess.run("sort -u %s | sort -k 2,2 -k 3,3n" % p, shell=True, env=env, capture_output=True, text=True, check=True).stdout
    assert "".join(sig_extract.sort_sigs(lines)) == want


def test_task_and_bed_helpers():
    from volcanosv_amd import sig_extract
    tasks = sig_extract.make_tasks([("chr1", 25_000_000), ("chrM", 16_571), ("chr2", 20_000_000)], 10_000_000)
    assert tasks == [["chr1", 0, 10_000_000], ["chr1", 10_000_000, 20_000_000], ["chr1", 20_000_000, 25_000_000], ["chrM", 0, 16_571],
                     ["chr2", 0, 10_000_000], ["chr2", 10_000_000, 20_000_000]]
    assert sig_extract.acquire_clip_pos("10S5M2D3=1X7S") == [10, 7, 11] and sig_extract.acquire_clip_pos("5H20M") == [0, 0, 20]


@pytest.mark.gpu
def test_gpu_script_body_on_a_bam(doc, tmp_path):
    """All fixture reads written to a BAM (real SEQ, SA tags), run through volcanosv_amd.sig_extract.run; the three output files
    equal what the reference's per-read candidates give after its own formatting (SE:540-560) and `sort`."""
    import subprocess
    from volcanosv_amd import bam, sig_extract
    reads = sorted(doc["cases"]["plain"]["reads"] + doc["cases"]["split"]["reads"], key=lambda r: r["pos"])
    exp = {r["name"]: e for case in ("plain", "split") for r, e in zip(doc["cases"][case]["reads"], doc["cases"][case]["expected"])}
    recs = [dict(tid=0, pos=r["pos"], qname=r["name"], mapq=r["mapq"], flag=r["flag"], cigar=[tuple(c) for c in r["cigar"]], seq=r["seq"],
                 tags=({b"SA": r["sa"]} if r["sa"] else None)) for r in reads]
    path = str(tmp_path / "reads.bam")
    bam.write_bam(path, [("chr1", 30_000_000), ("chr2", 30_000_000)], recs)
    ref = tmp_path / "ref.fa"
    ref.write_text(">chr1\nA\n")
    out = sig_extract.run(path, str(ref), str(tmp_path / "work"), log=lambda *a: None)
    ins, dele = [], []
    for r in reads:
        for c in exp[r["name"]]:
            if len(c) == 6 and c[4] == "INS":
                ins.append("%s\t%s\t%d\t%d\t%s\t%s\n" % (c[4], c[5], c[0], c[1], c[2], c[3]))
            elif len(c) == 5 and c[3] == "DEL":
                dele.append("%s\t%s\t%d\t%d\t%s\n" % (c[3], c[4], c[0], c[1], c[2]))
    env = dict(os.environ, LC_ALL="C")
    for name, lines in (("INS.sigs", ins), ("DEL.sigs", dele)):
        p = tmp_path / ("exp_" + name)
        p.write_text("".join(lines))
        want = subprocess.run("sort -u %s | sort -k 2,2 -k 3,3n" % p, shell=True, env=env, capture_output=True, text=True, check=True).stdout
        assert open(out + name).read() == want, name
        assert len(want.splitlines()) > 200
    # reads.sigs: every read with mapq >= 20 whose 10 Mb task produced a candidate (SE:524-535), in file order per task
    tasks_with_cand = {r["pos"] // 10_000_000 for r in reads if any((len(c) == 6 and c[4] in ("INS",)) or len(c) in (5, 7) for c in exp[r["name"]])}
    want = []
    for t in sorted(tasks_with_cand, key=lambda t: ("_chr1_%d_%d" % (t * 10_000_000, (t + 1) * 10_000_000)).encode()):
        for r in reads:
            if r["pos"] // 10_000_000 == t and r["mapq"] >= 20:
                end = r["pos"] + sum(l for op, l in r["cigar"] if op in (0, 2, 3, 7, 8))
                want.append("chr1\t%d\t%d\t%d\t%s\n" % (r["pos"], end, 1 if r["flag"] in (0, 16) else 0, r["name"]))
    assert open(out + "reads.sigs").read() == "".join(want)
