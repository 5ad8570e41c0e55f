"""GPU parity tests (run on the MI355X box with -m gpu): HIP path through the C-ABI vs the CPU oracle and the
golden fixtures produced by the reference's own functions. Bit-exact on every stage table."""
import numpy as np
import pytest

from helpers import compare_contig_tables, load_fixture, rows, sel
from volcanosv_amd.abi import (DTYPE_CLR, DTYPE_HIFI, DTYPE_ONT, DTYPE_READS, DTYPE_SVIM, F_HP1, F_HP2, M_DEL, VsvError)
from volcanosv_amd.soa import RecordSoA

pytestmark = pytest.mark.gpu

CONTIG = ["contig_hifi_tiefree", "contig_hifi_stable", "contig_ont_tiefree", "contig_ont_stable",
          "contig_clr_tiefree", "contig_clr_stable"]


@pytest.fixture(scope="module")
def eng():
    from volcanosv_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def oracle_run(soa, dtype, params=None):
    from oracle import oracle
    return oracle.run(soa, params=params, dtype=dtype)


def assert_tables_equal(got, want, names):
    for k in names:
        assert got[k].shape == want[k].shape, (k, got[k].shape, want[k].shape)
        if not np.array_equal(got[k], want[k]):
            bad = np.nonzero(got[k] != want[k])[0]
            i = int(bad[0])
            raise AssertionError("table %s differs at row %d of %d: got %s want %s (%d rows differ)" %
                                 (k, i, len(want[k]), got[k][i], want[k][i], len(bad)))


def run_both(eng, soa, dtype, params=None):
    """HIP path vs oracle, once per work mapping of the CIGAR scan (vsv_params.scan_layout: record-aligned parts for reads,
    fixed-size parts with carried offsets for contigs): whatever the records look like, both are exact. The second run also takes the
    other execution mode of the stages behind the scan (vsv_params.split_overlap = OFF: split candidates on the handle's stream,
    sorts in 256-thread workgroups)."""
    from volcanosv_amd.abi import OVERLAP_OFF, SCAN_CONTIGS, SCAN_READS
    from volcanosv_amd.engine import default_params
    p = params if params is not None else default_params(dtype)
    st, want = oracle_run(soa, dtype, p)
    assert st == 0
    keep, keep_ov = p.scan_layout, p.split_overlap
    try:
        for layout in (SCAN_CONTIGS, SCAN_READS):
            p.scan_layout = layout
            if layout == SCAN_READS:
                p.split_overlap = OVERLAP_OFF
            eng.run(soa, p)
            got = eng.tables(dtype)
            try:
                assert_tables_equal(got, want, list(got.keys()))
            except AssertionError as e:
                raise AssertionError("scan_layout %d: %s" % (layout, e))
    finally:
        p.scan_layout, p.split_overlap = keep, keep_ov
    return got


@pytest.mark.parametrize("name", CONTIG)
def test_golden_contig(eng, name):
    doc, soa, dtype = load_fixture(name)
    got = run_both(eng, soa, dtype)
    compare_contig_tables(doc, soa, got)     # straight against the reference's outputs


@pytest.mark.parametrize("name", ["reads_tiefree", "reads_stable"])
def test_golden_reads(eng, name):
    doc, soa, dtype = load_fixture(name)
    got = run_both(eng, soa, dtype)
    for t, chrom in enumerate(doc["expected"]["chroms"]):
        assert rows(soa, got["reads"], dtype=DTYPE_READS, where=sel(t)) == doc["expected"]["per_chrom"][chrom]["merged"]


def test_svim_known_answers(eng):
    """svim-asm tests/test_intra.py:8-22 through the HIP kernel."""
    from volcanosv_amd.engine import default_params
    cases = [
        ([(5, 10), (4, 20), (0, 10), (7, 10), (8, 5), (0, 5), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "INS")]),
        ([(5, 10), (4, 20), (0, 30), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "DEL")]),
        ([(5, 10), (4, 20), (0, 30), (2, 40), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "DEL"), (70, 50, 50, "INS")]),
        ([(5, 10), (4, 20), (0, 30), (1, 40), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "INS"), (30, 90, 50, "DEL")]),
    ]
    soa = RecordSoA.from_tuples([(0, 0, i, 60, False, c[0]) for i, c in enumerate(cases)])
    p = default_params(DTYPE_SVIM)
    p.min_svlen = 30
    eng.run(soa, p)
    raw = eng.table("raw")
    for i, (_, want) in enumerate(cases):
        got = [(int(s["pos"]), int(s["q_start"]), int(s["svlen"]), "DEL" if s["meta"] & M_DEL else "INS") for s in raw if int(s["rec"]) == i]
        assert got == want


SYNTH = [
    ("hifi", 200000, DTYPE_HIFI, {}),
    ("hifi", 100000, DTYPE_HIFI, dict(chrom_len=2000000, events_per_record=0.2, site_step=1000)),
    ("hifi", 100000, DTYPE_ONT, dict(chrom_len=2000000, events_per_record=0.2, site_step=1000)),
    ("hifi", 100000, DTYPE_CLR, dict(chrom_len=2000000, events_per_record=0.2, site_step=1000)),
    ("hifi", 100000, DTYPE_READS, dict(chrom_len=2000000, events_per_record=0.2, site_step=1000)),
    ("ont", 20000, DTYPE_ONT, dict(chrom_len=1000000)),
    ("ont", 20000, DTYPE_READS, dict(chrom_len=1000000)),
    ("contig", 60, DTYPE_HIFI, dict(chrom_len=20000000)),
    ("contig", 60, DTYPE_CLR, dict(chrom_len=20000000)),
    ("contig", 3, DTYPE_SVIM, dict(chrom_len=20000000)),
]


@pytest.mark.parametrize("shape,n,dtype,kw", SYNTH)
def test_synthetic_vs_oracle(eng, shape, n, dtype, kw):
    from volcanosv_amd import synth
    t, nq, nt = synth.generate(n, shape, seed=7, **kw)
    soa = synth.to_soa(t, nq)
    got = run_both(eng, soa, dtype)
    assert len(got["raw"]) > 0


def test_multi_tid(eng):
    from volcanosv_amd import synth
    parts = []
    for tid in range(3):
        t, nq, _ = synth.generate(30000, "hifi", seed=100 + tid, tid=tid, chrom_len=1500000, events_per_record=0.2, site_step=1000)
        parts.append((t, nq))
    t, nq = synth.concat(parts)
    soa = synth.to_soa(t, nq)
    got = run_both(eng, soa, DTYPE_HIFI)
    assert set(np.unique(got["calls"]["sig"]["tid"])) == {0, 1, 2}


def test_tid_hint_trims_keys_without_changing_results(eng):
    """vsv_records.tid_lo: a shard that holds chromosomes 5..6 of 7 sorts on tid - 5 (fewer key bits, fewer radix passes); the
    tables are the ones of the unhinted run and of the oracle. A single-chromosome view uses only_tid. A hint the data
    violates is reported, never silently mis-sorted."""
    from volcanosv_amd import synth
    from volcanosv_amd.engine import default_params
    parts = []
    for tid in (5, 6):
        t, nq, _ = synth.generate(30000, "hifi", seed=200 + tid, tid=tid, chrom_len=1500000, events_per_record=0.2, site_step=1000)
        parts.append((t, nq))
    t, nq = synth.concat(parts)
    soa = synth.to_soa(t, nq)
    soa.n_tids = 7
    want = run_both(eng, soa, DTYPE_HIFI)

    class Hinted:                      # same arrays, hinted struct
        def __init__(self, soa, lo):
            self.soa, self.lo = soa, lo

        def as_struct(self):
            r = self.soa.as_struct()
            r.tid_lo = self.lo
            return r

    for dtype in (DTYPE_HIFI, DTYPE_READS):
        p = default_params(dtype)
        eng.run(soa, p)
        base = eng.tables(dtype)
        eng.run(Hinted(soa, 5), p)
        assert_tables_equal(eng.tables(dtype), base, list(base.keys()))
    assert set(np.unique(want["calls"]["sig"]["tid"])) == {5, 6}
    with pytest.raises(VsvError) as e:
        eng.run(Hinted(soa, 6), default_params(DTYPE_HIFI))          # records of tid 5 lie below the hint
    assert e.value.status == -1 and "tid" in str(e.value)
    with pytest.raises(VsvError):
        eng.run(Hinted(soa, 7), default_params(DTYPE_HIFI))          # tid_lo >= n_tids
    one, nq1 = parts[1]
    s1 = synth.to_soa(one, nq1)
    s1.n_tids = 7
    base = run_both(eng, s1, DTYPE_HIFI)
    s1.only_tid = 6
    eng.run(s1, default_params(DTYPE_HIFI))
    assert_tables_equal(eng.tables(DTYPE_HIFI), base, list(base.keys()))


@pytest.mark.parametrize("kind,min_ok,min_err", [("basic", 150, 5), ("defects", 150, 40), ("collectors", 190, 0), ("dense", 12, 0)])
def test_random_small_inputs_statuses_and_tables(eng, kind, min_ok, min_err):
    """Hundreds of small random inputs (helpers.fuzz_case_basic / fuzz_case_defects: dense contigs where signatures collide,
    cluster and pair all the time; random tags, mapq, strands, data types, every threshold of vsv_params; planted defects that
    make the reference raise): the status equals the oracle's, and so does every table. `tools/fuzz_case.py KIND K` replays a case."""
    from helpers import fuzz_cases
    n_err = n_ok = 0
    seen = set()
    for case, soa, dtype, p in fuzz_cases(kind):
        st_o, want = oracle_run(soa, dtype, p)
        for layout in (1, 2):                   # both work mappings of the scan (vsv_params.scan_layout)
            p.scan_layout = layout
            try:
                eng.run(soa, p)
                st_g = 0
            except VsvError as e:
                st_g = e.status
            assert st_g == st_o, (kind, case, dtype, layout, st_g, st_o)
            if st_o == 0:
                got = eng.tables(dtype)
                try:
                    assert_tables_equal(got, want, list(got.keys()))
                except AssertionError as e:
                    raise AssertionError("%s case %d dtype %d layout %d: %s" % (kind, case, dtype, layout, e))
        seen.add(st_o)
        if st_o == 0:
            n_ok += 1
        else:
            n_err += 1
    assert n_ok >= min_ok and n_err >= min_err
    if kind == "defects":
        assert {0, -5, -6, -8} <= seen          # ok, reference_end assert, read-length assert, ZeroDivisionError


def test_engines_in_flight_do_not_interfere():
    """Three handles on three streams, each with a different input in flight at the same time (run_async, then finish): every
    engine returns the tables of its own input. Handles share nothing (include/volcanosv.h: no global mutable state)."""
    import torch
    from helpers import fuzz_cases
    from volcanosv_amd.engine import Engine
    cases = [c for c in fuzz_cases("basic", upto=89)]
    want = {}
    for case, soa, dtype, p in cases:
        want[case] = oracle_run(soa, dtype, p)
    streams = [torch.cuda.Stream() for _ in range(3)]
    engs = [Engine(0, stream=s.cuda_stream) for s in streams]
    try:
        n_ok = 0
        for i in range(0, len(cases), 3):
            batch = cases[i:i + 3]
            started = []
            for e, (case, soa, dtype, p) in zip(engs, batch):
                if want[case][0] == 0:
                    e.run_async(soa, p)
                    started.append((e, case, dtype))
            for e, case, dtype in started:
                e.finish()
                got = e.tables(dtype)
                assert_tables_equal(got, want[case][1], list(got.keys()))
                n_ok += 1
        assert n_ok > 60
    finally:
        for e in engs:
            e.close()


def test_long_records_with_both_haplotype_tags(eng):
    """Mb-scale records (every chunk of the scan lies inside one record) whose names carry hp1 AND hp2
    on every third record (two rows per signature), plus a low-mapq record and the reads / svim / collector op tables."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_CUTESV, DTYPE_READS, DTYPE_SVIM, F_HP1, F_HP2
    from volcanosv_amd.engine import default_params
    t, nq, nt = synth.generate(240, "contig", seed=41, chrom_len=60_000_000)
    t["flag"][::3] |= (F_HP1 | F_HP2)
    t["mapq"][5] = 3
    soa = synth.to_soa(t, nq)
    got = run_both(eng, soa, DTYPE_HIFI)
    both = np.isin(got["raw"]["rec"], np.arange(0, 240, 3))
    assert both.sum() > 1000 and ((got["raw"]["meta"][both] & 4) != 0).sum() * 2 == both.sum()
    for dt in (DTYPE_READS, DTYPE_SVIM, DTYPE_CUTESV):
        p = default_params(dt)
        p.enable_split = 0
        eng.cigar_scan(soa, p)
        raw = eng.table("raw")
        st, want = oracle_run(soa, dt, p)
        assert st == 0 and np.array_equal(raw, want["raw"]) and len(raw) > 1000


def test_unaligned_device_views(eng):
    """Device arrays that start in the middle of an allocation (a slice of a larger tensor): the split stage's wide loads
    need 16-byte / 4-byte aligned qid / flag / mapq arrays and must take the element-wise path otherwise."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.engine import DeviceRecords, default_params
    t, nq, nt = synth.generate(60001, "hifi", seed=17, chrom_len=2000000, events_per_record=0.2, site_step=1000)
    for skip in (1, 3):
        host = {k: (v[skip:].clone() if k != "cigar" else v.clone()) for k, v in t.items()}
        dev = {k: v.cuda() for k, v in t.items()}
        view = {k: (v[skip:] if k != "cigar" else v) for k, v in dev.items()}       # pointers offset by `skip` elements
        assert view["qid"].data_ptr() % 16 != 0 and view["flag"].data_ptr() % 4 != 0
        p = default_params(DTYPE_HIFI)
        eng.run(DeviceRecords(view, nq, nt), p)
        got = eng.tables(DTYPE_HIFI)
        st, want = oracle_run(synth.to_soa(host, nq), DTYPE_HIFI, p)
        assert st == 0
        assert_tables_equal(got, want, list(got.keys()))
        assert len(got["split"]) > 0
    del dev, view
    torch.cuda.empty_cache()


def test_staged_api_equals_fused(eng):
    from volcanosv_amd import synth
    from volcanosv_amd.engine import default_params
    t, nq, _ = synth.generate(50000, "hifi", seed=3, chrom_len=1000000, events_per_record=0.2, site_step=1000)
    soa = synth.to_soa(t, nq)
    p = default_params(DTYPE_HIFI)
    eng.run(soa, p)
    fused = eng.tables(DTYPE_HIFI)
    eng.cigar_scan(soa, p)
    raw = eng.table("raw")
    eng.split_pairs(p)
    eng.sort_cluster(p)
    c1 = eng.table("cluster1")
    eng.merge_sources(p)
    eng.pair_haplotypes(p)
    staged = eng.tables(DTYPE_HIFI)
    assert np.array_equal(raw, fused["raw"]) and np.array_equal(c1, fused["cluster1"])
    assert_tables_equal(staged, fused, list(fused.keys()))
    # split_min_mapq changed between the stages: the candidates are rebuilt with it (they depend on it), like a fused run with it
    p2 = default_params(DTYPE_HIFI)
    p2.min_split_mapq = 60
    eng.run(soa, p2)
    fused2 = eng.tables(DTYPE_HIFI)
    eng.cigar_scan(soa, p)
    eng.split_pairs(p2)
    eng.sort_cluster(p2)
    eng.merge_sources(p2)
    eng.pair_haplotypes(p2)
    assert_tables_equal(eng.tables(DTYPE_HIFI), fused2, list(fused2.keys()))


def test_staged_api_stages_called_again(eng):
    """A staged stage may be called again on the same run, e.g. with another threshold (include/volcanosv.h): whatever it and the
    stages behind it produce equals a fused run with the thresholds that were in force when each stage last ran — on rows and (the
    element path of large tables consumes its inputs: VSV_BIG=1 child suites) on elements. Long-record and read-shaped scans."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import SCAN_CONTIGS, SCAN_READS
    from volcanosv_amd.engine import default_params
    t, nq, _ = synth.generate(30000, "hifi", seed=9, chrom_len=600000, events_per_record=0.3, site_step=500)
    soa = synth.to_soa(t, nq)
    for layout in (SCAN_READS, SCAN_CONTIGS):
        p = default_params(DTYPE_HIFI)
        p.scan_layout = layout
        q = default_params(DTYPE_HIFI)
        q.scan_layout = layout
        q.cluster_shift = 40
        r = default_params(DTYPE_HIFI)
        r.scan_layout = layout
        r.cluster_shift = 40
        r.pair_shift = 80
        st, want_p = oracle_run(soa, DTYPE_HIFI, p)
        st2, want_q = oracle_run(soa, DTYPE_HIFI, q)
        st3, want_r = oracle_run(soa, DTYPE_HIFI, r)
        assert st == 0 and st2 == 0 and st3 == 0
        names = ["cigar", "split", "cluster1", "merged", "calls"]
        eng.cigar_scan(soa, p)
        eng.split_pairs(p)
        eng.sort_cluster(p)
        eng.sort_cluster(q)                       # again, another shift: cluster1 is the second call's
        assert np.array_equal(eng.table("cluster1"), want_q["cluster1"])
        eng.merge_sources(q)
        eng.merge_sources(q)                      # again: same tables
        assert np.array_equal(eng.table("merged"), want_q["merged"])
        eng.pair_haplotypes(q)
        eng.pair_haplotypes(r)                    # again, another pair_shift
        got = {k: eng.table(k) for k in names + ["raw"]}
        assert_tables_equal(got, want_r, names + ["raw"])
        eng.sort_cluster(p)                       # and back to the defaults from the first stage on
        eng.merge_sources(p)
        eng.pair_haplotypes(p)
        got = {k: eng.table(k) for k in names + ["raw"]}
        assert_tables_equal(got, want_p, names + ["raw"])


def test_split_overlap_modes_give_the_same_tables(eng):
    """vsv_params.split_overlap: the fused run builds the split candidates on the handle's auxiliary stream beside the scan (AUTO) or
    on the handle's stream (OFF); a timing choice only. Alternating runs on one handle, read-shaped and contig-shaped input, all
    data types that have a split stage."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import OVERLAP_AUTO, OVERLAP_OFF
    from volcanosv_amd.engine import default_params
    for shape, n, kw in (("hifi", 60000, dict(chrom_len=2_000_000, events_per_record=0.3, site_step=1000)), ("contig", 40, dict(chrom_len=20_000_000))):
        t, nq, _ = synth.generate(n, shape, seed=77, **kw)
        soa = synth.to_soa(t, nq)
        for dtype in (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS):
            _, want = oracle_run(soa, dtype)
            for mode in (OVERLAP_AUTO, OVERLAP_OFF, OVERLAP_AUTO):
                p = default_params(dtype)
                p.split_overlap = mode
                eng.run(soa, p)
                got = eng.tables(dtype)
                assert_tables_equal(got, want, list(got.keys()))


def test_edge_cases(eng):
    from volcanosv_amd.engine import default_params
    p = default_params(DTYPE_HIFI)
    # empty input
    soa = RecordSoA.from_tuples([])
    soa = RecordSoA(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint32), np.zeros(1, np.uint64),
                    np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint32))
    eng.run(soa, p)
    assert len(eng.table("calls")) == 0
    # single record, single op
    soa = RecordSoA.from_tuples([(0, 5, "x_hp1", 60, False, [(0, 100)])])
    run_both(eng, soa, DTYPE_HIFI)
    # ragged: 1-op records, a record straddling several 2048-op parts, event as first / last op, both hp tags
    big = []
    for i in range(3000):
        big += [(0, 40), (1 if i % 2 else 2, 31 if i % 500 == 0 else 3)]
    big.append((0, 10))
    recs = [(0, 10, "a_hp1", 60, False, [(1, 45), (0, 100)]),
            (0, 11, "b_hp2", 60, True, [(0, 100), (2, 77)]),
            (0, 12, "c_hp1", 60, False, [(0, 7)]),
            (0, 13, "d_hp1_hp2", 60, False, big),
            (0, 14, "e_hp2", 60, False, [(5, 9), (1, 300), (0, 5), (1, 300), (4, 3)]),
            (0, 15, "f_hp1", 49, False, [(0, 10), (1, 100), (0, 10)]),
            (0, 16, "g", 60, False, [(0, 10), (1, 100), (0, 10)])]
    for i in range(300):
        recs.append((0, 20 + i, "z%d_hp%d" % (i, 1 + i % 2), 60, False, [(0, 3)] if i % 3 else [(0, 3), (2, 50 + i), (0, 2)]))
    soa = RecordSoA.from_tuples(recs)
    run_both(eng, soa, DTYPE_HIFI)
    run_both(eng, soa, DTYPE_READS)
    # op count an exact multiple of the part size (2048) and of the chunk size (256)
    recs = [(0, i, "r%d_hp1" % i, 60, False, [(0, 5), (2, 30 + i % 7)] * 128) for i in range(64)]
    soa = RecordSoA.from_tuples(recs)
    assert soa.n_ops % 2048 == 0
    run_both(eng, soa, DTYPE_HIFI)


def test_scan_restaging_and_checkpoint_walks(eng):
    """Shapes that exercise the lazy parts of cigar_scan_emit: (a) thousands of 1-3-op records per 4096-op part (the LDS
    record table restages and the 64-record lookup window moves many times inside one part), (b) one multi-million-op record
    whose signatures are thousands of chunks apart (checkpoint walks over long runs of candidate-free chunks), (c) records
    that start on the last / first op of a 256-op chunk, (d) signatures as first and last op of a part."""
    rng = np.random.default_rng(77)
    recs = []
    pos = 100
    for i in range(20000):                                    # (a)
        pos += int(rng.integers(1, 40))
        k = int(rng.integers(0, 3))
        cig = [(0, int(rng.integers(1, 60)))]
        if k >= 1:
            cig += [(int(rng.integers(1, 3)), int(rng.choice([3, 29, 30, 31, 200]))), (0, int(rng.integers(1, 60)))]
        if k == 2:
            cig = [(4, 5)] + cig
        recs.append((0, pos, "t%d_hp%d" % (i, 1 + i % 2), 60 if i % 11 else 10, bool(i % 2), cig))
    big = []                                                  # (b)
    for i in range(1_200_000):
        big.append((0, int(rng.integers(1, 30))))
        big.append((1 if i % 2 else 2, 2 if i % 150_000 else 64))
    big.append((0, 9))
    recs.append((0, pos + 1000, "big_hp1", 60, False, big))
    for i in range(600):                                      # (c) op counts 255 / 256 / 257 in rotation
        n = 255 + i % 3
        cig = [(0, 2), (2, 30 + i % 5)] * ((n - 1) // 2) + [(0, 2)] * (1 + (n - 1) % 2)
        recs.append((0, pos + 2000 + i, "c%d_hp2" % i, 60, False, cig))
    recs.append((0, pos + 5000, "edge_hp1", 60, False, [(1, 50)] + [(0, 3), (2, 1)] * 2047 + [(0, 3), (2, 80)]))   # (d)
    soa = RecordSoA.from_tuples(recs)
    got = run_both(eng, soa, DTYPE_HIFI)
    assert len(got["raw"]) > 5000
    for dt in (DTYPE_READS,):
        run_both(eng, soa, dt)


def test_error_statuses(eng):
    from volcanosv_amd.engine import default_params
    soa = RecordSoA.from_tuples([(0, 10, "a_hp1", 60, False, [(0, 50), (7, 10), (0, 50)])])
    with pytest.raises(VsvError) as e:
        eng.run(soa, default_params(DTYPE_HIFI))
    assert e.value.status == -5          # VSV_E_REFEND, reference assert at Hifi.py:396
    eng.run(soa, default_params(DTYPE_READS))
    soa = RecordSoA.from_tuples([(0, 10, "a_hp1", 60, False, [(0, 500), (4, 100)]), (0, 900, "a_hp1", 60, False, [(4, 400), (0, 300)])])
    with pytest.raises(VsvError) as e:
        eng.run(soa, default_params(DTYPE_HIFI))
    assert e.value.status == -6          # VSV_E_READLEN, reference assert at Hifi.py:331
    # empty CIGAR
    soa = RecordSoA(np.array([1, 2], np.int32), np.zeros(2, np.int32), np.array([0, 1], np.uint32), np.array([0, 0, 1], np.uint64),
                    np.array([60, 60], np.uint8), np.array([F_HP1, F_HP1], np.uint8), np.array([(5 << 4)], np.uint32))
    with pytest.raises(VsvError) as e:
        eng.run(soa, default_params(DTYPE_HIFI))
    assert e.value.status == -4
    # caller bugs the kernels must survive: a query id beyond n_qids (the name table has no bit for it), CIGAR offsets that run
    # past the array (CLR walks every tagged record's CIGAR before the scan reports them)
    from volcanosv_amd import synth
    t, nq, _ = synth.generate(5000, "hifi", seed=77, chrom_len=2_000_000)
    good = synth.to_soa(t, nq)
    good.n_qids = 100
    with pytest.raises(VsvError) as e:
        eng.run(good, default_params(DTYPE_HIFI))
    assert e.value.status == -1 and "qid" in str(e.value)
    good.n_qids = nq
    off = good.cigar_off.copy()
    off[1000:2000] += np.uint64(1 << 40)
    good.cigar_off = off
    for dt in (DTYPE_CLR, DTYPE_HIFI):
        with pytest.raises(VsvError):
            eng.run(good, default_params(dt))


def test_seq_length_assert(eng):
    """`if read.seq: assert len(read.seq)==offset_contig` (H:397-398, RS:123-124): the BAM readers flag records whose stored SEQ
    length differs from the CIGAR's query length (VSV_F_SEQ_MISMATCH); a flagged record raises VSV_E_SEQLEN exactly when the
    reference walks it (haplotype tag + mapq, CLR gate), in both scan layouts, and never when it does not; the status equals the
    oracle's (which tests/test_reference_live.py pins to the reference)."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import F_SEQ_MISMATCH, SCAN_CONTIGS, SCAN_READS
    from volcanosv_amd.engine import default_params
    t, nq, _ = synth.generate(30000, "hifi", seed=91, chrom_len=3_000_000)
    base = synth.to_soa(t, nq)
    hp = (base.flag & (F_HP1 | F_HP2)) != 0
    walked = int(np.flatnonzero(hp & (base.mapq >= 50))[1234])
    low_q = int(np.flatnonzero(hp & (base.mapq < 50))[7])
    for dtype in (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS):
        for rec, want in ((walked, -10), (low_q, 0)):
            soa = synth.to_soa(t, nq)
            soa.flag = soa.flag.copy()                       # to_soa hands out views of the generator's tensors
            soa.flag[rec] |= F_SEQ_MISMATCH
            st_o, _ = oracle_run(soa, dtype)
            assert st_o == want, (dtype, rec, st_o)
            for layout in (SCAN_READS, SCAN_CONTIGS):
                p = default_params(dtype)
                p.scan_layout = layout
                try:
                    eng.run(soa, p)
                    st = 0
                except VsvError as e:
                    st = e.status
                assert st == want, (dtype, rec, layout, st)
    # CLR: a flagged record the gate rejects is never walked (C:427-431)
    gated = RecordSoA.from_tuples([(0, 100, "a_hp1", 60, False, [(0, 50), (1, 40), (0, 50), (1, 40), (0, 50)]),
                                   (0, 900, "b_hp2", 60, False, [(0, 500), (2, 60), (0, 500)])])
    gated.flag[0] |= F_SEQ_MISMATCH
    assert oracle_run(gated, DTYPE_CLR)[0] == 0 and oracle_run(gated, DTYPE_HIFI)[0] == -10
    eng.run(gated, default_params(DTYPE_CLR))
    with pytest.raises(VsvError) as e:
        eng.run(gated, default_params(DTYPE_HIFI))
    assert e.value.status == -10


def test_clr_gate_edge_cases(eng):
    """The CLR gate (ins_pct / var_dist, C:53-70, applied at C:422-433) inside the scan: EMPTY M / I ops (the only way to a zero
    ins_pct denominator with M ops present) send the part to the separate gate pass; records without M ops, with nothing but
    empty ops, with I ops only; a record longer than the chunks the scan keeps gate words for. Status and tables equal the
    oracle's in both scan layouts."""
    from volcanosv_amd import synth
    from volcanosv_amd.engine import default_params
    t, nq, _ = synth.generate(6000, "hifi", seed=123, chrom_len=1_500_000, events_per_record=0.5)
    base = synth.to_soa(t, nq)
    hp = np.flatnonzero((base.flag & (F_HP1 | F_HP2)) != 0)
    rng = np.random.default_rng(5)

    def variant(edit):
        soa = synth.to_soa(t, nq)
        soa.cigar = soa.cigar.copy()
        edit(soa)
        return soa

    def ops_of(soa, r):
        return soa.cigar[int(soa.cigar_off[r]):int(soa.cigar_off[r + 1])]

    def empty_some_m(soa):                       # an empty M op here and there: results as without the fusion
        ms = np.flatnonzero((soa.cigar & 15) == 0)
        soa.cigar[ms[rng.integers(0, len(ms), 40)]] &= np.uint32(15)

    def all_empty(soa):                          # every M and I op of one tagged record empty: m + ins == 0 -> ZeroDivisionError (C:61)
        seg = ops_of(soa, int(hp[100]))
        seg[(seg & 15) <= 1] &= np.uint32(15)

    def only_insertions(soa):                    # no M op: var_dist divides by zero (C:70)
        seg = ops_of(soa, int(hp[200]))
        seg[(seg & 15) == 0] |= np.uint32(1)

    def untagged_without_m(soa):                 # the gate is never asked for an untagged record
        r = int(hp[300])
        soa.flag = soa.flag.copy()
        soa.flag[r] &= np.uint8(~(F_HP1 | F_HP2) & 255)
        seg = ops_of(soa, r)
        seg[(seg & 15) == 0] |= np.uint32(2)

    wants = {"empty_some_m": 0, "all_empty": -8, "only_insertions": -8, "untagged_without_m": 0}
    for name, edit in (("empty_some_m", empty_some_m), ("all_empty", all_empty), ("only_insertions", only_insertions),
                       ("untagged_without_m", untagged_without_m)):
        soa = variant(edit)
        st_o, want = oracle_run(soa, DTYPE_CLR)
        assert st_o == wants[name], (name, st_o)
        for layout in (1, 2):
            p = default_params(DTYPE_CLR)
            p.scan_layout = layout
            try:
                eng.run(soa, p)
                st = 0
            except VsvError as e:
                st = e.status
            assert st == st_o, (name, layout, st, st_o)
            if st == 0:
                got = eng.tables(DTYPE_CLR)
                assert_tables_equal(got, want, list(got.keys()))
    # one record of 30000 ops among reads (more chunks than the scan keeps gate words for): the run repeats with the separate pass
    long_ops = []
    for i in range(15000):
        long_ops += [(0, 37), (2 if i % 50 else 1, 1 if i % 500 else 60)]
    recs = [(0, 100 + 50 * i, "q%d_hp%d" % (i, 1 + i % 2), 60, False, [(0, 3000), (1, 45), (0, 2000)]) for i in range(300)]
    recs.insert(150, (0, 5000, "long_hp1", 60, False, long_ops))
    recs.sort(key=lambda r: r[1])
    soa = RecordSoA.from_tuples(recs)
    run_both(eng, soa, DTYPE_CLR)


def test_capacity_overflow_reports_required_count():
    from volcanosv_amd import synth
    from volcanosv_amd.engine import Engine, default_params
    t, nq, _ = synth.generate(50000, "hifi", seed=3, chrom_len=1000000, events_per_record=0.5, site_step=500)
    soa = synth.to_soa(t, nq)
    e = Engine(0, max_sigs=2048)
    try:
        with pytest.raises(VsvError) as ex:
            e.run(soa, default_params(DTYPE_HIFI))
        assert ex.value.status == -3
        need = e.last_count()
        assert need > 2048
        e.reserve(0, 0, need + 16)
        e.run(soa, default_params(DTYPE_HIFI))
        from oracle import oracle
        st, want = oracle.run(soa, dtype=DTYPE_HIFI)
        assert np.array_equal(e.table("calls"), want["calls"])
    finally:
        e.close()
    # grow=True (what the drivers use): the same overflow is answered by reserving the reported count and running again,
    # synchronously or at finish() of an asynchronous run
    for use_async in (False, True):
        with Engine(0, max_sigs=2048, grow=True) as g:
            if use_async:
                g.run_async(soa, default_params(DTYPE_HIFI))
                g.finish()
            else:
                g.run(soa, default_params(DTYPE_HIFI))
            assert np.array_equal(g.table("calls"), want["calls"]) and np.array_equal(g.table("raw"), want["raw"])


def test_full_size_config2_properties(eng):
    """BASELINE.json config 2 at full size (10 M HiFi-like records, device-resident): idempotence, sortedness,
    call-table checksum equal to the CPU oracle's."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.engine import DeviceRecords, default_params
    t, nq, nt = synth.generate(10_000_000, "hifi", seed=20250330, device="cuda")
    dr = DeviceRecords(t, nq, nt)
    p = default_params(DTYPE_HIFI)
    eng.run(dr, p)
    a = eng.tables(DTYPE_HIFI)
    eng.run(dr, p)
    b = eng.tables(DTYPE_HIFI)
    assert_tables_equal(a, b, list(a.keys()))                      # idempotent / deterministic
    calls = a["calls"]
    assert len(calls) > 10000
    key = calls["sig"]["tid"].astype(np.int64) << 32 | calls["sig"]["pos"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)                               # sorted by (tid, pos)
    soa = synth.to_soa(t, nq)
    st, want = oracle_run(soa, DTYPE_HIFI)
    assert st == 0
    assert_tables_equal(a, want, list(a.keys()))
    del t
    torch.cuda.empty_cache()


def test_full_size_config3_properties():
    """BASELINE.json config 3 at full size (50 M ONT-like records, ~9.1e9 CIGAR ops, 38 GB device-resident): the raw, split, merged
    and call tables of the WHOLE input equal the CPU oracle's bit for bit (the oracle runs on a host copy in a thread beside the
    GPU checks); determinism, sortedness of the call table; and, as the fast failure, every table of a 200 k-record prefix."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_ONT
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params
    from concurrent.futures import ThreadPoolExecutor
    t, nq, nt = synth.generate(50_000_000, "ont", seed=20250331, device="cuda")
    assert t["cigar"].numel() > 9_000_000_000
    p = default_params(DTYPE_ONT)
    # the CPU oracle over ALL 50 M records (one core, ~a minute: windowed C), on a host copy, while the GPU checks below run
    pool = ThreadPoolExecutor(max_workers=1)
    full = pool.submit(oracle_run, synth.to_soa({n: v.cpu() for n, v in t.items()}, nq), DTYPE_ONT, p)
    with Engine(0, max_sigs=1 << 23) as e:
        dr = DeviceRecords(t, nq, nt, max_pos=synth.CHR10_LEN + 200000)
        e.run(dr, p)
        a = {k: e.table(k) for k in ("raw", "split", "merged", "calls")}
        e.run(dr, p)
        b = {k: e.table(k) for k in ("raw", "split", "merged", "calls")}
        assert_tables_equal(a, b, ["raw", "split", "merged", "calls"])
        assert len(a["raw"]) > 500_000 and len(a["calls"]) > 300_000
        key = a["calls"]["sig"]["tid"].astype(np.int64) << 32 | a["calls"]["sig"]["pos"].astype(np.int64)
        assert np.all(np.diff(key) >= 0)
        k = 200_000
        n_ops = int(t["cigar_off"][k])
        sl = {name: (v[: k + 1] if name == "cigar_off" else v[:n_ops] if name == "cigar" else v[:k]) for name, v in t.items()}
        nq_k = int(sl["qid"].max()) + 1
        e.run(DeviceRecords(sl, nq_k, nt, max_pos=synth.CHR10_LEN + 200000), p)
        got = e.tables(DTYPE_ONT)
        st, want = oracle_run(synth.to_soa({n: v.cpu() for n, v in sl.items()}, nq_k), DTYPE_ONT, p)
        assert st == 0
        assert_tables_equal(got, want, list(got.keys()))          # (the fast failure: the prefix)
    del t, sl, dr
    torch.cuda.empty_cache()
    st, want = full.result()
    pool.shutdown(wait=True)
    assert st == 0 and len(want["calls"]) > 300_000
    assert_tables_equal(a, want, ["raw", "split", "merged", "calls"])             # every row of the full-size run, bit for bit


@pytest.mark.parametrize("name", ["bnd_a", "bnd_b"])
def test_bnd_branch_golden(eng, name):
    """Complex_SV breakend branch: HIP kernels vs the oracle (bit-exact rows) and vs the svim-asm reference outputs."""
    from test_bnd_oracle import check_against_golden, load
    from oracle import oracle
    doc, seg = load(name)
    cand, calls = eng.bnd(seg)
    ocand, ocalls = oracle.run_bnd(seg)
    assert np.array_equal(cand, ocand) and np.array_equal(calls, ocalls)
    check_against_golden(doc, seg, cand, calls)


def test_bnd_synthetic_config5(eng):
    """Config-5 shaped stream: many split contigs over 22 chromosomes, hp2 copies jittered by +-300 bp, dense partitions."""
    from oracle import oracle
    from volcanosv_amd import bnd
    rng = np.random.default_rng(5)
    contigs = [("chr%d" % (i + 1), 50_000_000 + 1_000_000 * i) for i in range(22)]
    reads = []
    for e in range(20000):
        t1, t2 = int(rng.integers(0, 22)), int(rng.integers(0, 22))
        p1, p2 = int(rng.integers(100000, 40_000_000)), int(rng.integers(100000, 40_000_000))
        r1, r2 = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        L = 40000

        def read(hap, a, b):
            segs = [[t1, a - 20000, a, 0, 20000, L, 0], [t2, b, b + 20000, 20000, 40000, L, 0]]
            if r1:
                segs[0] = [t1, a, a + 20000, L - 20000, L, L, 1]
            if r2:
                segs[1] = [t2, b - 20000, b, 0, L - 20000, L, 1]
            return {"hap": hap, "name": "PS%d_hp%d" % (e, hap), "segs": segs}
        reads.append(read(1, p1, p2))
        if rng.random() < 0.8:
            reads.append(read(2, p1 + int(rng.integers(-300, 301)), p2 + int(rng.integers(-300, 301))))
        if e % 100 == 0:
            for k in range(12):
                reads.append(read(1 + k % 2, p1 + k, p2 + k))
    reads.sort(key=lambda r: r["hap"])
    seg = bnd.SegmentSoA(reads, contigs)
    cand, calls = eng.bnd(seg)
    ocand, ocalls = oracle.run_bnd(seg)
    assert len(cand) > 30000 and np.array_equal(cand, ocand) and np.array_equal(calls, ocalls)
    gts = (calls["meta"] >> 4) & 3
    assert (gts == 3).sum() > 5000


def test_bnd_pair_rows_after_exchange(eng):
    """vsv_bnd_set_candidates + vsv_bnd_pair: the per-rank step after the multi-GPU exchange (config 5)."""
    from test_bnd_oracle import load
    from oracle import oracle
    doc, seg = load("bnd_b")
    cand = eng.bnd_candidates(seg)
    hap2 = (cand["meta"] & 4) != 0
    rows = cand[np.lexsort((np.arange(len(cand)), cand["read"], hap2))]
    sub = rows[rows["src_tid"] % 2 == 0]                  # the contigs one of two ranks would own
    got = eng.bnd_pair_rows(sub, seg.contig_rank)
    want = oracle.run_bnd_pair(sub, seg.contig_rank)
    assert len(got) > 10 and np.array_equal(got, want)


def test_dense_runs_use_the_wave_cooperative_kernels(eng):
    """Dense regions (hundreds of signatures within 100 bp, runs and stretches > 48 rows): cluster_long / pair_long."""
    from volcanosv_amd import synth
    t, nq, nt = synth.generate(3000, "contig", seed=21, chrom_len=30_000_000)     # ~450x contig coverage
    soa = synth.to_soa(t, nq)
    from volcanosv_amd.engine import Engine
    e = Engine(0, max_sigs=1 << 23)
    try:
        got = run_both(e, soa, DTYPE_HIFI)
        assert len(got["raw"]) > 30000
        t, nq, nt = synth.generate(300000, "ont", seed=22, chrom_len=600_000, events_per_record=0.3, site_step=2000)
        soa = synth.to_soa(t, nq)
        run_both(e, soa, DTYPE_ONT)
    finally:
        e.close()


def test_overflow_batches_after_runs_without_any():
    """A part of the scan sends its rows as one batch when it has streamed its ops; a part with more rows than its stage holds (32
    rows read-shaped, 192 descriptors long-record) files the batches in front of the last one in an overflow list, placed by a
    launch of its own — which a handle skips while its runs file none. A handle that has only seen sparse input and then meets
    insertions packed into every read (hundreds of rows per 4096-op part) repeats that ONE run with the launch (vsv_rerun_count),
    keeps it for the next runs, and its tables equal the oracle's throughout. Both scan layouts."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import SCAN_CONTIGS, SCAN_READS
    from volcanosv_amd.engine import Engine, default_params
    from volcanosv_amd.soa import RecordSoA
    t, nq, _ = synth.generate(20000, "hifi", seed=31, chrom_len=2_000_000, events_per_record=0.05, site_step=1000)
    sparse = synth.to_soa(t, nq)
    recs = []
    for i in range(600):                                    # 40 insertions of 60-99 bp per read of 81 ops, 150 bp apart
        cig = [(0, 200)]
        for k in range(40):
            cig += [(1, 60 + (i + k) % 40), (0, 150)]
        recs.append((0, 1000 + 37 * i, "d%d_hp%d" % (i, 1 + i % 2), 60, False, cig))
    dense = RecordSoA.from_tuples(recs)
    dense.max_pos = 1 << 20
    for layout in (SCAN_READS, SCAN_CONTIGS):
        p = default_params(DTYPE_HIFI)
        p.scan_layout = layout
        st, want_s = oracle_run(sparse, DTYPE_HIFI, p)
        st2, want_d = oracle_run(dense, DTYPE_HIFI, p)
        assert st == 0 and st2 == 0 and len(want_d["raw"]) == 600 * 40
        with Engine(0) as e:
            for _ in range(2):
                e.run(sparse, p)
                assert_tables_equal(e.tables(DTYPE_HIFI), want_s, list(want_s.keys())[:6])
            assert e.rerun_count() == 0
            e.run(dense, p)
            got = e.tables(DTYPE_HIFI)
            assert_tables_equal(got, want_d, list(got.keys()))
            n = e.rerun_count()
            assert n in (1, 2), n       # the batches (a 4096-op part holds ~2000 of these rows, an 8192-op one ~4000) — and the sort buckets, sized for the sparse tables
            e.run(dense, p)
            assert_tables_equal(e.tables(DTYPE_HIFI), want_d, list(got.keys()))
            assert e.rerun_count() == n                              # the handle keeps the launch now
            e.run(sparse, p)
            assert_tables_equal(e.tables(DTYPE_HIFI), want_s, list(want_s.keys())[:6])


def test_config4_shape_22_chromosomes(eng):
    """BASELINE config 4 in miniature: 22 tids with records on every one of them (hg19-ordered lengths scaled down) through the
    HIP path — (a) the whole multi-chromosome SoA in one call against the oracle, (b) sharded like bench.py --config 4 / the WGS
    driver: shard.lpt_assign gives every rank its chromosomes, a rank runs them one at a time with the tid_lo / n_tids key hints,
    and the union of the per-chromosome call tables is the unsharded call table."""
    from volcanosv_amd import shard, synth
    from volcanosv_amd.engine import default_params
    parts, lens = [], [int(l // 60) + 60_000 for l in synth.HG19_LEN]
    for tid in range(22):
        t, nq, _ = synth.generate(4000 + 150 * (tid % 5), "hifi", seed=400 + tid, tid=tid, chrom_len=lens[tid], events_per_record=0.15, site_step=1000)
        parts.append((t, nq))
    t, nq = synth.concat(parts)
    soa = synth.to_soa(t, nq)
    soa.n_tids = 22
    want = run_both(eng, soa, DTYPE_HIFI)
    assert len(np.unique(want["calls"]["sig"]["tid"])) == 22
    run_both(eng, soa, DTYPE_READS)
    p = default_params(DTYPE_HIFI)
    for world in (3, 8):
        owner = shard.lpt_assign([int(pt[0]["pos"].numel()) for pt in parts], world)
        assert max(np.bincount(owner, minlength=world)) == -(-22 // world)          # equal-sized chromosomes: 3 on the busiest of 8 ranks
        got = []
        first_rec = np.concatenate(([0], np.cumsum([int(pt[0]["pos"].numel()) for pt in parts])))
        for rank in range(world):
            for tid in [c for c in range(22) if owner[c] == rank]:
                one = synth.to_soa(parts[tid][0], parts[tid][1])
                one.n_tids, one.only_tid, one.max_pos = 22, tid, lens[tid] + 100000
                eng.run(one, p)
                c = eng.table("calls").copy()
                c["sig"]["rec"] += np.uint32(first_rec[tid])                      # record indices of the shard -> of the whole input
                split = c["sig"]["rec2"] != 0xFFFFFFFF
                c["sig"]["rec2"][split] += np.uint32(first_rec[tid])
                got.append(c)
        allc = np.concatenate(got)
        allc = allc[np.argsort((allc["sig"]["tid"].astype(np.int64) << 32) | (allc["sig"]["pos"].astype(np.int64) + 65536), kind="stable")]
        # a/b index the merged table of the call's own run: compare the signature, genotype and member counts
        assert np.array_equal(allc["sig"], want["calls"]["sig"]) and np.array_equal(allc["gt"], want["calls"]["gt"])


def test_config5_exchange_with_the_hip_engine(tmp_path):
    """BASELINE config 5 over two ranks (gloo rendezvous, both ranks on cuda:0), HIP kernels as the compute on every rank:
    candidates on the owner of the primary alignment (vsv_bnd_segments), shard.exchange_bnd to the owner of the source contig,
    vsv_bnd_set_candidates + vsv_bnd_pair there, gather — equal to the single-process HIP run and to the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "tests", "_bnd_shard_worker.py"), "--engine", "hip", "--events", "6000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "BND_SHARD_OK" in r.stdout


def test_bucket_sort_overflow_falls_back_to_the_lsd_passes():
    """The signature tables are sorted by one counting pass into key-range buckets + an LDS sort per bucket; a bucket that does not
    fit (here: VSV_BK_CAP=48 rows instead of 4096) raises a device flag and the run is repeated through the LSD radix passes. The
    result must not depend on which of the two happened; VSV_SORT=lsd never uses the buckets. The bucket sort itself has two forms:
    two launches (every bucket owns a region of slots, a tile reserves room per bucket with one atomic, the LDS sort restores the input
    order of the chunks first: rs_slot_scatter / bk_slot_sort, the default) and three (histogram, self-scanning scatter, LDS sort:
    VSV_BK_SLOTS=0); both run here, with and without the overflow."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import run_both\n"
            "from volcanosv_amd import synth\n"
            "from volcanosv_amd.engine import Engine\n"
            "from volcanosv_amd.abi import DTYPE_HIFI, DTYPE_READS\n"
            "t, nq, _ = synth.generate(60000, 'hifi', seed=3, chrom_len=3000000, events_per_record=0.3, site_step=1000)\n"
            "soa = synth.to_soa(t, nq)\n"
            "with Engine(0) as e:\n"
            "    for k in range(3):\n"
            "        g = run_both(e, soa, DTYPE_HIFI)\n"
            "    run_both(e, soa, DTYPE_READS)\n"
            "print('SORT_OK', len(g['calls']))\n") % (root, os.path.join(root, "tests"))
    outs = []
    # (VSV_BK_TIEMAX=0: rows of equal key whose chunks arrived out of tile order are put back by tile-number passes — the form for runs of
    # more than 64 equal keys — instead of by rank inside the run)
    for env_extra in ({"VSV_BK_CAP": "48"}, {"VSV_SORT": "lsd"}, {}, {"VSV_BK_TIEMAX": "0"}, {"VSV_BK_SLOTS": "0"}, {"VSV_BK_SLOTS": "0", "VSV_BK_CAP": "48"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra), cwd=root)
        assert r.returncode == 0 and "SORT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1


def test_clr_fallback_followed_by_a_bucket_overflow():
    """Both whole-run fallbacks in one call: a CLR part longer than the fused gate's state (the run is repeated with the separate
    gate pass) whose repetition then overflows a bucket of the bucket sort (VSV_BK_CAP=2: repeated again through the LSD passes).
    finish() resolves them one after the other; neither flag may reach the success path with half-written tables."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import oracle_run, assert_tables_equal\n"
            "from volcanosv_amd.soa import RecordSoA\n"
            "from volcanosv_amd.engine import Engine, default_params\n"
            "from volcanosv_amd.abi import DTYPE_CLR\n"
            "long_ops = []\n"
            "for i in range(15000):\n"
            "    long_ops += [(0, 37), (2 if i %% 50 else 1, 1 if i %% 500 else 60)]\n"
            "recs = [(0, 100 + 50 * i, 'q%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 3000), (1, 45 + i %% 7), (0, 2000)]) for i in range(300)]\n"
            "recs.insert(150, (0, 5000, 'long_hp1', 60, False, long_ops))\n"
            "recs.sort(key=lambda r: r[1])\n"
            "soa = RecordSoA.from_tuples(recs)\n"
            "p = default_params(DTYPE_CLR)\n"
            "p.scan_layout = 1\n"                                    # record-aligned parts: the long record is one part of > 96 chunks
            "st, want = oracle_run(soa, DTYPE_CLR, p)\n"
            "assert st == 0\n"
            "with Engine(0) as e:\n"
            "    e.run(soa, p)\n"
            "    got = e.tables(DTYPE_CLR)\n"
            "    assert_tables_equal(got, want, list(got.keys()))\n"
            "    n = e.rerun_count()\n"
            "    e.run(soa, p)\n"                                    # the handle keeps the separate gate and the LSD passes: no repetition
            "    assert_tables_equal(e.tables(DTYPE_CLR), want, list(got.keys()))\n"
            "    print('FALLBACKS_OK', n, e.rerun_count())\n") % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VSV_DEBUG="1", VSV_BK_CAP="2"), cwd=root)
    assert r.returncode == 0 and "FALLBACKS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    n_first, n_second = [int(x) for x in r.stdout.strip().splitlines()[-1].split()[1:]]
    assert n_first == 2 and n_second == 2


def test_inputs_produced_on_another_stream():
    """include/volcanosv.h: device-resident inputs are read on the handle's stream. A caller whose arrays are still being produced
    on another stream orders the two on the device with vsv_wait_for_stream (Engine.wait_for_stream) instead of a host
    synchronisation: records generated on a side stream behind a long-running kernel, engine on its own stream, no host sync in
    between — the tables are the oracle's."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params
    dev = torch.device("cuda", 0)
    prod, own = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    t_host, nq, nt = synth.generate(300000, "hifi", seed=55, chrom_len=20_000_000)
    want_soa = synth.to_soa(t_host, nq)
    st, want = oracle_run(want_soa, DTYPE_HIFI)
    assert st == 0
    pinned = {k: v.pin_memory() for k, v in t_host.items()}
    with Engine(0, stream=own.cuda_stream) as e:
        for _ in range(3):
            with torch.cuda.stream(prod):
                busy = torch.randn(8192, 8192, device=dev)
                for _k in range(6):
                    busy = busy @ busy * 1e-4                      # keeps the producer stream busy for a while
                t = {k: v.to(dev, non_blocking=True) for k, v in pinned.items()}   # the records arrive behind it
            recs = DeviceRecords(t, nq, nt, sync=False)
            e.wait_for_stream(prod.cuda_stream)
            e.run(recs, default_params(DTYPE_HIFI))
            got = e.tables(DTYPE_HIFI)
            assert_tables_equal(got, want, list(got.keys()))
            del t


def test_bnd_device_rows_produced_on_torchs_stream_for_an_engine_with_its_own():
    """Engine.bnd_pair_device copies candidate rows on the HANDLE's stream; in the multi-GPU exchange they have just been produced
    on torch's current stream (index, sort, all-to-all). An engine on a stream of its own (bench engs[1..]) must order the two on
    the device: rows that arrive behind a long-running kernel on torch's stream, no host synchronisation in between."""
    import torch
    from oracle import oracle
    from volcanosv_amd import synth
    from volcanosv_amd.abi import BND_DTYPE
    from volcanosv_amd.engine import Engine
    dev = torch.device("cuda", 0)
    seg, _ = synth.generate_bnd(4000, seed=78)
    cand, _ = oracle.run_bnd(seg)
    hap2 = (cand["meta"] & 4) != 0
    rows = cand[np.lexsort((np.arange(len(cand)), cand["read"], hap2))]       # collection order
    want = oracle.run_bnd_pair(rows, seg.contig_rank)
    pinned = torch.from_numpy(np.frombuffer(rows.tobytes(), dtype=np.uint8).copy()).pin_memory()
    rank_t = torch.from_numpy(np.ascontiguousarray(seg.contig_rank)).to(dev)
    torch.cuda.synchronize()
    own = torch.cuda.Stream(device=dev)
    with Engine(0, stream=own.cuda_stream) as e:
        for _ in range(3):
            busy = torch.randn(8192, 8192, device=dev)
            for _k in range(6):
                busy = busy @ busy * 1e-4                                        # torch's current stream stays busy for a while
            rows_t = pinned.to(dev, non_blocking=True)                           # ... and the rows arrive behind it
            got = e.bnd_pair_device(rows_t, rank_t, dev)
            assert np.array_equal(np.frombuffer(got.cpu().numpy().tobytes(), dtype=BND_DTYPE), want)
            del rows_t


def test_pairing_in_rounds_equals_the_sequential_walk():
    """Dense merged tables (signatures closer than 2 * pair_shift for a whole chromosome) are paired in rounds of reservations
    instead of one wave walking the list; VSV_PAIR=rounds forces that path from a handle's first run. Every table must equal the
    oracle's (the literal first-come-first-served loop of pair_sig, H:552-569) on dense and on ordinary inputs, and the dense
    fuzz family must pass under it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import run_both, oracle_run, assert_tables_equal\n"
            "from helpers import fuzz_cases\n"
            "from volcanosv_amd import synth\n"
            "from volcanosv_amd.engine import Engine\n"
            "from volcanosv_amd.abi import DTYPE_HIFI, DTYPE_ONT\n"
            "with Engine(0, max_sigs=1 << 23) as e:\n"
            "    t, nq, _ = synth.generate(4000, 'contig', seed=23, chrom_len=8_000_000)\n"      # ~8000x coverage: one stretch per list
            "    g = run_both(e, synth.to_soa(t, nq), DTYPE_HIFI)\n"
            "    n1 = len(g['calls'])\n"
            "    t, nq, _ = synth.generate(200000, 'ont', seed=24, chrom_len=900_000, events_per_record=0.3, site_step=1500)\n"
            "    run_both(e, synth.to_soa(t, nq), DTYPE_ONT)\n"
            "    t, nq, _ = synth.generate(100000, 'hifi', seed=25)\n"
            "    run_both(e, synth.to_soa(t, nq), DTYPE_HIFI)\n"
            "    for case, soa, dtype, p in fuzz_cases('dense'):\n"
            "        st, want = oracle_run(soa, dtype, p)\n"
            "        e.run(soa, p)\n"
            "        got = e.tables(dtype)\n"
            "        assert_tables_equal(got, want, list(got.keys()))\n"
            "print('PAIR_OK', n1)\n") % (root, os.path.join(root, "tests"))
    outs = []
    for env_extra in ({"VSV_PAIR": "rounds"}, {"VSV_PAIR": "walk"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env_extra), cwd=root)
        assert r.returncode == 0 and "PAIR_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and int(outs[0].split()[1]) > 50000


def test_full_size_config4_call_tables():
    """BASELINE.json config 4 at N = 1 and FULL size: 22 chromosomes (hg19 lengths) x 20 M HiFi-like records, generated with
    bench.py's seeds and run exactly as bench.py config4 runs them (three engines taking turns, tid_lo / n_tids / max_pos key hints,
    split_overlap off) — every chromosome's call table and raw table equal the CPU oracle's on all 20 M records (the oracle runs of
    several chromosomes overlap on the host cores, ctypes releases the GIL)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from volcanosv_amd import shard, synth
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params
    dev = torch.device("cuda", 0)
    n_chrom, n_per = 22, 20_000_000
    owner = shard.lpt_assign([n_per] * n_chrom, 1)
    assert set(owner) == {0}
    p = default_params(DTYPE_HIFI)
    p.split_overlap = 1
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(2)]
    engs = [Engine(0, stream=s.cuda_stream, max_sigs=1 << 22) for s in streams]
    pool = ThreadPoolExecutor(max_workers=8)
    pending, total_calls = [], 0

    def check(c, got, fut):
        st, want = fut.result()
        assert st == 0, c
        try:
            assert_tables_equal(got, want, ["raw", "calls"])
        except AssertionError as e:
            raise AssertionError("chromosome %d: %s" % (c, e))
        return len(want["calls"])
    try:
        busy = {}
        for c in range(n_chrom + len(engs)):
            e = engs[c % len(engs)]
            if c % len(engs) in busy:                               # the engine's previous chromosome: collect before it is reused
                pc, t, nq = busy.pop(c % len(engs))
                e.finish()
                got = {k: e.table(k) for k in ("raw", "calls")}
                soa = synth.to_soa(t, nq)                           # host copy of the same 20 M records for the oracle
                del t
                pending.append((pc, got, pool.submit(oracle_run, soa, DTYPE_HIFI, p)))
                while len(pending) > 6:
                    total_calls += check(*pending.pop(0))
            if c < n_chrom:
                t, nq, _ = synth.generate(n_per, "hifi", seed=20250328 + 4 + 1000 * c, tid=c, chrom_len=synth.HG19_LEN[c], device=dev)
                recs = DeviceRecords(t, nq, c + 1, max_pos=synth.HG19_LEN[c] + 200000, tid_lo=c)
                e.run_async(recs, p)
                busy[c % len(engs)] = (c, t, nq)
        for item in pending:
            total_calls += check(*item)
        assert total_calls > 22 * 50_000
    finally:
        pool.shutdown(wait=True)
        for e in engs:
            e.close()
        torch.cuda.empty_cache()


def test_full_size_config5_breakends():
    """BASELINE.json config 5 at full size (1 M events x 2 haplotypes, bench.py's seed, ~6.7 M aligned segments): candidates,
    collection-order exchange and pairing on device rows as bench.py config5 runs them at N = 1, against oracle.run_bnd."""
    import torch
    from oracle import oracle
    from volcanosv_amd import bnd, shard, synth
    from volcanosv_amd.abi import BND_DTYPE
    from volcanosv_amd.engine import Engine
    dev = torch.device("cuda", 0)
    eng = Engine(0, max_sigs=1 << 24)                                # bench.py's row capacity for this workload
    seg, primary_tid = synth.generate_bnd(1_000_000, seed=20250328 + 5)
    assert len(seg.q_start) > 6_000_000
    dseg = bnd.DeviceSegments(seg, dev)
    gid_t = torch.arange(len(seg.hap), dtype=torch.int64, device=dev)
    owner_t = torch.zeros(len(synth.HG19_LEN), dtype=torch.int64, device=dev)
    rank_t = torch.from_numpy(np.ascontiguousarray(seg.contig_rank)).to(dev)
    got = []
    for _ in range(2):                                               # twice: deterministic
        cand = eng.bnd_candidates_device(dseg, dev)
        rows = shard.exchange_bnd_device(cand, gid_t, owner_t, dev)
        calls = eng.bnd_pair_device(rows, rank_t, dev)
        got.append((np.frombuffer(cand.cpu().numpy().tobytes(), dtype=BND_DTYPE), np.frombuffer(calls.cpu().numpy().tobytes(), dtype=BND_DTYPE)))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    ocand, ocalls = oracle.run_bnd(seg)
    assert len(ocalls) > 1_000_000
    assert np.array_equal(got[0][0], ocand)
    fields = ("src_tid", "src_pos", "dst_tid", "dst_pos", "read", "read2", "meta")
    canon = lambda c: c[np.lexsort(tuple(c[f] for f in reversed(fields)))]
    a, b = canon(got[0][1]), canon(ocalls)
    assert len(a) == len(b)
    for f in fields:
        assert np.array_equal(a[f], b[f]), f                         # the same calls (slot order vs key order)
    eng.close()
    del dseg
    torch.cuda.empty_cache()


def test_full_size_row2c_contigs():
    """SURVEY §8d row 2c at full size (200 k contig-like records of ~16 k ops, 3.2 G CIGAR ops, bench.py's seed): the raw, merged and
    call tables of the WHOLE pile equal the CPU oracle's bit for bit (the oracle runs on a host copy in a thread beside the GPU
    checks); determinism over three runs, sortedness and internal consistency; and, as the fast failure, every table of the
    20 k-record prefix."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params
    from concurrent.futures import ThreadPoolExecutor
    t, nq, nt = synth.generate(200_000, "contig", seed=20250328 + 6, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
    assert t["cigar"].numel() > 3_000_000_000
    p = default_params(DTYPE_HIFI)
    # the CPU oracle over ALL 200 k contigs (3.2 G ops on one core: tens of seconds), on a host copy, while the GPU checks below run
    pool = ThreadPoolExecutor(max_workers=1)
    full = pool.submit(oracle_run, synth.to_soa({n: v.cpu() for n, v in t.items()}, nq), DTYPE_HIFI, p)
    with Engine(0, max_sigs=1 << 24) as e:
        dr = DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0)
        runs = []
        for _ in range(3):                                           # (the handle's first run sorts with other kernels than the later ones)
            e.run(dr, p)
            runs.append({k: e.table(k) for k in ("raw", "merged", "calls")})
        for other in runs[1:]:
            assert_tables_equal(other, runs[0], ["raw", "merged", "calls"])
        a = runs[0]
        assert len(a["raw"]) > 6_000_000 and len(a["calls"]) > 2_000_000
        key = a["calls"]["sig"]["pos"].astype(np.int64)
        assert np.all(np.diff(key) >= 0)                             # one chromosome: sorted by pos
        mk = ((a["merged"]["meta"].astype(np.int64) & 4) << 40) | (a["merged"]["pos"].astype(np.int64) + 65536)
        assert np.all(np.diff(mk) >= 0)                              # merged: (hap, pos)
        c = a["calls"]
        paired = c["gt"] == 2
        assert paired.sum() > 100_000
        m = a["merged"]
        assert np.all((m["meta"][c["a"][paired]] & 4) == 0) and np.all((m["meta"][c["b"][paired]] & 4) != 0)
        assert len(np.unique(c["b"][paired])) == paired.sum()        # an hp2 row pairs at most once
        # every merged row is in exactly one call: hp1 rows through a, hp2 rows through b
        assert len(c) == len(m) - paired.sum()
        k = 20_000
        n_ops = int(t["cigar_off"][k])
        sl = {name: (v[: k + 1] if name == "cigar_off" else v[:n_ops] if name == "cigar" else v[:k]) for name, v in t.items()}
        nq_k = int(sl["qid"].max()) + 1
        e.run(DeviceRecords(sl, nq_k, 1, max_pos=synth.CHR10_LEN + 200000), p)
        got = e.tables(DTYPE_HIFI)
        st, want = oracle_run(synth.to_soa({n: v.cpu() for n, v in sl.items()}, nq_k), DTYPE_HIFI, p)
        assert st == 0 and len(want["calls"]) > 150_000
        assert_tables_equal(got, want, list(got.keys()))          # (the fast failure: the prefix)
    del t, sl, dr
    torch.cuda.empty_cache()
    st, want = full.result()
    pool.shutdown(wait=True)
    assert st == 0 and len(want["calls"]) > 2_000_000
    assert_tables_equal(a, want, ["raw", "merged", "calls"])       # every row of the full-size run, bit for bit


def test_cold_engine_takes_the_element_path_on_a_contig_pile():
    """The drop-in CLI runs one chromosome per process (Raw_variant_call.py:65-73): every run is the first of its handle. A fused run
    of a handle without history waits for the scan once and takes the row count from it, so a pile of contigs (here 20 k contigs,
    ~630 k raw signatures... above the row path's ~1.3 M-row limit only at full size: the limit is lowered for the test through the
    row capacity hint) runs its stages on elements from the first run on: vsv_path_counts reports it, and the tables equal the
    oracle's. A second engine that has seen the input before reports the same path without a wait."""
    import torch
    from volcanosv_amd import synth
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params
    t, nq, nt = synth.generate(45_000, "contig", seed=20250328 + 6, tid=0, chrom_len=synth.CHR10_LEN, device="cuda")
    p = default_params(DTYPE_HIFI)
    dr = DeviceRecords(t, nq, 1, max_pos=synth.CHR10_LEN + 200000, tid_lo=0)
    with Engine(0, max_sigs=1 << 23) as e:
        assert e.path_counts() == (0, 0)
        e.run(dr, p)
        cold = {k: e.table(k) for k in ("raw", "cigar", "merged", "calls")}
        assert len(cold["raw"]) > 1_350_000
        assert e.path_counts() == (1, 1) and e.rerun_count() == 0           # elements on the first run, after one wait for the scan
        e.run(dr, p)
        warm = {k: e.table(k) for k in ("raw", "cigar", "merged", "calls")}
        assert e.path_counts() == (2, 1)
        assert_tables_equal(warm, cold, list(cold.keys()))
    k = 6_000
    n_ops = int(t["cigar_off"][k])
    sl = {name: (v[: k + 1] if name == "cigar_off" else v[:n_ops] if name == "cigar" else v[:k]) for name, v in t.items()}
    nq_k = int(sl["qid"].max()) + 1
    with Engine(0, max_sigs=1 << 22) as e:                                  # a cold engine on the prefix the oracle finishes in seconds
        e.run(DeviceRecords(sl, nq_k, 1, max_pos=synth.CHR10_LEN + 200000), p)
        got = e.tables(DTYPE_HIFI)
        assert e.path_counts()[1] == 1
    st, want = oracle_run(synth.to_soa({n: v.cpu() for n, v in sl.items()}, nq_k), DTYPE_HIFI, p)
    assert st == 0 and len(want["calls"]) > 40_000
    assert_tables_equal(got, want, list(got.keys()))
    del t, sl, dr
    torch.cuda.empty_cache()


def test_element_path_falls_back_to_rows_beyond_its_32_bit_limits():
    """The element kernels compute positions and lengths in 32 bits; a length outside [0, 2^30) (here: five insertions of 2^28 - 1
    bases folded into one signature of 1.3e9, H:91-97) raises a device flag where the elements are built and the same input runs
    again on rows, with the 64-bit predicates. Tables equal the oracle's; one repetition, then the handle stays on rows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import oracle_run, assert_tables_equal\n"
            "from volcanosv_amd.soa import RecordSoA\n"
            "from volcanosv_amd.engine import Engine, default_params\n"
            "from volcanosv_amd.abi import DTYPE_HIFI\n"
            "big = (1 << 28) - 1\n"
            "recs = [(0, 100 + 50 * i, 'q%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 3000), (1, 45 + i %% 7), (0, 2000)]) for i in range(200)]\n"
            "recs.insert(100, (0, 5000, 'huge_hp1', 60, False, [(0, 100)] + [(1, big), (0, 10)] * 5 + [(0, 50)]))\n"
            "recs.sort(key=lambda r: r[1])\n"
            "soa = RecordSoA.from_tuples(recs)\n"
            "soa.max_pos = 1 << 20\n"
            "p = default_params(DTYPE_HIFI)\n"
            "st, want = oracle_run(soa, DTYPE_HIFI, p)\n"
            "assert st == 0 and int(want['cigar']['svlen'].max()) > (1 << 30)\n"
            "with Engine(0) as e:\n"
            "    e.run(soa, p)\n"
            "    got = e.tables(DTYPE_HIFI)\n"
            "    assert_tables_equal(got, want, list(got.keys()))\n"
            "    n = e.rerun_count()\n"
            "    e.run(soa, p)\n"
            "    assert_tables_equal(e.tables(DTYPE_HIFI), want, list(got.keys()))\n"
            "    print('LIMITS_OK', n, e.rerun_count())\n") % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VSV_DEBUG="1", VSV_BIG="1"), cwd=root)
    assert r.returncode == 0 and "LIMITS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.strip().splitlines()[-1].split()[1:] == ["1", "1"]


@pytest.mark.parametrize("slim_sort", ["merge", "lsd"])
def test_element_path_on_the_parity_cases(slim_sort):
    """Tables of more than ~1.3 M rows run their sort / cluster / merge / pair stages on 16-byte elements (csrc/slim_path.hip); the
    handle picks the path from the row counts of its previous run, so the small parity cases above never reach it. VSV_BIG=1 forces it
    for every size: the golden fixtures, the synthetic shapes, multi-chromosome inputs, the key hints, the three fuzz families and the
    dense piles go through it in ONE child process (the same assertions against the oracle and the reference's outputs). On that path
    the split stage of read-shaped input works per candidate (cand_info / split_eval_info, csrc/sig_stages.hip) instead of per pair.
    Twice: with the sorts behind the clusterings and the pairing as rank-inside-the-class + merge (sl_merge_sort, the default: keys of up
    to 31 bits, anything its windows cannot decide repeats on the passes), and with the LSD passes for every sort (VSV_SLIM_SORT=lsd)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("golden_contig or golden_reads or synthetic_vs_oracle or multi_tid or tid_hint or random_small or dense_runs or config4_shape or clr_gate or "
           "split_overlap or edge_cases or unaligned_device_views or staged_api")
    env = dict(os.environ, VSV_DEBUG="1", VSV_BIG="1")
    need = 25
    if slim_sort == "lsd":                  # (the passes are also what every first sort and every fallback runs: a shorter list here)
        env["VSV_SLIM_SORT"] = "lsd"
        sel = "golden_contig or synthetic_vs_oracle or multi_tid or random_small or dense_runs or staged_api"
        need = 12
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=800, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    tail = r.stdout.strip().splitlines()[-1]
    assert " passed" in tail and "failed" not in tail and int(tail.split(" passed")[0].split()[-1]) >= need, tail


@pytest.mark.parametrize("form", ["lds", "global"])
def test_first_element_sort_in_position_buckets(form):
    """The first sort of the element path: one counting pass into ~2000 position buckets (a monotone map of the stage-1 key: list pair,
    then cigar / split list, then position up to the max_pos hint) + a complete sort of every bucket, in LDS where it fits
    (sl_bucket_lds) and by the same passes in global memory where it does not (sl_bucket_global: same result, slower; the handle then
    takes the LSD passes for a while and vsv_sort1_slow_count reports it). `global` lowers the LDS limit to 4 elements
    (VSV_SORT1_CAP), so nearly every bucket takes that form. Shapes: HiFi-like reads on a 2 Mb chromosome (cigar and split lists), two
    chromosomes with key hints, and 12000 reads whose insertions start within 60 bp — one bucket holds them all, beyond the real LDS
    limit in both forms. Tables equal the oracle's; a second run (on the passes after a slow bucket) too."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import oracle_run, assert_tables_equal\n"
            "from volcanosv_amd import synth\n"
            "from volcanosv_amd.soa import RecordSoA\n"
            "from volcanosv_amd.engine import Engine, default_params\n"
            "from volcanosv_amd.abi import DTYPE_HIFI\n"
            "p = default_params(DTYPE_HIFI)\n"
            "slow = []\n"
            "def check(soa, name):\n"
            "    st, want = oracle_run(soa, DTYPE_HIFI, p)\n"
            "    assert st == 0 and len(want['calls']) > 50, name\n"
            "    with Engine(0) as e:\n"
            "        for rep in range(2):\n"
            "            e.run(soa, p)\n"
            "            got = e.tables(DTYPE_HIFI)\n"
            "            assert_tables_equal(got, want, list(got.keys()))\n"
            "            if rep == 0: slow.append(e.sort1_slow_count())\n"
            "        assert e.path_counts()[0] >= 2, name\n"
            "t, nq, _ = synth.generate(60000, 'hifi', seed=77, chrom_len=2000000, events_per_record=0.5, site_step=700)\n"
            "a = synth.to_soa(t, nq); a.max_pos = 2200000\n"
            "check(a, 'reads')\n"
            "parts = [synth.generate(20000, 'hifi', seed=78 + c, tid=c, chrom_len=1500000, events_per_record=0.5, site_step=900) for c in range(2)]\n"
            "t2, nq2 = synth.concat([(x[0], x[1]) for x in parts])\n"
            "b = synth.to_soa(t2, nq2); b.max_pos = 1700000; b.n_tids = 2\n"
            "check(b, 'two chromosomes')\n"
            "recs = [(0, 100000 + (i %% 50), 'q%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 500), (1, 60 + i %% 9), (0, 700)]) for i in range(12000)]\n"
            "recs += [(0, 300000 + 997 * i, 'r%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 400), (2, 80 + i %% 5), (0, 300)]) for i in range(300)]\n"
            "recs.sort(key=lambda r: r[1])\n"
            "c = RecordSoA.from_tuples(recs); c.max_pos = 1 << 20\n"
            "check(c, 'pile')\n"
            "print('BUCKETS_OK', *slow)\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ, VSV_DEBUG="1", VSV_BIG="1")
    if form == "global":
        env["VSV_SORT1_CAP"] = "4"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "BUCKETS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    slow = [int(x) for x in r.stdout.strip().splitlines()[-1].split()[1:]]
    assert slow[2] == 1                                   # the pile: one bucket beyond LDS whatever the limit
    assert slow[:2] == ([1, 1] if form == "global" else [0, 0])


def test_rank_and_merge_sort_gives_up_on_a_pile_inside_one_shift():
    """sl_merge_sort ranks an element against the slots whose anchors lie within one shift of its key, from an LDS window of 2048 + 2 x 128
    slots. 8000 reads whose insertions all start within 90 bp — one cluster per haplotype whose representative (the first longest
    member, H:236-247) lies 89 bp behind its seed, with thousands of slots in between — is more than a window can decide: the run
    raises ERRB_MERGE_FALLBACK and repeats on the LSD passes (one repetition; the handle then stays on the passes for a while). Tables
    equal the oracle's either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_gpu_parity import oracle_run, assert_tables_equal\n"
            "from volcanosv_amd.soa import RecordSoA\n"
            "from volcanosv_amd.engine import Engine, default_params\n"
            "from volcanosv_amd.abi import DTYPE_HIFI\n"
            "recs = [(0, 1000, 'q%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 100 + (i // 2) %% 90), (1, 150 if (i // 2) %% 90 == 89 else 100), (0, 2000)]) for i in range(8000)]\n"
            "recs += [(0, 50000 + 700 * i, 'r%%d_hp%%d' %% (i, 1 + i %% 2), 60, False, [(0, 300), (2, 60 + i %% 5), (0, 300)]) for i in range(300)]\n"
            "soa = RecordSoA.from_tuples(recs)\n"
            "soa.max_pos = 1 << 20\n"
            "p = default_params(DTYPE_HIFI)\n"
            "st, want = oracle_run(soa, DTYPE_HIFI, p)\n"
            "assert st == 0 and len(want['calls']) > 100\n"
            "with Engine(0) as e:\n"
            "    e.run(soa, p)\n"
            "    got = e.tables(DTYPE_HIFI)\n"
            "    assert_tables_equal(got, want, list(got.keys()))\n"
            "    n = e.rerun_count()\n"
            "    e.run(soa, p)\n"
            "    assert_tables_equal(e.tables(DTYPE_HIFI), want, list(got.keys()))\n"
            "    print('MERGE_FALLBACK_OK', n, e.rerun_count())\n") % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VSV_DEBUG="1", VSV_BIG="1"), cwd=root)
    assert r.returncode == 0 and "MERGE_FALLBACK_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.strip().splitlines()[-1].split()[1:] == ["1", "1"]
