"""Shared test helpers: fixture loading and table -> reference-list conversion."""
import json
import os

import numpy as np

from volcanosv_amd import sigtable
from volcanosv_amd.soa import RecordSoA
from volcanosv_amd.abi import DTYPE_BY_NAME, DTYPE_READS, M_DEL, M_HP2, M_SPLIT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        doc = json.load(f)
    chroms = doc["expected"]["chroms"]
    tid = {c: i for i, c in enumerate(chroms)}
    recs = [(tid[r[0]], r[1], r[2], r[3], r[4], r[5]) for r in doc["records"]]
    soa = RecordSoA.from_tuples(recs, tid_names=chroms)
    return doc, soa, DTYPE_BY_NAME[doc["dtype"]]


def rows(soa, table, dtype=None, where=None):
    out = []
    for s in table:
        if where is None or where(s):
            out.append(sigtable.sig_fields(soa, s, dtype))
    return out


def sel(tid=None, hap=None, is_del=None, split=None):
    def f(s):
        if tid is not None and int(s["tid"]) != tid:
            return False
        if hap is not None and bool(s["meta"] & M_HP2) != (hap == 2):
            return False
        if is_del is not None and bool(s["meta"] & M_DEL) != is_del:
            return False
        if split is not None and bool(s["meta"] & M_SPLIT) != split:
            return False
        return True
    return f


def compare_contig_tables(doc, soa, tabs):
    """Asserts every stage table equals the reference outputs stored in a contig_* fixture."""
    exp = doc["expected"]
    for t, chrom in enumerate(exp["chroms"]):
        pc = exp["per_chrom"][chrom]
        for hp in (1, 2):
            c1 = pc["cluster1_hp%d" % hp]
            assert rows(soa, tabs["cluster1"], where=sel(t, hp, True, False)) == c1["del_cigar"], (chrom, hp, "del_cigar")
            assert rows(soa, tabs["cluster1"], where=sel(t, hp, False, False)) == c1["ins_cigar"], (chrom, hp, "ins_cigar")
            assert rows(soa, tabs["cluster1"], where=sel(t, hp, True, True)) == c1["del_split"], (chrom, hp, "del_split")
            assert rows(soa, tabs["cluster1"], where=sel(t, hp, False, True)) == c1["ins_split"], (chrom, hp, "ins_split")
            # raw split signatures in emission order: list of [del_list, ins_list] per pair call
            flat = [s for pair in pc["split_raw_hp%d" % hp] for lst in pair for s in lst]
            assert rows(soa, tabs["split"], where=sel(t, hp)) == flat, (chrom, hp, "split_raw")
            assert rows(soa, tabs["merged"], where=sel(t, hp)) == pc["merged_hp%d" % hp], (chrom, hp, "merged")
        got = [sigtable.call_fields(soa, c, tabs["merged"]) for c in tabs["calls"] if int(c["sig"]["tid"]) == t]
        assert got == pc["paired"], (chrom, "paired")


# ---- randomized small inputs (tests/test_gpu_parity.py, tools/fuzz_case.py) ---------------------------------------------------
_SOAK = 100003 * int(os.environ.get("VSV_FUZZ_SEED", "0"))
def _fuzz_flags_mapq(soa, rng, strands=True):
    from volcanosv_amd.abi import F_HP1, F_HP2
    k = soa.n_records
    soa.mapq = np.where(rng.random(k) < 0.2, rng.integers(0, 61, k), soa.mapq).astype(np.uint8)
    fl = soa.flag.copy()
    flip = rng.random(k) < 0.15
    fl[flip] = (fl[flip] & ~np.uint8(F_HP1 | F_HP2)) | rng.choice(np.array([0, F_HP1, F_HP2, F_HP1 | F_HP2], np.uint8), int(flip.sum()))
    if strands:
        fl ^= (rng.random(k) < 0.1).astype(np.uint8)
    soa.flag = fl


def fuzz_case_basic(case, rng):
    """1-400 records on a 60-300 kb contig (signatures collide, cluster and pair all the time), random haplotype tags / mapq /
    strands / data type / thresholds, now and then an =/X/N op. Returns (soa, dtype, params)."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_CLR, DTYPE_HIFI, DTYPE_ONT, DTYPE_READS
    from volcanosv_amd.engine import default_params
    shape = ("hifi", "ont")[int(rng.integers(0, 2))]
    n = int(rng.integers(1, 401)) if shape == "hifi" else int(rng.integers(1, 81))
    t, nq, _ = synth.generate(n, shape, seed=5000 + case + _SOAK, chrom_len=int(rng.integers(60_000, 300_000)) + 40_000,
                              events_per_record=float(rng.choice([0.0, 0.05, 0.5, 2.0])), site_step=int(rng.choice([200, 1000, 5000])))
    soa = synth.to_soa(t, nq)
    _fuzz_flags_mapq(soa, rng)
    if rng.random() < 0.2 and soa.n_ops:                              # one M becomes '=', 'X' or 'N'
        cig = soa.cigar.copy()
        ms = np.flatnonzero((cig & 15) == 0)
        if len(ms):
            j = int(ms[rng.integers(0, len(ms))])
            cig[j] = (cig[j] & ~np.uint32(15)) | np.uint32(rng.choice([7, 8, 3]))
            soa.cigar = cig
    dtype = (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS)[int(rng.integers(0, 4))]
    p = default_params(dtype)
    p.min_svlen = int(rng.choice([30, 30, 10, 50]))
    p.cluster_shift = int(rng.choice([100, 100, 10, 1000]))
    p.pair_shift = int(rng.choice([200, 200, 0, 2000]))
    p.min_cigar_mapq = int(rng.choice([50, 50, 0, 60]))
    return soa, dtype, p


def fuzz_case_defects(case, rng):
    """1-3 chromosomes, every threshold of vsv_params varied, and at most one planted defect: an =/X/N op (reference_end assert,
    H:396), a record without M ops (CLR: ZeroDivisionError, C:61/70), a clipped end that grows (split mates of unequal read
    length, H:331). Returns (soa, dtype, params)."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_CLR, DTYPE_HIFI, DTYPE_ONT, DTYPE_READS
    from volcanosv_amd.engine import default_params
    parts = []
    n_t = int(rng.choice([1, 1, 2, 3]))
    shape = ("hifi", "ont")[int(rng.integers(0, 2))]
    for tid in range(n_t):
        n = int(rng.integers(1, 201)) if shape == "hifi" else int(rng.integers(1, 41))
        t, nq, _ = synth.generate(n, shape, seed=9000 + 7 * case + tid + _SOAK, tid=tid, chrom_len=int(rng.integers(60_000, 200_000)) + 40_000,
                                  events_per_record=float(rng.choice([0.05, 0.5, 2.0])), site_step=int(rng.choice([200, 1000])))
        parts.append((t, nq))
    t, nq = synth.concat(parts) if n_t > 1 else parts[0]
    soa = synth.to_soa(t, nq)
    k = soa.n_records
    _fuzz_flags_mapq(soa, rng, strands=False)
    defect = rng.choice(["none", "none", "none", "op", "no_m", "readlen", "clip"])
    cig = soa.cigar.copy()
    if defect == "op":
        ms = np.flatnonzero((cig & 15) == 0)
        if len(ms):
            j = int(ms[rng.integers(0, len(ms))])
            cig[j] = (cig[j] & ~np.uint32(15)) | np.uint32(rng.choice([7, 8, 3]))
    elif defect == "no_m":
        r = int(rng.integers(0, k))
        seg = cig[int(soa.cigar_off[r]):int(soa.cigar_off[r + 1])]
        seg[(seg & 15) == 0] |= np.uint32(7)
    elif defect in ("readlen", "clip"):
        sc = np.flatnonzero(((cig & 15) == 4) | ((cig & 15) == 5))
        if len(sc):
            j = int(sc[rng.integers(0, len(sc))])
            cig[j] += np.uint32(16 * int(rng.integers(1, 50)))
    soa.cigar = cig
    dtype = (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS)[int(rng.integers(0, 4))]
    p = default_params(dtype)
    p.min_svlen = int(rng.choice([30, 30, 10, 50]))
    p.cluster_shift = int(rng.choice([100, 100, 10, 1000]))
    p.pair_shift = int(rng.choice([200, 200, 0, 2000]))
    p.min_cigar_mapq = int(rng.choice([50, 50, 0, 60]))
    p.min_split_mapq = int(rng.choice([p.min_split_mapq, 0, 60]))
    p.max_split_svlen = int(rng.choice([50000, 50000, 500]))
    p.enable_split = int(rng.random() < 0.9)
    p.pair_window = int(rng.choice([1000, 1000, 100, 5000]))
    return soa, dtype, p


def fuzz_case_collectors(case, rng):
    """The two collector op tables (svim-asm analyze_cigar_indel, SV/SVIM_intra.py:8-30; sig_extract parse_read, SE:438-493): records
    with =/X/N/P ops, unmapped / secondary / supplementary / host-skipped flags, every merge threshold varied."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_CUTESV, DTYPE_SVIM, F_SKIP
    from volcanosv_amd.engine import default_params
    shape = ("hifi", "ont")[int(rng.integers(0, 2))]
    n = int(rng.integers(1, 301)) if shape == "hifi" else int(rng.integers(1, 61))
    t, nq, _ = synth.generate(n, shape, seed=13000 + case + _SOAK, chrom_len=int(rng.integers(60_000, 300_000)) + 40_000,
                              events_per_record=float(rng.choice([0.05, 0.5, 3.0])), site_step=int(rng.choice([200, 1000])))
    soa = synth.to_soa(t, nq)
    k = soa.n_records
    soa.mapq = np.where(rng.random(k) < 0.3, rng.integers(0, 61, k), soa.mapq).astype(np.uint8)
    fl = soa.flag.copy()
    for bit, prob in ((2, 0.1), (16, 0.1), (32, 0.05), (F_SKIP, 0.1), (1, 0.3)):     # SUPP, SECONDARY, UNMAPPED, SKIP, REVERSE
        fl ^= (rng.random(k) < prob).astype(np.uint8) * np.uint8(bit)
    soa.flag = fl
    cig = soa.cigar.copy()
    ms = np.flatnonzero((cig & 15) == 0)
    sel = ms[rng.random(len(ms)) < 0.1]
    cig[sel] = (cig[sel] & ~np.uint32(15)) | rng.choice(np.array([7, 8, 3, 6], np.uint32), len(sel))
    soa.cigar = cig
    dtype = (DTYPE_SVIM, DTYPE_CUTESV)[int(rng.integers(0, 2))]
    p = default_params(dtype)
    p.min_svlen = int(rng.choice([p.min_svlen, 5, 10, 100]))
    p.min_cigar_mapq = int(rng.choice([p.min_cigar_mapq, 0, 50]))
    p.merge_ins_threshold = int(rng.choice([100, 0, 30, 1000]))
    p.merge_del_threshold = int(rng.choice([0, 40, 1000]))
    return soa, dtype, p


def fuzz_case_dense(case, rng):
    """Contig-shaped records (thousands of ops each) piled on a short chromosome at recurring sites: sorted lists with runs of tens
    to thousands of rows inside the cluster / pairing distance, which is what the wave-cooperative cluster_long / pair_long
    kernels exist for; cluster and pairing distances varied."""
    from volcanosv_amd import synth
    from volcanosv_amd.abi import DTYPE_CLR, DTYPE_HIFI, DTYPE_ONT
    from volcanosv_amd.engine import default_params
    n = int(rng.integers(100, 500))
    t, nq, _ = synth.generate(n, "contig", seed=17000 + case + _SOAK, chrom_len=int(rng.choice([200_000, 300_000, 1_000_000])),
                              site_step=int(rng.choice([200, 1000])))
    soa = synth.to_soa(t, nq)
    dtype = (DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR)[int(rng.integers(0, 3))]
    p = default_params(dtype)
    p.cluster_shift = int(rng.choice([100, 300, 1000]))
    p.pair_shift = int(rng.choice([200, 1000, 2000]))
    p.pair_window = int(rng.choice([1000, 300, 5000]))
    return soa, dtype, p


FUZZ = {"dense": (fuzz_case_dense, 9, 12), "basic": (fuzz_case_basic, 20250403, 240), "defects": (fuzz_case_defects, 77, 300), "collectors": (fuzz_case_collectors, 5, 200)}


def fuzz_cases(kind, upto=None):
    """Yields (case, soa, dtype, params) for the named family, in the fixed order the tests use."""
    make, seed, count = FUZZ[kind]
    seed += int(os.environ.get("VSV_FUZZ_SEED", "0"))          # soak runs: other seeds, VSV_FUZZ_SCALE times as many cases
    count *= int(os.environ.get("VSV_FUZZ_SCALE", "1"))
    rng = np.random.default_rng(seed)
    for case in range(count if upto is None else upto + 1):
        yield (case,) + make(case, rng)
