"""Randomized parity of the post-filter rows on small, dense inputs: every call collides with many
signatures, duplicates and zero-length entries occur, every threshold is varied. HIP entry point vs the oracle's literal loops."""
import os

import numpy as np
import pytest

SEED = int(os.environ.get("VSV_FUZZ_SEED", "0"))       # soak runs: other seeds, VSV_FUZZ_SCALE times as many cases
SCALE = int(os.environ.get("VSV_FUZZ_SCALE", "1"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from volcanosv_amd.engine import Engine
    with Engine(0) as e:
        yield e


def test_support_join_random(eng):
    """FP_filter_v1.eval_sig (FP:106-123)."""
    from oracle import oracle
    rng = np.random.default_rng(101 + SEED)
    for case in range(80 * SCALE):
        nc, ns = int(rng.integers(1, 200)), int(rng.integers(0, 2000))
        span = int(rng.choice([2000, 20000, 200000]))
        cpos, clen = rng.integers(0, span, nc), rng.integers(1, 600, nc)
        spos, slen = np.sort(rng.integers(0, span, ns)), rng.integers(1, 600, ns)
        kw = dict(max_comp_svlen=int(rng.choice([250, 100, 1 << 30])), max_dist=int(rng.choice([1000, 100, 5000])),
                  max_shift=int(rng.choice([500, 50, 0, 5000])), min_size_sim=float(rng.choice([0.5, 0.0, 0.9, 1.0])))
        st, want = oracle.run_support(cpos, clen, spos, slen, oracle.default_support_params(**kw))
        assert st == 0
        got = eng.support_join(cpos, clen, spos, slen, eng.support_params(**kw))
        assert np.array_equal(got, want), (case, kw)


def test_signature_coverage_random(eng):
    """calc_ins_call_cov / calc_del_call_cov (calculate_signature_support.py:81-125, 138-280)."""
    from oracle import oracle
    rng = np.random.default_rng(102 + SEED)
    for case in range(80 * SCALE):
        nc, ns = int(rng.integers(1, 150)), int(rng.integers(0, 1500))
        span = int(rng.choice([3000, 30000, 300000]))
        fl = int(rng.choice([1000, 0, 10, 5000]))
        cpos = rng.integers(0, span, nc)
        spos = np.sort(rng.integers(0, span, ns))
        slen = rng.integers(0, 800, ns)
        if ns and rng.random() < 0.5:
            slen[rng.integers(0, ns, max(1, ns // 50))] = int(rng.integers(5000, 50000))     # a few very long signatures
        st, want = oracle.run_cov_ins(cpos, spos, slen, fl)
        assert st == 0
        assert np.array_equal(eng.support_cov_ins(cpos, spos, slen, fl), want), ("ins", case, fl)
        clen = rng.integers(1, 3000, nc)
        st, want, _ = oracle.run_cov_del(cpos, cpos + clen, spos, spos + slen, -slen, fl)
        assert st == 0
        assert np.array_equal(eng.support_cov_del(cpos, cpos + clen, spos, spos + slen, -slen, fl), want), ("del", case, fl)


def test_redundancy_pairs_random(eng):
    """remove_redundancy.py match_del_chr / match_ins_chr (RR:75-125): windowed pairs, DEL overlap / size tests, INS edit distance."""
    from oracle import oracle
    rng = np.random.default_rng(103 + SEED)
    for case in range(60 * SCALE):
        n = int(rng.integers(1, 120))
        span = int(rng.choice([500, 5000, 50000]))
        pos = np.sort(rng.integers(0, span, n))
        kw = dict(dist_thresh=int(rng.choice([500, 50, 5000])), dist_thresh_del=int(rng.choice([500, 50, 5000])),
                  overlap_thresh=float(rng.choice([0.5, 0.0, 0.9])), size_sim_thresh=float(rng.choice([0.5, 0.1, 0.95])),
                  size_sim_thresh_del=float(rng.choice([0.5, 0.1, 0.95])), seq_sim_thresh=float(rng.choice([0.5, 0.3, 0.8])))
        svlen = rng.integers(30, 3000, n)                       # |len(REF) - len(ALT)|, as remove_redundancy.py passes it
        st, want = oracle.run_redundancy_pairs(True, pos, svlen, params=oracle.default_redundancy_params(**kw))
        assert st == 0
        assert np.array_equal(eng.redundancy_pairs(True, pos, svlen, params=eng.redundancy_params(**kw)), want), ("del", case, kw)
        # INS: ALT strings from a few templates with point edits, so that similar and dissimilar pairs both occur
        templ = [rng.integers(1, 5, int(rng.integers(30, 400))).astype(np.uint8) for _ in range(4)]
        seqs = []
        for _ in range(n):
            s = templ[int(rng.integers(0, 4))].copy()
            k = int(rng.integers(0, max(1, len(s) // 3)))
            s[rng.integers(0, len(s), k)] = rng.integers(1, 5, k)
            cut = int(rng.integers(0, max(1, len(s) // 4)))
            seqs.append(s[cut:])
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        blob = np.concatenate(seqs)
        ilen = np.array([len(s) for s in seqs])
        st, want = oracle.run_redundancy_pairs(False, pos, ilen, blob, off, oracle.default_redundancy_params(**kw))
        assert st == 0
        assert np.array_equal(eng.redundancy_pairs(False, pos, ilen, blob, off, eng.redundancy_params(**kw)), want), ("ins", case, kw)


def test_cutesv_split_random(eng):
    """sig_extract.py analysis_split_read (SE:193-319) on random segment lists: 2-9 segments per read on 1-2 chromosomes, both
    strands, overlapping and gapped on the read and on the reference."""
    from oracle import oracle
    from volcanosv_amd.sig_extract import SplitSegments
    rng = np.random.default_rng(104 + SEED)
    for case in range(60 * SCALE):
        reads = []
        for r in range(int(rng.integers(1, 80))):
            ql = int(rng.integers(2000, 30000))
            k = int(rng.integers(2, 10))
            segs = []
            if rng.random() < 0.7:
                # a chain: segments tile the read in order, on one chromosome and strand, with a read-side gap (inserted
                # sequence), a reference-side gap (deleted sequence), a small overlap or nothing between neighbours
                rev = int(rng.random() < 0.3)
                x, ref = int(rng.integers(0, 200)), int(rng.integers(10000, 200000))
                for j in range(k):
                    ln = int(rng.integers(100, max(101, ql // k)))
                    segs.append([x, x + ln, ref, ref + ln + int(rng.integers(-20, 20)), 0, rev])
                    kind = rng.choice(["none", "ins", "del", "olap", "both"])
                    x += ln + (int(rng.integers(30, 3000)) if kind in ("ins", "both") else -int(rng.integers(0, 40)) if kind == "olap" else 0)
                    ref = segs[-1][3] + (int(rng.integers(30, 5000)) if kind in ("del", "both") else -int(rng.integers(0, 40)) if kind == "olap" else 0)
                ql = max(ql, x + 10)
            else:
                cuts = np.sort(rng.integers(0, ql, 2 * k))
                ref = int(rng.integers(10000, 200000))
                for j in range(k):
                    a, b = int(cuts[2 * j]), int(cuts[2 * j + 1]) + 1
                    if rng.random() < 0.3 and j > 0:
                        a = max(0, a - int(rng.integers(0, 300)))                    # overlap with the previous segment on the read
                    rs = ref + int(rng.integers(-2000, 6000))
                    re_ = max(rs + 1, rs + (b - a) + int(rng.integers(-50, 50)))
                    segs.append([a, b, rs, re_, int(rng.integers(0, 2)) if rng.random() < 0.2 else 0, int(rng.random() < 0.25)])
                    ref = re_ if rng.random() < 0.8 else ref
            if rng.random() < 0.5:
                segs = [segs[i] for i in rng.permutation(len(segs))]
            reads.append((r, ql, segs))
        seg = SplitSegments(reads)
        for size, parts, sv in ((100000, 7, 30), (-1, -1, 30), (2000, 3, 50), (100000, 7, 1)):
            want = oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, sv, size, parts)
            got = eng.cutesv_split(seg, seg.read_len, seg.read_rec, sv, size, parts)
            assert np.array_equal(got, want), (case, size, parts, sv, len(got), len(want))


def test_bnd_segments_and_pairing_random(eng):
    """Complex_SV breakend branch (SV/SVIM_inter.py:62-258, SV/SVIM_COMBINE.py:15-32, 143-161) on random multi-segment reads:
    2-6 segments over 1-4 contigs whose names sort differently as strings ('chr10' < 'chr2'), both strands and haplotypes,
    breakpoints near contig ends, partitions from singletons to more than ten members, every tolerance varied."""
    from oracle import oracle
    from volcanosv_amd import bnd
    rng = np.random.default_rng(105 + SEED)
    n_cand = n_calls = 0
    for case in range(80 * SCALE):
        nt = int(rng.integers(1, 5))
        names = list(rng.permutation(["chr1", "chr2", "chr10", "chrX", "chr21"])[:nt])
        contigs = [(str(nm), int(rng.integers(50_000, 400_000))) for nm in names]
        sites = [(int(rng.integers(0, nt)), int(rng.integers(0, 50_000))) for _ in range(int(rng.integers(1, 8)))]
        reads = []
        for r in range(int(rng.integers(1, 120))):
            L = int(rng.integers(5000, 40000))
            k = int(rng.integers(2, 7))
            cuts = np.sort(rng.choice(np.arange(100, L - 100), k - 1, replace=False))
            bounds = [0] + [int(c) for c in cuts] + [L]
            segs = []
            for j in range(k):
                qa, qb = bounds[j], bounds[j + 1]
                qa = max(0, qa + int(rng.integers(-80, 80))) if j else 0
                if qb <= qa:
                    qb = qa + 1
                t, base = sites[int(rng.integers(0, len(sites)))]
                clen = contigs[t][1]
                rs = int(np.clip(base + int(rng.integers(-400, 400)), 0, clen - 2))
                re_ = int(min(clen, rs + max(1, qb - qa + int(rng.integers(-30, 30)))))
                segs.append([t, rs, re_, qa, min(qb, L), L, int(rng.random() < 0.4)])
            reads.append({"hap": 1 + int(rng.random() < 0.5), "name": "PS%d_hp_r%d" % (case, r), "segs": segs})
        reads.sort(key=lambda x: x["hap"])
        seg = bnd.SegmentSoA(reads, contigs)
        kw = dict(query_gap_tolerance=int(rng.choice([50, 0, 500])), query_overlap_tolerance=int(rng.choice([50, 0, 500])),
                  reference_gap_tolerance=int(rng.choice([50, 0, 1000])), reference_overlap_tolerance=int(rng.choice([50, 0, 1000])),
                  max_sv_size=int(rng.choice([100000, 1000])), min_sv_size=int(rng.choice([40, 1, 200])),
                  partition_max_distance=int(rng.choice([1000, 100, 10000])), pair_distance=int(rng.choice([900, 0, 5000])),
                  max_partition=int(rng.choice([10, 3, 16])))
        po, pg = oracle.default_bnd_params(), eng._bnd_params()
        for name, v in kw.items():
            setattr(po, name, v)
            setattr(pg, name, v)
        ocand, ocalls = oracle.run_bnd(seg, po)
        cand, calls = eng.bnd(seg, pg)
        assert np.array_equal(cand, ocand), (case, kw, len(cand), len(ocand))
        assert np.array_equal(calls, ocalls), (case, kw, len(calls), len(ocalls))
        n_cand += len(cand)
        n_calls += len(calls)
    assert n_cand > 1000 and n_calls > 500
    pg.max_partition = 17                      # the pairing kernel holds a partition in registers: at most 16 members
    with pytest.raises(Exception, match="max_partition"):
        eng.bnd(seg, pg)
