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
        exp = mg.run_contig(style, recs, stable=True, dumps=True)
        soa = _soa(recs, exp["chroms"])
        st, tabs = oracle.run(soa, dtype=DTYPE_BY_NAME[style])
        assert st == 0, (style, seed)
        compare_contig_tables({"expected": exp}, soa, tabs)
        # the signature dump files of <out>/signature/ (write_sig_cigar / write_sig_split), byte for byte
        from volcanosv_amd import sigtable
        for t_, chrom_ in enumerate(exp["chroms"]):
            got_d = sigtable.signature_dump_texts(soa, tabs["cigar"], tabs["split"], chrom_, tid=t_)
            assert got_d == exp["per_chrom"][chrom_]["dumps"], (style, seed, chrom_)
        # the VCF text of write_vcf (H:678-714): REF / ALT from the same synthetic sequences, ids, INFO, genotype
        import test_vcf_bam as tv
        from volcanosv_amd import vcf
        dc_contig = tv.fixture_inputs({"records": recs})
        for t, chrom in enumerate(exp["chroms"]):
            calls = tabs["calls"][tabs["calls"]["sig"]["tid"] == t]
            lines = vcf.vcf_lines(soa, calls, tabs["merged"], tv.synth_seq(chrom), dc_contig)
            assert tv.digest(lines) == [list(r) for r in mg.jsonable(exp["per_chrom"][chrom]["vcf"])], (style, seed, chrom)
        n_calls += len(tabs["calls"])
    assert n_calls > 100


def test_reads_path_on_fresh_random_inputs():
    mg = _mg()
    rng = np.random.default_rng(4)
    n_rows = 0
    for seed in range(4000, 4006):
        recs = mg.make_read_records(seed, n=int(rng.integers(50, 500)), n_chrom=int(rng.integers(1, 3)), tie_rich=bool(rng.random() < 0.5))
        exp = mg.run_reads(recs, stable=True, dumps=True)
        soa = _soa(recs, exp["chroms"])
        st, tabs = oracle.run(soa, dtype=DTYPE_READS)
        assert st == 0
        from volcanosv_amd import sigtable
        for t, chrom in enumerate(exp["chroms"]):
            got = rows(soa, tabs["reads"], dtype=DTYPE_READS, where=sel(t))
            assert got == exp["per_chrom"][chrom]["merged"], (seed, chrom)
            assert sigtable.reads_dump_texts(soa, tabs["cigar"], tabs["split"], chrom, tid=t) == exp["per_chrom"][chrom]["dumps"], (seed, chrom)
            text = "".join(sigtable.reads_sig_lines(soa, tabs["reads"][tabs["reads"]["tid"] == t]))
            assert text == "".join("\t".join(str(x) for x in r) + "\n" for r in exp["per_chrom"][chrom]["merged"]), (seed, chrom)
            n_rows += len(got)
    assert n_rows > 100


def _golden_module(name):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    return __import__(name)


def test_fp_filter_eval_sig_on_fresh_random_inputs():
    """FP_filter_v1.eval_sig (FP:106-123) with its thresholds as arguments: the oracle's literal loop on new lists."""
    mg = _golden_module("make_golden_fp")
    ns = mg.load_functions()
    rng = np.random.default_rng(6)
    for seed in range(50, 90):
        calls, sigs = mg.make_lists(seed, int(rng.integers(1, 300)), int(rng.integers(0, 3000)), int(rng.choice([20_000, 300_000, 3_000_000])),
                                    boundary=bool(rng.random() < 0.5))
        prm = (int(rng.choice([1000, 100, 5000])), int(rng.choice([250, 100, 1200])), int(rng.choice([500, 50, 2000])), float(rng.choice([0.5, 0.3, 0.0, 1.0])))
        want = [int(x) for x in ns["eval_sig"](calls, sigs, *prm)]
        p = oracle.default_support_params(max_dist=prm[0], max_comp_svlen=prm[1], max_shift=prm[2], min_size_sim=prm[3])
        st, got = oracle.run_support([c[2] for c in calls], [c[3] for c in calls], [s[2] for s in sigs], [s[3] for s in sigs], p)
        assert st == 0 and got.tolist() == want, (seed, prm)


def test_signature_coverage_on_fresh_random_inputs(tmp_path):
    """calculate_signature_support.py calc_ins_call_cov / calc_del_call_cov (CS:81-125, 138-280) through its own load_vcf / load_sig."""
    mg = _golden_module("make_golden_cov")
    rng = np.random.default_rng(7)
    for seed in range(60, 72):
        ties = bool(rng.random() < 0.5)
        flank = int(rng.choice([1000, 1000, 200, 5000]))
        ns = mg.load_functions(stable=ties, flanking=flank)
        vcf, sig = mg.make_text(seed, int(rng.integers(30, 300)), int(rng.integers(50, 2500)), int(rng.choice([100_000, 600_000])), ties)
        vp = tmp_path / ("c%d.vcf" % seed)
        vp.write_text("".join(vcf))
        (tmp_path / "INS.sigs").write_text("".join(sig["INS"]))
        (tmp_path / "DEL.sigs").write_text("".join(sig["DEL"]))
        s_ins, s_del = ns["load_sig"](str(tmp_path / "INS.sigs"), "INS"), ns["load_sig"](str(tmp_path / "DEL.sigs"), "DEL")
        c_ins, c_del = ns["load_vcf"](str(vp), "INS"), ns["load_vcf"](str(vp), "DEL")
        if "chr1" in c_ins and "chr1" in s_ins:
            want = ns["calc_ins_call_cov"](c_ins["chr1"], s_ins["chr1"])
            pos = np.array(list(want.keys()))
            sp = np.array([int(l.split()[2]) for l in sig["INS"]])
            sl = np.array([int(l.split()[3]) for l in sig["INS"]])
            o = np.argsort(sp, kind="stable")
            st, cov = oracle.run_cov_ins(pos, sp[o], sl[o], flank)
            assert st == 0 and cov.tolist() == [int(v) for v in want.values()], (seed, "ins")
        if "chr1" in c_del and "chr1" in s_del:
            want = {tuple(int(x) for x in k): int(v) for k, v in ns["calc_del_call_cov"](c_del["chr1"], s_del["chr1"]).items()}
            sp = np.array([int(l.split()[2]) for l in sig["DEL"]])
            sl = np.array([int(l.split()[3]) for l in sig["DEL"]])
            o = np.argsort(sp, kind="stable")
            sp, sl = sp[o], sl[o]
            cs = np.array([c[0] for c in c_del["chr1"]])
            ce = np.array([c[1] for c in c_del["chr1"]])
            st, cov, has = oracle.run_cov_del(cs, ce, sp, sp + sl, -sl, flank)
            assert st == 0
            got = {(int(a), int(b)): int(c) for a, b, c, h in zip(cs, ce, cov, has) if h}
            assert got == want, (seed, "del", flank)


def test_remove_redundancy_links_on_fresh_random_inputs(tmp_path):
    """remove_redundancy.py match_del_chr / match_ins_chr (RR:127-197) with their thresholds as arguments (edlib stood in for by the
    Levenshtein DP, networkx real): the link lists of the host mirror over the oracle's pair test."""
    from test_remove_redundancy import OracleEngine
    from volcanosv_amd import remove_redundancy as rr
    mg = _golden_module("make_golden_redundancy")
    ns = mg.load_functions()
    rng = np.random.default_rng(8)
    eng = OracleEngine()
    n_links = 0
    for seed in range(30, 36):
        lines = mg.make_vcf(seed, int(rng.integers(10, 60)))
        vp = tmp_path / ("r%d.vcf" % seed)
        vp.write_text("".join(lines))
        del_sig, ins_sig, vcf_dc, header = ns["vcf_to_sig"](str(vp))
        d2, i2, _, _ = rr.vcf_to_sig(str(vp))
        kw = dict(dist_thresh=int(rng.choice([500, 100, 3000])), dist_thresh_del=int(rng.choice([3000, 300])), overlap_thresh=float(rng.choice([0.0, 0.5])),
                  size_sim_thresh=float(rng.choice([0.5, 0.2, 0.9])), size_sim_thresh_del=float(rng.choice([0.1, 0.6])), seq_sim_thresh=float(rng.choice([0.5, 0.3, 0.8])))
        p = eng.redundancy_params(**kw)
        for chrom in sorted({s[0] for s in del_sig}):
            want = ns["match_del_chr"]([s for s in del_sig if s[0] == chrom], kw["dist_thresh_del"], kw["size_sim_thresh_del"], kw["overlap_thresh"])
            got = rr.match_chr([s for s in d2 if s[0] == chrom], True, eng, p)
            assert [list(l) for l in got] == [list(l) for l in want], (seed, chrom, "del", kw)
            n_links += len(want)
        for chrom in sorted({s[0] for s in ins_sig}):
            want = ns["match_ins_chr"]([s for s in ins_sig if s[0] == chrom], kw["dist_thresh"], kw["size_sim_thresh"], kw["seq_sim_thresh"])
            got = rr.match_chr([s for s in i2 if s[0] == chrom], False, eng, p)
            assert [list(l) for l in got] == [list(l) for l in want], (seed, chrom, "ins", kw)
            n_links += len(want)
    assert n_links > 50


def test_breakend_branch_on_fresh_random_inputs():
    """svim-asm analyze_read_segments / CandidateBreakend / form_partitions / pair_haplotypes_breakends (real scipy linkage) on new
    read sets: per-read candidates, paired calls and the VCF text of the oracle's breakend branch."""
    import json
    from test_bnd_oracle import check_against_golden
    from volcanosv_amd import bnd
    mg = _golden_module("make_golden_bnd")
    svim = mg.load_svim()
    rng = np.random.default_rng(9)
    for seed in range(100, 108):
        reads = mg.make_reads(seed, n_events=int(rng.integers(20, 200)), dense=int(rng.integers(1, 5)))
        doc, _, _ = mg.make_doc(reads, svim)
        doc = json.loads(json.dumps(doc))                       # the same normalisation a committed fixture goes through
        seg = bnd.SegmentSoA(doc["reads"], [tuple(c) for c in doc["contigs"]])
        cand, calls = oracle.run_bnd(seg)
        check_against_golden(doc, seg, cand, calls)


def test_sig_extract_parse_read_on_fresh_random_inputs():
    """sig_extract.py parse_read (SE:438-493) on new reads: the CIGAR candidates with random length / merge thresholds (oracle tables
    + the host mirror's text step) and the split-read candidates of reads with SA tags."""
    import test_sig_extract as ts
    from volcanosv_amd import sig_extract
    mg = _golden_module("make_golden_sigextract")
    ns = mg.load_functions()
    rng = np.random.default_rng(10)
    n_cig = n_spl = 0
    for rep in range(6):
        reads = mg.make_plain_reads(rng, int(rng.integers(20, 150)))
        siglen, mdel, mins = int(rng.choice([10, 5, 30, 50])), int(rng.choice([0, 50, 500])), int(rng.choice([100, 20, 0, 1000]))
        exp = [mg.norm(ns["parse_read"](mg.FakeRead(d), "chr1", 30, 20, 7, 500, siglen, mdel, mins, 100000)) for d in reads]
        soa = ts.build_soa(reads)
        st, tabs = oracle.run(soa, params=sig_extract.params(min_mapq=20, min_siglength=siglen, merge_del_threshold=mdel, merge_ins_threshold=mins))
        assert st == 0
        got = sig_extract.cigar_candidates(soa, tabs["raw"], tabs["cigar"], lambda rec: reads[rec]["seq"], "chr1")
        for i, w in enumerate(ts.cigar_expected(exp)):
            assert got.get(i, []) == w, (rep, i, siglen, mdel, mins)
            n_cig += len(w)
    for rep in range(4):
        reads = mg.make_split_reads(rng, int(rng.integers(30, 200)))
        exp = [mg.norm(ns["parse_read"](mg.FakeRead(d), "chr1", 30, 20, 7, 500, 10, 0, 100, 100000)) for d in reads]
        doc = {"cases": {"split": {"reads": reads, "expected": exp}}}
        rd, soa, seg, names = ts.split_inputs(doc)
        rows = oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec, 30, 100000, 7)
        got = sig_extract.split_candidates(soa, rows, lambda rec: rd[rec]["seq"], lambda t: names[t])
        want = ts.split_expected(doc)
        assert got == want, rep
        n_spl += sum(len(v) for v in want.values())
    assert n_cig > 300 and n_spl > 30


def test_gt_correction_on_fresh_random_inputs(tmp_path):
    """correct_gt_del_real_data.py / correct_gt_ins_real_data.py on new call / signature / read sets: the oracle's support and span
    scans, and the host replay's TSV / VCF text, against the reference functions' outputs."""
    import json
    import test_gt_correction as tg
    mg = _golden_module("make_golden_gt")
    for k, (seed, dtype) in enumerate(((31, "Hifi"), (32, "ONT"), (33, "Hifi"))):
        case = json.loads(json.dumps(mg.make_case("live%d" % k, seed, dtype)))
        (tmp_path / ("s%d" % k)).mkdir()
        (tmp_path / ("t%d" % k)).mkdir()
        tg.test_oracle_scans_match_reference([case], tmp_path / ("s%d" % k))
        tg.test_host_replay_text_level([case], tmp_path / ("t%d" % k))


def test_error_sites_match_the_reference():
    """Where the reference raises, the oracle reports the matching status — and where it does not, neither does the oracle:
    an '=' op in an eligible record trips `assert offset_ref == read.reference_end` (H:396) on Hifi; on CLR the same record passes
    silently when the gate (C:425-427) rejects it first and raises when the gate lets it through; a haplotype-tagged CLR record
    without M ops divides by zero (C:61) whatever its mapq."""
    mg = _mg()
    M, I, D, EQ, S = 0, 1, 2, 7, 4
    base = [("chr1", 1000, "PS1_hp1_a", 60, False, [(M, 500), (D, 60), (M, 500)]),
            ("chr1", 5000, "PS1_hp2_b", 60, False, [(M, 400), (I, 80), (M, 600)])]

    def outcome(style, recs):
        try:
            exp = mg.run_contig(style, recs, stable=True)
            ref = 0
        except AssertionError:
            ref = -5
        except ZeroDivisionError:
            ref = -8
        chroms = ["chr1"]
        st, _ = oracle.run(_soa(recs, chroms), dtype=DTYPE_BY_NAME[style])
        return ref, st

    eq_ok_gate = ("chr1", 9000, "PS1_hp1_c", 60, False, [(M, 300), (EQ, 50), (M, 300)])                 # ins_pct 0: gate passes
    eq_gated = ("chr1", 9000, "PS1_hp1_c", 60, False, [(M, 50), (I, 40), (EQ, 50), (M, 50), (I, 40), (M, 50)])   # ins 80/230 > 0.13, mean M 50 < 200
    no_m_lowq = ("chr1", 9000, "PS1_hp1_c", 10, False, [(S, 20), (I, 60), (S, 20)])                    # tagged, mapq 10, no M op
    assert outcome("Hifi", base) == (0, 0)
    assert outcome("Hifi", base + [eq_ok_gate]) == (-5, -5)
    assert outcome("Hifi", base + [eq_gated]) == (-5, -5)              # Hifi has no gate
    assert outcome("CLR", base + [eq_ok_gate]) == (-5, -5)
    assert outcome("CLR", base + [eq_gated]) == (0, 0)                 # the gate rejects the record before the assert is reached
    assert outcome("CLR", base + [no_m_lowq]) == (-8, -8)
    assert outcome("Hifi", base + [no_m_lowq]) == (0, 0)               # mapq 10: not eligible, never walked
    # split mates: `assert rl1 == rl2` (H:331) when the two alignments of one name disagree about the read length
    mate1 = ("chr1", 20000, "PS2_hp1_s", 60, False, [(M, 400), (S, 300)])
    mate2_ok = ("chr1", 21000, "PS2_hp1_s", 60, False, [(S, 400), (M, 300)])
    mate2_bad = ("chr1", 21000, "PS2_hp1_s", 60, False, [(S, 400), (M, 310)])
    for style in ("Hifi", "ONT", "CLR"):
        assert outcome(style, base + [mate1, mate2_ok]) == (0, 0)
        ref, st = outcome(style, base + [mate1, mate2_bad])
        assert ref == -5 and st == -6                                  # an AssertionError there, VSV_E_READLEN here
    # `if read.seq: assert len(read.seq)==offset_contig` (H:397-398, O:408-409, C:430-431): a stored SEQ of another length than the
    # CIGAR's query length. The BAM readers mark such records (VSV_F_SEQ_MISMATCH); the assert is only reached by records that are walked.
    from volcanosv_amd.abi import F_SEQ_MISMATCH

    def outcome_seq(style, recs, seq_lens):
        """seq_lens[i]: stored SEQ length of record i (None = '*')."""
        try:
            mg.run_contig(style, [r + (seq_lens[i],) for i, r in enumerate(recs)], stable=True)
            ref = 0
        except AssertionError:
            ref = -5
        except ZeroDivisionError:
            ref = -8
        soa = _soa(recs, ["chr1"])
        for i, (r, sl) in enumerate(zip(recs, seq_lens)):
            qlen = sum(l for op, l in r[5] if op in (0, 1, 4, 7, 8))
            if sl and sl != qlen:                                     # what volcanosv_amd.bam / the device reader compute
                soa.flag[i] |= F_SEQ_MISMATCH
        st, _ = oracle.run(soa, dtype=DTYPE_BY_NAME[style])
        return ref, st

    walked = ("chr1", 9000, "PS1_hp1_c", 60, False, [(S, 10), (M, 300), (I, 50), (M, 300)])     # query length 660
    low_q = ("chr1", 9000, "PS1_hp1_c", 20, False, [(S, 10), (M, 300), (I, 50), (M, 300)])
    untagged = ("chr1", 9000, "PS1_c", 60, False, [(S, 10), (M, 300), (I, 50), (M, 300)])
    for style in ("Hifi", "ONT", "CLR"):
        assert outcome_seq(style, base + [walked], [None, None, 660]) == (0, 0)
        assert outcome_seq(style, base + [walked], [None, None, None]) == (0, 0)                # SEQ '*': nothing to compare
        ref, st = outcome_seq(style, base + [walked], [None, None, 661])
        assert ref == -5 and st == -10                                                          # AssertionError there, VSV_E_SEQLEN here
        assert outcome_seq(style, base + [low_q], [None, None, 661]) == (0, 0)                  # never walked
        assert outcome_seq(style, base + [untagged], [None, None, 661]) == (0, 0)
    assert outcome_seq("CLR", base + [eq_gated], [None, None, 5]) == (0, 0)                     # the gate comes first
    ref, st = outcome_seq("Hifi", base + [eq_ok_gate], [None, None, 5])
    assert ref == -5 and st == -5                                                               # reference_end assert (H:396) before the SEQ one


def test_filter_tra_merge_on_fresh_random_inputs(tmp_path):
    """Complex_SV/filter_tra.py load_raw_vcf / cluster_bnd / merge_bnd (:32-116), AST-extracted: the merged VCF text of the host mirror
    (bnd.merge_bnd_lines) on random breakend lines — dense positions, both bracket types, repeated keys, neighbours that share a
    centre key through the reference's own avg_pos2 slip (:64)."""
    import ast
    from collections import defaultdict
    from volcanosv_amd import bnd
    src = open(os.path.join(REF, "bin/VolcanoSV-vc/Complex_SV/filter_tra.py")).read()
    fdefs = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef)]
    ns = {"np": np, "os": os, "defaultdict": defaultdict}
    exec(compile(ast.Module(body=fdefs, type_ignores=[]), "filter_tra.py", "exec"), ns)
    rng = np.random.default_rng(11)
    n_merged = 0
    for case in range(40):
        lines = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"]
        n = int(rng.integers(2, 80))
        span = int(rng.choice([300, 3000, 100000]))
        body = []
        for k in range(n):
            c1, c2 = "chr%d" % rng.integers(1, 3), "chr%d" % rng.integers(1, 4)
            p1, p2 = int(rng.integers(1, span)), int(rng.integers(1, span))
            alt = [("N[%s:%d[", "["), ("]%s:%d]N", "]"), ("[%s:%d[N", "["), ("N]%s:%d]", "]")][int(rng.integers(0, 4))][0] % (c2, p2)
            body.append((c1, p1, "%s\t%d\tsvim_asm.BND.%d\tN\t%s\t.\tPASS\tSVTYPE=BND;READS=PS%d_hp1_x\tGT\t%s\n" %
                         (c1, p1, k, alt, k, rng.choice(["0/1", "1/0", "1/1"]))))
        if rng.random() < 0.5:
            body += body[: max(1, n // 10)]                                   # repeated keys
        body.sort(key=lambda x: (x[0], x[1]))                                 # the svim VCF is position-sorted
        lines += [b[2] for b in body]
        lines.insert(5, "chr1\t50\tsvim_asm.INS.1\tN\t<INS>\t.\tPASS\tSVTYPE=INS\tGT\t0/1\n")      # non-BND lines are dropped
        vp = tmp_path / ("t%d.vcf" % case)
        vp.write_text("".join(lines))
        dc, header, l1, l2 = ns["load_raw_vcf"](str(vp))
        if not l1 or not l2:
            continue                                                          # the reference indexes bnd_list[0] of an empty list
        outp = tmp_path / ("o%d.vcf" % case)
        ns["merge_bnd"](dc, str(outp), header, 100, l1, l2)
        hdr, got = bnd.merge_bnd_lines(lines, 100)
        assert "".join(hdr + got) == outp.read_text(), case
        n_merged += sum(1 for g in got if g.rstrip().endswith("1/1"))
    assert n_merged > 20


def test_svim_vcf_header_equals_write_final_vcf(tmp_path):
    """The header of variants.vcf: svim-asm's write_final_vcf (SVIM_COMBINE.py:379-425, AST-extracted, default type list, no
    candidates) against bnd.vcf_header, line for line except the time stamp."""
    import ast
    import re
    import time
    import types
    from volcanosv_amd import bnd
    sv = os.path.join(REF, "bin/VolcanoSV-vc/Complex_SV/svim-asm-1.0.2/src/svim_asm")
    tree = ast.parse(open(os.path.join(sv, "SVIM_COMBINE.py")).read())
    fdefs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("write_final_vcf", "sorted_nicely")]
    from collections import defaultdict
    ns = {"time": time, "re": re, "defaultdict": defaultdict}
    exec(compile(ast.Module(body=fdefs, type_ignores=[]), "SVIM_COMBINE.py", "exec"), ns)
    contigs = [("chr1", 248956422), ("chr10", 133797422), ("chrX", 156040895)]
    for qn in (True, False):
        opts = types.SimpleNamespace(working_dir=str(tmp_path), tandem_duplications_as_insertions=False, interspersed_duplications_as_insertions=False,
                                     query_names=qn, sample="Sample", symbolic_alleles=False)
        ns["write_final_vcf"]([], [], [], [], [], [], "1.0.2", [c[0] for c in contigs], [c[1] for c in contigs],
                              ["DEL", "INS", "INV", "DUP:TANDEM", "DUP:INT", "BND"], types.SimpleNamespace(close=lambda: None), opts)
        want = open(tmp_path / "variants.vcf").read().splitlines(True)
        got = bnd.vcf_header(contigs, qn, "Sample")
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g == w or (g.startswith("##fileDate=") and w.startswith("##fileDate="))


def test_phase_vcf_of_both_drivers(tmp_path):
    """phase_vcf of volcanosv-vc-large-indel.py (:202-231) and of volcanosv-vc-complex-sv.py (:67-97), AST-extracted, against
    pipeline.phase_large_indel / phase_complex on random VCF text: PS ids from contig names, '/' -> '|' genotypes, the PS INFO
    line spliced six lines before the header's end."""
    import ast
    from volcanosv_amd import bnd, pipeline

    def extract(path, name):
        tree = ast.parse(open(path).read())
        ns = {}
        exec(compile(ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name], type_ignores=[]), path, "exec"), ns)
        return ns[name]

    rng = np.random.default_rng(12)
    # large indel
    ref_fn = extract(os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL/volcanosv-vc-large-indel.py"), "phase_vcf")
    header = ["##fileformat=VCFv4.2\n", "##source=x\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"]
    lines = list(header)
    for k in range(200):
        hp = int(rng.integers(1, 3))
        reg = "PS%d_hp%d_ctg%d:%d-%d" % (rng.integers(1, 99999), hp, rng.integers(0, 9), rng.integers(1, 9999), rng.integers(1, 9999))
        gt = str(rng.choice(["0/1", "1/1"]))
        if gt == "1/1":
            reg += ",PS%d_hp%d_ctg1:5-6" % (rng.integers(1, 99999), 3 - hp)
        lines.append("chr1\t%d\tvolcano.chr1.INS.%d\tA\tAC\t20\tPASS\tSVLEN=1;SVTYPE=INS;TIG_REGION=%s;QUERY_STRAND=+\tGT\t%s\n" % (rng.integers(1, 10**8), k, reg, gt))
    (tmp_path / "in.vcf").write_text("".join(lines))
    (tmp_path / "hdr").write_text("".join(header))
    ref_fn(str(tmp_path / "in.vcf"), str(tmp_path / "out.vcf"), str(tmp_path / "hdr"))
    assert "".join(pipeline.phase_large_indel(lines, header)) == (tmp_path / "out.vcf").read_text()
    # complex
    ref_fn = extract(os.path.join(REF, "bin/VolcanoSV-vc/Complex_SV/volcanosv-vc-complex-sv.py"), "phase_vcf")
    lines = bnd.vcf_header([("chr1", 1000), ("chr2", 2000)], True, "Sample")
    for k in range(200):
        gt = str(rng.choice(["0/1", "1/0", "1/1", "./."]))
        lines.append("chr1\t%d\tvolcanosv.BND.%d\tN\tN[chr2:%d[\t.\tPASS\tSVTYPE=BND;READS=PS%d_hp1_c,PS%d_hp2_d\tGT\t%s\n" %
                     (rng.integers(1, 999), k, rng.integers(1, 1999), rng.integers(1, 99999), rng.integers(1, 99999), gt))
    (tmp_path / "c.vcf").write_text("".join(lines))
    ref_fn(str(tmp_path / "c.vcf"), str(tmp_path / "c_out.vcf"))
    assert "".join(pipeline.phase_complex(lines)) == (tmp_path / "c_out.vcf").read_text()


def test_generate_vcf_header_of_the_large_indel_driver(tmp_path):
    """generate_vcf_header of volcanosv-vc-large-indel.py (:104-131, with the reference's own header_info data file) against
    pipeline.generate_vcf_header: the per-run VCF header, for one chromosome and for all."""
    import ast
    import subprocess
    from volcanosv_amd import pipeline
    li = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
    tree = ast.parse(open(os.path.join(li, "volcanosv-vc-large-indel.py")).read())
    keep = ("create_fai", "extract_contigs_from_fai", "generate_vcf_header")
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1 some description\n" + "ACGT" * 30 + "\nAC\n>chr2\n" + "G" * 77 + "\n>chr10\nACGTACGT\n")
    pipeline.write_fai(str(fa))                                   # the reference shells out to samtools faidx for this file
    for chr_num, prefix in ((None, "Sample"), (2, "HG002"), (10, "x")):
        ns = {"os": os, "subprocess": subprocess, "code_dir": li, "prefix": prefix}
        exec(compile(ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in keep], type_ignores=[]), "d.py", "exec"), ns)
        ns["generate_vcf_header"](str(fa), str(tmp_path / "ref_hdr"), chr_num)
        pipeline.generate_vcf_header(str(fa), str(tmp_path / "my_hdr"), chr_num, prefix)
        assert (tmp_path / "my_hdr").read_text() == (tmp_path / "ref_hdr").read_text(), (chr_num, prefix)
