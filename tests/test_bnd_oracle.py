"""Pins the BND branch of the oracle against the reference's svim-asm functions (tests/golden/bnd_*.json, produced by
tests/golden/make_golden_bnd.py from SVIM_inter.analyze_read_segments, SVCandidate.CandidateBreakend and the
AST-extracted form_partitions / pair_haplotypes_breakends / sorted_nicely with the real scipy linkage)."""
import json
import os

import pytest

from volcanosv_amd import bnd
from volcanosv_amd.abi import B_DST_FWD, B_SRC_FWD

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        doc = json.load(f)
    return doc, bnd.SegmentSoA(doc["reads"], [tuple(c) for c in doc["contigs"]])


def cand_fields(seg, c):
    names = [x[0] for x in seg.contigs]
    m = int(c["meta"])
    return [names[int(c["src_tid"])], int(c["src_pos"]), "fwd" if m & B_SRC_FWD else "rev", names[int(c["dst_tid"])], int(c["dst_pos"]),
            "fwd" if m & B_DST_FWD else "rev"]


def check_against_golden(doc, seg, cand, calls):
    per_read = [[] for _ in doc["reads"]]
    for c in cand:
        per_read[int(c["read"])].append(cand_fields(seg, c))
    assert per_read == doc["expected"]["per_read"]
    got = sorted(bnd.call_fields(seg, c) for c in calls)
    assert got == doc["expected"]["paired"]
    assert bnd.vcf_lines(seg, calls) == doc["expected"]["vcf"]


@pytest.mark.parametrize("name", ["bnd_a", "bnd_b"])
def test_bnd_oracle_matches_svim(name):
    from oracle import oracle
    doc, seg = load(name)
    cand, calls = oracle.run_bnd(seg)
    assert doc["expected"]["n_dropped_partitions"] >= 3
    check_against_golden(doc, seg, cand, calls)


def test_is_similar_truth_table():
    """svim-asm tests/test_inter.py:8-11 (is_similar is used by the DUP branch; kept as a pinned predicate)."""
    sim = lambda c1, s1, e1, c2, s2, e2: c1 == c2 and abs(s1 - s2) < 20 and abs(e1 - e2) < 20
    assert not sim("chrI", 0, 100, "chrII", 0, 100)
    assert sim("chrI", 0, 100, "chrI", 0, 100) and sim("chrI", 0, 100, "chrI", 10, 90)
    assert not sim("chrI", 0, 100, "chrI", 21, 100)


def test_sa_tag_parsing_matches_reference_fixture():
    """svim-asm tests/chimeric_read_errors.sam: malformed SA entries are skipped, negative MAPQ becomes 0
    (SVIM_COLLECT.py:19-20, 37-40; expectations of tests/test_satag.py:36-54)."""
    sa = "chr1,100,+,50M50S,-5,1;bad,entry;chr2,200,-,30S70M,60,0;"
    got = bnd.parse_sa(sa, lambda n: {"chr1": 0, "chr2": 1}[n])
    assert [(g[0], g[1], g[2], g[4]) for g in got] == [(0, 99, False, 0), (1, 199, True, 60)]
    assert bnd.cigar_stats(got[1][3]) == (70, 30, 100, 100)


def test_filter_tra_merge():
    """filter_tra.py:70-116: consecutive lines of one bracket type within 100 bp on both mates collapse to the first, GT 1/1."""
    L = lambda c, p, alt, gt: "%s\t%d\tid\tN\t%s\t.\tPASS\tSVTYPE=BND\tGT\t%s\n" % (c, p, alt, gt)
    lines = ["##h\n", L("chr1", 100, "N[chr2:500[", "1/0"), L("chr1", 150, "N[chr2:560[", "0/1"), L("chr1", 400, "N[chr2:561[", "0/1"),
             L("chr1", 120, "N]chr2:510]", "0/1")]
    hdr, body = bnd.merge_bnd_lines(lines)
    assert hdr == ["##h\n"] and len(body) == 3
    assert body[0].split()[-1] == "1/1" and body[0].split()[1] == "100"


def test_reference_chimeric_read_bam_fixtures():
    """The reference's own SA-tag fixtures (svim-asm tests/chimeric_read.bam, chimeric_read_errors.bam — data files of
    tests/test_satag.py, copied verbatim as fixtures) through the BAM ingest (C-ABI vsv_bam_*) and the SA parser:
    test_satag.py:13-33 (4 records; the primary's SA tag reconstructs the three other alignments field by field) and
    :36-54 (an SA entry with too many fields is skipped; a negative MAPQ becomes 0)."""
    from volcanosv_amd import bam
    ops = "MIDNSHP=X"
    s = bam.read_bam(os.path.join(GOLDEN, "chimeric_read.bam"))
    assert s.n_records == 4
    tid_of = lambda n: s.tid_names.index(n)
    cig = lambda i: [(int(w) & 15, int(w) >> 4) for w in s.cigar[int(s.cigar_off[i]):int(s.cigar_off[i + 1])]]
    prim = [i for i in range(4) if not (s.flag[i] & 2)]                    # the non-supplementary record
    assert len(prim) == 1
    sa = bnd.parse_sa(s.sa_tags[prim[0]], tid_of)
    others = [i for i in range(4) if i != prim[0]]
    assert len(sa) == 3
    # as test_satag.py:25-35: SA entry k reconstructs alignment k+1
    for (tid, pos0, rev, c2, mq), i in zip(sa, others):
        assert int(s.tid[i]) == tid and int(s.pos[i]) == pos0
        assert c2 == cig(i), "".join("%d%s" % (l, ops[o]) for o, l in c2)
        assert bool(s.flag[i] & 1) == rev and int(s.mapq[i]) == mq
        assert int(s.sam_flags[i]) == (2064 if rev else 2048)              # SVIM_COLLECT.py:35-38
        assert bnd.cigar_stats(c2) == bnd.cigar_stats(cig(i))             # reference_end / query_alignment_start,end
    e = bam.read_bam(os.path.join(GOLDEN, "chimeric_read_errors.bam"))
    tid_e = lambda n: e.tid_names.index(n)
    assert len(bnd.parse_sa(e.sa_tags[0], tid_e)) == 2                     # first SA entry has too many fields
    neg = bnd.parse_sa(e.sa_tags[1], tid_e)
    assert len(neg) == 1 and neg[0][4] == 0                                # negative MAPQ -> 0
