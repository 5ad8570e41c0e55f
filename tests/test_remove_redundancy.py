"""remove_redundancy.py.

CPU: the oracle (windowed pair predicates + Levenshtein DP) against tests/golden/remove_redundancy.json — link lists,
components and the two output VCFs produced by the reference's own functions (tests/golden/make_golden_redundancy.py; edlib
restated as the Levenshtein DP, networkx real). GPU: vsv_redundancy_pairs (Myers bit-parallel edit distance, one wave per
pair) against the oracle, the reference link lists and the VCF text; long-sequence pairs against the DP."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "remove_redundancy.json")


@pytest.fixture(scope="module")
def doc():
    with open(GOLDEN) as f:
        return json.load(f)


class OracleEngine:
    """Engine look-alike over the oracle, so the host mirror's text logic can be checked without a GPU."""

    def redundancy_params(self, **kw):
        from oracle import oracle
        return oracle.default_redundancy_params(**{k: v for k, v in kw.items() if v is not None})

    def redundancy_pairs(self, is_del, pos, svlen, seq=None, seq_off=None, params=None):
        from oracle import oracle
        st, pairs = oracle.run_redundancy_pairs(is_del, pos, svlen, seq, seq_off, params)
        assert st == 0
        return pairs

    def close(self):
        pass


def check_text_level(doc, eng, tmp_path):
    from volcanosv_amd import remove_redundancy as rr
    for case in doc["cases"]:
        d = tmp_path / case["name"]
        d.mkdir()
        vcf = d / "in.vcf"
        vcf.write_text("".join(case["vcf"]))
        del_sig, ins_sig, dc, header = rr.vcf_to_sig(str(vcf))
        p = eng.redundancy_params()
        got = rr.match_chr([s for s in del_sig if s[0] == "chr1"], True, eng, p)
        assert [list(l) for l in got] == case["links_del_chr1"]
        got = rr.match_chr([s for s in ins_sig if s[0] == "chr1"], False, eng, p)
        assert [list(l) for l in got] == case["links_ins_chr1"] and len(got) > 20
        nodes_del, nodes_ins = rr.run(str(vcf), str(d / "out"), engine=eng)
        assert [sorted(n) for n in nodes_del] == case["nodes_del"] and [sorted(n) for n in nodes_ins] == case["nodes_ins"]
        assert (d / "out" / "volcano_variant_no_redundancy.vcf").read_text() == case["no_redundancy"]
        assert (d / "out" / "volcano_variant_redundancy.vcf").read_text() == case["redundancy"]


def test_oracle_edit_distance_known_answers(doc):
    from oracle import oracle
    for a, b, d in doc["edit_distance"]:
        assert oracle.levenshtein(a, b) == d


def test_oracle_matches_reference_text_level(doc, tmp_path):
    check_text_level(doc, OracleEngine(), tmp_path)


def test_zero_length_call_is_an_error():
    from oracle import oracle
    st, _ = oracle.run_redundancy_pairs(True, [100, 200], [0, 0])
    assert st == -8                                     # VSV_E_ZERODIV: get_size_sim divides by max(svlen) (RR:88-90)


@pytest.mark.gpu
def test_gpu_matches_reference_text_level(doc, tmp_path):
    from volcanosv_amd.engine import Engine
    with Engine(0) as eng:
        check_text_level(doc, eng, tmp_path)


@pytest.mark.gpu
def test_gpu_edit_distance_vs_dp_including_long_sequences():
    """Pairs with every block count from 1 to > 64 (the super-block path, > 4096 symbols) against the oracle's DP, plus
    parameter variations and the error statuses."""
    from oracle import oracle
    from volcanosv_amd.abi import VsvError
    from volcanosv_amd.engine import Engine
    rng = np.random.default_rng(7)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 500, 1000, 4095, 4096, 4097, 5000, 9000]
    seqs, pos = [], []
    p0 = 1000
    for ln in lens:                                     # three related calls per length: identical / mutated / unrelated
        base = rng.integers(0, 4, ln).astype(np.uint8)
        mut = base.copy()
        k = max(1, ln // 10)
        mut[rng.integers(0, ln, k)] = rng.integers(0, 5, k)
        mut = np.delete(mut, rng.integers(0, len(mut), min(len(mut) - 1, ln // 20)))
        other = rng.integers(0, 4, max(1, int(ln * 0.8))).astype(np.uint8)
        for s in (base, mut, other):
            seqs.append(s)
            pos.append(p0)
            p0 += 10
        p0 += 5000
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    blob = np.concatenate(seqs)
    svlen = [len(s) for s in seqs]
    with Engine(0) as eng:
        for kw in (dict(), dict(seq_sim_thresh=0.95), dict(seq_sim_thresh=0.0, size_sim_thresh=0.0, dist_thresh=100000)):
            got = eng.redundancy_pairs(False, pos, svlen, blob, off, eng.redundancy_params(**kw))
            st, want = oracle.run_redundancy_pairs(False, pos, svlen, blob, off, oracle.default_redundancy_params(**kw))
            assert st == 0 and np.array_equal(got, want), kw
        assert len(want) > 100                           # the last setting links every pair inside the (huge) window
        got = eng.redundancy_pairs(True, pos, svlen, params=eng.redundancy_params(dist_thresh_del=20, size_sim_thresh_del=0.5))
        st, want = oracle.run_redundancy_pairs(True, pos, svlen, params=oracle.default_redundancy_params(dist_thresh_del=20, size_sim_thresh_del=0.5))
        assert st == 0 and np.array_equal(got, want) and len(want) > 5
        assert eng.redundancy_pairs(True, [], []).shape == (0, 2) and eng.redundancy_pairs(True, [5], [40]).shape == (0, 2)
        with pytest.raises(VsvError) as e:
            eng.redundancy_pairs(True, [200, 100], [40, 40])
        assert e.value.status == -7
        with pytest.raises(VsvError) as e:
            eng.redundancy_pairs(True, [100, 200], [40, 0])
        assert e.value.status == -8
