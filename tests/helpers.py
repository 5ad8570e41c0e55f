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
