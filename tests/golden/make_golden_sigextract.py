#!/usr/bin/env python3
"""Generates tests/golden/sig_extract.json.gz from the REFERENCE's sig_extract.py functions (build container only).

parse_read (SE:438-493) and everything it calls are AST-extracted (FunctionDefs + the two module-level tables) and run on
duck-typed reads; the `cigar` package the script imports is absent here, so acquire_clip_pos gets a 10-line stand-in
Cigar.items() (CIGAR string -> [(len, op)], which is all SE:330 uses). The fixture holds the reads (pos, SAM flag, mapq,
CIGAR, sequence, SA tag) and, per read, the candidate list parse_read returned.

Usage:  python tests/golden/make_golden_sigextract.py            (needs /root/reference)
"""
import ast
import json
import os
import re
import types

import numpy as np

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))
OPS = "MIDNSHP=X"


class _Cigar:
    def __init__(self, s):
        self.s = s

    def items(self):
        return [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", self.s)]


def load_functions():
    tree = ast.parse(open(os.path.join(LI, "sig_extract.py")).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) or
            (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") in ("dic_starnd", "flag_signal"))]
    ns = {"cigar": types.SimpleNamespace(Cigar=_Cigar)}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "sig_extract.py", "exec"), ns)
    return ns


class FakeRead:
    def __init__(self, d):
        self.query_name = d["name"]
        self.flag = d["flag"]
        self.mapq = d["mapq"]
        self.reference_start = d["pos"]
        self.cigar = [tuple(c) for c in d["cigar"]]
        self.reference_end = d["pos"] + sum(l for op, l in self.cigar if op in (0, 2, 3, 7, 8))
        self.query_sequence = d["seq"]
        self.query_length = len(d["seq"])
        self._sa = d.get("sa")

    def get_tags(self):
        return [("NM", 3)] + ([("SA", self._sa)] if self._sa else [])


def rand_seq(rng, n):
    return "".join(rng.choice(list("ACGT"), n))


def cigar_string(c):
    return "".join("%d%s" % (l, OPS[op]) for op, l in c)


def make_plain_reads(rng, n):
    """CIGAR-only reads: indels of 1..400 bp, pairs closer than the merge thresholds, clips, =/X/N/P ops, low mapq, short reads."""
    reads = []
    pos = 1000
    for i in range(n):
        pos += int(rng.integers(1, 3000))
        k = int(rng.integers(1, 14))
        c = []
        lead = int(rng.integers(0, 4))
        if lead == 1:
            c.append((4, int(rng.integers(1, 300))))
        elif lead == 2:
            c.append((5, int(rng.integers(1, 300))))
        for j in range(k):
            c.append((int(rng.choice([0, 0, 0, 7, 8])), int(rng.choice([1, 5, 40, 90, 101, 250]))))
            if j < k - 1:
                op = int(rng.choice([1, 1, 2, 2, 3, 6]))
                ln = int(rng.choice([1, 5, 9, 10, 11, 29, 30, 60, 150, 400]))
                c.append((op, ln))
                if rng.integers(3) == 0:     # a second signal right behind the first (merge candidates)
                    c.append((0, int(rng.choice([1, 20, 99, 100, 101]))))
                    c.append((int(rng.choice([1, 2])), int(rng.choice([10, 15, 45]))))
        if rng.integers(4) == 0:
            c.append((4 if lead != 2 else 5, int(rng.integers(1, 200))))
        # merge equal neighbouring ops the generator may have produced
        cc = []
        for op, ln in c:
            if cc and cc[-1][0] == op:
                cc[-1] = (op, cc[-1][1] + ln)
            else:
                cc.append((op, ln))
        qlen = sum(l for op, l in cc if op in (0, 1, 4, 7, 8))
        flag = int(rng.choice([0, 0, 16, 16, 2048, 2064, 256, 272]))
        mapq = int(rng.choice([60, 60, 60, 20, 19, 0]))
        reads.append(dict(name="read%d" % i, flag=flag, mapq=mapq, pos=pos, cigar=cc, seq=rand_seq(rng, qlen), sa=None))
    return reads


def make_split_reads(rng, n):
    """Primary alignments with SA tags: 2..5 segments, same / different chromosome and strand, planted INS / DEL gaps."""
    reads = []
    pos0 = 50_000
    for i in range(n):
        pos0 += int(rng.integers(2000, 9000))
        nseg = int(rng.choice([2, 2, 2, 3, 3, 4, 5, 9]))
        total = 0
        segs = []
        rpos = pos0
        strand = "+" if rng.integers(2) else "-"
        for s in range(nseg):
            qlen = int(rng.integers(300, 900))
            gap_kind = int(rng.integers(4))
            qgap, rgap = 0, 0
            if s > 0:
                if gap_kind == 0:
                    qgap = int(rng.choice([0, 35, 120, 800]))        # unaligned read bases between segments -> INS
                    rgap = int(rng.choice([-5, 0, 20, 99, 101]))
                elif gap_kind == 1:
                    rgap = int(rng.choice([30, 45, 500, 20000, 150000]))  # reference gap -> DEL
                    qgap = int(rng.choice([0, 10, 101]))
                elif gap_kind == 2:
                    rgap = int(rng.integers(-300, 300))
                    qgap = int(rng.integers(0, 200))
            qs = total + qgap
            rs = rpos + rgap
            chrom = "chr1" if rng.integers(10) else "chr2"
            st = strand if rng.integers(8) else ("-" if strand == "+" else "+")
            segs.append(dict(qs=qs, qe=qs + qlen, rs=rs, re=rs + qlen, chrom=chrom, strand=st, mapq=int(rng.choice([60, 60, 30, 20, 5]))))
            total = qs + qlen
            rpos = rs + qlen
        total += int(rng.integers(0, 100))
        # read coordinates are forward-read coordinates; a '-' segment's CIGAR clips are mirrored
        def seg_cigar(sg):
            left, right = sg["qs"], total - sg["qe"]
            if sg["strand"] == "-":
                left, right = right, left
            c = []
            if left:
                c.append((4, left))
            c.append((0, sg["qe"] - sg["qs"]))
            if right:
                c.append((4, right))
            return c
        prim = int(rng.integers(nseg))
        p = segs[prim]
        sa = ""
        for k, sg in enumerate(segs):
            if k != prim:
                sa += "%s,%d,%s,%s,%d,%d;" % (sg["chrom"], sg["rs"] + 1, sg["strand"], cigar_string(seg_cigar(sg)), sg["mapq"], 0)
        if p["chrom"] != "chr1":
            continue                       # the primary is fetched from chr1
        reads.append(dict(name="split%d" % i, flag=16 if p["strand"] == "-" else 0, mapq=int(rng.choice([60, 60, 60, 19])), pos=p["rs"],
                          cigar=seg_cigar(p), seq=rand_seq(rng, total), sa=sa))
    return reads


def norm(cands):
    out = []
    for c in cands:
        out.append([x if isinstance(x, (int, str)) else float(x) for x in c])
    return out


def main():
    ns = load_functions()
    rng = np.random.default_rng(20250328)
    doc = {"params": dict(SV_size=30, min_mapq=20, max_split_parts=7, min_read_len=500, min_siglength=10, merge_del_threshold=0,
                          merge_ins_threshold=100, MaxSize=100000), "cases": {}}
    for name, reads in (("plain", make_plain_reads(rng, 250)), ("split", make_split_reads(rng, 400))):
        exp = []
        for d in reads:
            exp.append(norm(ns["parse_read"](FakeRead(d), "chr1", 30, 20, 7, 500, 10, 0, 100, 100000)))
        doc["cases"][name] = dict(reads=reads, expected=exp)
        kinds = {}
        for e in exp:
            for c in e:
                k = c[-2] if len(c) != 7 else c[5]
                kinds[k] = kinds.get(k, 0) + 1
        print(name, len(reads), "reads ->", kinds)
    # a second parameter set on the plain reads (merge_del 50, merge_ins 20, min_siglength 30)
    reads = doc["cases"]["plain"]["reads"]
    doc["cases"]["plain_params2"] = dict(params=dict(min_siglength=30, merge_del_threshold=50, merge_ins_threshold=20),
                                         expected=[norm(ns["parse_read"](FakeRead(d), "chr1", 30, 20, 7, 500, 30, 50, 20, 100000)) for d in reads])
    import gzip
    with gzip.open(os.path.join(HERE, "sig_extract.json.gz"), "wt", compresslevel=9) as f:
        json.dump(doc, f)


if __name__ == "__main__":
    main()
