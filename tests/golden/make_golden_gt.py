#!/usr/bin/env python3
"""Generates tests/golden/gt_correction.json from the REFERENCE's correct_gt_del_real_data.py / correct_gt_ins_real_data.py
functions (build container only). Top-level FunctionDefs are AST-extracted; pysam.AlignmentFile(...).fetch(chrom, start, end)
is a stand-in over an in-memory read list with htslib's overlap rule (start < end_region and end > start_region), joblib's
Parallel/delayed run sequentially. The fixture holds the inputs (VCF lines, .sigs lines, reads as (chrom, start, end)) and
what the reference functions produced: support / match-id lists, spanning-read depths, the TSV tables and the re-genotyped
VCF lines.

Usage:  python tests/golden/make_golden_gt.py            (needs /root/reference)
"""
import ast
import json
import os
import tempfile
import types
from collections import defaultdict

import numpy as np
import pandas as pd

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))

READS = {}            # chrom -> [(start, end, name)]


class FakeRead:
    def __init__(self, s, e, n):
        self.reference_start, self.reference_end, self.qname = s, e, n


class FakeAlignmentFile:
    def __init__(self, path, *a, **k):
        pass

    def fetch(self, chrom, start, end):
        return [FakeRead(s, e, n) for s, e, n in READS.get(chrom, []) if s < end and e > start]


def load_functions(script, extra):
    tree = ast.parse(open(os.path.join(LI, script)).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef)], type_ignores=[])
    ns = {"np": np, "pd": pd, "defaultdict": defaultdict, "tqdm": lambda x, **k: x, "gzip": __import__("gzip"),
          "pysam": types.SimpleNamespace(AlignmentFile=FakeAlignmentFile),
          "Parallel": lambda n_jobs=1: (lambda gen: list(gen)), "delayed": lambda f: f, "n_thread": 1}
    ns.update(extra)
    exec(compile(mod, script, "exec"), ns)
    return ns


def make_inputs(seed):
    rng = np.random.default_rng(seed)
    vcf = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"]
    sig_del, sig_ins, k = [], [], 0
    READS.clear()
    for chrom in ("chr1", "chr2", "chr10"):
        reads = []
        p = 5000
        while p < 600_000:
            p += int(rng.integers(50, 900))
            reads.append((p, p + int(rng.integers(3000, 25000)), "r%s_%d" % (chrom, len(reads))))
        READS[chrom] = reads
        pos = 20_000
        for _ in range(60):
            pos += int(rng.integers(800, 9000))
            typ = "DEL" if rng.integers(2) else "INS"
            ln = int(rng.choice([30, 45, 80, 200, 600, 999, 1000, 1001, 2500, 9000]))
            gt = str(rng.choice(["0/1", "1/1", "1/1", "0/1", "./."]))
            vcf.append("%s\t%d\tvolcano%d\tN\tN\t.\tPASS\tSVTYPE=%s;SVLEN=%d;TIG=x\tGT:DP\t%s:7\n" % (chrom, pos, k, typ, -ln if typ == "DEL" else ln, gt))
            k += 1
            # read signatures around the call: supporting (similar size, near), off-size, far; duplicates share (pos, len)
            for _ in range(int(rng.integers(0, 25))):
                sp = pos + int(rng.choice([0, 0, 3, -7, 120, -480, 499, 500, 501, 2000, -2600])) + int(rng.integers(-3, 4))
                sl = max(10, int(ln * float(rng.choice([1.0, 1.0, 0.95, 0.61, 0.6, 0.59, 1.66, 1.67, 3.0]))))
                (sig_del if typ == "DEL" else sig_ins).append((chrom, sp, sl, "read%d" % rng.integers(0, 10**6)))
    # a signature on a non-numeric chromosome (the INS loader skips it, the DEL loader keeps it)
    sig_ins.append(("chrX", 500, 80, "readx"))
    sig_del.append(("chrX", 500, 80, "readx"))

    def sig_lines(sigs, typ, with_seq):
        lines = set()
        for c, p, l, r in sigs:
            lines.add("%s\t%s\t%d\t%d\t%s%s\n" % (typ, c, p, l, r, "\tACGT" if with_seq else ""))
        return sorted(lines, key=lambda s: (s.split("\t")[1].encode(), int(s.split("\t")[2]), s.encode()))     # sort -k2,2 -k3,3n, LC_ALL=C

    return vcf, sig_lines(sig_del, "DEL", False), sig_lines(sig_ins, "INS", True)


def make_case(name, seed, dtype):
    """Inputs + every output of the two reference scripts for one seed."""
    if True:
        vcf, dsig, isig = make_inputs(seed)
        reads = {c: [[s, e] for s, e, _ in v] for c, v in READS.items()}
        case = dict(name=name, dtype=dtype, vcf=vcf, del_sigs=dsig, ins_sigs=isig, reads=reads)
        with tempfile.TemporaryDirectory() as d:
            vp = os.path.join(d, "in.vcf")
            open(vp, "w").writelines(vcf)
            dp, ip = os.path.join(d, "DEL.sigs"), os.path.join(d, "INS.sigs")
            open(dp, "w").writelines(dsig)
            open(ip, "w").writelines(isig)
            # ---- DEL script ----
            ns = load_functions("correct_gt_del_real_data.py", {"vtype": "DEL"})
            vars_comp = ns["load_vcf"](vp)
            sig_list = ns["load_sig"](dp)
            case["del_support"] = [int(x) for x in ns["match_varlist_siglist"](sig_list, vars_comp, 0.6, 2.3)]
            case["del_depth"] = [float(ns["check_full_cover_reads"]("x.bam", v[2].split()[0], int(v[2].split()[1]), v[1])) for v in vars_comp]
            tsv = os.path.join(d, "bnd_del_real.tsv")
            ns["vars_comp"] = vars_comp
            df = ns["extract_sig_support"](dp, vars_comp, "x.bam", tsv, 0.6, 2.3, None)
            case["del_tsv"] = open(tsv).read()
            para = ns["read_para"](os.path.join(LI, "para", "GT_correction_para_%s_DEL.txt" % dtype))
            df = pd.read_csv(tsv, sep="\t")
            df["new_gt"] = ns["correct_gt_eval"](df, para["t_large_11"], para["t_small_11"], para["t_large_01"], para["t_small_01"])
            df.to_csv(tsv + ".newgt", sep="\t", index=False)
            case["del_newgt_tsv"] = open(tsv + ".newgt").read()
            ns["write_new_gt_vcf"](vp, vp + ".newgt.DEL", df)
            case["del_newgt_vcf"] = open(vp + ".newgt.DEL").read()
            # ---- INS script ----
            ns = load_functions("correct_gt_ins_real_data.py", {"vtype": "INS"})
            sv_list = ns["load_vcf"](vp)
            cnt, match = ns["extract_sig_support"](sv_list, ip, 2.3, 0.6)
            case["ins_support"], case["ins_match"] = [int(x) for x in cnt], [int(x) for x in match]
            case["ins_gte30auto"] = open(ip + ".gte30auto").read()
            case["ins_depth"] = [int(ns["check_full_cover_reads"]("x.bam", "chr" + str(v[0]), v[1], 100)) for v in sv_list]
            tsv = os.path.join(d, "bnd_ins_real.tsv")
            ns["write_new_df"](sv_list, ip, "x.bam", None, tsv, 2.3, 0.6, 100, 1)
            case["ins_tsv"] = open(tsv).read()
            para = ns["read_para"](os.path.join(LI, "para", "GT_correction_para_%s_INS.txt" % dtype))
            df = pd.read_csv(tsv, sep="\t")
            df["new_gt"] = ns["correct_gt_eval"](df, para["t_large_11"], para["t_small_11"], para["t_large_01"], para["t_small_01"])
            df.to_csv(tsv + ".newgt", sep="\t", index=False)
            case["ins_newgt_tsv"] = open(tsv + ".newgt").read()
            ns["write_new_gt_vcf"](vp, vp + ".newgt.INS", df)
            case["ins_newgt_vcf"] = open(vp + ".newgt.INS").read()
        return case


def main():
    out = []
    for name, seed, dtype in (("a", 21, "Hifi"), ("b", 22, "ONT")):
        case = make_case(name, seed, dtype)
        out.append(case)
        print(name, "DEL vars", len(case["del_support"]), "sum support", sum(case["del_support"]), "| INS vars", len(case["ins_support"]), "sum support", sum(case["ins_support"]),
              "| changed GT (DEL)", sum(1 for a, b in zip(case["del_newgt_tsv"].splitlines()[1:], case["del_tsv"].splitlines()[1:]) if a.split("\t")[-1] != b.split("\t")[2]))
    with open(os.path.join(HERE, "gt_correction.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
