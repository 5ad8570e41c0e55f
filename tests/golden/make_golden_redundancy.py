#!/usr/bin/env python3
"""Generates tests/golden/remove_redundancy.json from the REFERENCE's remove_redundancy.py functions (build container only).

Top-level FunctionDefs are AST-extracted (the script runs argparse + the job at import). Third-party pieces:
  * networkx (pinned 2.8.6 in requirement.yaml; 3.4.2 here) is used as is (connected_components).
  * edlib (pinned 1.3.9) is NOT installed in this image: edit_sim's `edlib.align(seq1, seq2)["editDistance"]` (RR:75-81, default
    mode NW = global alignment, unit costs) is the Levenshtein distance, restated here as the textbook DP and injected as `edlib`.
Ties: sort_sig_per_chr uses numpy's unstable argsort and pick_best_sv_one_cluster takes the first longest member in `list(set)`
order (hash-seed dependent). The generated calls have distinct positions per chromosome and distinct lengths inside every
component (checked), so the fixture does not depend on either.

Usage:  python tests/golden/make_golden_redundancy.py            (needs /root/reference)
"""
import ast
import json
import os
import sys
import tempfile
import types

import networkx as nx
import numpy as np

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))


def levenshtein(a, b):
    """Textbook unit-cost edit distance, one DP row per character of `a` (the insertion chain of a row is a running minimum)."""
    bb = np.frombuffer(b.encode(), dtype=np.uint8)
    idx = np.arange(len(b) + 1)
    prev = idx.copy()
    for i, ch in enumerate(a.encode(), 1):
        x = np.empty(len(b) + 1, dtype=np.int64)
        x[0] = i
        x[1:] = np.minimum(prev[:-1] + (bb != ch), prev[1:] + 1)
        prev = np.minimum.accumulate(x - idx) + idx
    return int(prev[-1])


def load_functions():
    tree = ast.parse(open(os.path.join(LI, "remove_redundancy.py")).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef)], type_ignores=[])
    fake_edlib = types.SimpleNamespace(align=lambda s1, s2: {"editDistance": levenshtein(s1, s2)})
    sys.modules["edlib"] = fake_edlib
    ns = {"np": np, "nx": nx, "tqdm": lambda x, **k: x, "os": os}
    exec(compile(mod, "remove_redundancy.py", "exec"), ns)
    return ns


def mutate(rng, s, rate):
    out = []
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(str(rng.choice(list("ACGT"))))
            continue
        out.append(ch)
        if u > 1 - rate / 3:
            out.append(str(rng.choice(list("ACGT"))))
    return "".join(out) or "A"


def make_vcf(seed, n_sites):
    rng = np.random.default_rng(seed)
    lines = ["##fileformat=VCFv4.2\n", "##source=volcano\n", "##contig=<ID=chr1>\n",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"]
    used, k = set(), 0
    used_len = {"INS": set(), "DEL": set()}          # SV lengths are unique per type: no equal-longest members in any component
    body = []
    for chrom in ("chr1", "chr2", "chrX"):
        pos = 10_000
        for _ in range(n_sites):
            pos += int(rng.integers(200, 6000))
            typ = "DEL" if rng.integers(2) else "INS"
            ln = int(rng.choice([35, 60, 120, 200, 260]))
            seq = "".join(rng.choice(list("ACGT"), ln))
            copies = int(rng.choice([1, 1, 2, 3, 4]))
            for c in range(copies):                      # redundant calls of one event: jittered position / length / sequence
                p = pos + (0 if c == 0 else int(rng.integers(-700, 700)) * (6 if typ == "DEL" and rng.integers(3) == 0 else 1))
                while (chrom, p) in used:
                    p += 1
                used.add((chrom, p))
                if typ == "INS":
                    s = seq if c == 0 else mutate(rng, seq, float(rng.choice([0.05, 0.3, 0.9])))
                    s = s[: max(31, int(len(s) * float(rng.choice([1.0, 1.0, 0.7, 0.45]))) - c)]
                    while len(s) in used_len["INS"]:
                        s = s + str(rng.choice(list("ACGT")))
                    used_len["INS"].add(len(s))
                    ref, alt = ("a", "a" + s.lower()) if rng.integers(4) == 0 else ("A", "A" + s)
                else:
                    l2 = max(31, int(ln * float(rng.choice([1.0, 0.9, 0.5, 0.08]))) - c)
                    while l2 in used_len["DEL"]:
                        l2 += 1
                    used_len["DEL"].add(l2)
                    ref, alt = "A" + "".join(rng.choice(list("ACGT"), l2)), "A"
                body.append("%s\t%d\tvolcano%d\t%s\t%s\t.\tPASS\tSVTYPE=%s;SVLEN=%d;TIG=t%d\tGT\t%s\n" %
                            (chrom, p, k, ref, alt, typ, len(alt) - len(ref), k, "0/1" if rng.integers(2) else "1/1"))
                k += 1
    rng.shuffle(body)
    return lines + body


def main():
    ns = load_functions()
    out = []
    for name, seed, n in (("a", 11, 60), ("b", 12, 90)):
      if True:
        lines = make_vcf(seed, n)
        with tempfile.TemporaryDirectory() as d:
            vp = os.path.join(d, "in.vcf")
            open(vp, "w").writelines(lines)
            del_sig, ins_sig, vcf_dc, header = ns["vcf_to_sig"](vp)
            ns["vcf_dc"] = vcf_dc                               # write_vcf reads the module-level name (RR:232)
            links_del_chr1 = ns["match_del_chr"]([s for s in del_sig if s[0] == "chr1"], 3000, 0.1, 0)
            links_ins_chr1 = ns["match_ins_chr"]([s for s in ins_sig if s[0] == "chr1"], 500, 0.5, 0.5)
            nodes_del = ns["match_del"](del_sig, 3000, 0.1, 0)
            nodes_ins = ns["match_ins"](ins_sig, 500, 0.5, 0.5)
            tie = False
            for nodes in nodes_del + nodes_ins:                 # tie-free check (see the header)
                ll = [abs(len(vcf_dc[i][3]) - len(vcf_dc[i][4])) for i in nodes]
                tie |= ll.count(max(ll)) != 1
            assert not tie
            r_del, x_del = ns["pick_best_sv"](vcf_dc, nodes_del)
            r_ins, x_ins = ns["pick_best_sv"](vcf_dc, nodes_ins)
            ns["write_vcf"](d, "volcano_variant", header, r_del, x_del, r_ins, x_ins)
            nrd = open(os.path.join(d, "volcano_variant_no_redundancy.vcf")).read()
            rd = open(os.path.join(d, "volcano_variant_redundancy.vcf")).read()
        out.append(dict(name=name, vcf=lines, links_del_chr1=[list(l) for l in links_del_chr1], links_ins_chr1=[list(l) for l in links_ins_chr1],
                        nodes_del=[sorted(n) for n in nodes_del], nodes_ins=[sorted(n) for n in nodes_ins], no_redundancy=nrd, redundancy=rd))
        print(name, "seed", seed, "calls", len(lines) - 4, "del comps", len(nodes_del), "ins comps", len(nodes_ins), "kept", nrd.count("\n") - 5, "removed", rd.count("\n") - 5)
    # edit-distance known answers from the same DP (the stand-in the reference functions ran with)
    rng = np.random.default_rng(5)
    ed = []
    for _ in range(40):
        a = "".join(rng.choice(list("ACGTN"), int(rng.integers(1, 400))))
        b = mutate(rng, a, float(rng.choice([0.0, 0.1, 0.5]))) if rng.integers(3) else "".join(rng.choice(list("ACGT"), int(rng.integers(1, 400))))
        ed.append([a, b, levenshtein(a, b)])
    ed += [["A", "A", 0], ["A", "C", 1], ["ACGT", "", 4], ["KITTEN", "SITTING", 3]]
    with open(os.path.join(HERE, "remove_redundancy.json"), "w") as f:
        json.dump(dict(cases=out, edit_distance=ed), f)


if __name__ == "__main__":
    main()
