#!/usr/bin/env python3
"""Generates tests/golden/fp_filter_*.json from the REFERENCE's FP_filter_v1.py functions (build container only).

The script runs argparse and a main loop at import time, so — like make_golden.py — its top-level FunctionDefs are
extracted with `ast`, compiled and called. The fixtures hold only inputs (call / read-signature lists, VCF lines) and
the outputs `eval_sig` / `filter_vcf` returned.

Usage:  python tests/golden/make_golden_fp.py            (needs /root/reference)
"""
import ast
import json
import os
import tempfile

import numpy as np

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_functions():
    tree = ast.parse(open(os.path.join(LI, "FP_filter_v1.py")).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef)], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, "FP_filter_v1.py", "exec"), ns)
    return ns


def make_lists(seed, n_calls, n_sigs, span, boundary=False):
    rng = np.random.default_rng(seed)
    sizes = np.array([0, 1, 30, 45, 50, 60, 99, 100, 120, 125, 200, 249, 250, 251, 300, 500, 1200])
    cpos = np.sort(rng.integers(1000, span, n_calls))
    calls = [["chr1", "DEL" if rng.integers(2) else "INS", int(p), int(rng.choice(sizes))] for p in cpos]
    spos = rng.integers(0, span + 2000, n_sigs)
    if boundary:  # read signatures exactly at the +-max_dist / +-max_shift edges of some calls, sizes at the 0.5 edge
        extra = []
        for c in calls[:: max(1, n_calls // 40)]:
            for d in (-1001, -1000, -501, -500, 0, 500, 501, 1000, 1001):
                for f in (0.5, 2.0, 1.0):
                    extra.append((c[2] + d, int(c[3] * f)))
                extra.append((c[2] + d, max(0, int(c[3] * 0.5) - 1)))
                extra.append((c[2] + d, 2 * c[3] + 1))
        spos = np.concatenate([spos, np.array([e[0] for e in extra], dtype=np.int64)])
        slen = np.concatenate([rng.choice(sizes, n_sigs), np.array([e[1] for e in extra], dtype=np.int64)])
    else:
        slen = rng.choice(sizes, n_sigs)
    order = np.argsort(spos, kind="stable")
    sigs = [["chr1", "DEL" if rng.integers(2) else "INS", int(spos[i]), int(slen[i]), "read%d" % i] for i in order]
    return calls, sigs


def main():
    ns = load_functions()
    cases = []
    for name, seed, nc, nsig, span, boundary, prm in [
        ("sparse", 1, 300, 900, 2_000_000, False, (1000, 250, 500, 0.5)),
        ("dense", 2, 400, 6000, 300_000, False, (1000, 250, 500, 0.5)),
        ("edges", 3, 200, 500, 400_000, True, (1000, 250, 500, 0.5)),
        ("params", 4, 300, 3000, 200_000, True, (700, 300, 350, 0.3)),       # eval_sig's own defaults 300 / 0.3
        ("empty_sigs", 5, 20, 0, 100_000, False, (1000, 250, 500, 0.5)),
    ]:
        calls, sigs = make_lists(seed, nc, nsig, span, boundary)
        max_dist, max_comp, max_shift, min_sim = prm
        sup = ns["eval_sig"](calls, sigs, max_dist, max_comp, max_shift, min_sim)
        cases.append(dict(name=name, params=dict(max_dist=max_dist, max_comp_svlen=max_comp, max_shift=max_shift, min_size_sim=min_sim),
                          calls=[[c[2], c[3]] for c in calls], sigs=[[s[2], s[3]] for s in sigs], support=[int(x) for x in sup]))
    # filter_vcf on VCF text + a reads_sig file (Large_INDEL/FP_filter_v1.py:135-147)
    rng = np.random.default_rng(9)
    calls, sigs = make_lists(7, 120, 700, 250_000, True)
    lines = []
    for c in calls:
        if c[1] == "DEL":
            ref, alt = "A" + "C" * c[3], "A"
        else:
            ref, alt = "A", "A" + "G" * c[3]
        lines.append("chr1\t%d\tvolcano%d\t%s\t%s\t.\tPASS\tSVTYPE=%s;SVLEN=%d\tGT\t%s\n" % (c[2], len(lines), ref, alt, c[1], c[3], "0/1" if rng.integers(2) else "1/1"))
    with tempfile.TemporaryDirectory() as d:
        sp = os.path.join(d, "chr1_reads_sig.txt")
        with open(sp, "w") as f:
            for s in sigs:
                f.write("\t".join(str(x) for x in s) + "\n")
        kept = ns["filter_vcf"](lines, "chr1", sp, 1000, 250, 500, 0.5)
    out = dict(eval_sig=cases, filter_vcf=dict(vcf_lines=lines, reads_sig_lines=["\t".join(str(x) for x in s) + "\n" for s in sigs],
                                               kept=[str(x) for x in kept]))
    with open(os.path.join(HERE, "fp_filter.json"), "w") as f:
        json.dump(out, f)
    for c in cases:
        print(c["name"], len(c["calls"]), len(c["sigs"]), "zero-support:", sum(1 for x in c["support"] if x == 0), "max:", max(c["support"]))
    print("filter_vcf kept", len(kept), "of", len(lines))


if __name__ == "__main__":
    main()
