#!/usr/bin/env python3
"""Generates tests/golden/signature_support.json from the REFERENCE's calculate_signature_support.py functions
(build container only): load_vcf / load_sig on generated text, calc_ins_call_cov, calc_del_call_cov, and the CSV the script
body writes (pandas). Top-level FunctionDefs are AST-extracted (the script runs argparse + the whole job at import).

calc_del_call_cov orders the call regions with numpy's default (unstable) argsort and then mixes sorted and original call
indices (CS:171, 213-241); on a position-sorted VCF without equal starts both index spaces coincide. The "tiefree" case
uses the unmodified numpy; the "ties" case (several calls per start) injects the same stable-argsort proxy as
make_golden.py — the declared canonical order.

Usage:  python tests/golden/make_golden_cov.py            (needs /root/reference)
"""
import ast
import json
import os
import tempfile
from collections import defaultdict

import numpy as np
import pandas as pd

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))


class StableNumpy:
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, *args, **kw):
        kw["kind"] = "stable"
        return np.argsort(a, *args, **kw)


def load_functions(stable, flanking=1000, min_size=30):
    tree = ast.parse(open(os.path.join(LI, "calculate_signature_support.py")).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef)], type_ignores=[])
    ns = {"np": StableNumpy() if stable else np, "defaultdict": defaultdict, "flanking": flanking, "min_size": min_size}
    exec(compile(mod, "calculate_signature_support.py", "exec"), ns)
    return ns


def make_text(seed, n_calls, n_sigs, span, ties):
    rng = np.random.default_rng(seed)
    sizes = np.array([30, 31, 45, 60, 100, 250, 251, 400, 1000, 3000, 12000])
    if ties:
        cpos = np.sort(rng.choice(rng.integers(2000, span, n_calls // 3), n_calls))
    else:
        cpos = np.sort(rng.choice(np.arange(2000, span), n_calls, replace=False))
    vcf = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"]
    for i, p in enumerate(cpos):
        t = "DEL" if rng.integers(2) else "INS"
        ln = int(rng.choice(np.concatenate([sizes, [10, 29]])))          # some below min_size: dropped by load_vcf
        sv = -ln if t == "DEL" else ln
        vcf.append("chr1\t%d\tvolcano%d\tN\tN\t.\tPASS\tSVTYPE=%s;SVLEN=%d;TIG=x\tGT\t%s\n" % (p, i, t, sv, "0/1" if rng.integers(2) else "1/1"))
    sig = {"INS": [], "DEL": []}
    for t in ("INS", "DEL"):
        spos = np.sort(rng.integers(0, span + 3000, n_sigs))
        # cluster some signatures exactly at the +-flanking edges of calls
        edge = []
        for p in cpos[:: max(1, n_calls // 25)]:
            for d in (-1001, -1000, -999, 0, 999, 1000, 1001):
                edge.append(int(p) + d)
        spos = np.sort(np.concatenate([spos, np.array(edge, dtype=np.int64)]))
        for k, p in enumerate(spos):
            ln = int(rng.choice(sizes))
            if t == "INS":
                sig[t].append("INS\tchr1\t%d\t%d\tread%d\tACGT\n" % (p, ln, k))
            else:
                sig[t].append("DEL\tchr1\t%d\t%d\tread%d\n" % (p, ln, k))
    return vcf, sig


def jsonable(d, tup=False):
    return [[list(k) if tup else int(k), float(v)] for k, v in d.items()]


def main():
    out = []
    for name, seed, nc, nsig, span, ties in [("tiefree", 1, 300, 2500, 600_000, False), ("ties", 2, 300, 2500, 300_000, True),
                                             ("sparse", 3, 120, 200, 5_000_000, False)]:
        ns = load_functions(stable=ties)
        vcf, sig = make_text(seed, nc, nsig, span, ties)
        with tempfile.TemporaryDirectory() as d:
            vp = os.path.join(d, "calls.vcf")
            open(vp, "w").writelines(vcf)
            open(os.path.join(d, "INS.sigs"), "w").writelines(sig["INS"])
            open(os.path.join(d, "DEL.sigs"), "w").writelines(sig["DEL"])
            dc_sig_ins = ns["load_sig"](os.path.join(d, "INS.sigs"), "INS")
            dc_sig_del = ns["load_sig"](os.path.join(d, "DEL.sigs"), "DEL")
            dc_call_ins = ns["load_vcf"](vp, "INS")
            dc_call_del = ns["load_vcf"](vp, "DEL")
            ins = ns["calc_ins_call_cov"](dc_call_ins["chr1"], dc_sig_ins["chr1"])
            dele = ns["calc_del_call_cov"](dc_call_del["chr1"], dc_sig_del["chr1"])
            # script body with -chr 1 (CS:330-378)
            final_info = []
            for call in dc_call_ins["chr1"]:
                start, end, svlen, svid, gt, svtype = call
                final_info.append([start, end, svlen, svid, gt, svtype, ins[start] if start in ins else 0])
            for call in dc_call_del["chr1"]:
                start, end, svlen, svid, gt, svtype = call
                final_info.append([start, end, svlen, svid, gt, svtype, dele[(start, end)] if (start, end) in dele else 0])
            df = pd.DataFrame(final_info, columns=['start', 'end', 'svlen', 'svid', 'gt', 'svtype', 'cov'])
            df['rel_cov'] = df['cov'] / df['svlen']
            cp = os.path.join(d, "o.csv")
            df.to_csv(cp, index=False)
            csv_text = open(cp).read()
            # filter_vcf_by_sig_cov_insdel.py on that CSV: the whole script is run as a subprocess with the reference's own
            # filter_para.csv (it has no functions to extract); kept ids per (dtype, vtype)
            import subprocess, sys
            os.rename(cp, os.path.join(d, "calls_cutesv_sig_support_mins30_fl1000.csv"))
            kept = {}
            for dt in ("hifi", "ont"):
                for vt in ("DEL", "INSDEL"):
                    subprocess.check_call([sys.executable, os.path.join(LI, "filter_vcf_by_sig_cov_insdel.py"), "-i", vp, "-d", dt, "-a", "volcano", "-v", vt])
                    kept["%s_%s" % (dt, vt)] = [l.split()[2] for l in open(vp.replace(".vcf", "_filter_%s.vcf" % vt)) if l[0] != "#"]
        out.append(dict(name=name, cov_filter_kept=kept, flanking=1000, min_size=30, vcf=vcf, ins_sigs=sig["INS"], del_sigs=sig["DEL"],
                        call_ins=[list(c) for c in dc_call_ins["chr1"]], call_del=[list(c) for c in dc_call_del["chr1"]],
                        ins_cov=jsonable(ins), del_cov=jsonable(dele, True), csv=csv_text))
        print(name, "ins calls", len(dc_call_ins["chr1"]), "del calls", len(dc_call_del["chr1"]), "nonzero ins", sum(1 for v in ins.values() if v),
              "del entries", len(dele))
    with open(os.path.join(HERE, "signature_support.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
