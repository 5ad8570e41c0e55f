#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (copied back under gpurun_out/) into the small files kept under profiles/.

  pmc    --fetch DIR --write DIR [--sq DIR] --kernel NAME --records N --ops N --sigs N --out FILE
         per-dispatch FETCH_SIZE / WRITE_SIZE of one kernel from separate --pmc passes; HBM bytes per launch with the
         gfx950 corrections of MI355X_MICROARCH.md (KiB units; FETCH_SIZE doubled for a wide coalesced stream).
  trace  --dir DIR --steps K --out-csv FILE --out-txt FILE
         kernel_stats.csv copy + one-step breakdown (launches, busy time) from kernel_trace.csv.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def counter_rows(d, kernel):
    rows = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:60]


def cmd_pmc(a):
    fetch = counter_rows(a.fetch, a.kernel).get("FETCH_SIZE", [])
    write = counter_rows(a.write, a.kernel).get("WRITE_SIZE", [])
    if not fetch or not write:
        sys.exit("no FETCH_SIZE/WRITE_SIZE rows for kernel %r" % a.kernel)
    fetch, write = fetch[a.skip:], write[a.skip:]
    out = {
        "kernel": a.label or a.kernel,
        "workload": a.workload,
        "records": a.records, "n_ops": a.ops,
        "collection": "separate rocprofv3 passes: --pmc FETCH_SIZE ; --pmc WRITE_SIZE%s (each with --kernel-trace only)" % (" ; SQ_*" if a.sq else ""),
        "units": "FETCH_SIZE/WRITE_SIZE are KiB (MI355X_MICROARCH.md HBM section: bytes = counter*1024); FETCH_SIZE reads "
                 "exactly half of a wide coalesced 16 B/lane stream on gfx950 -> doubled",
        "FETCH_SIZE_per_dispatch": fetch, "WRITE_SIZE_per_dispatch": write,
    }
    if a.sq:
        sq = counter_rows(a.sq, a.kernel)
        out["SQ_counters_mean"] = {k: sum(v[a.skip:]) / max(1, len(v[a.skip:])) for k, v in sorted(sq.items())}
    fb = 2 * 1024 * sum(fetch) / len(fetch)
    wb = 1024 * sum(write) / len(write)
    alg = 24 * a.records + 4 * a.ops + 32 * a.sigs
    out.update(fetch_bytes_corrected=fb, write_bytes=wb, traffic_bytes_per_launch=fb + wb, algorithmic_bytes_per_launch=alg,
               traffic_over_algorithmic=(fb + wb) / alg)
    # which version of the kernel's source this was taken from: bench.py refuses the lookup for any other (the GPU box has no .git)
    import hashlib
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volcanosv_amd", "csrc", "cigar_scan.hip")
    out["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    json.dump(out, open(a.out, "w"), indent=1)
    print("traffic %.4f GB per launch = %.3f x algorithmic" % ((fb + wb) / 1e9, (fb + wb) / alg))


def cmd_trace(a):
    stats = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
    trace = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
    if not stats or not trace:
        sys.exit("no kernel_stats/kernel_trace under " + a.dir)
    shutil.copy(stats[-1], a.out_csv)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace[-1]))]
    rows.sort()
    # one step = [partition_search of step i, partition_search of step i + 1)
    starts = [i for i, r in enumerate(rows) if "partition_search" in r[2] or "partition_kernel" in r[2]]
    if len(starts) < 2:
        sys.exit("fewer than two steps in the trace")
    lo, hi = starts[-2], starts[-1]
    step = rows[lo:hi]
    agg = collections.OrderedDict()
    for s, e, n in step:
        k = short(n)
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
    busy = sum(v[1] for v in agg.values())
    span = (step[-1][1] - step[0][0]) / 1e3
    with open(a.out_txt, "w") as f:
        f.write("%s: %d launches, %.1f us busy, %.1f us span\n" % (a.title, len(step), busy, span))
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write("%-44s n=%3d us=%9.1f\n" % (k, n, us))
        if a.timeline:
            f.write("\nin launch order: start (us since the step began), duration, idle time in front of the launch\n")
            t0, prev = step[0][0], step[0][0]
            for s, e, n in step:
                f.write("%9.1f %8.1f %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev) / 1e3, short(n)))
                prev = max(prev, e)
    print(open(a.out_txt).read())


ap = argparse.ArgumentParser()
sub = ap.add_subparsers(dest="cmd", required=True)
p = sub.add_parser("pmc")
p.add_argument("--fetch", required=True); p.add_argument("--write", required=True); p.add_argument("--sq")
p.add_argument("--kernel", default="cigar_scan_emit"); p.add_argument("--label"); p.add_argument("--workload", default="")
p.add_argument("--records", type=int, required=True); p.add_argument("--ops", type=int, required=True); p.add_argument("--sigs", type=int, required=True)
p.add_argument("--skip", type=int, default=0); p.add_argument("--out", required=True)
p = sub.add_parser("trace")
p.add_argument("--dir", required=True); p.add_argument("--out-csv", required=True); p.add_argument("--out-txt", required=True)
p.add_argument("--timeline", action="store_true", help="append the launches of the step in order with their gaps")
p.add_argument("--title", default="one step of bench.py config 2 (10 M records), kernels in launch-aggregated form")
a = ap.parse_args()
{"pmc": cmd_pmc, "trace": cmd_trace}[a.cmd](a)
