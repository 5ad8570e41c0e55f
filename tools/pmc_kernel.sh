#!/bin/bash
# per-dispatch PMC counters of one kernel (run on the GPU box): tools/pmc_kernel.sh <tag> <kernel substring> "<counters>" <bench.py args...>
tag=$1; kern=$2; ctrs=$3; shift 3
export TMPDIR=/tmp
out=/tmp/pmc_$tag
rm -rf $out; mkdir -p gpurun_out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- python3 bench.py "$@" --steps 2 --warmup 2 --reps 1 --extras none --cpu-sample 0 > gpurun_out/${tag}_pmc.log 2>&1 || { tail -20 gpurun_out/${tag}_pmc.log; exit 1; }
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s n=%d last=%.4g mean=%.4g" % (k, len(v), v[-1], sum(v) / len(v)))
PY
rm -rf $out
