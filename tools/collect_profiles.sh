#!/bin/bash
# round profiles (run on the GPU box through gpurun): one-step kernel breakdowns, kernel statistics, PMC traffic of the two scan kernels
# and the bench lines, all into gpurun_out/ under the names profiles/ keeps.   tools/collect_profiles.sh <round tag, e.g. r03>
R=${1:-r04}
export TMPDIR=/tmp VSV_DEBUG=1
mkdir -p gpurun_out
VSV_SPLIT_STREAM=main tools/prof_step.sh ${R}_row2c_contig200k --config 6 --streams 1 > /dev/null
VSV_SPLIT_STREAM=main tools/prof_step.sh ${R}_bench_config2_streams1_mainstream --config 2 --streams 1 > /dev/null
tools/prof_step.sh ${R}_bench_config2_default --config 2 --streams 3 > /dev/null
VSV_SPLIT_STREAM=main tools/prof_step.sh ${R}_config3_ont50m --config 3 --streams 1 > /dev/null
pmc() {  # tag kernel records ops sigs label args...
  tag=$1; kern=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o run -- python3 bench.py "$@" --streams 1 --steps 2 --warmup 2 --reps 1 --extras none --cpu-sample 0 > gpurun_out/${tag}_$c.log 2>&1
  done
}
pmc ${R}_pmc_long cigar_scan_long --config 6
python3 bench.py --config 6 --streams 1 --steps 2 --warmup 2 --reps 1 --extras none --cpu-sample 0 2>/dev/null > gpurun_out/${R}_tmp_c6.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${R}_tmp_c6.json"))
w = d["config"]["workload"]
import re
ops = int(re.search(r"(\d+) CIGAR ops", w).group(1)); sigs = int(re.search(r"(\d+) raw signatures", w).group(1)); recs = d["config"]["records_per_gpu"]
open("gpurun_out/${R}_tmp_c6.args", "w").write("--records %d --ops %d --sigs %d" % (recs, ops, sigs))
PY
python3 tools/summarize_rocprof.py pmc --fetch /tmp/pmc_FETCH_SIZE --write /tmp/pmc_WRITE_SIZE --kernel cigar_scan_long --label "cigar_scan_long<0>" --workload "bench.py --config 6 (200 k contig-like records)" $(cat gpurun_out/${R}_tmp_c6.args) --skip 1 --out gpurun_out/${R}_pmc_cigar_scan_long_contig200k.json
rm -rf /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${R}_tmp_c6.*
python3 bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/${R}_bench_row2c.json 2>> gpurun_out/${R}_bench_default.err
python3 bench.py --config 3 --steps 10 --extras none --cpu-sample 0 > gpurun_out/${R}_bench_config3.json 2>> gpurun_out/${R}_bench_default.err
ls -la gpurun_out | tail -30
