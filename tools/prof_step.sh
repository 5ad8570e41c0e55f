#!/bin/bash
# one-step kernel breakdown of a bench workload under rocprofv3 (run on the GPU box through gpurun):
#   tools/prof_step.sh <tag> <bench.py args...>   -> gpurun_out/<tag>_stats.csv, gpurun_out/<tag>_step.txt
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py "$@" --steps 3 --warmup 2 --reps 1 --extras none --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then tail -30 gpurun_out/${tag}_prof.log; rm -rf $out; exit $rc; fi
python3 tools/summarize_rocprof.py trace --dir $out --out-csv gpurun_out/${tag}_stats.csv --out-txt gpurun_out/${tag}_step.txt --timeline --title "$tag: one step" > /dev/null
rc=$?
if [ $rc -ne 0 ]; then find $out | head -20; tail -5 gpurun_out/${tag}_prof.log; fi
rm -rf $out
head -45 gpurun_out/${tag}_step.txt
exit $rc
