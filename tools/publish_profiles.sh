#!/bin/bash
# gpurun_out/<round>_* (tools/collect_profiles.sh) -> profiles/ under the names the rounds keep:   tools/publish_profiles.sh r04
R=${1:-r04}
for f in gpurun_out/${R}_*_stats.csv; do [ -f "$f" ] && cp "$f" profiles/$(basename "${f%_stats.csv}")_kernel_stats.csv; done
for f in gpurun_out/${R}_*_step.txt; do [ -f "$f" ] && cp "$f" profiles/$(basename "${f%_step.txt}")_step_breakdown.txt; done
for f in gpurun_out/${R}_pmc_*.json gpurun_out/${R}_bench_*.json; do [ -f "$f" ] && cp "$f" profiles/; done
ls profiles | grep "^${R}"
