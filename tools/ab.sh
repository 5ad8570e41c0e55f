#!/bin/bash
# A/B timing of two builds on the SAME box (boxes differ by +-5 %): tools/ab.sh <kernel substring> <bench args...>
# expects volcanosv_amd/libvolcanosv_hip.so (B) and volcanosv_amd/libvolcanosv_hip_A.so (A, a copy of the earlier build)
kern=$1; shift
export VSV_DEBUG=1
for v in A B A B; do
  if [ $v = A ]; then export VSV_LIB=$PWD/volcanosv_amd/libvolcanosv_hip_A.so; else unset VSV_LIB; fi
  tools/prof_step.sh ab_$v "$@" > /dev/null
  echo "$v $(grep "$kern" gpurun_out/ab_${v}_step.txt | head -1)  span: $(head -1 gpurun_out/ab_${v}_step.txt | sed 's/.*busy, //')"
done
