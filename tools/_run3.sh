export VSV_DEBUG=1
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
tail -5 gpurun_out/r04_collect.log
ls gpurun_out | grep "^r04" | head -40
