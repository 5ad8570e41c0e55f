export VSV_DEBUG=1 PYTHONUNBUFFERED=1
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
head -12 gpurun_out/r04_row2c_contig200k_step.txt
