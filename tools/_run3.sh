export VSV_DEBUG=1
( time timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r4t_tests.log 2>&1 ) 2> gpurun_out/r4t_time.txt || { tail -60 gpurun_out/r4t_tests.log; exit 1; }
tail -14 gpurun_out/r4t_tests.log; cat gpurun_out/r4t_time.txt
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
