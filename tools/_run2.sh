export VSV_DEBUG=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cold_engine or full_size_row2c or long_records or split_overlap or dense" > gpurun_out/r4s_tests.log 2>&1 || { tail -60 gpurun_out/r4s_tests.log; exit 1; }
tail -3 gpurun_out/r4s_tests.log
VSV_SPLIT_STREAM=main timeout -k 10 200 tools/prof_step.sh r4s_c6 --config 6 --streams 1 > /dev/null && head -3 gpurun_out/r4s_c6_step.txt
for c in 6 6; do
timeout -k 10 400 python bench.py --config $c --extras none --cpu-sample 0 --reps 3 > gpurun_out/r4s_bench$c.json 2> gpurun_out/r4s_bench$c.err; python -c "
import json
d=json.load(open('gpurun_out/r4s_bench$c.json'))
print('config$c: 4 engines ms/step %.4f (%.4f..%.4f) single %.4f cold %.4f' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['single_engine_ms_per_step'], d['cold_ms_per_step']), 'roofline %.3f' % d['roofline']['frac'], 'reruns', d['reruns'])
"
done
