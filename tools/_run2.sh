export VSV_DEBUG=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or random or split_overlap or edge or multi_tid or engines or full_size_config3 or config4_shape" > gpurun_out/r4r_tests.log 2>&1 || { tail -60 gpurun_out/r4r_tests.log; exit 1; }
tail -3 gpurun_out/r4r_tests.log
for v in A B A B; do
  if [ $v = A ]; then export VSV_LIB=$PWD/volcanosv_amd/libvolcanosv_hip_A.so; else unset VSV_LIB; fi
  for c in 2 3; do
  timeout -k 10 400 python bench.py --config $c --steps $([ $c = 3 ] && echo 10 || echo 50) --extras none --cpu-sample 0 --reps 3 > gpurun_out/r4r_bench$c.json 2> gpurun_out/r4r_bench$c.err; python -c "
import json
d=json.load(open('gpurun_out/r4r_bench$c.json'))
print('$v config$c: 4 engines ms/step %.4f (%.4f..%.4f) single %.4f' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['single_engine_ms_per_step']), 'roofline %.3f' % d['roofline']['frac'], 'reruns', d['reruns'])
"
  done
done
