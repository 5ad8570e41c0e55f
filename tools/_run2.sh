export VSV_DEBUG=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or cold or staged or split_overlap or engines_in_flight or clr" > gpurun_out/r4m_tests.log 2>&1 || { tail -60 gpurun_out/r4m_tests.log; exit 1; }
tail -3 gpurun_out/r4m_tests.log
for c in 2 3 6; do
timeout -k 10 400 python bench.py --config $c --steps $([ $c = 3 ] && echo 10 || echo 50) --extras none --cpu-sample 0 --reps 3 > gpurun_out/r4m_bench$c.json 2> gpurun_out/r4m_bench$c.err; python -c "
import json
d=json.load(open('gpurun_out/r4m_bench$c.json'))
print('config$c: 4 engines ms/step %.4f (%.4f..%.4f) single %.4f cold %.4f ratio %.3f' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['single_engine_ms_per_step'], d['cold_ms_per_step'], d['cold_ms_per_step']/d['single_engine_ms_per_step']), d['path'], d['cold_path'], 'roofline %.3f' % d['roofline']['frac'], 'reruns', d['reruns'])
"
done
