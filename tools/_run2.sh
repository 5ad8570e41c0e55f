export VSV_DEBUG=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or random or long_records or edge or restaging or staged or cold_engine or dense" > gpurun_out/r4j_tests.log 2>&1 || { tail -60 gpurun_out/r4j_tests.log; exit 1; }
tail -3 gpurun_out/r4j_tests.log
run() { # label env...
  lbl=$1; shift
  env "$@" timeout -k 10 150 python bench.py --config 6 --streams 1 --steps 6 --warmup 2 --reps 1 --extras none --cpu-sample 0 2> gpurun_out/r4j_$lbl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$lbl', 'ms_per_step', round(d['ms_per_step'],3), 'scan_ms', round(r.get('avg_launch_ms', 0),3), 'frac', round(r['frac'],3), 'reruns', d.get('reruns'))
"
}
run new
VSV_SPLIT_STREAM=main timeout -k 10 200 tools/prof_step.sh r4j_new --config 6 --streams 1 > /dev/null && head -14 gpurun_out/r4j_new_step.txt
timeout -k 10 300 python bench.py --config 6 --extras none --cpu-sample 0 --reps 3 > gpurun_out/r4j_bench6.json 2> gpurun_out/r4j_bench6.err; python -c "
import json
d=json.load(open('gpurun_out/r4j_bench6.json'))
print('4 engines ms/step', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], 'single', d['single_engine_ms_per_step'], 'value', d['value'], 'frac', d['config']['whole_path_frac_of_hbm_peak'], 'roofline', d['roofline']['frac'])
"
