export VSV_DEBUG=1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or random or long_records or edge or restaging or error or capacity or multi_tid" > gpurun_out/r4g_tests.log 2>&1 || { tail -40 gpurun_out/r4g_tests.log; exit 1; }
tail -2 gpurun_out/r4g_tests.log
run() { # label env...
  lbl=$1; shift
  env "$@" timeout -k 10 150 python bench.py --config 6 --streams 1 --steps 6 --warmup 2 --reps 1 --extras none --cpu-sample 0 2> gpurun_out/r4g_$lbl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$lbl', 'ms_per_step', round(d['ms_per_step'],3), 'scan_ms', round(r.get('avg_launch_ms', 0),3), 'frac', round(r['frac'],3), 'reruns', d.get('reruns'))
"
}
run pool VSV_K1L=pool
run new
run nofast VSV_K1_ABLATE=2
run noemit VSV_K1_ABLATE=1
run new2
