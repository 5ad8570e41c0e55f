export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "position_buckets or cold_engine or full_size_row2c" --durations=5 --timeout=300 --timeout-method=thread > gpurun_out/r4x_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4x_tests.log | tail -40; exit 1; }
tail -8 gpurun_out/r4x_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4x_c6 --config 6 --streams 1 | head -75
VSV_SORT1_BITS=-1 python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4x_bench6_passes.json 2> gpurun_out/r4x_bench6.err
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4x_bench6.json 2>> gpurun_out/r4x_bench6.err
VSV_SORT1_BITS=-1 python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4x_bench6_passes2.json 2>> gpurun_out/r4x_bench6.err
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4x_bench6_2.json 2>> gpurun_out/r4x_bench6.err
python3 - <<'PY'
import json
for f in ["r4x_bench6_passes","r4x_bench6","r4x_bench6_passes2","r4x_bench6_2"]:
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["single_engine_ms_per_step"], d["cold_ms_per_step"], d["reruns"])
PY
