export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "position_buckets or cold_engine or full_size_row2c or staged" --timeout=300 --timeout-method=thread > gpurun_out/r4c_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4c_tests.log | tail -40; exit 1; }
tail -3 gpurun_out/r4c_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4c_c6 --config 6 --streams 1 | head -24
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4c_bench6.json 2> gpurun_out/r4c_bench6.err
python3 - <<'PY'
import json
for f in ["r4c_bench6"]:
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["single_engine_ms_per_step"], d["cold_ms_per_step"], d["reruns"], d["roofline"]["avg_launch_ms"])
PY
