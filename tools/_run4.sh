export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "full_size_config3 or element_path_on_the_parity_cases" --timeout=400 --timeout-method=thread --durations=5 > gpurun_out/r4f_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4f_tests.log | tail -40; exit 1; }
tail -8 gpurun_out/r4f_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4f_c3 --config 3 --streams 1 | head -40
VSV_SLIM8=0 python3 bench.py --config 3 --steps 10 --extras none --cpu-sample 0 > gpurun_out/r4f_bench3_old.json 2> gpurun_out/r4f_bench3.err
python3 bench.py --config 3 --steps 10 --extras none --cpu-sample 0 > gpurun_out/r4f_bench3.json 2>> gpurun_out/r4f_bench3.err
python3 - <<'PY'
import json
for f in ["r4f_bench3_old","r4f_bench3"]:
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["single_engine_ms_per_step"], d["cold_ms_per_step"], d["reruns"], d["roofline"]["avg_launch_ms"])
PY
