export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "full_size_row2c or cold_engine or element_path or dense_runs or pairing_in_rounds or full_size_config3" --timeout=400 --timeout-method=thread --durations=5 > gpurun_out/r4d_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4d_tests.log | tail -40; exit 1; }
tail -12 gpurun_out/r4d_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4d_c6 --config 6 --streams 1 | head -8
VSV_PAIR_FORM=wave python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4d_bench6_old.json 2> gpurun_out/r4d_bench6.err
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4d_bench6.json 2>> gpurun_out/r4d_bench6.err
VSV_PAIR_FORM=wave python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4d_bench6_old2.json 2>> gpurun_out/r4d_bench6.err
python3 bench.py --config 6 --extras none --cpu-sample 0 > gpurun_out/r4d_bench6_2.json 2>> gpurun_out/r4d_bench6.err
python3 - <<'PY'
import json
for f in ["r4d_bench6_old","r4d_bench6","r4d_bench6_old2","r4d_bench6_2"]:
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["single_engine_ms_per_step"], d["cold_ms_per_step"], d["reruns"], d["roofline"]["avg_launch_ms"])
PY
