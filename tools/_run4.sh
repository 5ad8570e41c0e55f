export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "golden or synthetic_vs_oracle or multi_tid or bucket_sort_overflow or random_small or clr_ or position_buckets or cold_engine or tid_hint or staged or split_overlap or full_size_config2" --durations=5 --timeout=300 --timeout-method=thread > gpurun_out/r4z_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4z_tests.log | tail -40; exit 1; }
tail -8 gpurun_out/r4z_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4z_c2 --config 2 --streams 1 | head -70
VSV_BK_SLOTS=0 python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4z_bench2_old.json 2> gpurun_out/r4z_bench2.err
python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4z_bench2.json 2>> gpurun_out/r4z_bench2.err
VSV_BK_SLOTS=0 python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4z_bench2_old2.json 2>> gpurun_out/r4z_bench2.err
python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4z_bench2_2.json 2>> gpurun_out/r4z_bench2.err
python3 - <<'PY'
import json
for f in ["r4z_bench2_old","r4z_bench2","r4z_bench2_old2","r4z_bench2_2"]:
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["single_engine_ms_per_step"], d["cold_ms_per_step"], d["reruns"])
PY
