export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "any_query_id or edge_cases or clr_ or full_size_config2 or config4_shape or golden or random_small" --timeout=300 --timeout-method=thread --durations=5 > gpurun_out/r4n_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4n_tests.log | tail -40; exit 1; }
tail -5 gpurun_out/r4n_tests.log
VSV_SPLIT_STREAM=main bash tools/prof_step.sh r4n_c2 --config 2 --streams 1 | sed -n 1,24p
for v in table dense table dense; do
if [ $v = table ]; then export VSV_SPLIT_FIND=table; else unset VSV_SPLIT_FIND; fi
python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4n_bench2_$v.json 2> gpurun_out/r4n_bench2.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4n_bench2_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["single_engine_ms_per_step"], d["reruns"])
PY
done
