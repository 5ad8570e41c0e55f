export VSV_DEBUG=1 PYTHONUNBUFFERED=1
for v in A B A B; do
if [ $v = A ]; then export VSV_LIB=$PWD/volcanosv_amd/libvolcanosv_hip_A.so; else unset VSV_LIB; fi
python3 bench.py --extras none --cpu-sample 0 > gpurun_out/r4k_bench2_$v.json 2> gpurun_out/r4k_bench2.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4k_bench2_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["single_engine_ms_per_step"])
PY
done
