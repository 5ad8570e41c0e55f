import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volcanosv_amd import synth
from volcanosv_amd.abi import DTYPE_BY_NAME
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
t, nq, nt = synth.generate(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, "hifi", seed=5, device="cuda")
dr = DeviceRecords(t, nq, nt); eng = Engine(0); p = default_params(DTYPE_BY_NAME[sys.argv[2] if len(sys.argv) > 2 else 'Hifi'])
for _ in range(3): eng.run(dr, p)
enq=[]; fin=[]
for _ in range(10):
    torch.cuda.synchronize(); a=time.perf_counter(); eng.run_async(dr,p); b=time.perf_counter(); eng.finish(); c=time.perf_counter()
    enq.append((b-a)*1e3); fin.append((c-b)*1e3)
print("enqueue ms", sorted(enq)[5], "finish ms", sorted(fin)[5])
