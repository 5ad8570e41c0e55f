import sys, time
sys.path.insert(0, '/root/repo')
import torch
from volcanosv_amd import synth
from volcanosv_amd.abi import DTYPE_ONT
from volcanosv_amd.engine import DeviceRecords, Engine, default_params
t0 = time.time()
t, nq, nt = synth.generate(50_000_000, "ont", seed=20250331, device="cuda")
torch.cuda.synchronize(); print("gen s", time.time() - t0, "ops", t["cigar"].numel(), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
dr = DeviceRecords(t, nq, nt, max_pos=synth.CHR10_LEN + 200000)
eng = Engine(0, max_sigs=1 << 24)
p = default_params(DTYPE_ONT)
t0 = time.time(); eng.run(dr, p); print("run s", time.time() - t0, "scan ms", eng.scan_ms())
t0 = time.time(); eng.run(dr, p); print("run s", time.time() - t0, "scan ms", eng.scan_ms())
tabs = eng.tables(DTYPE_ONT)
print({k: len(v) for k, v in tabs.items()})
b = 24 * dr.n_records + 4 * dr.n_ops + 32 * len(tabs["raw"])
print("K1 GB/s", b / eng.scan_ms() / 1e6)
