"""Multi-process path on CPU: chromosome sharding (LPT), index broadcast and call-table gather with the gloo backend,
world_size 2 — the same code bench.py runs over RCCL with one rank per GPU."""
import os
import socket
import subprocess
import sys

from volcanosv_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_assign_balances_and_is_deterministic():
    hg19 = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747,
            135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566]
    owner = shard.lpt_assign(hg19, 8)
    load = [sum(l for l, o in zip(hg19, owner) if o == r) for r in range(8)]
    assert max(load) / (sum(hg19) / 8) < 1.1            # LPT imbalance < 1.1 on hg19
    assert owner == shard.lpt_assign(hg19, 8)
    assert shard.lpt_assign([5, 5, 5], 1) == [0, 0, 0]
    eq = shard.lpt_assign([20] * 22, 8)                  # config 4: 22 equal chromosomes -> at most 3 per rank
    assert max(eq.count(r) for r in range(8)) == 3


def test_world_size_2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_shard_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARD_OK" in out.stdout and "GATHER_OK" in out.stdout and "DEFERRED_OK" in out.stdout


def test_bnd_cross_rank_join_world_size_2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_bnd_shard_worker.py")]
    out = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "BND_SHARD_OK" in out.stdout
