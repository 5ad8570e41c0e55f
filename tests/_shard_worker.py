"""Worker for tests/test_shard_gloo.py: world_size-2 gloo rehearsal of the chromosome-sharded job.
The compute on each rank is the CPU oracle (this is a test of the sharding/gather logic, not of the kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from volcanosv_amd import shard, synth  # noqa: E402
from volcanosv_amd.abi import DTYPE_HIFI  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    n_tid = 5
    sizes = [9000, 4000, 7000, 3000, 6000]
    # rank 0 owns the reference index (length, record count per tid) and broadcasts it
    index = torch.tensor([[1500000 + 1000 * t, sizes[t]] for t in range(n_tid)], dtype=torch.int64) if rank == 0 else None
    index = shard.broadcast_index(index if index is not None else torch.zeros(0, 0, dtype=torch.int64), dev)
    assert index.shape == (n_tid, 2) and int(index[3, 1]) == 3000
    owner = shard.lpt_assign([int(x) for x in index[:, 1]], world)
    mine = [t for t in range(n_tid) if owner[t] == rank]
    parts = []
    for t in mine:
        tens, nq, _ = synth.generate(int(index[t, 1]), "hifi", seed=500 + t, tid=t, chrom_len=int(index[t, 0]), events_per_record=0.2, site_step=1000)
        parts.append((tens, nq))
    if parts:
        tens, nq = synth.concat(parts)
        st, tabs = oracle.run(synth.to_soa(tens, nq), dtype=DTYPE_HIFI)
        assert st == 0
        calls = tabs["calls"]
    else:
        calls = np.zeros(0, dtype=tabs_dtype())
    allc = shard.gather_calls(calls, dev)
    if rank == 0:
        # single-process run over all chromosomes
        parts = []
        for t in range(n_tid):
            tens, nq, _ = synth.generate(int(index[t, 1]), "hifi", seed=500 + t, tid=t, chrom_len=int(index[t, 0]), events_per_record=0.2, site_step=1000)
            parts.append((tens, nq))
        tens, nq = synth.concat(parts)
        st, tabs = oracle.run(synth.to_soa(tens, nq), dtype=DTYPE_HIFI)
        want = tabs["calls"]
        assert len(allc) == len(want) and len(want) > 100
        # rec / a / b indices are shard-local; the VCF integer columns must be identical
        for f in ("pos", "svlen", "q_start", "q_end", "meta", "tid"):
            assert np.array_equal(allc["sig"][f], want["sig"][f]), f
        assert np.array_equal(allc["gt"], want["gt"])
        print("SHARD_OK %d calls from %d ranks, owners %s" % (len(allc), world, owner))
    else:
        assert allc is None
    # the gather primitive itself: exact-size transfers to rank 0 only, asynchronous handles, empty senders, several in flight
    mk = lambda r, k: torch.arange(0 if (r + k) % 3 == 0 else 1000 * (r + 1) + k, dtype=torch.int64).to(torch.uint8)
    pend = [shard.gather_bytes_start(mk(rank, k), dev) for k in range(4)]
    for k, h in enumerate(pend):
        bufs, counts = h.wait()
        assert counts == [int(mk(r, k).numel()) for r in range(world)]
        if rank == 0:
            assert all(torch.equal(bufs[r], mk(r, k)) for r in range(world))
        else:
            assert bufs is None                       # nobody but rank 0 receives anything
    # deferred counts: a start enqueues the count all-gather and returns — nothing is known on the host yet; the transfers are posted
    # later, interleaved with the starts of later steps (bench.py: `streams` steps later), in the same order on every rank
    pend = []
    for k in range(6):
        h = shard.gather_bytes_start(mk(rank, k + 2), dev)
        assert h.counts is None and h.bufs is None        # no readback, no transfer yet
        pend.append((k + 2, h))
        if len(pend) > 2:
            kk, hh = pend.pop(0)
            hh.post()                                      # explicit second half, then the wait
            bufs, counts = hh.wait()
            assert counts == [int(mk(r, kk).numel()) for r in range(world)]
            assert bufs is None if rank else all(torch.equal(bufs[r], mk(r, kk)) for r in range(world))
    for kk, hh in pend:
        bufs, counts = hh.wait()
        assert counts == [int(mk(r, kk).numel()) for r in range(world)]
        assert bufs is None if rank else all(torch.equal(bufs[r], mk(r, kk)) for r in range(world))
    if rank == 0:
        print("DEFERRED_OK")
    rows = shard.gather_rows_device(mk(rank, 1), dev)
    assert (rows is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(rows, torch.cat([mk(r, 1) for r in range(world)]))
        print("GATHER_OK")
    dist.barrier()
    dist.destroy_process_group()


def tabs_dtype():
    from volcanosv_amd.abi import CALL_DTYPE
    return CALL_DTYPE


if __name__ == "__main__":
    main()
