"""Multi-GPU sharding of the path: one process per GPU, chromosomes (tids) are the shard unit.

The reference scales the same way — one OS process per chromosome, joblib.Parallel over range(22)
(Large_INDEL/volcanosv-vc-large-indel.py:154-186, 268) — and its processes exchange data only through files.
Here ranks exchange two things over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests): the small reference index (contig names/lengths + per-tid record counts) is broadcast
from rank 0, and the per-rank call tables are gathered to rank 0 at the end. There is no collective on the
data path: signatures never cross chromosomes (fetch(chr_name), extract_contig_signature_Hifi.py:391).
"""
import numpy as np
import torch
import torch.distributed as dist

from .abi import CALL_DTYPE


def lpt_assign(weights, n_ranks):
    """Longest-processing-time assignment of chromosomes to ranks. weights[t] = records (or bases) of tid t.
    Returns owner[t]. Deterministic: ties go to the lowest rank."""
    order = sorted(range(len(weights)), key=lambda t: (-int(weights[t]), t))
    load = [0] * n_ranks
    owner = [0] * len(weights)
    for t in order:
        r = min(range(n_ranks), key=lambda q: (load[q], q))
        owner[t] = r
        load[r] += int(weights[t])
    return owner


def broadcast_index(index, device):
    """index: int64 tensor [n_tid, k] (e.g. contig length, record count) valid on rank 0. Returns it on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return index
    shape = torch.tensor(list(index.shape) if dist.get_rank() == 0 else [0, 0], dtype=torch.int64, device=device)
    dist.broadcast(shape, 0)
    buf = index.to(device) if dist.get_rank() == 0 else torch.empty(tuple(shape.tolist()), dtype=torch.int64, device=device)
    dist.broadcast(buf, 0)
    return buf


def gather_calls(calls, device):
    """calls: numpy structured array (CALL_DTYPE) of this rank. Returns on rank 0 the concatenation over ranks sorted by
    (tid, pos) (stable, rank order breaks ties; tids are disjoint across ranks so the order is unique), None elsewhere.
    Counts go through an all_gather, rows through one padded all_gather of raw bytes (rows are 48 B, tables are
    KB-MB: latency-bound, so a single collective beats per-peer send/recv rings on xGMI)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return calls
    world = dist.get_world_size()
    n = torch.tensor([len(calls)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    row = CALL_DTYPE.itemsize
    mine = torch.zeros(mx * row, dtype=torch.uint8, device=device)
    if len(calls):
        mine[: len(calls) * row] = torch.from_numpy(np.frombuffer(calls.tobytes(), dtype=np.uint8).copy()).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    if dist.get_rank() != 0:
        return None
    parts = [np.frombuffer(bufs[r][: counts[r] * row].cpu().numpy().tobytes(), dtype=CALL_DTYPE) for r in range(world)]
    allc = np.concatenate(parts) if parts else np.zeros(0, CALL_DTYPE)
    key = (allc["sig"]["tid"].astype(np.int64) << 32) | (allc["sig"]["pos"].astype(np.int64) & 0xFFFFFFFF)
    return allc[np.argsort(key, kind="stable")]
