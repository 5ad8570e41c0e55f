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

from .abi import B_HAP2, BND_DTYPE, CALL_DTYPE


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


class _RootGather:
    """Handle of a variable-length gather to rank 0 in flight (gather_bytes_start): `wait()` completes it and returns, on rank 0,
    (list of per-rank uint8 tensors in rank order, list of byte counts); (None, counts) on the other ranks."""

    def __init__(self, src, device, count_work=None, counts_t=None, keep=None):
        self.src, self.device, self.count_work, self.counts_t, self.keep = src, device, count_work, counts_t, keep
        self.works, self.bufs, self.counts = [], None, None

    def post(self):
        """Second half of the start: the counts have arrived (their all-gather was enqueued by gather_bytes_start), the exact-size
        transfers are posted. Called by wait(); a caller may call it earlier, at a point every rank reaches in the same order."""
        if self.counts is not None:
            return
        if self.count_work is None:                       # single rank
            self.bufs, self.counts = [self.src], [int(self.src.numel())]
            return
        self.count_work.wait()
        self.counts = [int(c) for c in torch.cat(self.counts_t).tolist()]          # ONE readback, long after the collective was enqueued
        world, rank = dist.get_world_size(), dist.get_rank()
        ops = []
        if rank == 0:
            self.bufs = [self.src] + [torch.empty(self.counts[r], dtype=torch.uint8, device=self.device) for r in range(1, world)]
            ops = [dist.P2POp(dist.irecv, self.bufs[r], r) for r in range(1, world) if self.counts[r]]
        elif self.counts[rank]:
            ops = [dist.P2POp(dist.isend, self.src, 0)]
        self.works = dist.batch_isend_irecv(ops) if ops else []
        self.count_work, self.counts_t, self.keep = None, None, None

    def wait(self):
        self.post()
        for w in self.works:
            w.wait()
        self.works = []
        return self.bufs, self.counts


def gather_bytes_start(t_u8, device):
    """Variable-length gather of a uint8 tensor to RANK 0 ONLY (the reference collects once, on the parent:
    volcanosv-vc-large-indel.py:271-278): one all-gather of the byte counts (8 bytes per rank), then grouped point-to-point
    transfers — every rank r > 0 sends exactly its bytes to rank 0, rank 0 posts one exact-size receive per sender; nobody else
    receives anything (a padded all-gather would deliver every rank's table to every rank: 8x the xGMI traffic at N = 8, on links
    that are point-to-point). Nothing here waits: the count all-gather is ENQUEUED (async_op) and the handle returned; the counts are
    read and the transfers posted by the handle's `.post()` / `.wait()`, which the caller reaches when it needs the rows — in
    bench.py `streams` steps later, when the engine that produced them is about to be reused — so a step never stops for a
    rendezvous of the ranks or a host readback (round 3 did both per step). Every rank must reach the starts and the
    waits in the same order (they are collectives). t_u8 is kept alive until then."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return _RootGather(t_u8, device)
    world = dist.get_world_size()
    src = t_u8.to(device).contiguous()
    n = torch.tensor([int(src.numel())], dtype=torch.int64, device=device)      # (the size of a tensor is host knowledge: no readback)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    work = dist.all_gather(counts_t, n, async_op=True)
    return _RootGather(src, device, work, counts_t, n)


def gather_bytes(t_u8, device):
    return gather_bytes_start(t_u8, device).wait()


def _calls_as_u8(calls, device):
    if torch.is_tensor(calls):
        return calls
    return torch.from_numpy(np.frombuffer(calls.tobytes(), dtype=np.uint8).copy()) if len(calls) else torch.zeros(0, dtype=torch.uint8)


def gather_calls(calls, device, to_host=True):
    """calls: this rank's call table — a numpy CALL_DTYPE array (host path, gloo tests) or a uint8 torch tensor holding the
    rows on `device` (GPU path: the bytes go GPU -> RCCL send/recv -> rank 0 with no host hop on the senders).
    Returns on rank 0 the concatenation over ranks in (tid, pos) order as a numpy array, None elsewhere. Every rank's
    table is already sorted by (tid, pos) and tids are disjoint across ranks, so the merge is a concatenation of the
    per-rank tables ordered by their tids (a stable sort only if two ranks interleave tids).
    Counts go through one all-gather, rows through exact-size point-to-point transfers to rank 0 only (gather_bytes_start)."""
    row = CALL_DTYPE.itemsize
    as_tensor = torch.is_tensor(calls)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if not as_tensor:
            return calls
        single = ([calls], [calls.numel() // row])
        return single if not to_host else finish_gather(single)
    bufs, counts = gather_bytes(_calls_as_u8(calls, device).to(device), device)
    if dist.get_rank() != 0:
        return None
    g = (bufs, [c // row for c in counts])
    return g if not to_host else finish_gather(g)     # to_host=False: rows stay on rank 0's device until finish_gather


def finish_gather(gathered):
    """Rank 0: (padded device buffers, counts) from gather_calls(..., to_host=False) -> merged numpy call table."""
    bufs, counts = gathered
    row = CALL_DTYPE.itemsize
    world = len(bufs)
    parts = [bufs[r][: counts[r] * row].cpu().numpy().view(CALL_DTYPE) for r in range(world) if counts[r]]
    if not parts:
        return np.zeros(0, CALL_DTYPE)
    spans = [(int(p["sig"]["tid"][0]), int(p["sig"]["tid"][-1])) for p in parts]
    order = sorted(range(len(parts)), key=lambda i: spans[i])
    allc = np.concatenate([parts[i] for i in order])
    if any(spans[order[i]][1] >= spans[order[i + 1]][0] for i in range(len(order) - 1)):   # interleaved tids: real merge
        key = (allc["sig"]["tid"].astype(np.int64) << 32) | (allc["sig"]["pos"].astype(np.int64) & 0xFFFFFFFF)
        allc = allc[np.argsort(key, kind="stable")]
    return allc


def _all_to_all_bytes(chunks, device):
    """chunks[r] = uint8 numpy array destined for rank r. Returns the list received from every rank. One counts
    all-to-all + one padded all-to-all (RCCL: grouped send/recv over xGMI); all-gather fallback for backends without it."""
    world = dist.get_world_size()
    sizes = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=device)
    rsizes = torch.empty_like(sizes)
    try:
        dist.all_to_all_single(rsizes, sizes)
        send = torch.from_numpy(np.concatenate(chunks) if sum(len(c) for c in chunks) else np.zeros(0, np.uint8)).to(device)
        recv = torch.empty(int(rsizes.sum().item()), dtype=torch.uint8, device=device)
        dist.all_to_all_single(recv, send, output_split_sizes=[int(x) for x in rsizes.tolist()], input_split_sizes=[int(x) for x in sizes.tolist()])
        out, o = [], 0
        r = recv.cpu().numpy()
        for n in rsizes.tolist():
            out.append(r[o:o + int(n)])
            o += int(n)
        return out
    except (RuntimeError, NotImplementedError):
        allsizes = [torch.empty_like(sizes) for _ in range(world)]
        dist.all_gather(allsizes, sizes)
        mx = max(int(t.max().item()) for t in allsizes)
        mine = torch.zeros(world * max(mx, 1), dtype=torch.uint8, device=device)
        for r, c in enumerate(chunks):
            if len(c):
                mine[r * mx: r * mx + len(c)] = torch.from_numpy(np.ascontiguousarray(c)).to(device)
        bufs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(bufs, mine)
        me = dist.get_rank()
        return [bufs[src][me * mx: me * mx + int(allsizes[src][me].item())].cpu().numpy() for src in range(world)]


def exchange_bnd(cand, read_base, owner_of_tid, device):
    """Complex_SV cross-rank breakpoint join. `cand`: this rank's breakend candidates (BND_DTYPE, rows produced from the
    primary alignments of the chromosomes it owns, svim_asm/SVIM_COLLECT.py:60-76); `read_base`: global id of this
    rank's read 0. Rows go to owner(src_tid) — partitions are keyed by the canonical source contig
    (svim_asm/SVCandidate.py:352-373, SVIM_COMBINE.py:17-26) — and come back ordered like the single-process collection:
    hp1 rows before hp2 rows, each by (global read id, pair order)."""
    cand = cand.copy()
    cand["read"] += np.uint32(read_base)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        got = cand
    else:
        world = dist.get_world_size()
        dest = np.array([owner_of_tid[int(t)] for t in cand["src_tid"]], dtype=np.int64) if len(cand) else np.zeros(0, np.int64)
        chunks = [np.frombuffer(cand[dest == r].tobytes(), dtype=np.uint8) for r in range(world)]
        parts = [np.frombuffer(p.tobytes(), dtype=BND_DTYPE) for p in _all_to_all_bytes(chunks, device)]
        got = np.concatenate(parts) if parts else np.zeros(0, BND_DTYPE)
    hap2 = (got["meta"] & B_HAP2) != 0
    order = np.lexsort((np.arange(len(got)), got["read"], hap2))      # stable: (hap, read id, original pair order)
    return got[order]


def exchange_bnd_device(cand_u8, read_gid, owner_of_tid_t, device):
    """exchange_bnd without leaving the GPUs: `cand_u8` = this rank's candidate rows (uint8 tensor [n * 32], vsv_bnd layout, local
    read indices), `read_gid` = int64 tensor local read -> global read id, `owner_of_tid_t` = int64 tensor tid -> rank. Rows travel
    to owner(src_tid) over one RCCL all-to-all (counts first) and come back in the single-process collection order: hp1 rows
    before hp2 rows, each by (global read id, pair order) — one stable sort of a 64-bit key on the device."""
    rows = cand_u8.view(-1, 32)
    w = rows.view(torch.int32)                                    # [n, 8]: src_tid, src_pos, dst_tid, dst_pos, read, read2, meta, pad
    if rows.shape[0]:
        w = w.clone()
        w[:, 4] = read_gid[w[:, 4].to(torch.int64)].to(torch.int32)     # local -> global read id (ids stay below 2^31)
    if dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        dest = owner_of_tid_t[w[:, 0].to(torch.int64)]
        order = torch.sort(dest, stable=True).indices
        w = w[order].contiguous()
        sizes = torch.bincount(dest, minlength=world).to(torch.int64)
        rsizes = torch.empty_like(sizes)
        dist.all_to_all_single(rsizes, sizes)
        send_split, recv_split = [int(x) for x in sizes.tolist()], [int(x) for x in rsizes.tolist()]
        recv = torch.empty((sum(recv_split), 8), dtype=torch.int32, device=device)
        dist.all_to_all_single(recv, w, output_split_sizes=recv_split, input_split_sizes=send_split)
        w = recv
    if w.shape[0]:
        key = ((w[:, 6].to(torch.int64) & B_HAP2) << 40) | (w[:, 4].to(torch.int64) & 0xFFFFFFFF)     # (hap, global read id)
        w = w[torch.sort(key, stable=True).indices].contiguous()
    return w.view(torch.uint8).view(-1)


def gather_rows_device(rows_u8, device):
    """Variable-length gather of device rows (uint8 tensor) to rank 0 only (gather_bytes_start: counts all-gather + exact-size
    send/recv); rank order. Returns the concatenated uint8 tensor on rank 0, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows_u8
    bufs, counts = gather_bytes(rows_u8, device)
    if dist.get_rank() != 0:
        return None
    return torch.cat([b for b, c in zip(bufs, counts) if c]) if any(counts) else torch.zeros(0, dtype=torch.uint8, device=device)


def gather_rows(rows, dtype, device):
    """Variable-length gather of structured numpy rows to rank 0 only (host path, gloo tests); rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    src = torch.from_numpy(np.frombuffer(rows.tobytes(), dtype=np.uint8).copy()) if len(rows) else torch.zeros(0, dtype=torch.uint8)
    bufs, counts = gather_bytes(src.to(device), device)
    if dist.get_rank() != 0:
        return None
    parts = [np.frombuffer(b.cpu().numpy().tobytes(), dtype=dtype) for b, c in zip(bufs, counts) if c]
    return np.concatenate(parts) if parts else np.zeros(0, dtype)
