"""world_size-2 gloo rehearsal of BASELINE config 5: breakend candidates are produced on the rank that owns the primary
alignment's chromosome, exchanged to the owner of the canonical source contig, paired there, and gathered to rank 0.
Compute = the CPU oracle (this tests the exchange logic)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from volcanosv_amd import bnd, shard  # noqa: E402
from volcanosv_amd.abi import BND_DTYPE  # noqa: E402


def main_hip(n_events):
    """The same exchange with the HIP engine as compute on every rank (all ranks share cuda:0; the collectives run over gloo):
    synthetic config-5 stream, result against the single-process HIP run and the oracle."""
    from volcanosv_amd import synth
    from volcanosv_amd.engine import Engine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    seg, primary_tid = synth.generate_bnd(n_events, seed=77)
    owner = shard.lpt_assign(synth.HG19_LEN, world)
    mine = np.flatnonzero(np.array(owner)[primary_tid] == rank)
    so = seg.seg_off.astype(np.int64)
    lens = (so[1:] - so[:-1])[mine]
    idx = np.repeat(so[:-1][mine], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
    off = np.zeros(len(mine) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    local = bnd.SegmentSoA.from_arrays(seg.contigs, off, seg.q_start[idx], seg.q_end[idx], seg.ref_id[idx], seg.ref_start[idx], seg.ref_end[idx],
                                       seg.is_reverse[idx], seg.hap[mine])
    with Engine(0) as eng:
        cand = eng.bnd_candidates(local).copy()
        if len(cand):
            cand["read"] = mine.astype(np.uint32)[cand["read"]]
        rows = shard.exchange_bnd(cand, 0, owner, dev)
        assert all(owner[int(t)] == rank for t in rows["src_tid"])
        calls = eng.bnd_pair_rows(rows, seg.contig_rank)
        allc = shard.gather_rows(calls, BND_DTYPE, dev)
        # the same exchange with the rows kept on the device (what bench.py config 5 times): all-to-all of torch tensors (gloo
        # moves them through the host here, RCCL does not), ordering by one stable device sort
        gdev = torch.device("cuda", 0)
        dseg = bnd.DeviceSegments(local, gdev)
        dcand = eng.bnd_candidates_device(dseg, gdev)
        gid_t = torch.from_numpy(mine.astype(np.int64)).to(gdev)
        owner_t = torch.tensor(owner, dtype=torch.int64, device=gdev)

        def exchange_on_device(cand_u8):
            w = cand_u8.view(-1, 32).view(torch.int32).clone()
            if w.shape[0]:
                w[:, 4] = gid_t[w[:, 4].to(torch.int64)].to(torch.int32)
            rows_np = np.frombuffer(w.cpu().numpy().tobytes(), dtype=BND_DTYPE)
            return shard.exchange_bnd(rows_np, 0, owner, dev)          # gloo leg of the rehearsal
        drows = exchange_on_device(dcand)
        assert np.array_equal(drows, rows), "device candidate rows differ from the host path"
        dcalls = eng.bnd_pair_device(torch.from_numpy(np.frombuffer(drows.tobytes(), dtype=np.uint8).copy()).to(gdev),
                                     torch.from_numpy(np.ascontiguousarray(seg.contig_rank)).to(gdev), gdev)
        assert np.array_equal(np.frombuffer(dcalls.cpu().numpy().tobytes(), dtype=BND_DTYPE), calls), "device pairing differs from the host path"
        if world == 1:
            one = shard.exchange_bnd_device(dcand, gid_t, owner_t, gdev)
            assert np.array_equal(np.frombuffer(one.cpu().numpy().tobytes(), dtype=BND_DTYPE), rows)
        if rank == 0:
            _, single = eng.bnd(seg)
            _, want = oracle.run_bnd(seg)
            key = lambda c: sorted(map(tuple, c[["src_tid", "src_pos", "dst_tid", "dst_pos", "read", "read2", "meta"]].tolist()))
            assert len(want) > n_events // 2 and key(allc) == key(single) == key(want)
            print("BND_SHARD_OK %d calls (HIP engine on %d ranks)" % (len(allc), world))
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "--engine" in sys.argv and sys.argv[sys.argv.index("--engine") + 1] == "hip":
        return main_hip(int(sys.argv[sys.argv.index("--events") + 1]) if "--events" in sys.argv else 3000)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "bnd_a.json")))
    contigs = [tuple(c) for c in doc["contigs"]]
    reads = sorted(doc["reads"], key=lambda r: r["hap"])                 # collection order: hp1 BAM, then hp2 BAM
    owner = shard.lpt_assign([c[1] for c in contigs], world)
    # global read ids follow the collection order; a rank holds the reads whose PRIMARY alignment lies on its chromosomes
    mine = [i for i, r in enumerate(reads) if owner[r["segs"][0][0]] == rank]
    seg = bnd.SegmentSoA([reads[i] for i in mine], contigs)
    cand, _ = oracle.run_bnd(seg)
    cand = cand.copy()
    cand["read"] = np.array(mine, dtype=np.uint32)[cand["read"]] if len(cand) else cand["read"]   # local -> global read id
    rows = shard.exchange_bnd(cand, 0, owner, dev)
    assert all(owner[int(t)] == rank for t in rows["src_tid"])
    calls = oracle.run_bnd_pair(rows, seg.contig_rank)
    allc = shard.gather_rows(calls, BND_DTYPE, dev)
    if rank == 0:
        full = bnd.SegmentSoA(reads, contigs)
        _, want = oracle.run_bnd(full)
        got = sorted(bnd.call_fields(full, c) for c in allc)
        assert got == sorted(bnd.call_fields(full, c) for c in want) == doc["expected"]["paired"]
        assert bnd.vcf_lines(full, allc) == doc["expected"]["vcf"]
        print("BND_SHARD_OK %d calls" % len(allc))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
