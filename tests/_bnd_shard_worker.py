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


def main():
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
