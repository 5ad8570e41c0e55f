#!/usr/bin/env python3
"""bench.py — alignment-records/s through SV-signature extraction + clustering + pairing on MI355X.

One step = one pass of the whole hot path (vsv_run_chromosome: cigar_scan_emit -> fold -> split pairs ->
sort/cluster x2 -> merge -> hap pairing) over one device-resident shard of BASELINE.json config 2
(10 M HiFi-like records of one chromosome). Steps are independent batches, so a rank keeps `--streams` engines
(one vsv_handle + HIP stream each, default 2) in flight round-robin: the latency-bound signature stages of one
batch overlap the bandwidth-bound cigar_scan_emit of the next, exactly as a rank that owns several chromosomes
runs them (volcanosv_amd/contig_signature.py). With N GPUs every rank owns one such chromosome shard (weak scaling,
no data-path collective); the per-rank call tables are gathered to rank 0 once, inside the timed region.

Prints ONE JSON line (see the task contract) with `roofline` for the dominant kernel (cigar_scan_emit, HBM-bound,
timed with HIP events on the launch stream inside the library) and `cpu_baseline` (the CPU oracle = C port of the
reference path, one core, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--records", type=int, default=10_000_000, help="records per GPU (config 2: 10 M)")
    ap.add_argument("--shape", default="hifi", choices=["hifi", "ont", "contig"])
    ap.add_argument("--dtype", default=None, help="Hifi | ONT | CLR | READS (default by shape)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="records of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-reps", type=int, default=5, help="repetitions of the CPU baseline (best is reported)")
    ap.add_argument("--max-sigs", type=int, default=1 << 22)
    ap.add_argument("--streams", type=int, default=2, help="engines (handle + HIP stream) in flight per rank")
    ap.add_argument("--host-threads", type=int, default=0, help="1: one host thread per engine enqueues its steps")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (rehearsal: all ranks on cuda:0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    rehearsal = args.dist_backend == "gloo"
    if rehearsal:
        local_rank = 0                      # every rank shares cuda:0; collectives run on host tensors
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev    # device of the collective buffers
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from volcanosv_amd import shard, synth
    from volcanosv_amd.abi import DTYPE_BY_NAME, SIG_DTYPE
    from volcanosv_amd.engine import DeviceRecords, Engine, default_params

    dtype_name = args.dtype or {"hifi": "Hifi", "ont": "ONT", "contig": "Hifi"}[args.shape]
    dtype = DTYPE_BY_NAME[dtype_name]
    config_idx = {"hifi": 2, "ont": 3, "contig": 6}[args.shape]
    # reference index (contig lengths + per-tid record counts): rank 0 owns it, RCCL broadcast to the others
    index = torch.tensor([[synth.CHR10_LEN, args.records]] * world, dtype=torch.int64, device=cdev)
    index = shard.broadcast_index(index, cdev)
    chrom_len, n_rec = int(index[rank, 0]), int(index[rank, 1])
    # synthetic shard generated directly in HBM (torch CUDA generator = Philox); tid = rank
    t, nq, _ = synth.generate(n_rec, args.shape, seed=20250328 + config_idx + 1000 * rank, tid=rank, chrom_len=chrom_len, device=dev)
    # sort-key hints from the reference index: the contig length, and that this shard holds one chromosome (tid = rank)
    recs = DeviceRecords(t, nq, rank + 1, max_pos=chrom_len + 200000, tid_lo=rank)
    n_streams = max(1, min(args.streams, max(1, args.steps)))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    engs = [Engine(local_rank, stream=s.cuda_stream, max_sigs=args.max_sigs) for s in streams]
    p = default_params(dtype)
    pool = None
    if args.host_threads and n_streams > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=n_streams)

    def run_steps(k, scan_ms):
        """k steps round-robin over the engines; a step's counters (status + table sizes) are read back, once, before
        its engine is reused. Returns the engine that ran the last step. With --host-threads each engine is driven by its
        own host thread (a step is ~95 kernel launches = 0.4 ms of enqueue on one core; the library calls release the GIL)."""
        if args.host_threads and n_streams > 1:
            def drive(j):
                ms = []
                for _ in range(j, k, n_streams):
                    engs[j].run_async(recs, p)
                    engs[j].finish()
                    ms.append(engs[j].scan_ms())
                return ms
            for ms in pool.map(drive, range(n_streams)):
                scan_ms.extend(ms)
            return engs[(k - 1) % n_streams]
        for i in range(k):
            e = engs[i % n_streams]
            if i >= n_streams:
                e.finish()
                scan_ms.append(e.scan_ms())
            e.run_async(recs, p)
        for i in range(max(0, k - n_streams), k):
            e = engs[i % n_streams]
            e.finish()
            scan_ms.append(e.scan_ms())
        return engs[(k - 1) % n_streams]

    weng = run_steps(max(args.warmup, n_streams), [])
    if world > 1 and dtype != DTYPE_BY_NAME["READS"]:
        # warm-up of the gather path as well (the first collective of a shape sets up its channels / staging buffers)
        if rehearsal:
            shard.gather_calls(weng.table("calls"), cdev)
        else:
            shard.finish_gather(shard.gather_calls(weng.table_torch("calls", dev), dev, to_host=False))
    torch.cuda.synchronize()
    scan_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng = run_steps(args.steps, scan_ms)
    if dtype == DTYPE_BY_NAME["READS"]:
        gathered = eng.table_torch("reads", dev)       # stays on the device like the call-table gather below; host copy after the timed region
    elif rehearsal:
        gathered = shard.gather_calls(eng.table("calls"), cdev)
    else:
        # final gather of the per-rank call tables: device-to-device copy out of the library, RCCL all-gather over
        # xGMI, one D2H of the merged table on rank 0
        gathered = shard.gather_calls(eng.table_torch("calls", dev), dev, to_host=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if isinstance(gathered, tuple):
        gathered = shard.finish_gather(gathered)     # host copy for the VCF writer, after the timed region
    elif torch.is_tensor(gathered):
        gathered = gathered.cpu().numpy().view(SIG_DTYPE)
    n_raw = len(eng.table("raw"))
    ceil_read = ceil_copy = None
    if rank == 0:
        ceil_read, ceil_copy = eng.stream_ceiling(t["cigar"], reps=5)
    alg_bytes = 24 * recs.n_records + 4 * recs.n_ops + 32 * n_raw      # SURVEY.md §8d, per launch
    scan_s = sum(scan_ms) / len(scan_ms) / 1e3
    achieved = alg_bytes / scan_s / 1e9

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_cigar_scan_emit_config2.json")
    if os.path.exists(pmc_path):
        # HBM bytes per launch of cigar_scan_emit from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # KiB units) on the same shape; scaled by the op count, which sets the traffic (24 B/record headers are read
        # only in part: pos + cigar_off low dwords).
        pmc = json.load(open(pmc_path))
        if args.shape == "hifi" and abs(recs.n_ops - pmc["n_ops"]) < 0.02 * pmc["n_ops"]:
            traffic = pmc["traffic_bytes_per_launch"] * recs.n_ops / pmc["n_ops"]
    if rank == 0:
        cpu = None
        if args.cpu_sample > 0:
            from oracle import oracle
            ns = min(args.cpu_sample, recs.n_records)
            host = {k: (v[: ns + 1] if k == "cigar_off" else v[:ns]).cpu() for k, v in t.items() if k != "cigar"}
            n_ops_s = int(host["cigar_off"][ns])
            host["cigar"] = t["cigar"][:n_ops_s].cpu()
            soa = synth.to_soa(host, nq)
            pc = oracle.default_params(dtype)
            best, total = None, 0.0
            for _ in range(max(1, args.cpu_reps)):
                c0 = time.perf_counter()
                st, tabs = oracle.run(soa, params=pc)
                c1 = time.perf_counter()
                assert st == 0
                total += c1 - c0
                best = c1 - c0 if best is None else min(best, c1 - c0)
            cpu = {"value": ns / best, "unit": "records/s", "cores": 1, "kind": "port",
                   "sample": "first %d records of the same shard through oracle/vsv_oracle.c (C port of the reference path, windowed "
                             "cluster/pair loops), best of %d runs, %.1f s of CPU work in all; the reference's own Python functions "
                             "measured 0.17 M records/s for extraction alone (BASELINE.md)" % (ns, max(1, args.cpu_reps), total)}
        total_records = recs.n_records * world * args.steps
        line = {
            "metric": "alignment-records/s through SV-signature+cluster; VCF bit-match vs CPU",
            "value": total_records / dt,
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "config2: %d %s-like records/GPU, 1 chromosome per GPU, dtype %s, %d CIGAR ops, %d raw signatures, %d calls gathered"
                                   % (recs.n_records, args.shape, dtype_name, recs.n_ops, n_raw, len(gathered) if gathered is not None else 0),
                       "records_per_gpu": recs.n_records, "parallelism": "chromosome-sharded x%d" % world,
                       "streams_per_gpu": n_streams},
            "roofline": {"bound": "hbm", "kernel": "cigar_scan_emit", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_s * 1e3,
                         # the library's own read-stream / copy kernels over the same CIGAR array, after the timed region
                         "measured_read_stream": ceil_read, "measured_copy_stream": ceil_copy,
                         "frac_of_measured_read_stream": achieved / ceil_read if ceil_read else None},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    for e in engs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
