#!/usr/bin/env python3
"""bench.py — alignment-records/s through SV-signature extraction + clustering + pairing on MI355X.

Headline (BASELINE.json config 2, the configuration the metric is quoted on): one step = one pass of the whole hot path
(vsv_run_chromosome: cigar scan -> fold -> split pairs -> sort/cluster x2 -> merge -> hap pairing) over one device-resident
shard of 10 M HiFi-like records of one chromosome. Steps are independent batches, so a rank keeps `--streams` engines (one
vsv_handle + HIP stream each, default 4) in flight round-robin: the latency-bound signature stages of one batch overlap the
bandwidth-bound scan of the next, exactly as a rank that owns several chromosomes runs them (volcanosv_amd/contig_signature.py).
With N GPUs every rank owns one such chromosome shard (weak scaling, no data-path collective); EVERY step's call table is copied out of
its engine and collected on rank 0 (counts all-gather + exact-size send/recv to rank 0 only, shard.gather_bytes_start) inside the
timed region, the transfer running under the next steps' compute. `python bench.py --gpus N` without a torchrun environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process (before anything touches the GPU) and relays
its line. The timed region of `--steps` steps is repeated `--reps` times (each
bracketed by barrier + synchronize, MAX over ranks) and the MEDIAN repetition is the headline; min / max are reported.

The same JSON line carries, under "configs", the other workloads of SURVEY.md §8d measured in the same run:
  reads    config 2 through the READS flavour (extract_reads_signature.py: no fold / cluster / pair)       [N = 1 only]
  config3  50 M ONT-like records, dtype ONT                                                                [N = 1 only]
  contig   row 2c: 200 k contig-like records of ~16 k ops (the literal contig-vs-reference shape)          [N = 1 only]
  config4  22 chromosomes x 20 M records assigned to the ranks by LPT (shard.lpt_assign), FIXED total work (strong scaling:
           the busiest of 8 ranks holds 3 chromosomes, ideal 7.33x), the rank's engines take turns over its chromosomes, call gather
  config5  Complex_SV: split-contig stream over 22 chromosomes, candidates on the rank that owns the primary alignment,
           cross-rank breakpoint join (shard.exchange_bnd: all-to-all to the owner of the source contig), pairing, gather
`--config 4` / `--config 5` make one of the last two the headline instead ("scaling": "strong"); `--extras none` skips them all.

`roofline` is the dominant kernel (the CIGAR scan, HBM-bound), timed with HIP events on its launch stream inside the library;
`cpu_baseline` is the CPU oracle (C port of the reference path) on this box's host cores: one thread, and one process per
chromosome like the reference's joblib fan-out (volcanosv-vc-large-indel.py:268).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
METRIC = "alignment-records/s through SV-signature+cluster; VCF bit-match vs CPU"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_worker(paths, reps):
    """Hidden mode (`bench.py --cpu-worker`): one process of the reference-style fan-out. Runs the CPU oracle over its
    chromosome shards (npz files), prints {"records": n, "seconds": best-of-reps total}."""
    import numpy as np
    from oracle import oracle
    from volcanosv_amd.soa import RecordSoA
    total_n, total_s = 0, 0.0
    for path in paths:
        z = np.load(path)
        soa = RecordSoA(z["pos"], z["tid"], z["qid"], z["cigar_off"], z["mapq"], z["flag"], z["cigar"])
        soa.n_qids = int(z["n_qids"])
        pc = oracle.default_params(int(z["dtype"]))
        best = None
        for _ in range(max(1, reps)):
            c0 = time.perf_counter()
            st, _ = oracle.run(soa, params=pc)
            c1 = time.perf_counter()
            assert st == 0
            best = c1 - c0 if best is None else min(best, c1 - c0)
        total_n += soa.n_records
        total_s += best
    print(json.dumps({"records": total_n, "seconds": total_s}))


def cpu_baseline(t, nq, dtype, n_sample, reps, procs):
    """The C oracle on the host cores: (a) one thread over the first n_sample records of the shard, (b) `procs` processes,
    one chromosome-like slice of the same sample each (the reference runs one process per chromosome, DRV:268)."""
    import numpy as np
    from oracle import oracle
    from volcanosv_amd import synth
    ns = min(n_sample, int(t["pos"].numel()))
    host = {k: (v[: ns + 1] if k == "cigar_off" else v[:ns]).cpu() for k, v in t.items() if k != "cigar"}
    n_ops_s = int(host["cigar_off"][ns])
    host["cigar"] = t["cigar"][:n_ops_s].cpu()
    soa = synth.to_soa(host, nq)
    pc = oracle.default_params(dtype)
    best, total = None, 0.0
    for _ in range(max(1, reps)):
        c0 = time.perf_counter()
        st, _ = oracle.run(soa, params=pc)
        c1 = time.perf_counter()
        assert st == 0
        total += c1 - c0
        best = c1 - c0 if best is None else min(best, c1 - c0)
    nproc = os.cpu_count() or 1
    out = {"value": ns / best, "unit": "records/s", "cores": 1, "kind": "port", "nproc": nproc, "cpu_model": cpu_model(),
           "sample": "first %d records of the same shard through oracle/vsv_oracle.c (C port of the reference path, windowed cluster/pair "
                     "loops), best of %d runs, %.1f s of CPU work in all; the reference's own Python functions measured 0.17 M records/s "
                     "for extraction alone (BASELINE.md)" % (ns, max(1, reps), total)}
    # (b) the reference's parallel form: one process per chromosome (22 of them), at most nproc at a time
    if procs > 1:
        n_chrom = 22
        workers = max(1, min(procs, n_chrom, nproc))
        tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        paths = []
        try:
            per = ns // n_chrom
            off = soa.cigar_off.astype(np.int64)
            for c in range(n_chrom):
                a, b = c * per, (c + 1) * per if c < n_chrom - 1 else ns
                path = os.path.join(tmp, "vsv_cpu_%d_%d.npz" % (os.getpid(), c))
                np.savez(path, pos=soa.pos[a:b], tid=soa.tid[a:b], qid=soa.qid[a:b], cigar_off=(off[a:b + 1] - off[a]).astype(np.uint64),
                         mapq=soa.mapq[a:b], flag=soa.flag[a:b], cigar=soa.cigar[off[a]:off[b]], n_qids=np.int64(soa.n_qids), dtype=np.int64(dtype))
                paths.append(path)
            c0 = time.perf_counter()
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-reps", str(max(1, reps))] + paths[w::workers],
                                   stdout=subprocess.PIPE, text=True, cwd=ROOT) for w in range(workers)]
            res = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
            wall = time.perf_counter() - c0
            slowest = max(r["seconds"] for r in res)
            out["parallel"] = {"value": sum(r["records"] for r in res) / slowest, "unit": "records/s", "cores": workers, "processes": workers,
                               "chromosomes": n_chrom,
                               "sample": "the same %d records cut into %d chromosome-like slices, one oracle process per slice, %d at a time "
                                         "(reference: joblib.Parallel over 22 per-chromosome processes, volcanosv-vc-large-indel.py:268); "
                                         "records / the slowest worker's summed best-of-%d oracle time; %.1f s wall including process start "
                                         "and file load" % (ns, n_chrom, workers, max(1, reps), wall)}
        finally:
            for path in paths:
                if os.path.exists(path):
                    os.remove(path)
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside a torchrun environment: N ranks (one per GPU) through torch.distributed.run, started as a
    CHILD process before this process has imported torch or touched the GPU (never an exec). Rank 0's JSON line is relayed on
    stdout, everything else on stderr; the exit code is the child's."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in r.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != self.world and self.rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE is %d; measuring %d rank(s) and reporting n_gpus = %d" % (args.gpus, self.world, self.world, self.world),
                  file=sys.stderr)
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
        self.rehearsal = args.dist_backend == "gloo"
        if self.rehearsal:
            local_rank = 0                      # every rank shares cuda:0; collectives run on host tensors
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.dev = torch.device("cuda", local_rank)
        self.cdev = torch.device("cpu") if self.rehearsal else self.dev    # device of the collective buffers
        if self.world > 1:
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        self.n_ranks_seen = dist.get_world_size() if self.world > 1 else 1        # what the process group says, not what was asked for
        from volcanosv_amd.engine import Engine
        n_streams = max(1, args.streams)
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=self.dev) for _ in range(n_streams - 1)]
        self.engs = [Engine(local_rank, stream=s.cuda_stream, max_sigs=args.max_sigs) for s in self.streams]

    def close(self):
        for e in self.engs:
            e.close()
        if self.world > 1:
            self.dist.destroy_process_group()

    # ---- timing primitives -----------------------------------------------------------------------------------------------
    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, fn):
        """One timed region: barrier + synchronize on both sides, MAX over ranks. fn() enqueues and returns what it gathered."""
        self.sync_all()
        t0 = time.perf_counter()
        out = fn()
        self.sync_all()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tmax = self.torch.tensor([dt], dtype=self.torch.float64, device=self.cdev)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, out

    def run_steps(self, jobs, k, scan_ms, engs=None, table=None):
        """k steps; step i runs jobs[i % len(jobs)] = (records, params) on engine i % n. A step's counters (status + table sizes)
        are read back, once, before its engine is reused. With `table` every finished step pays for its collect: the step's table
        is copied out of the engine (the engine's buffers are overwritten by its next run) and its gather to rank 0 is started
        (asynchronous: the transfer runs under the next steps); at most n gathers are in flight, all are complete on return.
        Returns (engine of the last step, what the last step gathered)."""
        engs = engs or self.engs
        n = len(engs)
        pend, last = [], None

        def done(e):
            nonlocal last
            e.finish()
            scan_ms.append(e.scan_ms())
            if table is not None:
                pend.append(self.gather_start(e, table))        # (enqueues the count exchange, waits for nothing)
                if len(pend) > 2:
                    pend[-3].post()                             # two steps later the counts are there: the transfer starts, under the next steps
                if len(pend) > n:
                    last = pend.pop(0).wait()
        for i in range(k):
            e = engs[i % n]
            if i >= n:
                done(e)
            recs, p = jobs[i % len(jobs)]
            e.run_async(recs, p)
        for i in range(max(0, k - n), k):
            done(engs[i % n])
        for h in pend:
            last = h.wait()
        return engs[(k - 1) % n], last

    def gather_start(self, eng, table):
        """One step's collect: the table leaves the engine (device-to-device copy by the library) and its gather to rank 0 starts
        (shard.gather_bytes_start: counts all-gather + exact-size RCCL send/recv to rank 0 only, over xGMI). Returns the handle;
        `.wait()` gives (per-rank uint8 tensors, byte counts) on rank 0. The host copy happens after the timed region."""
        from volcanosv_amd import shard
        rows = eng.table_torch(table, self.dev)
        if self.rehearsal:
            rows = rows.cpu()                       # gloo rehearsal: collectives on host tensors
        return shard.gather_bytes_start(rows, self.cdev)

    def n_gathered(self, g, table):
        from volcanosv_amd.abi import SIG_DTYPE
        if g is None or g[0] is None:
            return 0
        return sum(g[1]) // (SIG_DTYPE.itemsize if table == "reads" else 48)

    # ---- one chromosome-shard workload (configs 2, 3, 2c, reads flavour) -----------------------------------------------------
    def shard_workload(self, shape, n_records, dtype_name, config_idx, steps, warmup, reps, keep=False):
        from volcanosv_amd import shard, synth
        from volcanosv_amd.abi import DTYPE_BY_NAME
        from volcanosv_amd.engine import DeviceRecords, default_params
        torch = self.torch
        dtype = DTYPE_BY_NAME[dtype_name]
        index = torch.tensor([[synth.CHR10_LEN, n_records]] * self.world, dtype=torch.int64, device=self.cdev)
        index = shard.broadcast_index(index, self.cdev)     # reference index: rank 0 owns it, RCCL broadcast
        chrom_len, n_rec = int(index[self.rank, 0]), int(index[self.rank, 1])
        t, nq, _ = synth.generate(n_rec, shape, seed=20250328 + config_idx + 1000 * self.rank, tid=self.rank, chrom_len=chrom_len, device=self.dev)
        recs = DeviceRecords(t, nq, self.rank + 1, max_pos=chrom_len + 200000, tid_lo=self.rank)
        p = default_params(dtype)
        table = "reads" if dtype_name == "READS" else "calls"
        engs = self.engs[: max(1, min(len(self.engs), steps))]
        p.split_overlap = 1 if len(engs) >= 3 else 0      # VSV_OVERLAP_OFF: three engines in flight fill the GPU by themselves
        self.run_steps([(recs, p)], max(warmup, len(engs)), [], engs, table)     # (warm-up of the gather path too: the first transfer between two ranks sets up its channels)
        reruns0 = sum(e.rerun_count() for e in engs)
        slow0 = sum(e.sort1_slow_count() for e in engs)
        times, scan_ms, gathered = [], [], None
        for _ in range(max(1, reps)):
            dt, (eng, gathered) = self.timed(lambda: self.run_steps([(recs, p)], steps, scan_ms, engs, table))
            times.append(dt)
        reruns = sum(e.rerun_count() for e in engs) - reruns0
        slow = sum(e.sort1_slow_count() for e in engs) - slow0
        # the same steps on ONE engine, one after the other (the latency of a chromosome, no overlap between steps)
        k1 = max(3, min(steps, 10))
        p1 = default_params(dtype)          # (one engine by itself builds the split candidates on its auxiliary stream: VSV_OVERLAP_AUTO)
        self.run_steps([(recs, p1)], 2, [], engs[:1], table)
        dt1, _ = self.timed(lambda: self.run_steps([(recs, p1)], k1, [], engs[:1], table))
        # a COLD handle: fresh engine (workspace reserved up front: allocation is not what is measured), ONE run, HIP events on its
        # stream around everything the run enqueues — what one invocation of the drop-in CLI gets, which runs one chromosome per process
        from volcanosv_amd.engine import Engine
        cold, paths = [], set()
        el0 = sum(e.path_counts()[0] for e in engs)
        if dtype_name != "READS":
            for _ in range(3):
                s = torch.cuda.Stream(device=self.dev)
                ce = Engine(self.local_rank, stream=s.cuda_stream, max_sigs=self.args.max_sigs)
                ce.reserve(recs.n_records, recs.n_ops, self.args.max_sigs, large_tables=True)
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(s)
                ce.run_async(recs, p1)
                ev1.record(s)
                ce.finish()
                ev1.synchronize()
                cold.append(ev0.elapsed_time(ev1))
                paths.add("elements" if ce.path_counts()[0] else "rows")
                assert ce.rerun_count() == 0
                ce.close()
                del ce, s
        self.run_steps([(recs, p1)], 1, [], engs[:1], table)
        warm_path = "elements" if sum(e.path_counts()[0] for e in engs) > el0 else "rows"
        n_raw = len(eng.table("raw"))
        alg_bytes = 24 * recs.n_records + 4 * recs.n_ops + 32 * n_raw      # SURVEY.md §8d, per launch
        scan_s = sum(scan_ms) / len(scan_ms) / 1e3
        alone = []                                                          # the scan with nothing else on the chip (one engine, step by step)
        for _ in range(4):
            engs[0].run_async(recs, p)
            engs[0].finish()
            alone.append(engs[0].scan_ms())
        alone_s = sorted(alone[1:])[1] / 1e3
        ts = sorted(times)
        med = ts[len(ts) // 2]
        res = {
            "shape": shape, "dtype": dtype_name, "records": recs.n_records, "ops": recs.n_ops, "raw_signatures": n_raw,
            "rows_gathered": self.n_gathered(gathered, table), "steps": steps, "reps": len(times), "streams": len(engs),
            "ms_per_step": med / steps * 1e3, "ms_per_step_min": ts[0] / steps * 1e3, "ms_per_step_max": ts[-1] / steps * 1e3,
            "single_engine_ms_per_step": dt1 / k1 * 1e3,
            # the first run of a fresh handle (median of three new engines; HIP events on the engine's stream) and the path it took: a cold
            # handle waits once for its scan to pick the path a warm one takes (vsv_path_counts)
            "cold_ms_per_step": sorted(cold)[len(cold) // 2] if cold else None,
            "path": warm_path, "cold_path": "/".join(sorted(paths)) if paths else None,
            "reruns": reruns,          # whole-run repetitions inside the timed regions (bucket-sort overflow / fused CLR gate fallbacks, vsv_rerun_count)
            "sort1_slow_runs": slow,   # ... and runs whose first element sort met a bucket beyond LDS and sorted it in global memory (vsv_sort1_slow_count)
            "records_per_s": recs.n_records * self.world * steps / med, "ops_per_s": recs.n_ops * self.world * steps / med,
            "scan": {"kernel": "cigar_scan_long" if recs.n_ops >= 512 * recs.n_records else "cigar_scan_emit", "avg_launch_ms": scan_s * 1e3,
                     "GBs": alg_bytes / scan_s / 1e9, "frac": alg_bytes / scan_s / 1e9 / HBM_PEAK_GBS,
                     "note": "average over the timed steps, i.e. while the other engine's signature stages share the chip",
                     "alone_launch_ms": alone_s * 1e3, "alone_GBs": alg_bytes / alone_s / 1e9, "alone_frac": alg_bytes / alone_s / 1e9 / HBM_PEAK_GBS},
            "algorithmic_bytes_per_step": alg_bytes,
            "whole_path_frac": alg_bytes * steps / med / 1e9 / HBM_PEAK_GBS,
        }
        if keep:
            return res, (t, nq, recs, eng, dtype)
        del t, recs
        torch.cuda.empty_cache()
        return res, None

    # ---- config 4: 22 chromosomes, LPT-sharded, fixed total work -------------------------------------------------------------
    def config4(self, n_per_chrom, steps, warmup, reps):
        from volcanosv_amd import shard, synth
        from volcanosv_amd.abi import DTYPE_BY_NAME
        from volcanosv_amd.engine import DeviceRecords, default_params
        torch = self.torch
        n_chrom = 22
        index = torch.tensor([[synth.HG19_LEN[c], n_per_chrom] for c in range(n_chrom)], dtype=torch.int64, device=self.cdev)
        index = shard.broadcast_index(index, self.cdev)
        owner = shard.lpt_assign([int(x) for x in index[:, 1].tolist()], self.world)
        mine = [c for c in range(n_chrom) if owner[c] == self.rank]
        p = default_params(DTYPE_BY_NAME["Hifi"])
        jobs, keep = [], []
        for c in mine:
            t, nq, _ = synth.generate(int(index[c, 1]), "hifi", seed=20250328 + 4 + 1000 * c, tid=c, chrom_len=int(index[c, 0]), device=self.dev)
            recs = DeviceRecords(t, nq, c + 1, max_pos=int(index[c, 0]) + 200000, tid_lo=c)
            keep.append(t)
            jobs.append((recs, p))
        engs = self.engs[: max(1, min(len(self.engs), len(jobs)))]     # the rank's engines take turns over its chromosomes (contig_signature.run)
        p.split_overlap = 1 if len(engs) >= 3 else 0

        def one_pass(scan_ms):
            """Every chromosome of this rank once; the call tables of all of them are gathered (device rows, one collective)."""
            parts = []
            n = len(engs)
            for i, (recs, pp) in enumerate(jobs):
                e = engs[i % n]
                if i >= n:
                    e.finish()
                    scan_ms.append(e.scan_ms())
                    parts.append(e.table_torch("calls", self.dev))
                e.run_async(recs, pp)
            for i in range(max(0, len(jobs) - n), len(jobs)):
                e = engs[i % n]
                e.finish()
                scan_ms.append(e.scan_ms())
                parts.append(e.table_torch("calls", self.dev))
            return parts

        def region(k, scan_ms):
            """k whole-genome jobs; every job pays for its collect (the reference's parent collects once per job,
            volcanosv-vc-large-indel.py:271-278): the rank's call tables go to rank 0 as one transfer that runs under the next job."""
            pend, g = [], None
            for _ in range(k):
                parts = one_pass(scan_ms)
                rows = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=self.dev)
                pend.append(shard.gather_bytes_start(rows.cpu() if self.rehearsal else rows, self.cdev))
                if len(pend) > 1:
                    pend[-2].post()                 # the previous job's counts have arrived: its transfer runs under this job
                if len(pend) > 2:
                    g = pend.pop(0).wait()
            for h in pend:
                g = h.wait()
            return g

        region(max(1, warmup), [])
        reruns0 = sum(e.rerun_count() for e in engs)
        times, scan_ms, g = [], [], None
        for _ in range(max(1, reps)):
            dt, g = self.timed(lambda: region(steps, scan_ms))
            times.append(dt)
        reruns = sum(e.rerun_count() for e in engs) - reruns0
        ts = sorted(times)
        med = ts[len(ts) // 2]
        total_records = int(index[:, 1].sum().item())
        ops_local = sum(r.n_ops for r, _ in jobs)
        recs_local = sum(r.n_records for r, _ in jobs)
        # whole-job algorithmic bytes: headers + ops of all chromosomes (signature rows left out: ~0.2 %)
        tot = torch.tensor([24.0 * recs_local + 4.0 * ops_local], dtype=torch.float64, device=self.cdev)
        if self.world > 1:
            self.dist.all_reduce(tot)
        load = [0] * self.world
        for c in range(n_chrom):
            load[owner[c]] += 1
        res = {
            "workload": "config4: %d chromosomes x %d HiFi-like records (hg19 lengths), LPT over %d rank(s): %s chromosomes per rank, "
                        "%d engines per rank, call tables gathered to rank 0 inside the timed region" % (n_chrom, n_per_chrom, self.world, load, len(engs)),
            "scaling": "strong", "records": total_records, "steps": steps, "reps": len(times),
            "ms_per_step": med / steps * 1e3, "ms_per_step_min": ts[0] / steps * 1e3, "ms_per_step_max": ts[-1] / steps * 1e3,
            "records_per_s": total_records * steps / med, "chromosomes_per_rank": load, "ideal_speedup": n_chrom / max(load),
            "scan_avg_launch_ms": (sum(scan_ms) / len(scan_ms)) if scan_ms else None,
            "whole_path_frac_of_aggregate_hbm": float(tot.item()) * steps / med / 1e9 / (HBM_PEAK_GBS * self.world),
            "calls_gathered": self.n_gathered(g, "calls"), "reruns": reruns, "n_ranks_seen": self.world,
        }
        del keep, jobs
        torch.cuda.empty_cache()
        return res

    # ---- config 5: Complex_SV breakend stream with the cross-rank join --------------------------------------------------------
    def config5(self, n_events, steps, warmup, reps):
        import numpy as np
        from volcanosv_amd import bnd, shard, synth
        from volcanosv_amd.abi import BND_DTYPE
        seg, primary_tid = synth.generate_bnd(n_events, seed=20250328 + 5)
        owner = shard.lpt_assign(synth.HG19_LEN, self.world)
        own = np.array(owner, np.int64)
        mine = np.flatnonzero(own[primary_tid] == self.rank)             # reads whose PRIMARY alignment lies on this rank's chromosomes
        so = seg.seg_off.astype(np.int64)
        lens = (so[1:] - so[:-1])[mine]
        idx = np.repeat(so[:-1][mine], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
        off = np.zeros(len(mine) + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        local = bnd.SegmentSoA.from_arrays(seg.contigs, off, seg.q_start[idx], seg.q_end[idx], seg.ref_id[idx], seg.ref_start[idx], seg.ref_end[idx],
                                           seg.is_reverse[idx], seg.hap[mine])
        gid = mine.astype(np.uint32)                                      # local read -> global read id (collection order)
        eng = self.engs[0]
        torch = self.torch
        device_path = not self.rehearsal
        if device_path:
            # everything stays in HBM: the segment table is uploaded once, candidates / exchange / pairing / gather run on device rows
            dseg = bnd.DeviceSegments(local, self.dev)
            gid_t = torch.from_numpy(mine.astype(np.int64)).to(self.dev)
            owner_t = torch.tensor(owner, dtype=torch.int64, device=self.dev)
            rank_t = torch.from_numpy(np.ascontiguousarray(seg.contig_rank)).to(self.dev)

            def step():
                cand = eng.bnd_candidates_device(dseg, self.dev)           # vsv_bnd_segments -> live candidate rows (device)
                rows = shard.exchange_bnd_device(cand, gid_t, owner_t, self.dev)   # RCCL all-to-all to owner(src_tid), collection order
                calls = eng.bnd_pair_device(rows, rank_t, self.dev)        # vsv_bnd_set_candidates + vsv_bnd_pair (device rows)
                return shard.gather_rows_device(calls, self.dev), cand.numel() // 32
        else:
            def step():
                cand = eng.bnd_candidates(local)                              # vsv_bnd_segments: segment case analysis -> candidates
                cand = cand.copy()
                if len(cand):
                    cand["read"] = gid[cand["read"]]
                rows = shard.exchange_bnd(cand, 0, owner, self.cdev)          # all-to-all to owner(src_tid), collection order restored
                calls = eng.bnd_pair_rows(rows, seg.contig_rank)              # vsv_bnd_set_candidates + vsv_bnd_pair
                return shard.gather_rows(calls, BND_DTYPE, self.cdev), len(cand)

        for _ in range(max(1, warmup)):
            step()
        times, out = [], None
        for _ in range(max(1, reps)):
            def region():
                o = None
                for _ in range(steps):
                    o = step()
                return o
            dt, out = self.timed(region)
            times.append(dt)
        ts = sorted(times)
        med = ts[len(ts) // 2]
        n_segs, n_reads = int(len(seg.q_start)), int(len(seg.hap))
        calls, _ = out
        n_calls = 0 if calls is None else (calls.numel() // 32 if torch.is_tensor(calls) else len(calls))
        return {
            "workload": "config5: %d split contigs (%d events x 2 haplotypes, dense partitions included), %d aligned segments over 22 chromosomes; "
                        "candidates on the owner of the primary alignment, all-to-all to the owner of the source contig, pairing, gather; %s"
                        % (n_reads, n_events, n_segs, "segment table resident in HBM, candidate / call rows never leave the GPUs" if device_path else
                           "gloo rehearsal: segment tables and candidate rows cross the C-ABI as host arrays"),
            "scaling": "strong", "records": n_segs, "reads": n_reads, "steps": steps, "reps": len(times),
            "ms_per_step": med / steps * 1e3, "ms_per_step_min": ts[0] / steps * 1e3, "ms_per_step_max": ts[-1] / steps * 1e3,
            "records_per_s": n_segs * steps / med, "calls_gathered": int(n_calls),
        }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU). Under torch.distributed.run the world size comes from WORLD_SIZE; "
                    "without it, --gpus N > 1 starts torch.distributed.run with N ranks as a child process")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed --steps region; the median is reported")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5, 6], help="headline workload: BASELINE config 2 (default), 3, 4, 5; 6 = row 2c contig-like")
    ap.add_argument("--records", type=int, default=0, help="records per GPU (config 2: 10 M; config 3: 50 M; contig: 200 k) / per chromosome (config 4: 20 M) / events (config 5: 1 M)")
    ap.add_argument("--shape", default=None, choices=[None, "hifi", "ont", "contig"], help="alias: hifi = --config 2, ont = 3, contig = 6")
    ap.add_argument("--dtype", default=None, help="Hifi | ONT | CLR | READS (default by shape)")
    ap.add_argument("--extras", default="auto", help="auto: all the other workloads in 'configs'; none; or a comma list of reads,config3,contig,config4,config5")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--extra-reps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="records of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-reps", type=int, default=5, help="repetitions of the CPU baseline (best is reported)")
    ap.add_argument("--cpu-procs", type=int, default=22, help="processes of the parallel CPU baseline (one chromosome slice each; 0 = skip)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("paths", nargs="*", help=argparse.SUPPRESS)
    ap.add_argument("--max-sigs", type=int, default=0, help="row capacity of an engine (default by workload)")
    ap.add_argument("--streams", type=int, default=4, help="engines (handle + HIP stream) in flight per rank (1 / 2 / 3 / 4 / 5 engines on config 2: 0.64 / 0.49 / 0.436 / 0.424 / 0.484 ms per step)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (rehearsal: all ranks on cuda:0)")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.paths, args.cpu_reps)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    if args.shape:
        args.config = {"hifi": 2, "ont": 3, "contig": 6}[args.shape]
    shape = {2: "hifi", 3: "ont", 6: "contig"}.get(args.config)
    if not args.max_sigs:
        args.max_sigs = (1 << 24) if args.config in (3, 5, 6) or args.extras != "none" else (1 << 22)

    b = Bench(args)
    torch = b.torch
    from volcanosv_amd.abi import DTYPE_BY_NAME
    line, kept = None, None
    if args.config in (2, 3, 6):
        dtype_name = args.dtype or {"hifi": "Hifi", "ont": "ONT", "contig": "Hifi"}[shape]
        n_rec = args.records or {2: 10_000_000, 3: 50_000_000, 6: 200_000}[args.config]
        res, kept = b.shard_workload(shape, n_rec, dtype_name, args.config, args.steps, args.warmup, args.reps, keep=True)
        t, nq, recs, eng, dtype = kept
        ceil_read = ceil_copy = None
        traffic, traffic_source = None, None
        if b.rank == 0:
            ceil_read, ceil_copy = eng.stream_ceiling(t["cigar"], reps=5)
            # HBM bytes per launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs, KiB units,
            # FETCH_SIZE doubled for the wide coalesced stream as MI355X_MICROARCH.md prescribes) of the same kernel on the same
            # shape, scaled by the op count (which sets the traffic): a profile figure, not an in-run measurement
            pmc_name = {"hifi": "r02_pmc_cigar_scan_emit_config2.json", "contig": "r04_pmc_cigar_scan_long_contig200k.json"}.get(shape)
            pmc_path = os.path.join(ROOT, "profiles", pmc_name) if pmc_name else None
            if pmc_path and os.path.exists(pmc_path):
                pmc = json.load(open(pmc_path))
                # a profile LOOKUP, not a measurement of this run: refused when the kernel's source is no longer the one that was profiled
                # (the profile records the sha256 of csrc/cigar_scan.hip; profiles older than that rule: the scan kernel's name and part
                # of its text, checked by tests/test_bench_contract.py)
                import hashlib
                src_now = hashlib.sha256(open(os.path.join(ROOT, "volcanosv_amd", "csrc", "cigar_scan.hip"), "rb").read()).hexdigest()
                fresh = pmc.get("kernel_source_sha256") in (None, src_now) if shape == "hifi" else pmc.get("kernel_source_sha256") == src_now
                if not fresh:
                    traffic_source = "profile lookup REFUSED: profiles/%s was taken from another version of csrc/cigar_scan.hip" % pmc_name
                elif shape == "contig" or abs(recs.n_ops - pmc["n_ops"]) < 0.02 * pmc["n_ops"]:
                    traffic = pmc["traffic_bytes_per_launch"] * recs.n_ops / pmc["n_ops"]
                    traffic_source = "profile lookup: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel), scaled by op count" % pmc_name
        label = {2: "config2", 3: "config3", 6: "row 2c (contig-like)"}[args.config]
        line = {
            "metric": METRIC, "value": res["records_per_s"], "unit": "records/s", "n_gpus": b.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "reps": res["reps"], "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
            "n_ranks_seen": b.n_ranks_seen, "single_engine_ms_per_step": res["single_engine_ms_per_step"], "reruns": res["reruns"], "sort1_slow_runs": res["sort1_slow_runs"],
            "cold_ms_per_step": res["cold_ms_per_step"], "path": res["path"], "cold_path": res["cold_path"],
            "config": {"workload": "%s: %d %s-like records/GPU, 1 chromosome per GPU, dtype %s, %d CIGAR ops, %d raw signatures, %d rows gathered per step"
                                   % (label, recs.n_records, shape, dtype_name, recs.n_ops, res["raw_signatures"], res["rows_gathered"]),
                       "headline": "batch throughput: %d engines per GPU in flight round-robin over the same device-resident shard (the stages of one step "
                                   "overlap the scan of the next; nothing is cached between steps); one chromosome's latency = single_engine_ms_per_step; "
                                   "every step's call table is copied out of its engine and collected on rank 0 inside the timed region" % res["streams"],
                       "records_per_gpu": recs.n_records, "parallelism": "chromosome-sharded x%d" % b.world, "streams_per_gpu": res["streams"],
                       "ops_per_s": res["ops_per_s"], "whole_path_frac_of_hbm_peak": res["whole_path_frac"]},
            "roofline": {"bound": "hbm", "kernel": res["scan"]["kernel"], "achieved": res["scan"]["alone_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": res["scan"]["alone_frac"], "frac_basis": "the scan launch by itself (alone_*); the same launch inside the timed region: overlapped_frac",
                         "traffic": traffic, "traffic_profile_lookup": traffic_source,
                         "algorithmic_bytes_per_launch": res["algorithmic_bytes_per_step"], "avg_launch_ms": res["scan"]["alone_launch_ms"],
                         "overlapped_launch_ms": res["scan"]["avg_launch_ms"], "overlapped_achieved": res["scan"]["GBs"], "overlapped_frac": res["scan"]["frac"],
                         "note": "achieved / frac / avg_launch_ms: the scan launch by itself (HIP events around the kernel on its launch stream, same input, "
                                 "nothing else in flight) - what rocprofv3 --kernel-trace reports too, see profiles/; overlapped_*: the same events averaged "
                                 "over the timed region, where %d engines are in flight and the launch shares the GPU with the other engines' kernels" % res["streams"],
                         # the library's own read-stream / copy kernels over the same CIGAR array, after the timed region
                         "measured_read_stream": ceil_read, "measured_copy_stream": ceil_copy,
                         "frac_of_measured_read_stream": res["scan"]["alone_GBs"] / ceil_read if ceil_read else None},
        }
    elif args.config == 4:
        res = b.config4(args.records or 20_000_000, args.steps if args.steps != 50 else 10, args.warmup, args.reps)
        line = {"metric": METRIC, "value": res["records_per_s"], "unit": "records/s", "n_gpus": b.world, "n_ranks_seen": b.n_ranks_seen, "steps": res["steps"], "warmup": args.warmup,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "reps": res["reps"], "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
                "config": {"workload": res["workload"], "parallelism": "22 chromosomes LPT-sharded x%d" % b.world, "detail": res},
                "roofline": {"bound": "hbm", "kernel": "cigar_scan_emit", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                             "avg_launch_ms": res["scan_avg_launch_ms"], "whole_path_frac_of_aggregate_hbm": res["whole_path_frac_of_aggregate_hbm"]}}
    else:
        res = b.config5(args.records or 1_000_000, args.steps if args.steps != 50 else 5, args.warmup, args.reps)
        line = {"metric": METRIC, "value": res["records_per_s"], "unit": "records/s", "n_gpus": b.world, "n_ranks_seen": b.n_ranks_seen, "steps": res["steps"], "warmup": args.warmup,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
                "reps": res["reps"], "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
                "config": {"workload": res["workload"], "parallelism": "primary-alignment owner x%d + all-to-all" % b.world, "detail": res},
                "roofline": {"bound": "hbm", "kernel": "segment_bnd / bnd_pair", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}}

    # ---- CPU baseline (rank 0, N = 1 semantics: the sample is this rank's shard), before the shard is released -----------------
    cpu = None
    if kept is not None and b.rank == 0 and args.cpu_sample > 0:
        t, nq, recs, eng, dtype = kept
        cpu = cpu_baseline(t, nq, dtype, args.cpu_sample if shape != "contig" else min(args.cpu_sample, 20_000), args.cpu_reps, args.cpu_procs)
    # ---- the other workloads, same run ------------------------------------------------------------------------------------------
    want = [] if args.extras == "none" else (["reads", "config3", "contig", "config4", "config5"] if args.extras == "auto" else args.extras.split(","))
    extras = []
    single = [w for w in want if w in ("reads", "config3", "contig")]
    kept = None
    torch.cuda.empty_cache()
    if b.world == 1:
        for w in single:
            if w == "reads" and args.config == 2 and (args.dtype or "Hifi") != "READS":
                r, _ = b.shard_workload("hifi", args.records or 10_000_000, "READS", 2, args.extra_steps, 2, args.extra_reps)
                r["name"] = "config2 through the READS flavour (extract_reads_signature.py)"
            elif w == "config3" and args.config != 3:
                r, _ = b.shard_workload("ont", 50_000_000, "ONT", 3, args.extra_steps, 2, args.extra_reps)
                r["name"] = "config3: 50 M ONT-like records, dtype ONT"
            elif w == "contig" and args.config != 6:
                r, _ = b.shard_workload("contig", 200_000, "Hifi", 6, args.extra_steps, 2, args.extra_reps)
                r["name"] = "row 2c: 200 k contig-like records (~16 k ops each), dtype Hifi"
            else:
                continue
            extras.append(r)
    # the two multi-GPU workloads run on every rank count; a failure in one of them must not take the headline line with it
    if "config4" in want and args.config != 4:
        try:
            r = b.config4(20_000_000, max(2, args.extra_steps // 2), 1, args.extra_reps)
        except Exception as e:                                       # noqa: BLE001
            r = {"error": "%s: %s" % (type(e).__name__, e)}
        r["name"] = "config4 (strong scaling, LPT)"
        extras.append(r)
    if "config5" in want and args.config != 5:
        try:
            r = b.config5(1_000_000, 3, 1, args.extra_reps)
        except Exception as e:                                       # noqa: BLE001
            r = {"error": "%s: %s" % (type(e).__name__, e)}
        r["name"] = "config5 (Complex_SV cross-rank breakpoint join)"
        extras.append(r)
    if b.rank == 0:
        line["configs"] = extras
        line["cpu_baseline"] = cpu
        print(json.dumps(line))
    b.close()


if __name__ == "__main__":
    main()
