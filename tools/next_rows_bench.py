#!/usr/bin/env python3
"""Measurements for the SURVEY §8(f) rows built after the hot path: every kernel timed on the GPU (HIP events around the
C-ABI call, device-resident inputs where the entry point takes them) with the CPU oracle beside it on a bounded sample.
Prints one JSON line per row; `profiles/r01_next_rows.jsonl` keeps the output of the committed run."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from volcanosv_amd import sig_extract, synth  # noqa: E402
from volcanosv_amd.engine import DeviceRecords, Engine  # noqa: E402


def gpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def cpu_s(fn):
    t0 = time.perf_counter()
    r = fn()
    return time.perf_counter() - t0, r


def main():
    eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream, max_sigs=1 << 24)
    g = torch.Generator(device="cuda").manual_seed(1)
    out = []

    # ---- f-2 sig_extract, CIGAR stage: cigar_scan_emit<3> + combine_kernel on ONT-like reads ----------------------------
    t, nq, nt = synth.generate(2_000_000, "ont", seed=5, device="cuda")
    dr = DeviceRecords(t, nq, nt)
    p = sig_extract.params()
    ms = gpu_ms(lambda: eng.run(dr, p))
    n_raw, n_comb = len(eng.table("raw")), len(eng.table("cigar"))
    k = 200_000
    n_ops = int(t["cigar_off"][k])
    host = {name: (v[: k + 1] if name == "cigar_off" else v[:n_ops] if name == "cigar" else v[:k]).cpu() for name, v in t.items()}
    cs, _ = cpu_s(lambda: oracle.run(synth.to_soa(host, nq), params=p))
    out.append(dict(row="f-2 sig_extract CIGAR stage (parse_read + generate_combine_sigs)", records=dr.n_records, ops=dr.n_ops, raw_signals=n_raw,
                    combined=n_comb, gpu_ms=ms, gpu_records_per_s=dr.n_records / ms * 1e3, scan_ms=eng.scan_ms(),
                    scan_GBs=(24 * dr.n_records + 4 * dr.n_ops + 32 * n_raw) / eng.scan_ms() / 1e6,
                    cpu_oracle_records_per_s=k / cs, cpu_sample="first %d records, 1 core" % k))
    del t, dr
    torch.cuda.empty_cache()

    # ---- f-2 split branch: 2..6 segments per read ------------------------------------------------------------------------
    rng = np.random.default_rng(2)
    n_reads = 500_000
    reads = []
    for r in range(n_reads):
        ns = int(rng.integers(2, 7))
        q = np.sort(rng.integers(0, 20000, 2 * ns)).reshape(ns, 2)
        rs = np.sort(rng.integers(1_000_000, 1_200_000, ns))
        segs = [[int(q[i, 0]), int(q[i, 1]), int(rs[i]), int(rs[i] + q[i, 1] - q[i, 0]), int(rng.integers(0, 2) == 0), int(rng.integers(0, 4) == 0)] for i in range(ns)]
        reads.append((r, 20000, segs))
    seg = sig_extract.SplitSegments(reads)
    ms = gpu_ms(lambda: eng.cutesv_split(seg, seg.read_len, seg.read_rec), reps=3)
    cs, rows = cpu_s(lambda: oracle.run_cutesv_split(seg, seg.read_len, seg.read_rec))
    out.append(dict(row="f-2 sig_extract split branch (analysis_split_read), host arrays in, rows out (includes H2D/D2H)", reads=n_reads,
                    segments=int(seg.q_start.shape[0]), rows=len(rows), gpu_ms=ms, gpu_reads_per_s=n_reads / ms * 1e3, cpu_oracle_reads_per_s=n_reads / cs))

    # ---- f-3 support joins: 2e5 calls x 2e6 read signatures, device-resident ---------------------------------------------
    ns_, nc = 2_000_000, 200_000
    spos = torch.sort(torch.randint(0, 130_000_000, (ns_,), device="cuda", generator=g, dtype=torch.int32)).values
    slen = torch.randint(30, 400, (ns_,), device="cuda", generator=g, dtype=torch.int32)
    cpos = torch.sort(torch.randint(0, 130_000_000, (nc,), device="cuda", generator=g, dtype=torch.int32)).values
    clen = torch.randint(30, 260, (nc,), device="cuda", generator=g, dtype=torch.int32)
    sp = eng.support_params()
    ms = gpu_ms(lambda: eng.support_join(cpos, clen, spos, slen, sp))
    kc = 2000
    cs, _ = cpu_s(lambda: oracle.run_support(cpos[:kc].cpu().numpy(), clen[:kc].cpu().numpy(), spos.cpu().numpy(), slen.cpu().numpy()))
    window = float((torch.searchsorted(spos, cpos + 1000, right=True) - torch.searchsorted(spos, cpos - 1000)).float().mean())
    out.append(dict(row="f-3 FP_filter_v1.eval_sig (vsv_support_join)", calls=nc, read_signatures=ns_, mean_window=window, gpu_ms=ms,
                    gpu_calls_per_s=nc / ms * 1e3, algorithmic_GBs=(8 * nc + 8 * window * nc) / ms / 1e6,
                    cpu_oracle_calls_per_s=kc / cs, cpu_sample="%d calls, literal loop of the reference (scans the list from its start), 1 core" % kc))
    ms = gpu_ms(lambda: eng.support_cov_ins(cpos, spos, slen, 1000))
    cs, _ = cpu_s(lambda: oracle.run_cov_ins(cpos.cpu().numpy(), spos.cpu().numpy(), slen.cpu().numpy(), 1000))
    out.append(dict(row="f-3 calculate_signature_support INS (vsv_support_cov_ins)", calls=nc, signatures=ns_, gpu_ms=ms, gpu_calls_per_s=nc / ms * 1e3,
                    cpu_oracle_calls_per_s=nc / cs, cpu_sample="all calls, the reference's moving-index scan, 1 core"))
    cend, send = cpos + clen, spos + slen
    ms = gpu_ms(lambda: eng.support_cov_del(cpos, cend, spos, send, -slen, 1000))
    kc = 500
    cs, _ = cpu_s(lambda: oracle.run_cov_del(cpos[:kc].cpu().numpy(), cend[:kc].cpu().numpy(), spos.cpu().numpy(), send.cpu().numpy(), (-slen).cpu().numpy(), 1000))
    out.append(dict(row="f-3 calculate_signature_support DEL (vsv_support_cov_del)", calls=nc, signatures=ns_, gpu_ms=ms, gpu_calls_per_s=nc / ms * 1e3,
                    cpu_oracle_calls_per_s=kc / cs, cpu_sample="%d calls, four boundary scans restated as one pass per call, 1 core" % kc))

    # ---- f-4 remove_redundancy: INS pairs with 2 kb sequences -> edit-distance cell updates per second ------------------
    n_calls, L = 4000, 2000
    base = rng.integers(0, 4, (n_calls // 2, L)).astype(np.uint8)
    seqs = []
    for i in range(n_calls // 2):
        m = base[i].copy()
        idx = rng.integers(0, L, L // 10)
        m[idx] = rng.integers(0, 4, len(idx))
        seqs += [base[i], m]
    pos = np.repeat(np.arange(n_calls // 2) * 5000 + 1000, 2) + np.tile([0, 7], n_calls // 2)
    off = np.arange(n_calls + 1, dtype=np.uint64) * L
    blob = np.concatenate(seqs)
    svlen = np.full(n_calls, L, dtype=np.int32)
    rp = eng.redundancy_params()
    ms = gpu_ms(lambda: eng.redundancy_pairs(False, pos, svlen, blob, off, rp), reps=3)
    pairs = eng.redundancy_pairs(False, pos, svlen, blob, off, rp)
    kp = 40
    cs, _ = cpu_s(lambda: oracle.run_redundancy_pairs(False, pos[: 2 * kp], svlen[: 2 * kp], blob[: 2 * kp * L], off[: 2 * kp + 1]))
    cand = n_calls // 2
    out.append(dict(row="f-4 remove_redundancy INS matching (vsv_redundancy_pairs, Myers bit-parallel edit distance, wave per pair; includes H2D/D2H)",
                    calls=n_calls, candidate_pairs=cand, matched=len(pairs), seq_len=L, gpu_ms=ms, gpu_GCUPS=cand * L * L / ms / 1e6,
                    cpu_oracle_GCUPS=kp * L * L / cs / 1e9, cpu_sample="%d pairs, textbook DP, 1 core" % kp))
    for o in out:
        print(json.dumps(o))
    eng.close()


if __name__ == "__main__":
    main()
