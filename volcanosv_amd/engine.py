"""Thin host wrapper over the C-ABI (include/volcanosv.h): one Engine = one vsv_handle = one GPU stream.

The stage methods mirror the reference's function boundaries
(extract_contig_signature_Hifi.py: extract_signature_from_cigar :386, extract_sig_from_split_reads :421,
cluster_* :196-288, merge_all :492, pair_sig :548); `run()` is the fused per-chromosome body of :742-772.
"""
import ctypes as C

import numpy as np

from . import _lib
from .abi import (BND_DTYPE, E_CAPACITY, T_BND_CALL_SLOTS, T_BND_CALLS, T_BND_CAND, T_BND_SLOTS, T_CUTESV_SPLIT, BndParams, RedundancyParams, SupportParams, CALL_DTYPE, SIG_DTYPE, T_CALLS, T_CIGAR, T_CLUSTER1, T_MERGED, T_RAW, T_READS, T_SPLIT, DTYPE_BY_NAME,
                  DTYPE_CUTESV, DTYPE_READS, DTYPE_SVIM, Params, Records, VsvError)

_TABLE_IDS = {"raw": T_RAW, "cigar": T_CIGAR, "split": T_SPLIT, "cluster1": T_CLUSTER1, "merged": T_MERGED,
              "calls": T_CALLS, "reads": T_READS, "bnd_cand": T_BND_CAND, "bnd_calls": T_BND_CALLS, "bnd_slots": T_BND_SLOTS, "bnd_call_slots": T_BND_CALL_SLOTS, "cutesv_split": T_CUTESV_SPLIT}


def default_params(dtype):
    if isinstance(dtype, str):
        dtype = DTYPE_BY_NAME[dtype]
    p = Params()
    st = _lib.load().vsv_default_params(int(dtype), C.byref(p))
    if st:
        raise VsvError(st)
    return p


class DeviceRecords:
    """Record SoA already resident in HBM: a dict of torch CUDA tensors (pos,tid,qid,cigar_off,mapq,flag,cigar)."""

    def __init__(self, tensors, n_qids=0, n_tids=0, max_pos=0, tid_lo=0, sync=True):
        """max_pos / tid_lo are the sort-key hints of vsv_records: positions <= max_pos, tids in [tid_lo, n_tids).
        sync=False: the caller orders the producer before the engine itself (Engine.wait_for_stream)."""
        self.t = tensors
        # the engine reads these arrays on ITS stream: whatever produced them (normally torch's current stream) must be done
        if sync:
            import torch
            torch.cuda.current_stream(tensors["pos"].device).synchronize()
        self.max_pos, self.tid_lo = int(max_pos), int(tid_lo)
        self.n_records = int(tensors["pos"].numel())
        self.n_ops = int(tensors["cigar"].numel())
        self.n_qids, self.n_tids = int(n_qids), int(n_tids)

    def as_struct(self):
        r = Records()
        r.n_records, r.n_ops = self.n_records, self.n_ops
        for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar"):
            setattr(r, name, C.c_void_p(self.t[name].data_ptr()))
        r.on_device, r.n_qids, r.n_tids = 1, self.n_qids, self.n_tids
        r.max_pos, r.tid_lo = self.max_pos, self.tid_lo
        return r


class Engine:
    def __init__(self, device=0, stream=None, max_sigs=None, grow=False):
        """grow: when a run needs more signature rows than reserved (VSV_E_CAPACITY), reserve what the library reports and run the
        same input again instead of raising (the drivers use it; the reference has no such limit)."""
        self.grow = bool(grow)
        self._last = None
        self.lib = _lib.load()
        self.h = C.c_void_p()
        st = self.lib.vsv_create(int(device), C.c_void_p(stream or 0), C.byref(self.h))
        if st:
            raise VsvError(st, "vsv_create failed (no MI355X visible?)")
        if max_sigs:
            self.reserve(0, 0, max_sigs)
        self._keep = None

    def close(self):
        if self.h:
            self.lib.vsv_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st:
            raise VsvError(st, self.lib.vsv_last_error(self.h).decode())

    def wait_for_stream(self, hip_stream):
        """Device-side ordering: this engine's stream waits for what has been enqueued on `hip_stream` (a raw hipStream_t, e.g.
        torch.cuda.current_stream().cuda_stream) — for device-resident inputs another stream is still producing."""
        self._check(self.lib.vsv_wait_for_stream(self.h, C.c_void_p(hip_stream or 0)))

    def reserve(self, max_records, max_ops, max_sigs, large_tables=False):
        self._check(self.lib.vsv_reserve(self.h, int(max_records), int(max_ops), int(max_sigs)))
        if large_tables:
            self._check(self.lib.vsv_reserve_large_tables(self.h))

    def last_count(self):
        return int(self.lib.vsv_last_count(self.h))

    def rerun_count(self):
        """Whole-run repetitions vsv_finish() took on this engine so far (bucket-sort overflow / fused-CLR-gate fallbacks)."""
        return int(self.lib.vsv_rerun_count(self.h))

    def sort1_slow_count(self):
        """Runs whose first element sort met a bucket too large for LDS and sorted it in global memory (vsv_sort1_slow_count)."""
        return int(self.lib.vsv_sort1_slow_count(self.h))

    def path_counts(self):
        """(element_runs, cold_syncs) of this engine (vsv_path_counts): runs whose stages behind the split stage worked on 16-byte
        elements, and first runs that waited for the scan once to pick that path from the run's own row count."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.vsv_path_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- stage entry points --------------------------------------------------------------------------
    def _recs(self, soa):
        self._keep = soa  # keep host arrays / tensors alive while the GPU reads them
        return soa.as_struct()

    def cigar_scan(self, soa, params):
        r = self._recs(soa)
        self._check(self.lib.vsv_cigar_scan(self.h, C.byref(r), C.byref(params)))

    def split_pairs(self, params=None):
        self._check(self.lib.vsv_split_pairs(self.h, None, C.byref(params) if params is not None else None))

    def sort_cluster(self, params=None):
        self._check(self.lib.vsv_sort_cluster(self.h, C.byref(params) if params is not None else None))

    def merge_sources(self, params=None):
        self._check(self.lib.vsv_merge_sources(self.h, C.byref(params) if params is not None else None))

    def pair_haplotypes(self, params=None):
        self._check(self.lib.vsv_pair_haplotypes(self.h, C.byref(params) if params is not None else None))

    def _regrow(self, st):
        """VSV_E_CAPACITY with grow=True: reserve 1.25 x the reported row count and run the last input synchronously."""
        for _ in range(4):
            if st != E_CAPACITY or not self.grow or self._last is None:
                break
            self.reserve(0, 0, int(self.last_count() * 1.25) + 1024)
            soa, params = self._last
            r = self._recs(soa)
            st = self.lib.vsv_run_chromosome(self.h, C.byref(r), C.byref(params))
        self._check(st)

    def run(self, soa, params):
        r = self._recs(soa)
        self._last = (soa, params)
        self._regrow(self.lib.vsv_run_chromosome(self.h, C.byref(r), C.byref(params)))

    def run_async(self, soa, params):
        r = self._recs(soa)
        self._last = (soa, params)
        self._check(self.lib.vsv_run_chromosome_async(self.h, C.byref(r), C.byref(params)))

    def stream_ceiling(self, tensor, reps=5):
        """(read GB/s, copy GB/s) of the library's own streaming kernels over a torch device tensor: the measured ceiling the scan
        is compared with, next to the nominal 8 TB/s."""
        r, c = C.c_double(), C.c_double()
        self._check(self.lib.vsv_stream_ceiling(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size(), int(reps), C.byref(r), C.byref(c)))
        return r.value, c.value

    def finish(self):
        self._regrow(self.lib.vsv_finish(self.h))

    # ---- Complex_SV breakend branch (svim_asm/SVIM_inter.py, SVIM_COMBINE.py) ----------------------------------------
    def bnd(self, seg, params=None):
        """segments -> (candidates, calls): vsv_bnd_segments + vsv_bnd_pair."""
        p = params
        if p is None:
            p = BndParams()
            self._check(self.lib.vsv_default_bnd_params(C.byref(p)))
        self._keep = seg
        s = seg.as_struct()
        self._check(self.lib.vsv_bnd_segments(self.h, C.byref(s), C.byref(p)))
        cand = self.table("bnd_cand")
        self._check(self.lib.vsv_bnd_pair(self.h, C.byref(p)))
        return cand, self.table("bnd_calls")

    def bnd_candidates(self, seg, params=None):
        p = params or self._bnd_params()
        self._keep = seg
        s = seg.as_struct()
        self._check(self.lib.vsv_bnd_segments(self.h, C.byref(s), C.byref(p)))
        return self.table("bnd_cand")

    def bnd_pair_rows(self, rows, contig_rank, params=None):
        """Pairs candidate rows (numpy BND_DTYPE, collection order) — the per-rank step after the multi-GPU exchange."""
        p = params or self._bnd_params()
        rows = np.ascontiguousarray(rows)
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        self._check(self.lib.vsv_bnd_set_candidates(self.h, rows.ctypes.data_as(C.c_void_p), len(rows), rank.ctypes.data_as(C.c_void_p),
                                                    len(rank), 0))
        self._check(self.lib.vsv_bnd_pair(self.h, C.byref(p)))
        return self.table("bnd_calls")

    # ---- the same branch without leaving the GPU (multi-GPU exchange, bench config 5) ---------------------------------------
    def bnd_candidates_device(self, seg, device, params=None):
        """vsv_bnd_segments on a device-resident segment table (bnd.DeviceSegments): the live candidate rows as a uint8 torch
        tensor [n * 32] on `device`, in slot order (= read order, pair order)."""
        import torch
        p = params or self._bnd_params()
        self._keep = seg
        s = seg.as_struct()
        self._check(self.lib.vsv_bnd_segments(self.h, C.byref(s), C.byref(p)))
        rows = self.table_torch("bnd_slots", device).view(-1, 32)
        meta = rows[:, 24:28].contiguous().view(torch.int32).view(-1)
        return rows[(meta & 64) == 0].contiguous().view(-1)          # VSV_B_DEAD

    def bnd_pair_device(self, rows_u8, contig_rank_t, device, params=None):
        """vsv_bnd_set_candidates + vsv_bnd_pair on device-resident candidate rows (collection order); calls as a uint8 tensor."""
        import torch
        p = params or self._bnd_params()
        n = rows_u8.numel() // 32
        self._keep = (rows_u8, contig_rank_t)
        # the rows were just produced on torch's current stream (exchange_bnd_device: index, sort, all-to-all); the library copies
        # them on the HANDLE's stream: order the two on the device
        self.wait_for_stream(torch.cuda.current_stream(device).cuda_stream)
        self._check(self.lib.vsv_bnd_set_candidates(self.h, C.c_void_p(rows_u8.data_ptr() if n else 0) if n else None, n,
                                                    C.c_void_p(contig_rank_t.data_ptr()), int(contig_rank_t.numel()), 1))
        self._check(self.lib.vsv_bnd_pair(self.h, C.byref(p)))
        rows = self.table_torch("bnd_call_slots", device).view(-1, 32)
        meta = rows[:, 24:28].contiguous().view(torch.int32).view(-1)
        return rows[(meta & 64) == 0].contiguous().view(-1)

    def cutesv_split(self, seg, read_len, read_rec, sv_size=30, max_size=100000, max_split_parts=7):
        """sig_extract.py analysis_split_read (SE:193-319), INS/DEL candidates: SIG_DTYPE rows in (read, emission) order."""
        self._keep = seg
        s = seg.as_struct()
        rl = np.ascontiguousarray(read_len, dtype=np.int32)
        rr = np.ascontiguousarray(read_rec, dtype=np.uint32)
        self._check(self.lib.vsv_cutesv_split(self.h, C.byref(s), rl.ctypes.data_as(C.c_void_p), rr.ctypes.data_as(C.c_void_p), int(sv_size),
                                              int(max_size), int(max_split_parts)))
        self._cutesv_reads = len(rl)
        return self.table("cutesv_split")

    def cutesv_split_tra(self):
        """Per-read "yields a translocation candidate" flags (uint8) of the last cutesv_split call (vsv_cutesv_split_tra)."""
        out = np.zeros(self._cutesv_reads, dtype=np.uint8)
        self._check(self.lib.vsv_cutesv_split_tra(self.h, out.ctypes.data_as(C.c_void_p), len(out)))
        return out

    def redundancy_params(self, **kw):
        p = RedundancyParams()
        self._check(self.lib.vsv_default_redundancy_params(C.byref(p)))
        for k, v in kw.items():
            if v is not None:
                setattr(p, k, v)
        return p

    def redundancy_pairs(self, is_del, pos, svlen, seq=None, seq_off=None, params=None):
        """remove_redundancy.py match_del_chr / match_ins_chr for the calls of one chromosome (ascending pos): the matching
        pairs i < j as an (n_pairs, 2) uint32 array. seq/seq_off: concatenated ALT strings coded 0..15 (INS only)."""
        p = params or self.redundancy_params()
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        svlen = np.ascontiguousarray(svlen, dtype=np.int32)
        if not is_del:
            seq = np.ascontiguousarray(seq, dtype=np.uint8)
            seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        cap = max(1024, 4 * len(pos))
        while True:
            out = np.zeros((cap, 2), dtype=np.uint32)
            n = C.c_int64()
            st = self.lib.vsv_redundancy_pairs(self.h, 1 if is_del else 0, pos.ctypes.data_as(C.c_void_p), svlen.ctypes.data_as(C.c_void_p),
                                               None if is_del else seq.ctypes.data_as(C.c_void_p),
                                               None if is_del else seq_off.ctypes.data_as(C.c_void_p), len(pos), C.byref(p),
                                               out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
            if st == E_CAPACITY and n.value > cap:  # retry with the reported size
                cap = int(n.value)
                continue
            self._check(st)
            return out[: n.value]

    def bgzf_inflate(self, payloads, isizes, crcs=None):
        """Raw-deflate payloads of BGZF members (list of bytes) -> list of their inflated bytes, decoded on the GPU. crcs: the
        members' trailer CRC-32s; given, every member is checked on the GPU (a mismatch raises)."""
        n = len(payloads)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(p) for p in payloads], out=off[1:])
        comp = np.frombuffer(b"".join(payloads) or b"\0", dtype=np.uint8)
        isz = np.ascontiguousarray(isizes, dtype=np.uint32)
        out = np.zeros(max(1, int(isz.sum())), dtype=np.uint8)
        want = np.ascontiguousarray(crcs, dtype=np.uint32) if crcs is not None else None
        if want is not None:
            self._check(self.lib.vsv_bgzf_set_expected_crc(self.h, want.ctypes.data_as(C.c_void_p), n))
        try:
            self._inflate_call(comp, off, isz, n, out)
        finally:
            if want is not None:
                self.lib.vsv_bgzf_set_expected_crc(self.h, None, 0)
        o = np.concatenate(([0], np.cumsum(isz.astype(np.int64))))
        return [out[o[i]:o[i + 1]].tobytes() for i in range(n)]

    def _inflate_call(self, comp, off, isz, n, out):
        self._check(self.lib.vsv_bgzf_inflate(self.h, comp.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                              isz.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))

    def gt_support(self, var_pos, var_svlen, blk_lo, blk_hi, sig_pos, sig_svlen, sig_cnt, max_shift_ratio=2.3, min_size_sim=0.6):
        """Window sums of correct_gt_*_real_data.py (vsv_gt_support): (sum int64[n], lo int32[n], hi int32[n])."""
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (var_pos, var_svlen, blk_lo, blk_hi, sig_pos, sig_svlen, sig_cnt)]
        nv, ns = len(a[0]), len(a[4])
        s, lo, hi = np.zeros(nv, np.int64), np.zeros(nv, np.int32), np.zeros(nv, np.int32)
        p = [x.ctypes.data_as(C.c_void_p) for x in a]
        self._check(self.lib.vsv_gt_support(self.h, p[0], p[1], p[2], p[3], nv, p[4], p[5], p[6], ns, float(max_shift_ratio), float(min_size_sim),
                                            s.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)))
        return s, lo, hi

    def span_count(self, recs, q_tid, q_a, q_b):
        """Reads with reference_start < a and reference_end > b per query (vsv_span_count); recs = RecordSoA / DeviceRecords /
        DeviceRecordView sorted by (tid, pos)."""
        r = self._recs(recs)
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (q_tid, q_a, q_b)]
        out = np.zeros(len(a[0]), np.uint32)
        self._check(self.lib.vsv_span_count(self.h, C.byref(r), a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p),
                                            a[2].ctypes.data_as(C.c_void_p), len(a[0]), out.ctypes.data_as(C.c_void_p)))
        return out

    def support_params(self, **kw):
        p = SupportParams()
        self._check(self.lib.vsv_default_support_params(C.byref(p)))
        for k, v in kw.items():
            if v is not None:
                setattr(p, k, v)
        return p

    def support_join(self, call_pos, call_len, sig_pos, sig_len, params=None):
        """FP_filter_v1.eval_sig (Large_INDEL/FP_filter_v1.py:106-123) on the GPU: per-call read-signature support.
        numpy int32 arrays in, numpy uint32 out; torch device tensors in, torch int32 tensor out."""
        p = params or self.support_params()
        if hasattr(call_pos, "data_ptr"):
            import torch
            ts = [t.contiguous().to(torch.int32) for t in (call_pos, call_len, sig_pos, sig_len)]
            out = torch.empty(len(ts[0]), dtype=torch.int32, device=ts[0].device)
            # the library reads these on ITS stream: the conversions above (torch's current stream) must have finished; its
            # own stream is synchronised before the call returns, so `out` is complete for whoever reads it next
            torch.cuda.current_stream(ts[0].device).synchronize()
            self._check(self.lib.vsv_support_join(self.h, ts[0].data_ptr(), ts[1].data_ptr(), len(ts[0]), ts[2].data_ptr(), ts[3].data_ptr(),
                                                  len(ts[2]), C.byref(p), 1, out.data_ptr()))
            return out
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (call_pos, call_len, sig_pos, sig_len)]
        out = np.zeros(len(a[0]), dtype=np.uint32)
        ptr = [x.ctypes.data_as(C.c_void_p) for x in a]
        self._check(self.lib.vsv_support_join(self.h, ptr[0], ptr[1], len(a[0]), ptr[2], ptr[3], len(a[2]), C.byref(p), 0,
                                              out.ctypes.data_as(C.c_void_p)))
        return out

    def _int_arrays(self, arrays):
        """(pointers, keep-alive list, on_device, torch device or None) for a tuple of int32 arrays (numpy or torch CUDA)."""
        if hasattr(arrays[0], "data_ptr"):
            import torch
            ts = [t.contiguous().to(torch.int32) for t in arrays]
            torch.cuda.current_stream(ts[0].device).synchronize()      # see support_join: the library reads on its own stream
            return [t.data_ptr() for t in ts], ts, 1, ts[0].device
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in arrays]
        return [x.ctypes.data_as(C.c_void_p) for x in a], a, 0, None

    def support_cov_ins(self, call_pos, sig_pos, sig_len, flanking=1000):
        """calc_ins_call_cov (calculate_signature_support.py:81-125): int64 coverage per call; sig_pos ascending."""
        (cp,), keep_c, dev_c, device = self._int_arrays((call_pos,))
        (sp, sl), keep_s, dev_s, _ = self._int_arrays((sig_pos, sig_len))
        assert dev_c == dev_s
        n, m = len(keep_c[0]), len(keep_s[0])
        if dev_c:
            import torch
            out = torch.empty(n, dtype=torch.int64, device=device)
            self._check(self.lib.vsv_support_cov_ins(self.h, cp, n, sp, sl, m, int(flanking), 1, out.data_ptr()))
            return out
        out = np.zeros(n, dtype=np.int64)
        self._check(self.lib.vsv_support_cov_ins(self.h, cp, n, sp, sl, m, int(flanking), 0, out.ctypes.data_as(C.c_void_p)))
        return out

    def support_cov_del(self, call_start, call_end, sig_start, sig_end, sig_svlen, flanking=1000):
        """calc_del_call_cov (calculate_signature_support.py:138-280): int64 coverage per call; sig_start ascending."""
        (cs, ce), keep_c, dev_c, device = self._int_arrays((call_start, call_end))
        (ss, se, sv), keep_s, dev_s, _ = self._int_arrays((sig_start, sig_end, sig_svlen))
        assert dev_c == dev_s
        n, m = len(keep_c[0]), len(keep_s[0])
        if dev_c:
            import torch
            out = torch.empty(n, dtype=torch.int64, device=device)
            self._check(self.lib.vsv_support_cov_del(self.h, cs, ce, n, ss, se, sv, m, int(flanking), 1, out.data_ptr()))
            return out
        out = np.zeros(n, dtype=np.int64)
        self._check(self.lib.vsv_support_cov_del(self.h, cs, ce, n, ss, se, sv, m, int(flanking), 0, out.ctypes.data_as(C.c_void_p)))
        return out

    def _bnd_params(self):
        p = BndParams()
        self._check(self.lib.vsv_default_bnd_params(C.byref(p)))
        return p

    def scan_ms(self):
        ms = C.c_float()
        self._check(self.lib.vsv_last_scan_ms(self.h, C.byref(ms)))
        return float(ms.value)

    # ---- readback ------------------------------------------------------------------------------------
    def table(self, name):
        tid = _TABLE_IDS[name]
        n = C.c_int64()
        self._check(self.lib.vsv_table_count(self.h, tid, C.byref(n)))
        dt = CALL_DTYPE if name == "calls" else (BND_DTYPE if name.startswith("bnd") else SIG_DTYPE)
        out = np.zeros(int(n.value), dtype=dt)
        if n.value:
            self._check(self.lib.vsv_table_fill(self.h, tid, out.ctypes.data_as(C.c_void_p), n.value, 0))
        return out

    def table_torch(self, name, device):
        """Live table (calls / merged / reads) as a uint8 torch tensor [n_rows * row_bytes] on `device`, copied
        device-to-device by the library — the form the multi-GPU gather ships over RCCL without touching the host."""
        import torch
        tid = _TABLE_IDS[name]
        n = C.c_int64()
        self._check(self.lib.vsv_table_count(self.h, tid, C.byref(n)))
        row = (CALL_DTYPE if name == "calls" else SIG_DTYPE).itemsize
        out = torch.empty(max(int(n.value), 1) * row, dtype=torch.uint8, device=device)
        if n.value:
            # the library copies on the engine's stream and waits for it: the rows are complete when the call returns
            self._check(self.lib.vsv_table_fill(self.h, tid, C.c_void_p(out.data_ptr()), n.value, 1))
        return out[: int(n.value) * row]

    def tables(self, dtype):
        if dtype in (DTYPE_SVIM, DTYPE_CUTESV):
            names = ["raw", "cigar"]
        elif dtype == DTYPE_READS:
            names = ["raw", "cigar", "split", "reads"]
        else:
            names = ["raw", "cigar", "split", "cluster1", "merged", "calls"]
        return {k: self.table(k) for k in names}
