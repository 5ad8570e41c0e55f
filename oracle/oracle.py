"""ctypes binding of the CPU oracle (oracle/vsv_oracle.c). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package volcanosv_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from volcanosv_amd.abi import BND_DTYPE, CALL_DTYPE, SIG_DTYPE, BndParams, Params, Records, RedundancyParams, Segments, SupportParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Out(C.Structure):
    _fields_ = [(n, t) for pair in (
        ("raw", "n_raw"), ("cigar", "n_cigar"), ("split", "n_split"), ("cluster1", "n_cluster1"),
        ("merged", "n_merged"), ("calls", "n_calls"), ("reads", "n_reads"))
        for n, t in ((pair[0], C.c_void_p), (pair[1], C.c_int64))] + [("status", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libvsv_oracle.so")
    src = os.path.join(_HERE, "vsv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_run.argtypes = [C.POINTER(Records), C.POINTER(Params), C.c_int, C.POINTER(_Out)]
        _LIB.orc_run.restype = C.c_int
        _LIB.orc_free.argtypes = [C.POINTER(_Out)]
        _LIB.orc_default_params.argtypes = [C.c_int, C.POINTER(Params)]
        _LIB.orc_default_bnd_params.argtypes = [C.POINTER(BndParams)]
        _LIB.orc_bnd.argtypes = [C.POINTER(Segments), C.POINTER(BndParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        _LIB.orc_bnd_free.argtypes = [C.c_void_p, C.c_void_p]
        _LIB.orc_bnd_pair.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(BndParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        _LIB.orc_gt_support.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 4 + [C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _LIB.orc_span_count.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]
        _LIB.orc_levenshtein.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        _LIB.orc_levenshtein.restype = C.c_int64
        _LIB.orc_default_redundancy_params.argtypes = [C.POINTER(RedundancyParams)]
        _LIB.orc_redundancy_pairs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RedundancyParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        _LIB.orc_free_u32.argtypes = [C.c_void_p]
        _LIB.orc_cutesv_split.argtypes = [C.POINTER(Segments), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
        _LIB.orc_cov_ins.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        _LIB.orc_cov_del.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        _LIB.orc_default_support_params.argtypes = [C.POINTER(SupportParams)]
        _LIB.orc_support.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(SupportParams), C.c_void_p]
    return _LIB


def default_params(dtype):
    p = Params()
    lib().orc_default_params(int(dtype), C.byref(p))
    return p


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def run(soa, params=None, dtype=None, literal=False):
    """Runs the whole path on the host. Returns (status, dict of stage tables)."""
    p = params if params is not None else default_params(dtype)
    r = soa.as_struct()
    out = _Out()
    st = lib().orc_run(C.byref(r), C.byref(p), 1 if literal else 0, C.byref(out))
    tabs = {
        "raw": _copy(out.raw, out.n_raw, SIG_DTYPE), "cigar": _copy(out.cigar, out.n_cigar, SIG_DTYPE),
        "split": _copy(out.split, out.n_split, SIG_DTYPE), "cluster1": _copy(out.cluster1, out.n_cluster1, SIG_DTYPE),
        "merged": _copy(out.merged, out.n_merged, SIG_DTYPE), "calls": _copy(out.calls, out.n_calls, CALL_DTYPE),
        "reads": _copy(out.reads, out.n_reads, SIG_DTYPE),
    }
    lib().orc_free(C.byref(out))
    return st, tabs


def default_bnd_params():
    p = BndParams()
    lib().orc_default_bnd_params(C.byref(p))
    return p


def run_bnd(seg, params=None):
    """BND branch on the host: returns (candidates, calls) as BND_DTYPE arrays."""
    p = params if params is not None else default_bnd_params()
    s = seg.as_struct()
    a, b, na, nb = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
    st = lib().orc_bnd(C.byref(s), C.byref(p), C.byref(a), C.byref(na), C.byref(b), C.byref(nb))
    assert st == 0
    cand, calls = _copy(a.value, na.value, BND_DTYPE), _copy(b.value, nb.value, BND_DTYPE)
    lib().orc_bnd_free(a, b)
    return cand, calls


def run_bnd_pair(rows, contig_rank, params=None):
    """Pairing step alone on candidate rows in collection order (hp1 rows, then hp2 rows)."""
    p = params if params is not None else default_bnd_params()
    rows = np.ascontiguousarray(rows)
    rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
    b, nb = C.c_void_p(), C.c_int64()
    st = lib().orc_bnd_pair(rows.ctypes.data_as(C.c_void_p), len(rows), rank.ctypes.data_as(C.c_void_p), C.byref(p), C.byref(b), C.byref(nb))
    assert st == 0
    calls = _copy(b.value, nb.value, BND_DTYPE)
    lib().orc_bnd_free(b, None)
    return calls


def default_support_params(**kw):
    p = SupportParams()
    lib().orc_default_support_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def run_support(call_pos, call_len, sig_pos, sig_len, params=None):
    """FP_filter_v1.eval_sig on the host (literal loop). Returns (status, support uint32[n_calls])."""
    p = params if params is not None else default_support_params()
    a = [np.ascontiguousarray(x, dtype=np.int32) for x in (call_pos, call_len, sig_pos, sig_len)]
    out = np.zeros(len(a[0]), dtype=np.uint32)
    st = lib().orc_support(a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p), len(a[0]), a[2].ctypes.data_as(C.c_void_p),
                           a[3].ctypes.data_as(C.c_void_p), len(a[2]), C.byref(p), out.ctypes.data_as(C.c_void_p))
    return st, out


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def run_cov_ins(call_pos, sig_pos, sig_len, flanking=1000):
    """calc_ins_call_cov on the host: int64 coverage per call."""
    c, sp, sl = _i32(call_pos), _i32(sig_pos), _i32(sig_len)
    out = np.zeros(len(c), dtype=np.int64)
    st = lib().orc_cov_ins(c.ctypes.data_as(C.c_void_p), len(c), sp.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), len(sp),
                           int(flanking), out.ctypes.data_as(C.c_void_p))
    return st, out


def run_cov_del(call_start, call_end, sig_start, sig_end, sig_svlen, flanking=1000):
    """calc_del_call_cov on the host: (status, int64 coverage per call, has-entry flags)."""
    cs, ce, ss, se, sv = _i32(call_start), _i32(call_end), _i32(sig_start), _i32(sig_end), _i32(sig_svlen)
    out = np.zeros(len(cs), dtype=np.int64)
    has = np.zeros(len(cs), dtype=np.uint8)
    st = lib().orc_cov_del(cs.ctypes.data_as(C.c_void_p), ce.ctypes.data_as(C.c_void_p), len(cs), ss.ctypes.data_as(C.c_void_p),
                           se.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p), len(ss), int(flanking),
                           out.ctypes.data_as(C.c_void_p), has.ctypes.data_as(C.c_void_p))
    return st, out, has


def run_cutesv_split(seg, read_len, read_rec, sv_size=30, max_size=100000, max_split_parts=7, tra=False):
    """analysis_split_read INS/DEL candidates on the host: SIG_DTYPE rows in (read, emission) order. tra=True: also the per-read
    "yields a translocation candidate" flags (uint8)."""
    s = seg.as_struct()
    rl = np.ascontiguousarray(read_len, dtype=np.int32)
    rr = np.ascontiguousarray(read_rec, dtype=np.uint32)
    a, na = C.c_void_p(), C.c_int64()
    flags = np.zeros(len(rl), dtype=np.uint8)
    st = lib().orc_cutesv_split(C.byref(s), rl.ctypes.data_as(C.c_void_p), rr.ctypes.data_as(C.c_void_p), int(sv_size), int(max_size),
                                int(max_split_parts), C.byref(a), C.byref(na), flags.ctypes.data_as(C.c_void_p))
    assert st == 0
    rows = _copy(a.value, na.value, SIG_DTYPE)
    lib().orc_bnd_free(a, None)
    return (rows, flags) if tra else rows


def levenshtein(a, b):
    """Unit-cost global edit distance of two byte strings (edlib.align default mode)."""
    a, b = (x.encode() if isinstance(x, str) else bytes(x) for x in (a, b))
    return int(lib().orc_levenshtein(a, len(a), b, len(b)))


def default_redundancy_params(**kw):
    p = RedundancyParams()
    lib().orc_default_redundancy_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def run_redundancy_pairs(is_del, pos, svlen, seq=None, seq_off=None, params=None):
    """match_del_chr / match_ins_chr on the host: (status, (n_pairs, 2) uint32 pairs i < j)."""
    p = params if params is not None else default_redundancy_params()
    pos, svlen = _i32(pos), _i32(svlen)
    sq = np.ascontiguousarray(seq if seq is not None else [], dtype=np.uint8)
    so = np.ascontiguousarray(seq_off if seq_off is not None else [0], dtype=np.uint64)
    a, na = C.c_void_p(), C.c_int64()
    st = lib().orc_redundancy_pairs(1 if is_del else 0, pos.ctypes.data_as(C.c_void_p), svlen.ctypes.data_as(C.c_void_p),
                                    sq.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), len(pos), C.byref(p), C.byref(a), C.byref(na))
    if st:
        return st, np.zeros((0, 2), np.uint32)
    out = np.frombuffer((C.c_char * (na.value * 8)).from_address(a.value), dtype=np.uint32).reshape(-1, 2).copy() if na.value else np.zeros((0, 2), np.uint32)
    lib().orc_free_u32(a)
    return 0, out


def run_gt_support(var_chrom, var_pos, var_svlen, sig_chrom, sig_pos, sig_svlen, sig_cnt, max_shift_ratio=2.3, min_size_sim=0.6):
    """match_varlist_siglist / extract_sig_support on the host (literal scans): (support list, resume-index list)."""
    a = [_i32(x) for x in (var_chrom, var_pos, var_svlen, sig_chrom, sig_pos, sig_svlen, sig_cnt)]
    nv, ns = len(a[0]), len(a[3])
    cnt, match = np.zeros(nv, np.int64), np.zeros(nv, np.int64)
    p = [x.ctypes.data_as(C.c_void_p) for x in a]
    lib().orc_gt_support(p[0], p[1], p[2], nv, p[3], p[4], p[5], p[6], ns, float(max_shift_ratio), float(min_size_sim),
                         cnt.ctypes.data_as(C.c_void_p), match.ctypes.data_as(C.c_void_p))
    return cnt, match


def run_span_count(read_tid, read_start, read_end, q_tid, q_a, q_b):
    """count_reads_span_region on the host: reads with start < a and end > b among those fetch(chrom, a, b) returns."""
    a = [_i32(x) for x in (read_tid, read_start, read_end, q_tid, q_a, q_b)]
    out = np.zeros(len(a[3]), np.uint32)
    p = [x.ctypes.data_as(C.c_void_p) for x in a]
    lib().orc_span_count(p[0], p[1], p[2], len(a[0]), p[3], p[4], p[5], len(a[3]), out.ctypes.data_as(C.c_void_p))
    return out
