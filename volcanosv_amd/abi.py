"""Shared layouts of the C-ABI (include/volcanosv.h): ctypes structs + numpy dtypes.

The field order and sizes here must match include/volcanosv.h exactly; tests/test_abi.py checks
sizeof() against the compiled library.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 2

# dtype of the extractor (reference: extract_contig_signature_{Hifi,ONT,CLR}.py, extract_reads_signature.py,
# svim_asm/SVIM_intra.py)
DTYPE_HIFI, DTYPE_ONT, DTYPE_CLR, DTYPE_READS, DTYPE_SVIM, DTYPE_CUTESV = 0, 1, 2, 3, 4, 5
DTYPE_BY_NAME = {"Hifi": DTYPE_HIFI, "ONT": DTYPE_ONT, "CLR": DTYPE_CLR, "READS": DTYPE_READS, "SVIM": DTYPE_SVIM, "CUTESV": DTYPE_CUTESV}

F_REVERSE, F_SUPP, F_HP1, F_HP2, F_SECONDARY, F_UNMAPPED, F_SKIP, F_SEQ_MISMATCH = 1, 2, 4, 8, 16, 32, 64, 128
M_DEL, M_SPLIT, M_HP2, M_DEAD, M_QREV = 1, 2, 4, 8, 16

T_RAW, T_CIGAR, T_SPLIT, T_CLUSTER1, T_MERGED, T_CALLS, T_READS = range(7)
SCAN_AUTO, SCAN_READS, SCAN_CONTIGS = 0, 1, 2      # vsv_params.scan_layout
OVERLAP_AUTO, OVERLAP_OFF = 0, 1                     # vsv_params.split_overlap

STATUS = {
    0: "VSV_OK", -1: "VSV_E_INVALID", -2: "VSV_E_HIP", -3: "VSV_E_CAPACITY", -4: "VSV_E_EMPTY_CIGAR",
    -5: "VSV_E_REFEND", -6: "VSV_E_READLEN", -7: "VSV_E_UNSORTED", -8: "VSV_E_ZERODIV", -9: "VSV_E_NO_DEVICE", -10: "VSV_E_SEQLEN",
}

E_CAPACITY = -3        # vsv_status values the wrappers act on (the rest only travel inside VsvError)

SIG_DTYPE = np.dtype([
    ("pos", "<i4"), ("svlen", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"),
    ("rec", "<u4"), ("rec2", "<u4"), ("meta", "<u4"), ("tid", "<i4"),
])
CALL_DTYPE = np.dtype([("sig", SIG_DTYPE), ("a", "<i4"), ("b", "<i4"), ("gt", "<i4"), ("pad", "<i4")])
assert SIG_DTYPE.itemsize == 32 and CALL_DTYPE.itemsize == 48


class Records(C.Structure):
    _fields_ = [
        ("n_records", C.c_int64), ("n_ops", C.c_int64),
        ("pos", C.c_void_p), ("tid", C.c_void_p), ("qid", C.c_void_p), ("cigar_off", C.c_void_p),
        ("mapq", C.c_void_p), ("flag", C.c_void_p), ("cigar", C.c_void_p),
        ("on_device", C.c_int32), ("n_qids", C.c_int32), ("n_tids", C.c_int32), ("max_pos", C.c_int32),
        ("tid_lo", C.c_int32), ("reserved0", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("min_svlen", C.c_int32), ("min_cigar_mapq", C.c_int32),
        ("min_split_mapq", C.c_int32), ("max_split_svlen", C.c_int32), ("cluster_shift", C.c_int32),
        ("pair_shift", C.c_int32), ("pair_window", C.c_int32), ("enable_split", C.c_int32),
        ("merge_ins_threshold", C.c_int32), ("merge_del_threshold", C.c_int32), ("scan_layout", C.c_int32), ("split_overlap", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class Segments(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("n_segs", C.c_int64), ("seg_off", C.c_void_p),
        ("q_start", C.c_void_p), ("q_end", C.c_void_p), ("ref_id", C.c_void_p), ("ref_start", C.c_void_p), ("ref_end", C.c_void_p),
        ("is_reverse", C.c_void_p), ("hap", C.c_void_p), ("contig_len", C.c_void_p), ("contig_rank", C.c_void_p),
        ("n_tids", C.c_int32), ("on_device", C.c_int32),
    ]


class BndParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("min_sv_size", "max_sv_size", "query_gap_tolerance", "query_overlap_tolerance",
                                         "reference_gap_tolerance", "reference_overlap_tolerance", "partition_max_distance",
                                         "pair_distance", "max_partition")] + [("reserved", C.c_int32 * 7)]


class SupportParams(C.Structure):
    """vsv_support_params (FP_filter_v1.py:8-11 defaults)."""
    _fields_ = [("max_comp_svlen", C.c_int32), ("max_dist", C.c_int32), ("max_shift", C.c_int32), ("pad", C.c_int32),
                ("min_size_sim", C.c_double)]


class RedundancyParams(C.Structure):
    """vsv_redundancy_params (remove_redundancy.py:9-14 defaults)."""
    _fields_ = [("dist_thresh", C.c_int32), ("dist_thresh_del", C.c_int32), ("overlap_thresh", C.c_double), ("size_sim_thresh", C.c_double),
                ("size_sim_thresh_del", C.c_double), ("seq_sim_thresh", C.c_double)]


BND_DTYPE = np.dtype([("src_tid", "<i4"), ("src_pos", "<i4"), ("dst_tid", "<i4"), ("dst_pos", "<i4"), ("read", "<u4"), ("read2", "<u4"),
                      ("meta", "<u4"), ("pad", "<u4")])
B_SRC_FWD, B_DST_FWD, B_HAP2, B_GT_SHIFT, B_DEAD = 1, 2, 4, 4, 64
T_BND_CAND, T_BND_CALLS, T_BND_SLOTS, T_BND_CALL_SLOTS = 7, 8, 11, 12
T_CUTESV_SPLIT = 10


class VsvError(RuntimeError):
    """Raised when a C-ABI call returns a negative status. `.status` holds the vsv_status value; the
    reference raises AssertionError / IndexError / ZeroDivisionError at the cited lines instead."""

    def __init__(self, status, msg=""):
        self.status = int(status)
        super().__init__("%s (%d)%s" % (STATUS.get(int(status), "VSV_E_?"), status, (": " + msg) if msg else ""))
