"""ctypes loader of the in-tree HIP library (volcanosv_amd/libvolcanosv_hip.so).

There is NO CPU fallback: if the library is missing or cannot be loaded this raises, and every product
entry point goes through it. (The CPU oracle under oracle/ is test infrastructure and is never imported
from this package.)
"""
import ctypes as C
import os

from .abi import ABI_VERSION, BndParams, Params, Records, RedundancyParams, Segments, SupportParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvolcanosv_hip.so")
if os.environ.get("VSV_DEBUG") == "1" and os.environ.get("VSV_LIB"):      # A/B timing of two builds on the same box (tools/ab.sh)
    LIB_PATH = os.environ["VSV_LIB"]
_lib = None

# every symbol include/volcanosv.h declares
SYMBOLS = [
    "vsv_abi_version", "vsv_status_string", "vsv_create", "vsv_destroy", "vsv_bam_device_want_sa", "vsv_bam_device_sa_tags", "vsv_bam_device_want_seq", "vsv_bam_device_seq_slices", "vsv_stream_ceiling", "vsv_bgzf_set_expected_crc", "vsv_last_error", "vsv_last_count", "vsv_rerun_count", "vsv_sort1_slow_count", "vsv_path_counts", "vsv_reserve_large_tables",
    "vsv_default_params", "vsv_reserve", "vsv_wait_for_stream", "vsv_cutesv_split_tra", "vsv_cigar_scan", "vsv_split_pairs", "vsv_sort_cluster", "vsv_merge_sources",
    "vsv_pair_haplotypes", "vsv_run_chromosome", "vsv_run_chromosome_async", "vsv_finish", "vsv_table_count",
    "vsv_table_fill", "vsv_last_scan_ms", "vsv_default_bnd_params", "vsv_bnd_segments", "vsv_bnd_pair", "vsv_bnd_set_candidates",
    "vsv_cutesv_split", "vsv_gt_support", "vsv_span_count", "vsv_bgzf_inflate", "vsv_bam_set_inflate_device", "vsv_bam_parse_device", "vsv_copy_to_host", "vsv_bam_load_device", "vsv_bam_l_seq_device",
    "vsv_bam_sam_flags_device", "vsv_default_redundancy_params", "vsv_redundancy_pairs", "vsv_default_support_params", "vsv_support_join", "vsv_support_cov_ins", "vsv_support_cov_del",
    "vsv_bam_open", "vsv_bam_close", "vsv_bam_error", "vsv_bam_set_threads", "vsv_bam_n_refs", "vsv_bam_ref_name", "vsv_bam_ref_len", "vsv_bam_load",
    "vsv_bam_qnames", "vsv_bam_sa_tags", "vsv_bam_l_seq", "vsv_bam_sam_flags", "vsv_bam_set_keep_seq", "vsv_bam_seq",
]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP extension %s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or python volcanosv_amd/build.py); there is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7, requested by
    # torch under the name libamdhip64.so). If this library pulled in /opt/rocm's copy first, torch would load a
    # second runtime and its streams / device would not be ours. Importing torch first makes the dynamic loader
    # resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.vsv_abi_version.restype = C.c_int
    lib.vsv_status_string.restype = C.c_char_p
    lib.vsv_status_string.argtypes = [C.c_int]
    lib.vsv_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(H)]
    lib.vsv_destroy.argtypes = [H]
    lib.vsv_destroy.restype = None
    lib.vsv_last_error.argtypes = [H]
    lib.vsv_last_error.restype = C.c_char_p
    lib.vsv_last_count.argtypes = [H]
    lib.vsv_last_count.restype = C.c_int64
    lib.vsv_rerun_count.argtypes = [H]
    lib.vsv_rerun_count.restype = C.c_int64
    lib.vsv_sort1_slow_count.argtypes = [H]
    lib.vsv_sort1_slow_count.restype = C.c_int64
    lib.vsv_path_counts.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.vsv_reserve_large_tables.argtypes = [H]
    lib.vsv_default_params.argtypes = [C.c_int, C.POINTER(Params)]
    lib.vsv_reserve.argtypes = [H, C.c_int64, C.c_int64, C.c_int64]
    lib.vsv_wait_for_stream.argtypes = [H, C.c_void_p]
    lib.vsv_cutesv_split_tra.argtypes = [H, C.c_void_p, C.c_int64]
    for name in ("vsv_cigar_scan", "vsv_split_pairs", "vsv_run_chromosome", "vsv_run_chromosome_async"):
        getattr(lib, name).argtypes = [H, C.POINTER(Records), C.POINTER(Params)]
    for name in ("vsv_sort_cluster", "vsv_merge_sources", "vsv_pair_haplotypes"):
        getattr(lib, name).argtypes = [H, C.POINTER(Params)]
    lib.vsv_finish.argtypes = [H]
    lib.vsv_bam_device_want_sa.argtypes = [H, C.c_int]
    lib.vsv_bam_device_want_seq.argtypes = [H, C.c_int]
    lib.vsv_bam_device_seq_slices.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.vsv_bam_device_sa_tags.argtypes = [H, C.POINTER(C.c_int64)]
    lib.vsv_bam_device_sa_tags.restype = C.c_void_p
    lib.vsv_stream_ceiling.argtypes = [H, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.vsv_table_count.argtypes = [H, C.c_int, C.POINTER(C.c_int64)]
    lib.vsv_table_fill.argtypes = [H, C.c_int, C.c_void_p, C.c_int64, C.c_int]
    lib.vsv_last_scan_ms.argtypes = [H, C.POINTER(C.c_float)]
    lib.vsv_default_bnd_params.argtypes = [C.POINTER(BndParams)]
    lib.vsv_bnd_segments.argtypes = [H, C.POINTER(Segments), C.POINTER(BndParams)]
    lib.vsv_bnd_pair.argtypes = [H, C.POINTER(BndParams)]
    lib.vsv_bnd_set_candidates.argtypes = [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int]
    lib.vsv_cutesv_split.argtypes = [H, C.POINTER(Segments), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.vsv_default_redundancy_params.argtypes = [C.POINTER(RedundancyParams)]
    lib.vsv_redundancy_pairs.argtypes = [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RedundancyParams), C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64)]
    lib.vsv_bam_parse_device.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(Records),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vsv_gt_support.argtypes = [H] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64, C.c_double, C.c_double] + [C.c_void_p] * 3
    lib.vsv_span_count.argtypes = [H, C.POINTER(Records), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.vsv_copy_to_host.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int64]
    lib.vsv_bgzf_inflate.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.vsv_bgzf_set_expected_crc.argtypes = [H, C.c_void_p, C.c_int64]
    lib.vsv_support_cov_ins.argtypes = [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p]
    lib.vsv_support_cov_del.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_int, C.c_void_p]
    lib.vsv_default_support_params.argtypes = [C.POINTER(SupportParams)]
    lib.vsv_support_join.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(SupportParams),
                                     C.c_int, C.c_void_p]
    for name in SYMBOLS:
        if name not in ("vsv_destroy", "vsv_last_error", "vsv_last_count", "vsv_rerun_count", "vsv_sort1_slow_count", "vsv_status_string") and not name.startswith("vsv_bam"):
            getattr(lib, name).restype = C.c_int
    B = C.c_void_p
    lib.vsv_bam_open.argtypes = [C.c_char_p, C.POINTER(B)]
    lib.vsv_bam_open.restype = C.c_int
    lib.vsv_bam_close.argtypes = [B]
    lib.vsv_bam_close.restype = None
    lib.vsv_bam_set_threads.argtypes = [B, C.c_int]
    lib.vsv_bam_set_threads.restype = None
    lib.vsv_bam_error.argtypes = [B]
    lib.vsv_bam_error.restype = C.c_char_p
    lib.vsv_bam_n_refs.argtypes = [B]
    lib.vsv_bam_n_refs.restype = C.c_int
    lib.vsv_bam_ref_name.argtypes = [B, C.c_int]
    lib.vsv_bam_ref_name.restype = C.c_char_p
    lib.vsv_bam_ref_len.argtypes = [B, C.c_int]
    lib.vsv_bam_ref_len.restype = C.c_int64
    lib.vsv_bam_load_device.argtypes = [B, H, C.c_int, C.POINTER(Records)]
    lib.vsv_bam_load_device.restype = C.c_int
    lib.vsv_bam_load.argtypes = [B, C.c_int, C.POINTER(Records)]
    lib.vsv_bam_load.restype = C.c_int
    lib.vsv_bam_set_inflate_device.argtypes = [B, H]
    lib.vsv_bam_set_inflate_device.restype = None
    lib.vsv_bam_set_keep_seq.argtypes = [B, C.c_int]
    lib.vsv_bam_set_keep_seq.restype = None
    for name in ("vsv_bam_qnames", "vsv_bam_sa_tags", "vsv_bam_seq"):
        getattr(lib, name).argtypes = [B, C.POINTER(C.c_int64)]
        getattr(lib, name).restype = C.c_void_p
    for name in ("vsv_bam_l_seq", "vsv_bam_sam_flags", "vsv_bam_l_seq_device", "vsv_bam_sam_flags_device"):
        getattr(lib, name).argtypes = [B]
        getattr(lib, name).restype = C.c_void_p
    if lib.vsv_abi_version() != ABI_VERSION:
        raise ImportError("ABI mismatch: library %d, python %d" % (lib.vsv_abi_version(), ABI_VERSION))
    _lib = lib
    return lib
