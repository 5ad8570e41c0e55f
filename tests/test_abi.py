"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/volcanosv.h declares,
struct layouts match, and the product fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from volcanosv_amd import _lib
from volcanosv_amd.abi import CALL_DTYPE, SIG_DTYPE, Params, Records, VsvError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "volcanosv.h")).read()
    declared = set(re.findall(r"\b(vsv_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(lib):
    assert C.sizeof(Records) == 8 + 8 + 7 * 8 + 6 * 4
    assert C.sizeof(Params) == 16 * 4
    assert SIG_DTYPE.itemsize == 32 and CALL_DTYPE.itemsize == 48
    assert lib.vsv_abi_version() == 2


def test_default_params_are_reference_constants(lib):
    p = Params()
    assert lib.vsv_default_params(0, C.byref(p)) == 0
    assert (p.min_svlen, p.min_cigar_mapq, p.min_split_mapq, p.max_split_svlen) == (30, 50, 50, 50000)
    assert (p.cluster_shift, p.pair_shift, p.pair_window) == (100, 200, 1000)
    assert lib.vsv_default_params(3, C.byref(p)) == 0 and p.min_split_mapq == 0
    assert lib.vsv_default_params(99, C.byref(p)) == -1


def test_no_silent_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from volcanosv_amd.engine import Engine
    with pytest.raises(VsvError) as e:
        Engine(0)
    assert e.value.status == -9


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "volcanosv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".h"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "vsv_oracle" not in txt, f
