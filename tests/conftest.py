import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library honours its test hooks (VSV_SORT, VSV_PAIR, VSV_BK_CAP, VSV_BAM_WINDOW, ...) only under VSV_DEBUG=1 (csrc/vsv_env.h)
os.environ.setdefault("VSV_DEBUG", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _built_native_code():
    """Build the HIP library and the CPU oracle once per session (hipcc cross-compiles without a GPU). The product
    itself never builds or falls back on its own: volcanosv_amd/_lib.py raises if the .so is missing."""
    import __graft_entry__
    __graft_entry__.build()
