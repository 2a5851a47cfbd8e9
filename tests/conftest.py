import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The tests exercise the in-tree libnidx_gpu.so; on a fresh checkout (the .so is git-ignored) build it
    first — hipcc cross-compiles gfx950 with or without a GPU.  This compiles the product, it is not a fallback."""
    from nucliadb_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/nidx_oracle.h)."""
    from oracle import oracle

    oracle.build()
    return oracle
