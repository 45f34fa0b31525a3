import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# kernels compiled at graph creation (rf_jit.cpp) are kept on disk for the session: the parity tests build the same
# fused chains many times, in several processes
os.environ.setdefault("RF_JIT_CACHE_DIR", os.path.join(os.environ.get("TMPDIR", "/tmp"), "reforge_amd_jit_cache"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One rf_ctx on device 0 for the whole GPU session (one process, one context)."""
    import reforge_amd as rf

    c = rf.Context(0)
    yield c
    c.close()
