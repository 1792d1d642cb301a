import os
import sys

import pytest

# the block pipeline wants 8 hardware queues; the variable must be in the environment before the first HIP call of the
# process -- the library no longer sets it when it is loaded
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# torch ships its own HIP runtime: load it BEFORE libmpvss_hip.so pulls in the system one, otherwise a later
# torch.cuda initialisation in the same process finds no GPU (bench.py imports torch first for the same reason).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through its C ABI.  Fails (does not skip) when the library or GPU is missing:
    a GPU test must never pass on a silent fallback."""
    from mpvss_rs_amd import Engine
    eng = Engine(0)
    yield eng
    eng.close()
