import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Most GPU tests assert BYTE identities between renders of different batch sizes / workers / entry points.  Those hold for the batch-independent
# nets; a max_batch <= 2 render is low-latency (split-K) by default since round 5 and differs from them in fp32 summation order (<= 2 LSB).  The
# suite therefore pins the batch-independent nets, and the tests OF the default (tests/test_gpu_deoldify.py::test_low_latency_*) ask for it explicitly.
os.environ.setdefault("HAVC_LOW_LATENCY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One havc context on GPU 0 for the whole session (GPU tests only)."""
    from vsdeoldify_amd.render import get_context
    return get_context(0)
