import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")



# Round 6 (the GPU suite ran 864 s of the driver's 1 200 s): seeded state dicts are cached per process (vsdeoldify_amd.synth) and the
# renders of this session share ONE packed + uploaded weight blob per (state dict object, layout, precision) -- the product's own switch for worker contexts
# (render._shared_weights) -- instead of packing the same 225 M parameters (5 s fast, 20 s precise) in forty tests.  Every parity assertion is unchanged.
os.environ.setdefault("HAVC_SHARE_WEIGHTS", "1")
# The package default arithmetic is "precise" since round 6 (vsdeoldify_amd/precision.py).  The suite was written against the fast (fp16) mode with
# explicit precision="precise" wherever the precise mode is under test: it keeps running that way through the documented process-wide switch; the
# DEFAULT itself is checked by tests/test_host_logic.py (resolution order) and tests/test_gpu_precise.py (a render built without any switch).
os.environ.setdefault("HAVC_PRECISION", "fast")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One havc context on GPU 0 for the whole session (GPU tests only)."""
    from vsdeoldify_amd.render import get_context
    return get_context(0)
