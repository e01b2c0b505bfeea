import os
import sys

import pytest

# The suite's implicit mode is the EXACT one: tests that do not name a precision compare at 1e-5 .. 1e-6 against the oracle, and
# every test of a throughput mode names it.  (The library's own default is the parity-grade fp16 mode, moda_amd/nerf.py
# _initial_precision; tests/test_host_cpu.py and tests/test_gpu_default_mode.py check THAT.)
os.environ.setdefault("MODA_PRECISION", "fp32")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _gpu_quiesce(request):
    """GPU tests start from an idle device: nothing a previous test left in flight (side streams of the graph tests, frees in
    the caching allocator) overlaps the launches of this one."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    yield
