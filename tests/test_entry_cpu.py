"""CPU: the driver's build() entry point itself (hipcc cross-compile of every HIP source + the C-ABI binding and version
check) -- the check the driver runs without a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_graft_entry_build_runs():
    import __graft_entry__ as g
    assert g.build() is None
    from moda_amd import _lib
    assert _lib.load().moda_abi_version() == _lib.ABI_VERSION
