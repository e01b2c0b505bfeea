"""CPU: the driver's build() entry point -- its binding and ABI-version check after the compile step (the compile itself,
~3 minutes of hipcc, is what the driver runs; here it is replaced by the library already built in-tree)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_graft_entry_build_binds_and_checks_the_abi(monkeypatch):
    import __graft_entry__ as g
    from moda_amd import build as b, _lib
    calls = []
    monkeypatch.setattr(b, "build", lambda force=False, verbose=True: calls.append(force))
    assert g.build() is None
    assert calls == [True]                                   # every HIP source is recompiled, not reused
    assert _lib.load().moda_abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "moda_hip.h")).read()
    assert "moda_abi_version" in hdr
