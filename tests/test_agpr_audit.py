"""CPU: the machine code of the asm-owned-AGPR kernels (PrecBF16A, moda_amd/csrc/mlp_fused.hip) as built into libmoda_hip.so.
hipcc neither schedules nor pads what is inside an asm statement and does not know the literally named AGPRs, so the build itself is
audited (tools/agpr_audit.py: no scratch, no compiler access to the accumulator file, nothing touching an MFMA's destination before it
has landed, no VALU write in front of an MFMA operand read)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _audit_module():
    from moda_amd import isa_audit
    return isa_audit


def test_agpr_kernels_pass_the_isa_audit():
    from moda_amd import build
    lib = build.build()                       # (no-op when the library is up to date)
    au = _audit_module()
    ks = {k: v for k, v in au.disassemble(lib).items() if ("PrecBF16A" in k or "PrecF16A" in k) and "mlp_fused_kernel" in k}
    assert len(ks) == 4, sorted(ks)           # bf16 and fp16, the two ENDY forms each, of the four-wave two-column-block kernel
    for name, k in ks.items():
        assert k["scratch"] == 0 and k["agpr"] == 256, (name, k["scratch"], k["agpr"])
        n_mfma = sum(1 for t in k["ins"] if t.startswith(("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16")))
        assert n_mfma > 1000, n_mfma
        # every MFMA of the hidden layers takes its B operand from the accumulator file
        assert sum(1 for t in k["ins"] if t.startswith("v_mfma") and ", a[" in t) > 900
        assert au.audit(k["ins"], k["scratch"]) == []


def test_the_audit_finds_what_it_is_there_for():
    """The three hazards of round 5, as synthetic instruction streams."""
    au = _audit_module()
    mf = "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], a[0:3], v[0:15]"
    assert au.audit([mf, "v_mov_b32_e32 v40, v3"], 0)                                   # a copy right behind the chain
    assert au.audit([mf, "ds_read_b128 v[4:7], v90"], 0)                                # a register of the tile reused at once
    assert au.audit(["v_mov_b32_e32 v5, v33", mf], 0)                                   # a VALU write in front of the operand read
    assert au.audit(["v_accvgpr_read_b32 v9, a7"], 0) and au.audit(["v_accvgpr_write_b32 a7, v9"], 0) and au.audit([mf], 64)
    clean = ["v_mov_b32_e32 v5, v33", "s_nop 1", mf, "v_mfma_f32_32x32x16_bf16 v[32:47], v[16:19], a[64:67], v[32:47]",
             "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], a[4:7], v[0:15]", "s_nop 15", "v_cvt_pk_bf16_f32 v60, v0, v1",
             "v_pk_max_i16 v60, v60, 0", "v_accvgpr_write_b32 a[128], v60"]
    assert au.audit(clean, 0) == []


def test_the_generic_operand_rules():
    """Rules 4 and 5 are not patterns: ANY vector write (either register file) of ANY MFMA source fewer than two wait states ahead,
    and ANY recent MFMA result read as A / B or accumulated onto in part."""
    au = _audit_module()
    mf = "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], a[0:3], v[0:15]"
    cvt = "v_cvt_pk_bf16_f32 v60, v0, v1"
    assert au.audit([cvt, "v_accvgpr_write_b32 a2, v60", mf], 0)                        # accumulator-file B operand, 0 states
    assert au.audit([cvt, "v_accvgpr_write_b32 a2, v60", "s_nop 0", mf], 0)             # 1 state
    assert au.audit([cvt, "v_accvgpr_write_b32 a2, v60", "s_nop 1", mf], 0) == []       # 2 states: clean
    assert au.audit(["v_add_f32_e32 v7, v1, v2", "s_nop 0", mf], 0)                     # the C operand counts too
    assert au.audit(["v_fma_f32 v17, v1, v2, v3", "v_nop", mf], 0)                      # A operand, one unrelated instruction between
    a_from_d = "v_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], a[8:11], v[32:47]"
    assert au.audit([mf, a_from_d], 0) and au.audit([mf, "s_nop 11", a_from_d], 0) == []
    partial = "v_mfma_f32_32x32x16_bf16 v[8:23], v[24:27], a[8:11], v[8:23]"
    assert au.audit([mf, partial], 0)
    assert au.audit([mf, mf], 0) == []                                                  # a whole-tile accumulate chain is free


def test_the_build_stamps_the_audit_and_the_loader_reads_it(tmp_path, monkeypatch):
    from moda_amd import build, _lib
    build.build()
    assert build.audit_built_library(verbose=False)
    lines = open(build.AUDIT_STAMP).read().splitlines()
    assert lines[0] == "ok" and lines[-1] == build.source_hash()
    monkeypatch.delenv("MODA_MLP_AGPR", raising=False)
    monkeypatch.delenv("MODA_LIB_PATH", raising=False)
    _lib._check_agpr_audit()
    assert "MODA_MLP_AGPR" not in os.environ                                            # a clean stamp changes nothing
    bad = tmp_path / "agpr_audit.txt"
    bad.write_text("failed: synthetic\n" + build.source_hash())
    monkeypatch.setattr(build, "AUDIT_STAMP", str(bad))
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _lib._check_agpr_audit()
    assert os.environ.get("MODA_MLP_AGPR") == "0" and w                                 # the eight-wave form, loudly
    monkeypatch.delenv("MODA_MLP_AGPR", raising=False)
