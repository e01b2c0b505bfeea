"""Build libmoda_hip.so (gfx950) in-tree with hipcc.  `python -m moda_amd.build` or __graft_entry__.build()."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
BUILT_LIB = os.path.join(LIB_DIR, "libmoda_hip.so")        # what build() writes and needs_build() looks at -- always
# MODA_LIB_PATH: LOAD another build of the library (A/B timing of kernel variants inside one process tree / one gpurun call,
# tools/ab_build.sh + tools/ab_run.py).  It redirects only what _lib.load() opens: build() never writes a variant file.
LIB_PATH = os.environ.get("MODA_LIB_PATH") or BUILT_LIB
# per-file flags.  mlp_fused.hip: hipcc otherwise pairs neighbouring scalar f32 adds / muls into v_pk_*_f32, which needs its
# operands in aligned register pairs (a v_mov per operand) and issues slower beside MFMAs -- the fused skin-MLP + warp kernel
# 1.146 -> 1.088 ms, the 8 x 256 kernel unchanged (A/B on one box).  Not applied to the other files: it reorders fp32 sums in the
# exact-fp32 training kernels (the 1e-3 gradient fixture moved to 1.3e-3 on its noisiest scalar).
# -amdgpu-sched-strategy=max-ilp: the 8 x 256 kernel 12.36 -> 12.11 ms (its default schedule leaves 140 bytes of scratch per
# lane, this one 20; max-memory-clause 12.13), the other kernels of the file unchanged (A/B on one box, twice).
FILE_FLAGS = {"mlp_fused.hip": ("-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp")}
SOURCES = ("mlp_fused.hip", "render_kernels.hip", "train_kernels.hip", "gemm_bf16.hip", "gemm_x3.hip", "bwd64_chain.hip", "bwd256_fused.hip", "loss_kernels.hip", "prep_kernels.hip")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _flag_key():
    """Everything besides the sources that decides what an object contains: extra flags, per-file flags and their switch."""
    return (os.environ.get("MODA_HIPCC_FLAGS", "") + " | " + repr(sorted(FILE_FLAGS.items()))
            + " | no_file_flags=" + ("1" if os.environ.get("MODA_NO_FILE_FLAGS") else "0"))


def _stamp_matches():
    stamp = os.path.join(LIB_DIR, "flags.txt")
    return os.path.exists(stamp) and open(stamp).read() == _flag_key()


def source_hash():
    """sha256[:16] over everything that decides what the kernels are: the HIP sources, the two headers, the per-file flags.
    profiles/traffic.json is stamped with it by tools/pmc_summary.py, and bench.py reports `roofline.traffic` (HBM bytes from the
    PMC counters) only when the stamp equals the hash of the sources it runs from -- a profile of another build is labelled stale
    instead of passing as a measurement of this one."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, x) for x in SOURCES + ("moda_dev.h",)] + [os.path.join(ROOT, "include", "moda_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(BUILT_LIB) or not _stamp_matches():
        return True
    t = os.path.getmtime(BUILT_LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ("moda_dev.h",)] + [os.path.join(ROOT, "include", "moda_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=None):
    """hipcc --offload-arch=gfx950 -shared -fPIC: one code object per source (compiled in parallel; an object is reused
    when it is newer than its source, the header and the flags it was built with), linked into one .so."""
    if not force and not needs_build():
        return BUILT_LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    hdr = os.path.join(ROOT, "include", "moda_hip.h")
    dev_hdr = os.path.join(CSRC, "moda_dev.h")
    extra = os.environ.get("MODA_HIPCC_FLAGS", "").split()
    stamp = os.path.join(LIB_DIR, "flags.txt")
    same_flags = _stamp_matches()
    if not same_flags and os.path.exists(stamp):
        os.remove(stamp)               # objects of two flag sets must never be linked together after an interrupted build

    def compile_one(s):
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIB_DIR, s.replace(".hip", ".o"))
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(dev_hdr)):
            return obj
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-c", src, "-o", obj] + (list(FILE_FLAGS.get(s, ())) if not os.environ.get("MODA_NO_FILE_FLAGS") else []) + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", BUILT_LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    audit_built_library(verbose=verbose)
    open(stamp, "w").write(_flag_key())          # only a finished link vouches for the flags of what lies in lib/
    return BUILT_LIB


AUDIT_STAMP = os.path.join(LIB_DIR, "agpr_audit.txt")


def audit_built_library(verbose=True):
    """The asm-owned-AGPR kernels (the default 8 x 256 inference kernels) are correct only if hipcc kept out of a[0:255] and left
    the hand-placed wait states alone -- which no compiler diagnostic shows, and another hipcc version may change.  Every build
    therefore audits its own machine code (moda_amd/isa_audit.py) and stamps the result beside the library; `_lib.load()` reads
    the stamp, and anything but 'ok' makes the dispatch take the compiler-scheduled eight-wave form (MODA_MLP_AGPR=0), loudly."""
    from . import isa_audit
    try:
        res = isa_audit.audit_library(BUILT_LIB)
        bad = {k: v for k, v in res.items() if v}
        if len(res) < 4:
            text = f"failed: expected the 4 AGPR-form kernels in the library, found {len(res)}"
        elif bad:
            text = "failed:\n" + "\n".join(f"{k}: {f}" for k, v in bad.items() for f in v[:4])
        else:
            text = "ok"
    except Exception as e:        # no llvm-objdump / objcopy: unaudited is not audited
        text = f"failed: the audit could not run ({type(e).__name__}: {e})"
    open(AUDIT_STAMP, "w").write(text + "\n" + source_hash())
    if verbose or text != "ok":
        print(f"[moda_amd.build] ISA audit of the AGPR-form kernels: {text.splitlines()[0]}", flush=True)
    return text == "ok"


if __name__ == "__main__":
    build(force="--force" in sys.argv)
