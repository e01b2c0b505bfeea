"""ctypes binding of libmoda_hip.so (include/moda_hip.h).  The HIP library is the ONLY compute path of
this package: if it is missing, or a tensor is not on a GPU, calls fail loudly -- there is no CPU fallback."""
import ctypes
import os

import torch

from .build import LIB_PATH

_c = ctypes
_I64, _I32, _F32, _P = _c.c_int64, _c.c_int32, _c.c_float, _c.c_void_p


class MlpDesc(_c.Structure):
    _fields_ = [("W", _I32), ("D", _I32), ("n_out", _I32), ("flags", _I32), ("n_freq", _I32), ("reserved", _I32),
                ("window", _F32 * 16), ("overflow", _P)]


class GemmDesc(_c.Structure):
    _fields_ = [("A", _P), ("sam", _I64), ("sak", _I64), ("A2", _P), ("sam2", _I64), ("K1", _I64),
                ("B", _P), ("sbk", _I64), ("sbn", _I64), ("C", _P), ("ldc", _I64), ("M", _I64), ("N", _I64), ("K", _I64),
                ("bias", _P), ("rowbias", _P), ("ld_rowbias", _I64), ("rows_per_bias", _I64), ("mask_src", _P),
                ("ld_mask", _I64), ("act", _I32), ("accumulate", _I32), ("split_k", _I32), ("reserved", _I32), ("a_sum", _P),
                ("mask_bits", _P), ("ld_bits", _I64)]


class LossTerm(_c.Structure):       # moda_hip.h moda_loss_term
    _fields_ = [("x", _P), ("mask", _P), ("dx", _P), ("n", _I64), ("k", _I32), ("mask_kind", _I32), ("weight", _F32),
                ("reserved", _I32)]


class NerfTrainDesc(_c.Structure):
    _fields_ = [("D", _I32), ("W", _I32), ("P", _I32), ("C1", _I32), ("Cd", _I32), ("n_out", _I32), ("raw_feat", _I32),
                ("sigma_only", _I32), ("n_freq", _I32), ("reserved", _I32), ("window", _F32 * 16), ("M", _I64), ("R1", _I64),
                ("Rd", _I64)]


_SIGNATURES = {
    "moda_abi_version": (_c.c_int, []),
    "moda_stream_capture_id": (_c.c_uint64, [_P]),
    "moda_mlp_stream_bytes": (_I64, [_c.POINTER(MlpDesc)]),
    "moda_mlp_bias_floats": (_I64, [_c.POINTER(MlpDesc)]),
    "moda_mlp_fwd": (_c.c_int, [_c.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _P]),
    "moda_mlp_pack": (_c.c_int, [_c.POINTER(_P), _I32, _P, _I64, _I32, _P, _c.POINTER(_P), _I32, _P, _I64, _P, _P, _P]),
    "moda_fold_rows": (_c.c_int, [_I32, _c.POINTER(_P), _c.POINTER(_I64), _c.POINTER(_I64), _c.POINTER(_I64), _c.POINTER(_P),
                                  _c.POINTER(_I64), _c.POINTER(_I64), _c.POINTER(_I64), _c.POINTER(_P), _c.POINTER(_P),
                                  _c.POINTER(_I64), _c.POINTER(_P), _P]),
    "moda_linear_fwd": (_c.c_int, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I32, _P, _I64, _P]),
    "moda_embed_fwd": (_c.c_int, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _P]),
    "moda_bone_transform_fwd": (_c.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "moda_warp_workspace_floats": (_I64, [_I64, _I32, _I32]),
    "moda_skinning_fwd": (_c.c_int, [_P, _I32, _P, _P, _P, _I64, _I64, _I32, _P, _P, _P]),
    "moda_dqs_fwd": (_c.c_int, [_P, _I32, _P, _P, _I64, _I64, _I32, _P, _P]),
    "moda_warp_fwd": (_c.c_int, [_P, _I32, _P, _I32, _P, _P, _I32, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "moda_warp_frames_fwd": (_c.c_int, [_P, _I32, _P, _I64, _I32, _P, _P, _P, _I32, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "moda_warp_tiles": (_I32, [_I32]),
    "moda_warp_tables_fwd": (_c.c_int, [_P, _I64, _P, _I64, _I32, _P, _I32, _P, _P, _P, _P]),
    "moda_row_runs": (_c.c_int, [_P, _I64, _P, _I64, _I64, _P, _P, _P]),
    "moda_row_runs_multi": (_c.c_int, [_I32, _c.POINTER(_P), _c.POINTER(_I64), _I64, _P, _P, _P]),
    "moda_mlp_warp_fwd": (_c.c_int, [_c.POINTER(MlpDesc), _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P, _P,
                                     _I64, _I64, _P, _P]),
    "moda_sample_rays_fwd": (_c.c_int, [_P, _P, _P, _P, _P, _F32, _I32, _I64, _I64, _P, _P, _P]),
    "moda_points_fwd": (_c.c_int, [_P, _P, _P, _I64, _I64, _P, _P]),
    "moda_composite_fwd": (_c.c_int, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _I64, _I64,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _P, _P]),
    "moda_mlp_composite_fwd": (_c.c_int, [_c.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _P, _P,
                                          _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "moda_mlp_live_fwd": (_c.c_int, [_c.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P,
                                     _I64, _P]),
    "moda_sample_pdf_fwd": (_c.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "moda_merge_sort_fwd": (_c.c_int, [_P, _I32, _P, _I32, _I64, _P, _P]),
    "moda_merge_index_fwd": (_c.c_int, [_P, _I32, _P, _I32, _I64, _P, _P, _P]),
    "moda_merge_rows_fwd": (_c.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _P, _P]),
    "moda_vec_to_sim3_fwd": (_c.c_int, [_P, _I64, _P, _P, _P, _P]),
    "moda_dq_op": (_c.c_int, [_I32, _P, _P, _I64, _P, _P, _P]),
    "moda_gemm_f32": (_c.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P, _I32, _P, _I32, _I32, _P]),
    "moda_gemm_f32_ex": (_c.c_int, [_c.POINTER(GemmDesc), _P]),
    "moda_nerf_train_ws_floats": (_I64, [_c.POINTER(NerfTrainDesc)]),
    "moda_nerf_train_scratch_floats": (_I64, [_c.POINTER(NerfTrainDesc)]),
    "moda_nerf_train_fwd": (_c.c_int, [_c.POINTER(NerfTrainDesc), _P, _P, _P, _c.POINTER(_P), _P, _P, _P]),
    "moda_nerf_train_fwd_fused": (_c.c_int, [_c.POINTER(NerfTrainDesc), _P, _P, _P, _c.POINTER(_P), _P, _P, _P, _P, _P, _P]),
    "moda_mlp_dump_fwd": (_c.c_int, [_c.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _P, _I64,
                                     _P]),
    "moda_nerf_train_bwd": (_c.c_int, [_c.POINTER(NerfTrainDesc), _P, _P, _P, _c.POINTER(_P), _P, _P, _P, _P, _c.POINTER(_P),
                                       _P, _P, _P, _P]),
    "moda_segsum_f32": (_c.c_int, [_P, _I64, _I64, _I64, _I64, _P, _I64, _P]),
    "moda_colsum_f32": (_c.c_int, [_P, _I64, _I64, _I64, _P, _P]),
    "moda_embed_bwd": (_c.c_int, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _P, _P]),
    "moda_embed_jvp": (_c.c_int, [_P, _I64, _I32, _I32, _P, _P, _P, _I64, _P]),
    "moda_act_bwd": (_c.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "moda_project_fwd": (_c.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "moda_project_bwd": (_c.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P]),
    "moda_flow_render": (_c.c_int, [_P, _P, _P, _F32, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "moda_pts_exp": (_c.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "moda_composite_bwd": (_c.c_int, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _I64, _I64,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "moda_points_bwd": (_c.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "moda_warp_prepped_fwd": (_c.c_int, [_P, _I32, _P, _P, _P, _P, _I32, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P]),
    "moda_warp_prepped_bwd": (_c.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32,
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "moda_bone_prep": (_c.c_int, [_P, _I64, _P, _P, _P, _P]),
    "moda_bone_transform_bwd": (_c.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P]),
    "moda_dq_inverse_bwd": (_c.c_int, [_P, _P, _I64, _P, _P]),
    "moda_raycast": (_c.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "moda_rt_to_dq": (_c.c_int, [_P, _I64, _P, _P, _P, _P]),
    "moda_normalize_rows": (_c.c_int, [_P, _I64, _I32, _P, _P, _P, _P]),
    "moda_match_matrix": (_c.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _I32, _P]),
    "moda_match_sweep": (_c.c_int, [_P, _I64, _I64, _P, _I32, _F32, _P, _P, _I32, _P]),
    "moda_match_sinkhorn": (_c.c_int, [_P, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "moda_match_matrix_rows": (_c.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _P]),
    "moda_match_expect": (_c.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _I32, _P]),
    "moda_match_prob": (_c.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P]),
    "moda_match_ecols": (_c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F32, _P, _I32, _P]),
    "moda_match_dbar": (_c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _I32, _P, _P, _I64, _I64, _P, _P, _P, _I32, _P]),
    "moda_ray_loss": (_c.c_int, [_P] * 9 + [_I64, _I32] + [_P] * 11 + [_P]),
    "moda_masked_mean": (_c.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P]),
    "moda_loss_terms": (_c.c_int, [_P, _I32, _P, _P, _P]),
    "moda_row_dist": (_c.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "moda_dbg_poison_lds": (_c.c_int, [_c.c_uint32, _P]),
    "moda_fold_final": (_c.c_int, [_P, _I64, _P, _P, _P, _I64, _P, _P, _P]),
    "moda_s3im": (_c.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, _P, _P, _P, _P]),
    "moda_logsig_loss": (_c.c_int, [_P, _P, _I64, _F32, _F32, _P, _P, _P, _P]),
    "moda_sum_tensors": (_c.c_int, [_c.POINTER(_P), _I32, _I64, _P, _P]),
    "moda_affine3": (_c.c_int, [_P, _P, _P, _F32, _P, _I64, _P, _P]),
}

EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 9        # moda_abi_version() of the library these signatures describe (include/moda_hip.h)
_lib = None


def load():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m moda_amd.build` (hipcc, gfx950). "
                "moda_amd has no CPU or PyTorch fallback.")
        lib = _c.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.moda_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {lib.moda_abi_version()}, this package binds version "
                               f"{ABI_VERSION}: rebuild it with `python -m moda_amd.build`")
        _check_agpr_audit()
        _lib = lib
    return _lib


def _check_agpr_audit():
    """The default 8 x 256 inference kernels own their AGPRs and wait states in inline asm (mlp_fused.hip); the build audits their
    machine code and stamps the verdict (build.audit_built_library).  No stamp, a stamp of other sources, or a failed audit: the
    dispatch takes the compiler-scheduled eight-wave form instead (the library reads MODA_MLP_AGPR per call)."""
    import warnings
    from . import build
    if os.environ.get("MODA_LIB_PATH") or "MODA_MLP_AGPR" in os.environ:
        return                                   # an A/B build or an explicit choice: the caller's business
    try:
        lines = open(build.AUDIT_STAMP).read().splitlines()
    except OSError:
        lines = ["missing"]
    if lines[0] == "ok" and lines[-1] == build.source_hash():
        return
    os.environ["MODA_MLP_AGPR"] = "0"
    warnings.warn("moda_amd: the ISA audit of the AGPR-form 8x256 kernels is " + (lines[0] if lines[0] != "ok" else "of other sources")
                  + " -- using the compiler-scheduled eight-wave kernels (slower, always safe); rebuild with `python -m moda_amd.build`")


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def stream():
    return _c.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t, dtype=torch.float32):
    """A contiguous device tensor of `dtype`, or an error: the kernels only read GPU memory."""
    if not torch.is_tensor(t):
        raise TypeError(f"expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError("moda_amd ops take CUDA (ROCm) tensors; the HIP library is the only compute path")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def ptr(t):
    return None if t is None else _c.c_void_p(t.data_ptr())


def no_grad_only(*tensors):
    """For entry points that have no backward kernel (the fused inference kernels, the algebra helpers):
    refuse loudly rather than return tensors without a graph.  The differentiable route is moda_amd/autograd.py."""
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "this moda_amd entry point is inference-only: call it under torch.no_grad(), or use render_rays / "
            "NeRF.forward / Embedding.forward, which switch to the autograd route")


# ---- optional per-launch timing with events on the launch stream (bench.py's roofline leg) ----------
PROFILE = None   # None, or dict: kernel tag -> list of (start_event, end_event, units)


def profile_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    return ev


def profile_end(start, tag, units):
    if start is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record(torch.cuda.current_stream())
    PROFILE.setdefault(tag, []).append((start, end, units))


# ---- small constant tensors (bounds, lattices): built once per (device, values) so that steady-state calls issue no
#      host-to-device copy (which would also be illegal while a HIP graph is being captured) ----------------------------
_CONST = {}


def const_tensor(key, device, make):
    """Cached device tensor for an immutable host-side constant; `make()` returns a CPU tensor / array."""
    k = (key, str(device))
    t = _CONST.get(k)
    if t is None:
        v = make()
        t = (v if torch.is_tensor(v) else torch.as_tensor(v)).to(device=device, dtype=torch.float32).contiguous()
        _CONST[k] = t
    return t
