"""`Embedding` and `NeRF` with the reference's constructor arguments, attributes and state-dict names
(reference nnutils/nerf.py:13-75, 83-198), evaluated by the HIP library.

Two evaluation routes:
  * `NeRF.forward(x, xyz=None, sigma_only=False)` -- the reference signature on an already embedded
    input; one `moda_linear_fwd` launch per layer (compatibility route, exact fp32).
  * `NeRF.fused(xyz, code=..., dir_src=..., ...)` -- what the rendering path uses: positional encoding,
    every layer and the heads in ONE kernel (`moda_mlp_fwd`), fp32-exact or bf16 MFMA.
"""
import math

import torch
from torch import nn

from . import _lib as L
from . import mlp_pack as mp
from . import overflow

PRECISIONS = ("fp32", "bf16", "bf16x3", "fp16")


def _initial_precision():
    """The mode a process starts in: MODA_PRECISION if set, else 'fp16' -- the parity-grade throughput mode (within the 1e-4 bar of
    the reference on every inference fixture, G7 / G8 / G24 / G25, at ~11x the exact-fp32 mode's speed at config 2; overflow is
    reported, never saturated).  A caller that only swaps its imports renders in this mode; MODA_PRECISION=fp32 (or
    set_precision('fp32')) restores the exact-fp32 kernels."""
    import os
    mode = os.environ.get("MODA_PRECISION", "fp16")
    if mode not in PRECISIONS:
        raise ValueError(f"MODA_PRECISION={mode!r}: expected one of {PRECISIONS}")
    return mode


_PRECISION = _initial_precision()
_PREC_FLAGS = {"fp32": 0, "bf16": mp.MLP_BF16, "bf16x3": mp.MLP_BF16X3, "fp16": mp.MLP_F16}
_TAG = {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3", "fp16": "f16"}      # kernel names of the event profile (bench.py)
FP16_SPLIT_HEADS = __import__("os").environ.get("MODA_FP16_HEADS", "1") != "0"      # (0: single-fp16 heads, A/B)
_WARP_PRECISIONS = ("bf16", "fp16", "bf16x3")          # precisions the fused skin + warp kernel is instantiated for


def set_precision(mode):
    """'fp32' (exact fp32 MFMA, parity mode), 'bf16' (bf16 MFMA operands, fp32 accumulate: throughput mode) or 'bf16x3'
    (split-bf16: every operand as bf16 hi + lo, three MFMAs per product, exact sincosf encoding -- the parity-grade
    throughput mode: within 1e-4 of the reference at a third of the bf16 rate instead of a sixteenth) or 'fp16' (fp16 MFMA
    operands, fp32 accumulate: the bf16 mode's speed with 11 significand bits instead of 8; weights / activations outside
    fp16's range are REPORTED, `moda_amd.overflow`)."""
    global _PRECISION
    if mode not in PRECISIONS:
        raise ValueError(mode)
    _PRECISION = mode


def get_precision():
    return _PRECISION


def default_precision():
    """The precision a `NeRF.fused(...)` call without an explicit `precision=` runs in under the current mode.

    'fp16' is the mode of `render_rays`' per-sample hot loop, not of every entry point: fp16 operands (11 significand bits) are
    used exactly where the rendered outputs were measured within the 1e-4 bar of the reference -- the 8 x 256 colour / density
    network of a compositing pass (its outputs go through sigmoid, the density and the weighted sums: img 4e-5, depth 9e-6 at
    config 2) and the fused skin + warp kernels (the 64-wide network's logits only steer a softmax over bones: warped positions
    2e-5) -- and those call sites say so (`precision=hot_precision()`).  Everything else a caller can reach in this mode --
    `evaluate_mlp`, the feature / visibility / displacement networks, mesh queries, the hierarchical pre-pass -- returns RAW
    network outputs or feeds an ill-conditioned step (the inverse CDF of `sample_pdf`), where fp16 operands are 3-6e-4 off:
    those run split-bf16 ('bf16x3', ~1e-6)."""
    return "bf16x3" if _PRECISION == "fp16" else _PRECISION


def hot_precision():
    """What the hot-loop call sites of `render_rays` pass as `precision=`: the mode itself."""
    return _PRECISION


class precision_scope:
    """`with precision_scope(mode):` -- evaluate the calls inside in `mode` (None: no change)."""

    def __init__(self, mode):
        if mode is not None and mode not in PRECISIONS:
            raise ValueError(mode)
        self.mode = mode

    def __enter__(self):
        global _PRECISION
        self.saved = _PRECISION
        if self.mode is not None:
            _PRECISION = self.mode

    def __exit__(self, *exc):
        global _PRECISION
        _PRECISION = self.saved
        return False


def embedding_window(n_freqs, alpha):
    """w_k = 0.5 (1 + cos(pi clamp(alpha - k, 0, 1) + pi))  (nerf.py:63-68), as python floats."""
    alpha = float(alpha)
    out = []
    for k in range(n_freqs):
        w = min(max(alpha - k, 0.0), 1.0)
        out.append(0.5 * (1 + math.cos(math.pi * w + math.pi)))
    return out


class Embedding(nn.Module):
    """x -> (x, sin(2^k x), cos(2^k x), ...) with the frequency window (nerf.py:13-75)."""

    def __init__(self, in_channels, N_freqs, logscale=True, alpha=None):
        super().__init__()
        if not logscale:
            raise NotImplementedError("only logscale frequency bands are used by MoDA (moda.py:274-276)")
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.nfuncs = 2
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.alpha = self.N_freqs if alpha is None else alpha
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)

    def window(self):
        return embedding_window(self.N_freqs, self.alpha)

    def forward(self, x, normalize=False):
        if self.N_freqs <= 0:
            return x
        if torch.is_grad_enabled() and x.requires_grad:
            from .autograd import EmbedFn
            return EmbedFn.apply(x, self.N_freqs, self.window(), bool(normalize))
        shape = x.shape
        xf = L.dev(x).reshape(-1, shape[-1])
        out = torch.empty((xf.shape[0], self.out_channels), device=xf.device, dtype=torch.float32)
        win = (L._F32 * 16)(*(self.window() + [0.0] * (16 - self.N_freqs)))
        L.call("moda_embed_fwd", L.ptr(xf), xf.shape[0], shape[-1], self.N_freqs, win, int(normalize), L.ptr(out),
               out.stride(0), L.stream())
        return out.view(shape[:-1] + (self.out_channels,))


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, out_channels=3, skips=[4],
                 raw_feat=False, init_beta=1. / 100, activation=nn.ReLU(True), in_channels_code=0,
                 enable_semantic=False):
        super().__init__()
        if not isinstance(activation, nn.ReLU):
            raise NotImplementedError("the HIP kernels implement ReLU hidden activations (all MoDA nets use it)")
        if enable_semantic:
            raise NotImplementedError("enable_semantic is never set by MoDA (moda.py:271-273)")
        self.D, self.W = D, W
        self.in_channels_xyz = in_channels_xyz
        self.in_channels_dir = in_channels_dir
        self.in_channels_code = in_channels_code
        self.skips = list(skips)
        self.use_xyz = False
        self.enable_semantic = enable_semantic
        self.out_channels = out_channels
        self.weights_reg = []
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
                self.weights_reg.append(f"xyz_encoding_{i+1}")
            elif i in self.skips:
                layer = nn.Linear(W + in_channels_xyz, W)
                self.weights_reg.append(f"xyz_encoding_{i+1}")
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(layer, activation))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), activation)
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, out_channels))
        self.raw_feat = raw_feat
        self.beta = nn.Parameter(torch.Tensor([init_beta]))

    # ------------------------------------------------------------------ compatibility route
    @staticmethod
    def _linear(x, lin, act, col0=0, k=None, out=None, bias=None):
        w = L.dev(lin.weight)
        b = L.dev(lin.bias if bias is None else bias)
        k = x.shape[1] if k is None else k
        y = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32) if out is None else out
        L.call("moda_linear_fwd", L.ptr(x), x.shape[0], k, x.stride(0), L.ptr(w), w.shape[0], w.shape[1], col0,
               L.ptr(b), act, L.ptr(y), y.stride(0), L.stream())
        return y

    @staticmethod
    def _fold_many(jobs, run_start=None):
        """[(x (R, k), lin, col0, k, bias|None), ...] (at most four) -> [(R, O) = bias + x @ lin.weight[:, col0:col0+k]^T, ...]:
        the per-row code folds of a fused call in ONE launch (`moda_fold_rows`, exact fp32).  run_start (R,) int32: rows that
        repeat their predecessor are neither computed nor written (their consumer reads row run_start[r])."""
        # (one route whatever the row count: a row's result must not depend on how many rows the call has -- a batch rendered in
        #  chunks is bit-identical to the batch rendered whole)
        n = len(jobs)
        xs, ws, bs, ys = [], [], [], []
        for x, lin, col0, k, bias in jobs:
            w = L.dev(lin.weight).detach()
            b = L.dev(lin.bias if bias is None else bias).detach()
            x = L.dev(x)
            if x.stride(1) != 1:
                x = x.contiguous()
            xs.append(x); ws.append(w); bs.append(b)
            ys.append(torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32))
        P, I = L._P * n, L._I64 * n
        L.call("moda_fold_rows", n, P(*[t.data_ptr() for t in xs]), I(*[t.shape[0] for t in xs]), I(*[j[3] for j in jobs]),
               I(*[t.stride(0) for t in xs]), P(*[t.data_ptr() for t in ws]), I(*[t.shape[0] for t in ws]),
               I(*[t.stride(0) for t in ws]), I(*[j[2] for j in jobs]), P(*[t.data_ptr() for t in bs]),
               P(*[t.data_ptr() for t in ys]), I(*[t.stride(0) for t in ys]),
               None if run_start is None else P(*[run_start.data_ptr() if t.shape[0] == run_start.shape[0] else None for t in xs]),
               L.stream())
        return ys

    @staticmethod
    def _fold(x, lin, col0, k, bias=None):
        """(R, O) = bias + x @ lin.weight[:, col0:col0+k]^T -- one per-row code fold (see _fold_many)."""
        return NeRF._fold_many([(x, lin, col0, k, bias)])[0]

    def _needs_grad(self, *tensors):
        return torch.is_grad_enabled() and (any(torch.is_tensor(t) and t.requires_grad for t in tensors)
                                            or any(p.requires_grad for p in self.parameters()))

    def forward(self, x, xyz=None, sigma_only=False):
        """nerf.py:147-198 on an embedded input x (..., in_channels_xyz [+ in_channels_dir]).
        Under autograd every layer is a LinearFn (MFMA GEMM forward, GEMM backward) and activations are kept."""
        shape = x.shape
        if self._needs_grad(x):
            out = self._forward_rows(L.dev(x).reshape(-1, shape[-1]), sigma_only, train=True)
            return out.view(shape[:-1] + (out.shape[-1],))
        x2 = L.dev(x).reshape(-1, shape[-1])
        outs = []
        rows = 1 << 20   # bound the (rows, W) activation buffers
        for r0 in range(0, x2.shape[0], rows):
            outs.append(self._forward_rows(x2[r0:r0 + rows], sigma_only))
        out = torch.cat(outs, 0) if len(outs) != 1 else outs[0]
        return out.view(shape[:-1] + (out.shape[-1],))

    def _forward_rows(self, x, sigma_only, train=False):
        if train:
            from .autograd import LinearFn
            lin_ = lambda t, m, act: LinearFn.apply(t, m.weight, m.bias, act)
        else:
            lin_ = self._linear
        cx = self.in_channels_xyz
        input_xyz = x[:, :cx]
        h = input_xyz
        for i in range(self.D):
            lin = getattr(self, f"xyz_encoding_{i+1}")[0]
            if i in self.skips:
                h = torch.cat([input_xyz, h], -1)   # data movement only (nerf.py:175)
            h = lin_(h, lin, 1)
        sigma = lin_(h, self.sigma, 0)
        if sigma_only:
            return sigma
        final = lin_(h, self.xyz_encoding_final, 0)
        d_in = torch.cat([final, x[:, cx:cx + self.in_channels_dir]], -1)
        d = lin_(d_in, self.dir_encoding[0], 1)
        rgb = lin_(d, self.rgb[0], 0 if self.raw_feat else 2)
        return rgb if self.raw_feat else torch.cat([rgb, sigma], -1)

    # ------------------------------------------------------------------ training route
    def train_forward(self, xyz, embedding_xyz, code=None, dir_src=None, sigma_only=False):
        """NeRF([PE(xyz), code], dir_src) under autograd as one node (autograd.NerfFn): xyz (..., 3) sample positions,
        code (R, in_channels_xyz - 63) / dir_src (R', in_channels_dir) per-ray rows with R, R' in {1, rays}
        (samples of a ray are consecutive).  Same maths as Embedding.forward + evaluate_mlp's concatenation +
        NeRF.forward (nerf.py:35-75, geom_utils.py:33-50, nerf.py:147-198) without expanding per-ray inputs."""
        from .autograd import NerfFn, NerfSpec
        if self.skips != [4] or self.D < 5:
            raise NotImplementedError("train_forward implements skips=[4] (the only value MoDA uses)")
        P = embedding_xyz.out_channels
        C1 = self.in_channels_xyz - P
        Cd = self.in_channels_dir
        if sigma_only:
            dir_src = None                  # the dir branch is not evaluated (nerf.py:179-180)
        if (C1 > 0) != (code is not None) or ((Cd > 0 and not sigma_only) != (dir_src is not None)):
            raise ValueError("code / dir_src must be given exactly when the network has those input channels")
        lead = xyz.shape[:-1]
        M = 1
        for d in lead:
            M *= d
        for t, c, name in ((code, C1, "code"), (dir_src, Cd, "dir_src")):
            if t is not None and (t.shape[-1] != c or M % t.reshape(-1, c).shape[0] != 0):
                raise ValueError(f"{name}: expected (R, {c}) rows with R dividing {M} samples, got {tuple(t.shape)}")
        spec = NerfSpec(self.D, self.W, P, C1, 0 if sigma_only else Cd, self.out_channels, self.raw_feat,
                        embedding_xyz.N_freqs, embedding_xyz.window(), sigma_only=sigma_only)
        params = []
        for i in range(self.D):
            lin = getattr(self, f"xyz_encoding_{i+1}")[0]
            params += [lin.weight, lin.bias]
        params += [self.sigma.weight, self.sigma.bias, self.xyz_encoding_final.weight, self.xyz_encoding_final.bias,
                   self.dir_encoding[0].weight, self.dir_encoding[0].bias, self.rgb[0].weight, self.rgb[0].bias]
        spec.param_objs = params           # the Parameter objects: NerfFn's backward writes into their GradBucket views, if bound
        from .autograd import get_train_precision
        if (get_train_precision() == "bf16" and not sigma_only and self.W in (64, 128, 256) and embedding_xyz.N_freqs <= 10
                and 5 <= self.D <= 8 and 1 <= self.out_channels <= 64):
            # the packed bf16 weight stream of the fused kernel (a gather + one small GEMM per step; part of a captured graph)
            flags = mp.MLP_BF16 | (0 if self.raw_feat else (mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA))
            with torch.no_grad():
                spec.pack = self._packed(self._spec(embedding_xyz.N_freqs, flags), xyz.device)
        out = NerfFn.apply(spec, xyz.reshape(-1, 3), code, None if sigma_only else dir_src, *params)
        return out.view(lead + (out.shape[-1],))

    # ------------------------------------------------------------------ fused route
    def _spec(self, n_freq, flags):
        if self.skips != [4]:
            raise NotImplementedError("fused MLP kernel implements skips=[4] (the only value MoDA uses)")
        return mp.MlpSpec(W=self.W, D=self.D, n_out=self.out_channels, in_xyz=self.in_channels_xyz,
                          in_dir=self.in_channels_dir, n_freq=n_freq, flags=flags)

    def invalidate_packed(self):
        """Kept for callers of earlier versions: there is nothing to invalidate any more.  The MFMA weight stream is packed
        from the parameter tensors at EVERY fused call (one `moda_mlp_pack` launch + the two small products of the
        `xyz_encoding_final` fold), because no host-side key can see every parameter update: torch's fused AdamW and
        optimiser steps replayed from a captured HIP graph move no version counter (a cache keyed on `_version` rendered and
        trained on stale weights after such steps -- round 2's bf16 training-mode divergence)."""

    def _packed(self, spec, device):
        """Weight stream + bias block for `spec`, packed now from the current parameter values (see invalidate_packed)."""
        sd = dict(self.named_parameters())
        wn, bn = mp.weight_names(spec), mp.bias_names(spec)
        idx = mp.stream_index(spec)
        if not hasattr(idx, "_gpu") or idx._gpu[0] != str(device):
            wcode, bcode = idx.codes()
            idx._gpu = (str(device), torch.from_numpy(wcode).to(device), torch.from_numpy(bcode).to(device))
        # xyz_encoding_final is a Linear without activation in front of dir_encoding's Linear (nerf.py:184-187): the two
        # are one (W/2 x W) layer, Wd[:, :W] Wf with bias bd + Wd[:, :W] bf (mlp_pack.fold_final) -- a W x W layer per
        # sample that never has to be executed.  The product is an exact-fp32 MFMA GEMM of the library.
        src = {n: L.dev(sd[n]).detach() for n in wn}
        bd_folded = None
        if not spec.sigma_only:
            W = self.W
            wd, wf = src["dir_encoding.0.weight"], src["xyz_encoding_final.weight"]
            prod = torch.empty((W // 2, W), device=device, dtype=torch.float32)      # the stream reads only these W columns
            bd_folded = torch.empty((W // 2,), device=device, dtype=torch.float32)
            L.call("moda_fold_final", L.ptr(wd), wd.stride(0), L.ptr(wf), L.ptr(L.dev(self.xyz_encoding_final.bias).detach()),
                   L.ptr(L.dev(self.dir_encoding[0].bias).detach()), W, L.ptr(prod), L.ptr(bd_folded), L.stream())
            src["dir_encoding.0.weight"] = prod
        bsrc = [L.dev(sd[n]).detach() for n in bn]
        n_w = idx._gpu[1].numel()
        stream = torch.empty((n_w,), device=device,
                             dtype=torch.float16 if spec.f16 else (torch.bfloat16 if spec.bf16 else torch.float32))
        bias = torch.empty((idx._gpu[2].numel(),), device=device, dtype=torch.float32)
        wp = (L._P * len(wn))(*[src[n].data_ptr() for n in wn])
        bp = (L._P * len(bn))(*[t.data_ptr() for t in bsrc])
        L.call("moda_mlp_pack", wp, len(wn), L.ptr(idx._gpu[1]), n_w, 3 if spec.f16 else (2 if spec.x3 else int(spec.bf16)),
               L.ptr(stream), bp, len(bn), L.ptr(idx._gpu[2]), bias.numel(), L.ptr(bias), overflow.ptr() if spec.f16 else None,
               L.stream())
        assert stream.numel() * stream.element_size() == idx.stream_bytes
        return stream, bias, bd_folded

    def fused(self, xyz, n_freq=10, alpha=None, code=None, dir_src=None, flip=None, sigma_only=False,
              with_sigma=None, precision=None, sigmoid=None, out_tr_S=0, n_live=None):
        """out (..., n_out [+1]) = NeRF([PE(xyz), code], [dir_src]) in one kernel.

        xyz (..., 3); code (R, in_channels_xyz - 63) with R in {1, N rays, M samples} rows; dir_src
        (R', in_channels_dir) likewise; flip (...,) uint8/bool negates x before encoding (symm_shape).
        Rows are assigned to samples in order: sample m uses row m // (M / R).
        out_tr_S = S > 0 returns the output as (M/S, n_out, S) (channel-major per ray) instead of (..., n_out).
        n_live (rays,) int32 with xyz (rays, S, 3), S % 32 == 0: early ray termination (opt-in, not reference behaviour) --
        32-sample groups that start at or beyond n_live[ray] are not evaluated and their output rows are left as they are."""
        L.no_grad_only(xyz, code, dir_src, *self.parameters())
        precision = precision or default_precision()
        lead = xyz.shape[:-1]
        x = L.dev(xyz).reshape(-1, 3)
        M = x.shape[0]
        if with_sigma is None:
            with_sigma = not self.raw_feat
        if precision not in PRECISIONS:
            raise ValueError(precision)
        flags = _PREC_FLAGS[precision]
        if precision == "fp16":
            overflow.poll()
        if sigma_only:
            flags |= mp.MLP_SIGMA_ONLY
        else:
            if sigmoid is None:
                sigmoid = not self.raw_feat
            flags |= (mp.MLP_SIGMOID if sigmoid else 0) | (mp.MLP_WITH_SIGMA if with_sigma else 0)
            if precision == "fp16" and self.W == 256 and FP16_SPLIT_HEADS:
                # the 8 x 256 network's output error under fp16 is its rgb head's (128 terms, nothing behind it): that head with
                # split weights and activations, 3 MFMAs per product (mlp_fused.hip, HX)
                flags |= mp.MLP_F16_HEADS
        spec = self._spec(n_freq, flags)
        spec.check()
        stream, bias, bd_folded = self._packed(spec, x.device)
        W = self.W
        l1 = self.xyz_encoding_1[0]
        l5 = self.xyz_encoding_5[0]
        ld = self.dir_encoding[0]
        n_pe = spec.n_pe

        def fold(src, lin, col0, width, name, bias=None, defer=False):
            """(R, O) = bias + src @ lin.weight[:, col0:col0+width]^T ; R rows map to samples by division.
            defer: return (None, R, job) for _fold_many instead of launching."""
            if width == 0:
                if src is not None and src.shape[-1] != 0:
                    raise ValueError(f"{name}: network takes no such input")
                return L.dev(lin.bias if bias is None else bias).view(1, -1), 1
            if src is None:
                raise ValueError(f"{name}: the network expects {width} per-row channels")
            s2 = L.dev(src).reshape(-1, src.shape[-1])
            if s2.shape[1] != width:
                raise ValueError(f"{name}: expected {width} channels, got {s2.shape[1]}")
            R = s2.shape[0]
            if M % R != 0:
                raise ValueError(f"{name}: {R} rows do not divide {M} samples")
            if defer:
                return None, R, (s2, lin, col0, width, bias)
            return self._fold(s2, lin, col0, width, bias=bias), R

        # the (up to three) folds of this call go out as one launch
        jobs, slots = [], {}

        def want(name, src, lin, col0, width, what, bias=None):
            r = fold(src, lin, col0, width, what, bias=bias, defer=True)
            if r[0] is None:
                slots[name] = (len(jobs), r[1])
                jobs.append(r[2])
            else:
                slots[name] = r
        want("rb1", code, l1, n_pe, spec.n_code, "code")
        want("rb5", code, l5, n_pe, spec.n_code, "code")
        if sigma_only:   # the dir branch is not evaluated (nerf.py:179-180)
            slots["rbd"] = (L.dev(ld.bias).view(1, -1), 1)
        else:
            want("rbd", dir_src, ld, W, self.in_channels_dir, "dir_src", bias=bd_folded)   # bd + Wd[:, :W] bf
        outs = self._fold_many(jobs) if jobs else []
        get = lambda name: (outs[slots[name][0]], slots[name][1]) if isinstance(slots[name][0], int) else slots[name]
        (rb1, R1), (rb5, R5), (rbd, Rd) = get("rb1"), get("rb5"), get("rbd")
        n_cols = 1 if sigma_only else self.out_channels + (1 if with_sigma else 0)
        out = torch.empty((M, n_cols), device=x.device, dtype=torch.float32)
        fl = None
        if flip is not None:
            fl = L.dev(flip.reshape(-1), torch.uint8)
        desc = L.MlpDesc(W=W, D=self.D, n_out=self.out_channels, flags=flags, n_freq=n_freq, reserved=0,
                         overflow=overflow.ptr() if spec.f16 else None)
        win = embedding_window(n_freq, n_freq if alpha is None else alpha)
        for k in range(16):
            desc.window[k] = win[k] if k < n_freq else 0.0
        if M > 0 and n_live is not None:
            if xyz.dim() != 3 or out_tr_S:
                raise ValueError("n_live needs xyz (rays, S, 3) and the sample-major output layout")
            nl = L.dev(n_live, torch.int32).reshape(-1)
            if nl.shape[0] != xyz.shape[0]:
                raise ValueError(f"n_live: expected {xyz.shape[0]} entries, got {nl.shape[0]}")
            prof = L.profile_begin()
            L.call("moda_mlp_live_fwd", L._c.byref(desc), L.ptr(stream), L.ptr(bias), L.ptr(x), L.ptr(fl),
                   L.ptr(rb1), L.ptr(rb5), R1, M // R1, L.ptr(rbd), Rd, M // Rd, L.ptr(out), n_cols, M, L.ptr(nl), xyz.shape[1],
                   L.stream())
            L.profile_end(prof, f"mlp_fused_W{W}_{_TAG[spec.precision]}", M)
        elif M > 0:
            prof = L.profile_begin()
            L.call("moda_mlp_fwd", L._c.byref(desc), L.ptr(stream), L.ptr(bias), L.ptr(x), L.ptr(fl),
                   L.ptr(rb1), L.ptr(rb5), R1, M // R1, L.ptr(rbd), Rd, M // Rd, L.ptr(out), n_cols, int(out_tr_S), M, L.stream())
            L.profile_end(prof, f"mlp_fused_W{W}_{_TAG[spec.precision]}", M)
        if out_tr_S:
            return out.view(M // out_tr_S, n_cols, out_tr_S)
        return out.view(lead + (n_cols,))


    def fused_composite(self, xyz, z_vals, rays_d, beta, n_freq=10, alpha=None, dir_src=None, flip=None, noise=None, cyc=None,
                        want_weights=True, want_visibility=False):
        """The colour network AND the compositing of rendering.py:183-237 in one kernel (`moda_mlp_composite_fwd`, bf16 mode):
        xyz (N,S,3) with S in {32, 64, 128, 256}; z_vals (N,S); rays_d (N,3); noise (N,S)|None already scaled; cyc (N,S)|None.
        -> dict(rgb, depth, sil, weights, visibility, cyc_out) -- bit-identical to `fused` + `rendering.composite` -- or None
        when the kernel does not serve this network / shape (the caller then takes the two-kernel route)."""
        L.no_grad_only(xyz, dir_src, *self.parameters())
        if xyz.dim() != 3:
            return None
        N, S, _ = xyz.shape
        if (self.W != 256 or self.out_channels != 3 or self.raw_feat or S not in (32, 64, 128, 256) or N == 0
                or self.in_channels_xyz != 3 + 6 * n_freq or n_freq > 10):
            return None
        flags = mp.MLP_BF16 | mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA
        spec = self._spec(n_freq, flags)
        spec.check()
        x = L.dev(xyz).reshape(-1, 3)
        M = x.shape[0]
        ds = L.dev(dir_src).reshape(-1, dir_src.shape[-1])
        Rd = ds.shape[0]
        if ds.shape[1] != self.in_channels_dir or M % Rd or (Rd != 1 and (M // Rd) % 32):
            return None
        stream, bias, bd_folded = self._packed(spec, x.device)
        rbd = self._fold(ds, self.dir_encoding[0], self.W, self.in_channels_dir, bias=bd_folded)
        rb1 = L.dev(self.xyz_encoding_1[0].bias).view(1, -1)
        rb5 = L.dev(self.xyz_encoding_5[0].bias).view(1, -1)
        dev_ = x.device
        o = {"rgb": torch.empty((N, 3), device=dev_), "depth": torch.empty((N,), device=dev_), "sil": torch.empty((N,), device=dev_),
             "weights": torch.empty((N, S), device=dev_) if want_weights else None,
             "visibility": torch.empty((N, S), device=dev_) if want_visibility else None,
             "cyc_out": torch.empty((N,), device=dev_) if cyc is not None else None,
             "feat": None, "vis_out": None, "n_used": None}
        fl = None if flip is None else L.dev(flip.reshape(-1), torch.uint8)
        ns = None if noise is None else L.dev(noise).reshape(-1)
        cy = None if cyc is None else L.dev(cyc).reshape(-1)
        desc = L.MlpDesc(W=self.W, D=self.D, n_out=3, flags=flags, n_freq=n_freq, reserved=0)
        win = embedding_window(n_freq, n_freq if alpha is None else alpha)
        for k in range(16):
            desc.window[k] = win[k] if k < n_freq else 0.0
        prof = L.profile_begin()
        L.call("moda_mlp_composite_fwd", L._c.byref(desc), L.ptr(stream), L.ptr(bias), L.ptr(x), L.ptr(fl), L.ptr(rb1), L.ptr(rb5),
               1, M, L.ptr(rbd), Rd, M // Rd, L.ptr(L.dev(z_vals).reshape(-1)), L.ptr(L.dev(rays_d).reshape(-1, 3)),
               L.ptr(L.dev(beta).reshape(1)), L.ptr(ns), L.ptr(cy), S, M, L.ptr(o["rgb"]), L.ptr(o["depth"]), L.ptr(o["sil"]),
               L.ptr(o["weights"]), L.ptr(o["visibility"]), L.ptr(o["cyc_out"]), L.stream())
        L.profile_end(prof, f"mlp_fused_W{self.W}_bf16", M)
        return o


    def fused_warp_serves(self, S, embedding_xyz, precision=None):
        """Will `fused_warp` take a call with S samples per ray (the shape tests it makes before any launch)?"""
        precision = precision or _PRECISION
        return (precision in _WARP_PRECISIONS and self.W == 64 and S % 32 == 0 and S > 0 and self.out_channels <= 64 and self.raw_feat
                and self.in_channels_dir == 0 and embedding_xyz.N_freqs <= 10 and embedding_xyz.in_channels == 3
                and self.in_channels_xyz - (3 + 6 * embedding_xyz.N_freqs) >= 0)

    def fused_warp(self, xyz, embedding_xyz, code, bones, dq, skin_aux, backward, rays_per_set=1, pts_tf=None, cyc_ref=None,
                   precision=None, runs=None, runs_cover_code=False, want_xyz=True):
        """The skin net + skinning softmax + DQS warp as ONE kernel (`moda_mlp_warp_fwd`), one-MFMA modes (bf16 / fp16):
        xyz_out = DQS(softmax(gauss(bones, xyz) + self([PE(xyz), code])), pts_tf or xyz) -- the chain gauss_mlp_skinning
        (geom_utils.py:202-217) -> neu_dbs (:372-456) of rendering.py:304-319 (backward=True: inverse transforms, per-set
        bones) and :330-341 (backward=False, cyc_ref -> cycle distance).  The (rays, B, S) logits are never written.

        xyz (N,S,3) with S % 32 == 0; code (R,128) with R in {1, N/rays_per_set}; bones (B,10) shared or
        (N/rays_per_set,B,10); dq (N/rays_per_set, B*8).  Returns (xyz_out (N,S,3), cyc (N,S) | None), or None when the
        kernel does not serve this case (the caller then takes the two-kernel route).
        runs (N/rays_per_set,) int32: `moda_row_runs` of the sets, when the caller has it already (render_rays detects the runs of
        the reference's per-ray repeats ONCE per call, jointly over bone_rts and time_embedded); runs_cover_code: that partition
        is also valid for `code`'s rows, so the code folds run on run starts only.  `bones` rows that do not start a run are then
        never read (bone_transform(run_start=...)).  want_xyz=False (with cyc_ref): only the cycle distance is returned."""
        L.no_grad_only(xyz, code, bones, dq, skin_aux, pts_tf, *self.parameters())
        N, S, _ = xyz.shape
        B = self.out_channels
        k = int(rays_per_set)
        if (self.W != 64 or S % 32 != 0 or B > 64 or not self.raw_feat or self.in_channels_dir != 0 or N % k
                or embedding_xyz.N_freqs > 10 or embedding_xyz.in_channels != 3):
            return None
        nsets = N // k
        # a direct call without `precision=` runs the kernel in the current mode if that is one of its 16-bit modes (fp16 mode: this
        # kernel keeps fp16 operands, see default_precision), else in bf16 -- as before round 4; render_rays always says which
        precision = precision or (_PRECISION if _PRECISION in ("bf16", "fp16") else "bf16")
        if precision not in _WARP_PRECISIONS:
            return None
        flags = _PREC_FLAGS[precision]
        if precision == "fp16":
            overflow.poll()
            if FP16_SPLIT_HEADS:
                flags |= mp.MLP_F16_HEADS           # the last two layers with split operands: see mlp_fused.hip (HX)
        n_freq = embedding_xyz.N_freqs
        spec = self._spec(n_freq, flags)
        spec.check()
        x = L.dev(xyz).reshape(-1, 3)
        M = x.shape[0]
        if M == 0:
            return None
        stream, bias, bd_folded = self._packed(spec, x.device)
        c2 = L.dev(code).reshape(-1, code.shape[-1])
        R1 = c2.shape[0]
        if c2.shape[1] != spec.n_code or R1 not in (1, nsets):
            return None
        l1, l5, ld = self.xyz_encoding_1[0], self.xyz_encoding_5[0], self.dir_encoding[0]
        fold_runs = runs if (runs is not None and runs_cover_code and R1 == nsets and R1 > 1) else None
        rb1, rb5 = self._fold_many([(c2, l1, spec.n_pe, spec.n_code, None), (c2, l5, spec.n_pe, spec.n_code, None)], run_start=fold_runs)
        rbd = bd_folded                                  # dir bias with xyz_encoding_final's folded in (see _packed)
        bn = L.dev(bones).reshape(-1, B, 10)
        q = L.dev(dq).reshape(-1, B, 8)
        if q.shape[0] != nsets or bn.shape[0] not in (1, nsets):
            raise ValueError(f"fused_warp: expected {nsets} transform sets and 1 or {nsets} bone sets of {B} bones, got "
                             f"{q.shape[0]} and {bn.shape[0]}")
        tiles = L.load().moda_warp_tiles(B)
        qtab = torch.empty((bn.shape[0] * tiles * 320,), device=x.device, dtype=torch.float32)
        dqtab = torch.empty((nsets * tiles * 2048,), device=x.device, dtype=torch.uint8)
        # Many sets (the reference's layout: every frame's bone_rts row repeated for each of its rays, moda.py:1281-1311): the
        # operand tables are built once per RUN of identical consecutive sets and read at the run's first slot -- at config 2,
        # 256 slots of the 65 536 are ever written or read (0.13 GB of tables per warp otherwise)
        if runs is not None and (runs.shape[0] != nsets or runs.dtype != torch.int32):
            raise ValueError(f"fused_warp: runs must be ({nsets},) int32")
        if runs is None and nsets >= 512:
            runs = torch.empty((nsets,), device=x.device, dtype=torch.int32)
            ws = torch.empty(((nsets + 255) // 256,), device=x.device, dtype=torch.int32)
            per_set_bones = bn.shape[0] == nsets
            L.call("moda_row_runs", L.ptr(q), B * 8, L.ptr(bn) if per_set_bones else None, B * 10, nsets, L.ptr(runs), L.ptr(ws),
                   L.stream())
        L.call("moda_warp_tables_fwd", L.ptr(bn), bn.shape[0], L.ptr(q), nsets, 1 if backward else 0, L.ptr(L.dev(skin_aux)), B,
               L.ptr(qtab), L.ptr(dqtab), L.ptr(runs), L.stream())
        if not want_xyz and cyc_ref is None:
            raise ValueError("fused_warp: want_xyz=False needs cyc_ref")
        out = torch.empty((N, S, 3), device=x.device, dtype=torch.float32) if want_xyz else None
        cr = None if cyc_ref is None else L.dev(cyc_ref).reshape(-1, 3)
        cyc = torch.empty((N, S), device=x.device, dtype=torch.float32) if cr is not None else None
        pt = None if pts_tf is None else L.dev(pts_tf).reshape(-1, 3)
        desc = L.MlpDesc(W=self.W, D=self.D, n_out=B, flags=flags, n_freq=n_freq, reserved=2 if fold_runs is not None else 0,
                         overflow=overflow.ptr() if spec.f16 else None)           # 2: MODA_MLP_ROWS_AT_RUNS
        win = embedding_window(n_freq, embedding_xyz.alpha)
        for i in range(16):
            desc.window[i] = win[i] if i < n_freq else 0.0
        prof = L.profile_begin()
        L.call("moda_mlp_warp_fwd", L._c.byref(desc), L.ptr(stream), L.ptr(bias), L.ptr(x), L.ptr(rb1), L.ptr(rb5), R1, M // R1,
               L.ptr(rbd), L.ptr(qtab), 0 if bn.shape[0] == 1 and nsets != 1 else k, L.ptr(dqtab), k, L.ptr(pt), L.ptr(cr),
               L.ptr(out), L.ptr(cyc), S, M, L.ptr(runs), L.stream())
        L.profile_end(prof, f"mlp_warp_W64_{_TAG[precision]}", M)
        return out, cyc


class NeRFUnc(NeRF):
    """nerf.py:502-511: the uncertainty head is a plain NeRF evaluated on [PE(x, y, t), vid_code]."""

    def forward(self, x, xyz=None, sigma_only=False):
        return super().forward(x, sigma_only=sigma_only)
