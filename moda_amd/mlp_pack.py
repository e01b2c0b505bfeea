"""Weight-stream layout of the fused PE+MLP kernel (csrc/mlp_fused.hip) -- host side.

The kernel consumes weights as a flat sequence of 1 KiB *fragments* (64 lanes x 16 B), in exactly the
order its MFMAs issue, so that the LDS image is lane-linear and LDS-DMA can fill it.  This module
builds, once per network shape, the gather index that turns a reference `NeRF` state dict
(nnutils/nerf.py:109-140) into that stream, and the index of the LDS-resident bias block.

Fragment = A operand of one 32-row output tile for one k-group:
  lane l -> output row 32*rt + (l & 31), lane half h = l >> 5;
  fp32 : 4 elements = 4 consecutive v_mfma_f32_32x32x2_f32 steps, step s takes k index (h)
  bf16 : 8 elements = one v_mfma_f32_32x32x16_bf16, element j takes k index 8h + j.
Which input feature sits at a k index depends on the B operand's source:
  * PE slots (layer 1 and the skip layer): slot p = g*E + j, see `pe_slot_feature`;
  * an accumulator tile of the previous layer used directly as B operand: the MFMA C/D register map
    (row = (reg&3) + 8*(reg>>2) + 4h) fixes the feature of each k index, see `act_feature`.
`xyz_encoding_final` (a Linear without activation) is folded into `dir_encoding` on the host (`fold_final`): the stream
holds no fragments for it.  Layer order, row-tile / k order and the per-layer padding to whole ring chunks mirror the kernel and
`stream_shape()` in mlp_fused.hip; tests/test_mlp_pack.py replays the stream through a lane-level MFMA
model and checks it against the oracle.
"""
from dataclasses import dataclass

import numpy as np

MLP_BF16 = 1
MLP_SIGMOID = 2
MLP_WITH_SIGMA = 4
MLP_SIGMA_ONLY = 8
MLP_BF16X3 = 16     # split-bf16: operands as bf16 hi + lo, three MFMAs per product; bf16 fragment geometry, (hi, lo) pairs
MLP_F16 = 32        # fp16 operands (v_mfma_f32_32x32x16_f16): bf16 fragment geometry and stream layout, fp16 elements
MLP_F16_HEADS = 64  # with MLP_F16: head fragments as (rounding, residual) pairs -- the dir and rgb layers of the 64-wide network
                    # (fused skin + warp kernel), the rgb head alone of the 256-wide one (moda_mlp_fwd / _live_fwd)

FRAG_BYTES = 1024


@dataclass(frozen=True)
class MlpSpec:
    W: int              # hidden width (64, 128, 256)
    D: int              # xyz_encoding layers (5..8), skips == [4]
    n_out: int          # rgb head rows
    in_xyz: int         # in_channels_xyz of the module: 3+6*n_freq PE features (+ code channels)
    in_dir: int         # in_channels_dir of the module (dir embedding + env/appearance codes), may be 0
    n_freq: int = 10
    flags: int = 0

    @property
    def bf16(self):
        """bf16 fragment geometry (8 two-byte elements per lane and fragment): the bf16, split-bf16 and fp16 modes."""
        return bool(self.flags & (MLP_BF16 | MLP_BF16X3 | MLP_F16))

    @property
    def f16(self):
        return bool(self.flags & MLP_F16)

    @property
    def heads_split(self):
        return bool(self.flags & MLP_F16_HEADS)

    @property
    def precision(self):
        """Name of the arithmetic of this spec, as `set_precision` spells it."""
        return "fp16" if self.f16 else ("bf16x3" if self.x3 else ("bf16" if self.flags & MLP_BF16 else "fp32"))

    @property
    def x3(self):
        return bool(self.flags & MLP_BF16X3)

    @property
    def sigma_only(self):
        return bool(self.flags & MLP_SIGMA_ONLY)

    @property
    def with_sigma(self):
        return bool(self.flags & (MLP_WITH_SIGMA | MLP_SIGMA_ONLY))

    @property
    def n_pe(self):
        return 3 + 6 * self.n_freq

    @property
    def n_code(self):      # per-row code channels appended to the PE in the first / skip layer input
        return self.in_xyz - self.n_pe

    @property
    def NT(self):
        return self.W // 32

    @property
    def NTD(self):
        return max(self.NT // 2, 1)

    @property
    def chf(self):         # fragments per ring chunk
        return 8 if self.W == 64 else 16

    @property
    def subs(self):        # fragments per 32-feature activation tile
        return 2 if self.bf16 else 4

    @property
    def elems(self):       # elements per lane per fragment
        return 8 if self.bf16 else 4

    @property
    def peg(self):         # fragments covering the 64 PE slots
        return 32 // self.elems

    @property
    def nbias(self):
        return (self.D - 2) * self.W + (self.NT + 1) * 32 + 64

    def check(self):
        if self.W not in (64, 128, 256) or not (5 <= self.D <= 8) or not (1 <= self.n_out <= 64):
            raise NotImplementedError(f"fused MLP kernel is not instantiated for {self}")
        if bin(self.flags & (MLP_BF16 | MLP_BF16X3 | MLP_F16)).count("1") > 1:
            raise ValueError("MLP_BF16, MLP_BF16X3 and MLP_F16 are three modes, not options of each other")
        if self.heads_split and (not self.f16 or self.sigma_only or self.W == 128 or (self.W == 64 and self.with_sigma)):
            raise ValueError("MLP_F16_HEADS: the fp16 64-wide network with raw outputs (moda_mlp_warp_fwd) or the 256-wide one")
        if not (0 <= self.n_freq <= 10) or self.n_code < 0:
            raise NotImplementedError(f"unsupported positional encoding / input width in {self}")


def pe_slot_feature(p, h, n_freq):
    """Embedding feature index (nerf.py:58-72 order) of PE slot p in lane half h, or -1 for zero."""
    if p < 30:
        k, c = divmod(p, 3)
        return 3 + 6 * k + 3 * h + c if k < n_freq else -1
    if p == 30:
        return h          # x | y
    return 2 if h == 0 else -1   # z | pad


def act_feature(bf16, sub, j, h):
    """Row of a 32-row accumulator tile that element j of fragment `sub` multiplies (lane half h)."""
    if bf16:
        return 16 * sub + 8 * (j >> 2) + 4 * h + (j & 3)
    s = 4 * sub + j
    return (s & 3) + 8 * (s >> 2) + 4 * h


WEIGHT_ORDER = ("xyz_encoding_final.weight", "dir_encoding.0.weight", "sigma.weight", "rgb.0.weight")
BIAS_ORDER = ("xyz_encoding_final.bias", "dir_encoding.0.bias", "sigma.bias", "rgb.0.bias")


def weight_names(spec):
    return [f"xyz_encoding_{i+1}.0.weight" for i in range(spec.D)] + list(WEIGHT_ORDER)


def bias_names(spec):
    return [f"xyz_encoding_{i+1}.0.bias" for i in range(spec.D)] + list(BIAS_ORDER)


def weight_shapes(spec):
    W = spec.W
    shp = {}
    for i in range(spec.D):
        cin = spec.in_xyz if i == 0 else (W + spec.in_xyz if i == 4 else W)
        shp[f"xyz_encoding_{i+1}.0.weight"] = (W, cin)
    shp["xyz_encoding_final.weight"] = (W, W)
    shp["dir_encoding.0.weight"] = (W // 2, W + spec.in_dir)
    shp["sigma.weight"] = (1, W)
    shp["rgb.0.weight"] = (spec.n_out, W // 2)
    return shp


class StreamIndex:
    """Gather indices into cat([w.reshape(-1) for w in weight_names] + [0]) and the same for biases."""

    def __init__(self, spec):
        spec.check()
        self.spec = spec
        shapes = weight_shapes(spec)
        offs, off = {}, 0
        for n in weight_names(spec):
            offs[n] = off
            off += shapes[n][0] * shapes[n][1]
        self.zero = off                    # index of the appended zero
        self.n_weight_elems = off
        self._shapes, self._offs = shapes, offs
        frags = []                         # list of (64, E) int64 index arrays
        parts = []                         # split mode: every real fragment is emitted twice, as (hi, lo); else all 0

        pair_now = [spec.x3]               # every fragment as a (values, residuals) pair: split mode, or the current layer's

        def emit(fr):
            frags.append(fr)
            parts.append(0)
            if pair_now[0]:
                frags.append(fr)
                parts.append(1)
        lane = np.arange(64)
        self._r = (lane & 31)[:, None]
        self._h = (lane >> 5)[:, None]
        self._j = np.arange(spec.elems)[None, :]
        E = spec.elems
        self._pe_feat = [np.asarray([[pe_slot_feature(g * E + j, l >> 5, spec.n_freq) for j in range(E)]
                                     for l in range(64)], np.int64) for g in range(spec.peg)]
        self._act_feat = [np.asarray([[act_feature(spec.bf16, sub, j, l >> 5) for j in range(E)]
                                      for l in range(64)], np.int64) for sub in range(spec.subs)]

        def pad_layer():
            while len(frags) % spec.chf:
                frags.append(np.full((64, spec.elems), self.zero, np.int64))
                parts.append(0)

        # every layer is streamed one 32-row OUTPUT tile at a time: [PE groups] then [(t, s) over the input tiles]
        def act_segment(name, rts, n_in_tiles, col0, sigma_row=False, with_pe=False, with_act=True):
            for rt in rts:
                if with_pe:
                    for g in range(spec.peg):
                        emit(self._pe_frag(name, rt, g))
                if with_act:
                    for t in range(n_in_tiles):
                        for s in range(spec.subs):
                            emit(self._act_frag(name, rt, t, s, col0, sigma_row))

        NT, NTD = spec.NT, spec.NTD
        act_segment("xyz_encoding_1.0.weight", range(NT), NT, 0, with_pe=True, with_act=False); pad_layer()
        for i in (1, 2, 3):
            act_segment(f"xyz_encoding_{i+1}.0.weight", range(NT), NT, 0); pad_layer()
        act_segment("xyz_encoding_5.0.weight", range(NT), NT, spec.in_xyz, with_pe=True); pad_layer()
        for i in range(5, spec.D):
            act_segment(f"xyz_encoding_{i+1}.0.weight", range(NT), NT, 0); pad_layer()
        if spec.with_sigma:
            act_segment("sigma.weight", [0], NT, 0)
        pad_layer()
        if not spec.sigma_only:
            # xyz_encoding_final has no activation (nerf.py:184-187), so it is folded into dir_encoding on the host:
            # the tensor gathered here under the name dir_encoding.0.weight must be fold_final()'s product
            # MLP_F16_HEADS: these two layers with split operands (W = 64), the rgb head alone (W = 256)
            pair_now[0] = spec.x3 or (spec.heads_split and spec.W == 64)
            act_segment("dir_encoding.0.weight", range(NTD), NT, 0); pad_layer()
            pair_now[0] = spec.x3 or spec.heads_split
            act_segment("rgb.0.weight", range((spec.n_out + 31) // 32), NTD, 0); pad_layer()
            pair_now[0] = spec.x3
        self.widx = np.stack(frags, 0).reshape(-1)     # (nfrags * 64 * E,)
        self.nfrags = len(frags)
        self.nchunks = self.nfrags // spec.chf
        self.stream_bytes = self.nfrags * FRAG_BYTES
        self.part = np.asarray(parts, np.int8)         # per fragment: 0 = the values (split mode: their bf16 roundings), 1 = residuals

        # bias block: hidden layers 2..4, 6..D | final (W) + sigma at [W] padded to (NT+1)*32 | rgb padded to 64
        bshapes = {n: shapes[n.replace(".bias", ".weight")][0] for n in bias_names(spec)}
        boffs, off = {}, 0
        for n in bias_names(spec):
            boffs[n] = off
            off += bshapes[n]
        self.bzero = off
        self._boffs = boffs
        b = []
        for i in (1, 2, 3) + tuple(range(5, spec.D)):
            b += list(boffs[f"xyz_encoding_{i+1}.0.bias"] + np.arange(spec.W))
        fin = np.full((NT + 1) * 32, self.bzero, np.int64)
        fin[:spec.W] = boffs["xyz_encoding_final.bias"] + np.arange(spec.W)
        fin[spec.W] = boffs["sigma.bias"]
        b += list(fin)
        rgb = np.full(64, self.bzero, np.int64)
        rgb[:spec.n_out] = boffs["rgb.0.bias"] + np.arange(spec.n_out)
        b += list(rgb)
        self.bidx = np.asarray(b, np.int64)
        assert self.bidx.shape[0] == spec.nbias

    def _gather(self, name, rows, cols, valid):
        nrows, ncols = self._shapes[name]
        ok = valid & (rows < nrows) & (cols >= 0) & (cols < ncols)
        idx = self._offs[name] + np.where(ok, rows, 0) * ncols + np.where(ok, cols, 0)
        return np.where(ok, idx, self.zero).astype(np.int64)

    def _pe_frag(self, name, rt, g):
        spec = self.spec
        feat = self._pe_feat[g]
        rows = np.broadcast_to(32 * rt + self._r, (64, spec.elems))
        return self._gather(name, rows, feat, feat >= 0)

    def _act_frag(self, name, rt, t, sub, col0, sigma_row):
        spec = self.spec
        feat = 32 * t + self._act_feat[sub]
        rows = np.broadcast_to(32 * rt + self._r, (64, spec.elems))
        return self._gather(name, rows, col0 + feat, np.ones((64, spec.elems), bool))

    # ------------------------------------------------------------------ device packing (moda_mlp_pack)
    def codes(self):
        """(wcode, bcode) int32 tables of `moda_mlp_pack`: per stream / bias element (source id << 24) | element offset,
        -1 for a zero.  Source ids follow weight_names() / bias_names(); the source called dir_encoding.0.weight is the
        FOLDED (W/2, W) product Wd[:, :W] Wf (fold_final) -- the stream never reads dir_encoding's per-ray columns."""
        if getattr(self, "_codes", None) is None:
            spec = self.spec
            names = weight_names(spec)
            starts = np.asarray([self._offs[n] for n in names] + [self.zero], np.int64)
            sid = np.searchsorted(starts, self.widx, side="right") - 1
            off = self.widx - starts[np.minimum(sid, len(names) - 1)]
            d_id = names.index("dir_encoding.0.weight")
            ncols = self._shapes["dir_encoding.0.weight"][1]
            isd = (sid == d_id)
            row, col = off // ncols, off % ncols
            assert not np.any(isd & (col >= spec.W))
            off = np.where(isd, row * spec.W + col, off)
            assert off.max() < (1 << 24) and len(names) <= 16
            lo_part = np.repeat(self.part.astype(np.int64), 64 * spec.elems) << 30        # split mode: bit 30 = residual fragment
            wcode = np.where(self.widx == self.zero, -1, (sid << 24) | off | lo_part).astype(np.int32)
            bn = bias_names(spec)
            bstarts = np.asarray([self._boffs[n] for n in bn] + [self.bzero], np.int64)
            bs = np.searchsorted(bstarts, self.bidx, side="right") - 1
            boff = self.bidx - bstarts[np.minimum(bs, len(bn) - 1)]
            bcode = np.where(self.bidx == self.bzero, -1, (bs << 24) | boff).astype(np.int32)
            self._codes = (wcode, bcode)
        return self._codes

    def pack_codes_numpy(self, params):
        """What moda_mlp_pack computes, in numpy (fp32; the bf16 rounding is the caller's): params as for pack_numpy but
        with dir_encoding.0.weight already the folded (W/2, W) product."""
        wcode, bcode = self.codes()
        ws = [np.asarray(params[n], np.float32).reshape(-1) for n in weight_names(self.spec)]
        bs = [np.asarray(params[n], np.float32).reshape(-1) for n in bias_names(self.spec)]

        def take(code, srcs):
            out = np.zeros(code.shape, np.float32)
            ok = code >= 0
            sid, off = (code >> 24) & 15, code & 0xffffff            # (bit 30, the split mode's part flag, is the caller's)
            for i, a in enumerate(srcs):
                m = ok & (sid == i)
                out[m] = a[off[m]]
            return out
        return take(wcode, ws), take(bcode, bs)

    # ------------------------------------------------------------------ numpy packing (tests, oracle-side checks)
    def pack_numpy(self, params):
        spec = self.spec
        flat = np.concatenate([np.asarray(params[n], np.float32).reshape(-1) for n in weight_names(spec)]
                              + [np.zeros(1, np.float32)])
        bflat = np.concatenate([np.asarray(params[n], np.float32).reshape(-1) for n in bias_names(spec)]
                               + [np.zeros(1, np.float32)])
        return flat[self.widx], bflat[self.bidx]


def fold_final(params):
    """xyz_encoding_final is a Linear WITHOUT activation feeding dir_encoding's Linear (nerf.py:184-187), so
        dir_encoding(cat[final(h), d]) = ReLU(Wd[:, :W] (Wf h + bf) + Wd[:, W:] d + bd) = ReLU((Wd[:, :W] Wf) h + Wd[:, W:] d + bd')
    with bd' = bd + Wd[:, :W] bf: one (W/2 x W) layer instead of a (W x W) and a (W/2 x W) one -- 65,536 of the coarse net's
    601,600 MACs per sample never have to be executed.  Returns a copy of `params` whose dir_encoding tensors are the folded
    ones (numpy, float64 product rounded once); the stream gathers the dir layer from it and no `final` layer at all."""
    W = params["xyz_encoding_final.weight"].shape[0]
    wd = np.asarray(params["dir_encoding.0.weight"], np.float64)
    wf = np.asarray(params["xyz_encoding_final.weight"], np.float64)
    bf = np.asarray(params["xyz_encoding_final.bias"], np.float64)
    out = dict(params)
    out["dir_encoding.0.weight"] = np.concatenate([wd[:, :W] @ wf, wd[:, W:]], 1).astype(np.float32)
    out["dir_encoding.0.bias"] = (np.asarray(params["dir_encoding.0.bias"], np.float64) + wd[:, :W] @ bf).astype(np.float32)
    return out


def split_bf16(a, round_fn):
    """(hi, lo) of the split-bf16 mode: hi = bf16(a), lo = bf16(a - hi); round_fn is a float32 -> bf16-valued-float32 rounding
    (oracle.bf16_round)."""
    a = np.asarray(a, np.float32)
    hi = round_fn(a).astype(np.float32)
    return hi, round_fn((a - hi).astype(np.float32)).astype(np.float32)


def stream_x3(index, gathered, round_fn):
    """The device stream of the split mode from the gathered fp32 values of `index` (StreamIndex of an x3 spec, whose real
    fragments come in pairs): fragments with part 0 hold the bf16 roundings, part 1 the rounded residuals.  What
    moda_mlp_pack writes with bf16 = 2 (as float32 holding bf16-representable numbers)."""
    hi, lo = split_bf16(gathered, round_fn)
    lo_frag = np.repeat(index.part.astype(bool), 512)
    return np.where(lo_frag, lo, hi)


_INDEX_CACHE = {}


def stream_index(spec):
    if spec not in _INDEX_CACHE:
        _INDEX_CACHE[spec] = StreamIndex(spec)
    return _INDEX_CACHE[spec]
