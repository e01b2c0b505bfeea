"""Lane-level model of the fused MLP kernel's MFMA dataflow (test infrastructure).

Replays a packed weight stream (moda_amd/mlp_pack.py) through the gfx950 MFMA operand maps
(A/B/C-D lane layouts of v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16, as documented in the
CDNA4 guide) in the order csrc/mlp_fused.hip issues them.  It checks the packer's index algebra and
the accumulator-as-next-operand trick on the CPU, before anything runs on a GPU.
"""
import numpy as np

from moda_amd import mlp_pack as mp
from oracle import moda_oracle as orc


def _reg_row(reg, h):
    return (reg & 3) + 8 * (reg >> 2) + 4 * h


class _Stream:
    def __init__(self, spec, wstream):
        self.spec = spec
        self.f = wstream.reshape(-1, 64, spec.elems)
        self.i = 0

    def next(self):
        v = self.f[self.i]
        self.i += 1
        return v

    def end_layer(self):
        while self.i % self.spec.chf:
            self.i += 1


def _mma(spec, D, afrag, bop, rnd):
    """One fragment step.  afrag (64,E) A elements per lane; bop (64,E) B elements per lane; D (32,32)."""
    lane = np.arange(64)
    r, h = lane & 31, lane >> 5
    if spec.bf16:
        A = np.zeros((32, 16), np.float32)
        B = np.zeros((16, 32), np.float32)
        for j in range(8):
            A[r, 8 * h + j] = afrag[:, j]
            B[8 * h + j, r] = bop[:, j]
        D += rnd(A).astype(np.float64) @ rnd(B).astype(np.float64)
    else:
        for j in range(4):   # four k=2 MFMAs
            A = np.zeros((32, 2), np.float32)
            B = np.zeros((2, 32), np.float32)
            A[r, h] = afrag[:, j]
            B[h, r] = bop[:, j]
            D += A.astype(np.float64) @ B.astype(np.float64)


def _act_bop(spec, Dt, sub, relu):
    """B operand lanes built from accumulator tile Dt exactly as the kernel reuses its registers."""
    lane = np.arange(64)
    col, h = lane & 31, lane >> 5
    out = np.zeros((64, spec.elems), np.float32)
    for j in range(spec.elems):
        reg = (8 * sub + j) if spec.bf16 else (4 * sub + j)
        v = Dt[_reg_row(reg, h), col]
        out[:, j] = np.maximum(v, 0) if relu else v
    return out


def emulate(spec, wstream, bias, xyz, rb1, rb5, rbd, window):
    """xyz (n,3) with n a multiple of 32; rb1/rb5 (n,W) and rbd (n,W/2) already gathered per sample."""
    rnd = orc.bf16_round if spec.bf16 else (lambda a: a)
    n = xyz.shape[0]
    NT, NTD = spec.NT, spec.NTD
    emb = orc.embedding(xyz.astype(np.float32), spec.n_freq, None)
    if window is not None:
        w = np.concatenate([np.ones(3, np.float32)] + [np.repeat(window[k], 6) for k in range(spec.n_freq)])
        emb = emb * w
    lane = np.arange(64)
    col, h = lane & 31, lane >> 5
    outs = []
    for c0 in range(0, n, 32):
        st = _Stream(spec, wstream)
        # PE B operand per slot
        pe = np.zeros((32, 64), np.float32)   # [slot][lane]
        for p in range(32):
            for l in range(64):
                f = mp.pe_slot_feature(p, l >> 5, spec.n_freq)
                pe[p, l] = emb[c0 + (l & 31), f] if f >= 0 else 0.0

        # the kernel streams a layer one output tile at a time: [PE groups] then [(t, s) over the input tiles]
        def seg_act(D, rts, src, n_in, relu_in, with_pe=False, with_act=True):
            for rt in rts:
                if with_pe:
                    for g in range(spec.peg):
                        a = st.next()
                        bop = np.stack([pe[g * spec.elems + j] for j in range(spec.elems)], 1)
                        _mma(spec, D[rt], a, bop, rnd)
                if with_act:
                    for t in range(n_in):
                        for s in range(spec.subs):
                            _mma(spec, D[rt], st.next(), _act_bop(spec, src[t], s, relu_in), rnd)

        def init_rows(rb, ntile):   # rb (n, 32*ntile) -> tiles [rt](32 rows, 32 cols)
            blk = rb[c0:c0 + 32].astype(np.float64)    # (32 cols, rows)
            return [blk[:, 32 * rt:32 * rt + 32].T.copy() for rt in range(ntile)]

        def init_bias(off, ntile):
            return [np.repeat(bias[off + 32 * rt: off + 32 * rt + 32].astype(np.float64)[:, None], 32, 1) for rt in range(ntile)]

        D = init_rows(rb1, NT)
        seg_act(D, range(NT), None, NT, True, with_pe=True, with_act=False); st.end_layer()
        boff = 0
        for _ in range(3):
            Dn = init_bias(boff, NT); seg_act(Dn, range(NT), D, NT, True); st.end_layer(); D = Dn; boff += spec.W
        Dn = init_rows(rb5, NT); seg_act(Dn, range(NT), D, NT, True, with_pe=True); st.end_layer(); D = Dn
        for _ in range(spec.D - 5):
            Dn = init_bias(boff, NT); seg_act(Dn, range(NT), D, NT, True); st.end_layer(); D = Dn; boff += spec.W
        sig = None
        if spec.with_sigma:
            Ds = init_bias(boff + 32 * NT, 1)
            seg_act(Ds, [0], D, NT, True)
            sig = Ds[0][0, :].copy()
        if spec.sigma_only:
            st.end_layer()
            outs.append(sig[:, None])
            continue
        st.end_layer(); boff += (NT + 1) * 32
        # xyz_encoding_final is folded into the dir layer (mlp_pack.fold_final): its input is the last hidden layer (post-ReLU)
        Dd = init_rows(rbd, NTD); seg_act(Dd, range(NTD), D, NT, True); st.end_layer()
        nout_t = (spec.n_out + 31) // 32
        Do = init_bias(boff, nout_t); seg_act(Do, range(nout_t), Dd, NTD, True); st.end_layer()
        assert st.i == spec_nfrags(spec), (st.i, spec_nfrags(spec))
        rgb = np.concatenate(Do, 0)[:spec.n_out].T    # (32 cols, n_out)
        if spec.flags & mp.MLP_SIGMOID:
            rgb = 1 / (1 + np.exp(-rgb))
        if spec.with_sigma:
            rgb = np.concatenate([rgb, sig[:, None]], 1)
        outs.append(rgb)
    return np.concatenate(outs, 0)


def spec_nfrags(spec):
    return mp.stream_index(spec).nfrags
