// The hidden-layer chain of a 64-wide network's backward as ONE launch (bf16-storage training route, gfx950).
//
// Reference: the autograd backward of the hidden nn.Linear + ReLU layers of NeRF.forward (nnutils/nerf.py:166-178) for the
// 5 x 64 networks (nerf_skin, nerf_vis): for l = L .. L - n + 1, with dh_l the gradient at layer l's pre-activation and h_{l-1}
// the (post-ReLU) input of that layer,
//     dW_l += dh_l^T h_{l-1}        db_l += 1^T dh_l        dh_{l-1} = (dh_l W_l) (.) [h_{l-1} > 0]
// Launched per layer (gemm_bf16.hip: one dW and one dX kernel each) every dh_l is written to HBM once and read twice and every
// h twice: 138 MB per layer at M = 262144.  Here a workgroup keeps a 128-row tile of dh in LDS while it walks the layers: the dX
// product's accumulators are masked, rounded to bf16 and written straight back into the LDS image as the next layer's dh; HBM
// sees dh_L and h_{L-1} .. h_{L-n} once each on the way in and dh_{L-n} once on the way out (201 MB for four layers instead of
// 552 MB), and eight launches become two.  The weight-gradient tiles (n x 64 x 64 fp32) live in the accumulator registers across
// all tiles of the (persistent) workgroup and leave as PARTIAL tiles, one set per workgroup, that a second small kernel sums into
// the gradients -- atomics from 512 workgroups on the same 4096 addresses would serialise (gemm_bf16.hip has the numbers).
//
// One LDS image layout serves every use of a tile ([row m][64 columns] bf16, 128-byte rows, 16-byte chunks XOR-swizzled by the
// row as in gemm_bf16.hip): read with the transposing ds_read_b64_tr_b16 it is the k-slow operand of dW (k = m); read with
// ds_read_b128 along a row it is the k-fast operand of dX (k = the 64 columns); and lane l of the dX accumulator tile holds
// column m = its own image row, so the masked result is written back with plain 8-byte stores.  Same arithmetic as the
// per-layer kernels: bf16 operands, fp32 accumulation, dh rounded to bf16 between layers, db summed from the rounded values.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef CHAIN64_PIN
#define CHAIN64_PIN 1
#endif
constexpr int W = 64;                      // layer width
constexpr int RT = 128;                    // rows of a tile
constexpr int RB = W * 2;                  // bytes per image row
constexpr int IMG = RT * RB;               // 16 KB
constexpr int WIMG = W * RB;               // 8 KB
constexpr int PART = W * W + W;            // floats of one layer's partial: dW then db
constexpr int MAXL = 4;

struct ChainArgs {
    const unsigned short* dh_in; long long ld_in;     // (M, in_cols) bf16
    const unsigned short* h[MAXL];                    // h[j]: input activations of layer L - j, (M, h_cols[j]) bf16, leading dimension ld_h[j]
    long long ld_h[MAXL];
    const unsigned short* wb[MAXL];                   // wb[j]: bf16 [o][i] (w_rows[j] x w_cols[j], leading dimension w_ld[j]) weights of layer L - j
    unsigned short* dh_out; long long ld_out;         // (M, 64) bf16: gradient at the pre-activation of layer L - n
    float* part;                                      // [gridDim.x][n][PART]
    long long M;
    int n;
    // narrower operands are zero-padded to 64 columns / rows as they are staged (the heads of the network: 32-wide dz_rgb, dir
    // activations and weights); multiples of 8
    int in_cols, h_cols[MAXL], w_rows[MAXL], w_cols[MAXL], w_ld[MAXL];
};

DEVINL int sw(int row) { return ((row >> 1) & 1) << 2; }
DEVINL unsigned pk2(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}

template <int NL>         // layers chained: 4 (the hidden layers 4, 3, 2, 1) or 2 (the heads: rgb -> dir_encoding -> layer D-1); even
__global__ __launch_bounds__(256, 2) void chain64_kernel(ChainArgs a) {
    static_assert(NL % 2 == 0 && NL <= MAXL, "the register stages alternate by layer parity");
    __shared__ __attribute__((aligned(16))) unsigned char lds[IMG + IMG + NL * WIMG + 4 * W * 4];
    unsigned char* Zs = lds;                  // dh tile
    unsigned char* Hs = lds + IMG;            // activation tile
    unsigned char* Ws = lds + 2 * IMG;        // the NL weight images [k = o][r = i], resident for the whole kernel
    float* red = (float*)(lds + 2 * IMG + NL * WIMG);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;       // dW: 32 x 32 block of (o, i) per wave
    const int wc2 = (wave & 1) * 64;                             // dX: (i block wr) x (m block wc2 .. + 64)

    // transposing-read bases (gemm_bf16.hip): lane 4q+p of 16-lane group g supplies row 8h + q (+ 16u, + 4), elements
    // 16 (g & 1) + 4p .. + 3 of the 32-wide block that starts at column `cb`
    auto tr_base = [&](int cb) { const int ch = cb / 8 + 2 * (g & 1) + (p4 >> 1); return (8 * h + q4) * RB + 16 * (ch ^ sw(q4)) + 8 * (p4 & 1); };
    const int xb_o = tr_base(wr), yb_i = tr_base(wc);            // dW: X = dh (r = o block), Y = h (c = i block); dX: X = W (r = i block wr)
    typedef s16x4 __attribute__((address_space(3))) lds_s16x4;
    auto tr_frag = [&](const unsigned char* base, int off, int u) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u) * RB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u + 4) * RB));
        union { struct { s16x4 a, b; } s; bf16x8 v; } o;
        o.s.a = lo;
        o.s.b = hi;
        return o.v;
    };

    // staging map of a 128 x 64 tile: thread -> 16-byte chunk (tid & 7) of rows (tid >> 3) + 32 e
    const int sch = tid & 7, srow = tid >> 3;
    uint4 zr[4], hr[2][4];                    // register stages: the next tile's dh, the activation tiles of the next TWO steps
    auto fetch_tile = [&](const unsigned short* src, long long ld, int cols, long long r0, uint4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long r = r0 + srow + 32 * e;
            dst[e] = make_uint4(0u, 0u, 0u, 0u);
            if (r < a.M && 8 * sch < cols) dst[e] = *(const uint4*)(src + r * ld + 8 * sch);
        }
    };
    auto stash_tile = [&](unsigned char* img, const uint4 (&src)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = srow + 32 * e;
            *(uint4*)(img + row * RB + 16 * (sch ^ sw(row))) = src[e];
        }
    };
    // the weights: NL x 8 KB, staged once (a per-step register prefetch of them made every step wait for ALL loads in flight:
    // the compiler parked them in the dX accumulators' registers and had to move them out at once, and vmcnt counts in order)
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = srow + 32 * e;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < a.w_rows[l] && 8 * sch < a.w_cols[l]) v = *(const uint4*)(a.wb[l] + (long long)row * a.w_ld[l] + 8 * sch);
            *(uint4*)(Ws + l * WIMG + row * RB + 16 * (sch ^ sw(row))) = v;
        }

    f32x16 dw[NL];
    float dbs[NL];                            // this thread's share of db_l: column tid & 63 over the rows 32 (tid >> 6) .. + 31 of every tile
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        dbs[l] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[l][r] = 0.f;
    }

    // The activation tiles are a linear stream of steps (tile, layer); the tile of step s + 2 is fetched during step s into
    // register stage s & 1 (= layer & 1: NL is even), so a load has two steps' time to land -- with one step the step time was
    // the load latency (2 TB/s); the next tile's dh comes in during layer NL - 2.
    const long long ntiles = (a.M + RT - 1) / RT;
    long long t = blockIdx.x;
    if (t < ntiles) {
        fetch_tile(a.dh_in, a.ld_in, a.in_cols, t * RT, zr);
        fetch_tile(a.h[0], a.ld_h[0], a.h_cols[0], t * RT, hr[0]);
        fetch_tile(a.h[1], a.ld_h[1], a.h_cols[1], t * RT, hr[1]);
    }
    for (; t < ntiles; t += gridDim.x) {
        const long long r0 = t * RT, rn = (t + gridDim.x) * RT;
        const bool more = t + gridDim.x < ntiles;
        __syncthreads();                      // the previous tile's copy-out has read Zs
        stash_tile(Zs, zr);
        stash_tile(Hs, hr[0]);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            __syncthreads();                  // images of layer l complete
            if (l + 2 < NL) fetch_tile(a.h[l + 2], a.ld_h[l + 2], a.h_cols[l + 2], r0, hr[l & 1]);
            else if (more) fetch_tile(a.h[l + 2 - NL], a.ld_h[l + 2 - NL], a.h_cols[l + 2 - NL], rn, hr[l & 1]);
            if (l == NL - 2 && more) fetch_tile(a.dh_in, a.ld_in, a.in_cols, rn, zr);
            // db_l: column sums of the (bf16) dh image
            {
                const int col = tid & 63, rg = tid >> 6;
                float s = 0.f;
#pragma unroll 8
                for (int rr = 0; rr < 32; ++rr) {
                    const int row = 32 * rg + rr;
                    const unsigned short v = *(const unsigned short*)(Zs + row * RB + 16 * ((col >> 3) ^ sw(row)) + 2 * (col & 7));
                    s += __builtin_bit_cast(float, (unsigned)v << 16);
                }
                dbs[l] += s;
            }
            // dW_l += dh^T h: k = the 128 rows of the tile
#if CHAIN64_PIN
            {   // two k-steps read ahead, pinned: hipcc otherwise sinks each pair of reads to its MFMA (lgkmcnt(0) per MFMA)
                constexpr int AH = 2;
                bf16x8 xq[AH], yq[AH];
#pragma unroll
                for (int u = 0; u < AH; ++u) { xq[u] = tr_frag(Zs, xb_o, u); yq[u] = tr_frag(Hs, yb_i, u); }
#pragma unroll
                for (int u = 0; u < RT / 16; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
                    dw[l] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xq[u % AH], yq[u % AH], dw[l], 0, 0, 0);
                    if (u + AH < RT / 16) { xq[u % AH] = tr_frag(Zs, xb_o, u + AH); yq[u % AH] = tr_frag(Hs, yb_i, u + AH); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int u = 0; u < RT / 16; ++u) {
                const bf16x8 xa = tr_frag(Zs, xb_o, u), yv = tr_frag(Hs, yb_i, u);
                dw[l] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, yv, dw[l], 0, 0, 0);
            }
#endif
            // dX: T[i][m] = sum_o W[o][i] dh[m][o]; wave: i block wr, m blocks wc2, wc2 + 32
            f32x16 dx[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[j][r] = 0.f;
#pragma unroll
            for (int u = 0; u < W / 16; ++u) {
                const bf16x8 xa = tr_frag(Ws + l * WIMG, xb_o, u);   // (the i block of the dX product is the same wr as the o block of dW)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wc2 + 32 * j + (lane & 31);
                    const bf16x8 yv = *(const bf16x8*)(Zs + row * RB + 16 * ((2 * u + h) ^ sw(row)));
                    dx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, yv, dx[j], 0, 0, 0);
                }
            }
            __syncthreads();                  // every wave has read Zs (dW, dX, db): it may be overwritten
            // epilogue: lane = image row m; registers 4q .. 4q+3 = columns i0 = wr + 8q + 4h .. + 3: mask by h > 0, round, store
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc2 + 32 * j + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = row * RB + 16 * (((wr >> 3) + q) ^ sw(row)) + 8 * h;
                    const uint2 m = *(const uint2*)(Hs + off);
                    // bf16 > 0  <=>  its 16 bits, as a signed integer, > 0
                    const float v0 = ((short)(m.x & 0xffffu) > 0) ? dx[j][4 * q] : 0.f, v1 = ((int)m.x > 0xffff) ? dx[j][4 * q + 1] : 0.f;
                    const float v2 = ((short)(m.y & 0xffffu) > 0) ? dx[j][4 * q + 2] : 0.f, v3 = ((int)m.y > 0xffff) ? dx[j][4 * q + 3] : 0.f;
                    *(uint2*)(Zs + off) = make_uint2(pk2(v0, v1), pk2(v2, v3));
                }
            }
            __syncthreads();                  // masks read, new dh image complete: Hs / Ws may take the next layer's operands
            if (l + 1 < NL) stash_tile(Hs, hr[(l + 1) & 1]);
        }
        // copy-out of dh_{L-NL}: the image rows as 16-byte pieces
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = srow + 32 * e;
            const long long r = r0 + row;
            if (r < a.M) *(uint4*)(a.dh_out + r * a.ld_out + 8 * sch) = *(const uint4*)(Zs + row * RB + 16 * (sch ^ sw(row)));
        }
    }
    // partial tiles: C/D map lane l register r -> row (o) (r&3) + 8(r>>2) + 4(l>>5), column (i) l & 31
    float* part = a.part + (long long)blockIdx.x * NL * PART;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = wr + (r & 3) + 8 * (r >> 2) + 4 * h;
            part[l * PART + o * W + wc + (lane & 31)] = dw[l][r];
        }
        __syncthreads();
        red[(tid >> 6) * W + (tid & 63)] = dbs[l];
        __syncthreads();
        if (tid < W) part[l * PART + W * W + tid] = red[tid] + red[W + tid] + red[2 * W + tid] + red[3 * W + tid];
    }
}

// ---- the two ends of the chain: everything the positional encoding takes part in ---------------------------------------------------
// With dh_a the gradient at the skip layer's pre-activation (layer 4) and dh_b at layer 0's, PE the (M, 64) bf16 encoding rows
// (column 63 zero) and Wa / Wb the bf16 [o][pe] blocks of those layers' weights:
//     dWa[:, :63] += dh_a^T PE      dWb[:, :63] += dh_b^T PE   (db_b += 1^T dh_b)      d_pe = dh_a Wa + dh_b Wb  (fp32, never stored)
//     d_xyz[m, c] = d_pe[m, c] + sum_k w_k 2^k (cos(2^k x) d_pe[m, sin_k c] - sin(2^k x) d_pe[m, cos_k c])     (nerf.py:35-75 backward)
// -- two dW GEMMs, two dX GEMMs (one accumulating), 67 MB of d_pe written and read again and the embedding backward of the
// per-layer route as one pass over dh_a, dh_b and PE.
struct PeEndsArgs {
    const unsigned short *dha, *dhb; long long ld_dh;   // (M, 64) bf16
    const unsigned short* pe; long long ld_pe;          // (M, 64) bf16
    const unsigned short *wa, *wb;                      // bf16 [64 o][64 pe], contiguous
    const float* xyz;                                   // (M, 3)
    float* d_xyz;                                       // (M, 3)
    float* part;                                        // [gridDim.x][2][PART]
    long long M;
    int n_freq;
    float win[16];
};

__global__ __launch_bounds__(256, 2) void pe_ends64_kernel(PeEndsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * IMG + 2 * WIMG + 4 * W * 4];
    unsigned char* Za = lds;                  // dh_a tile  } after the products: the fp32 d_pe tile [m][64] (32 KB) lies over both
    unsigned char* Zb = lds + IMG;            // dh_b tile  }
    unsigned char* Ps = lds + 2 * IMG;        // PE tile
    unsigned char* Ws = lds + 3 * IMG;        // Wa, Wb images [k = o][r = pe]
    float* red = (float*)(lds + 3 * IMG + 2 * WIMG);
    float* dpe = (float*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32, wc2 = (wave & 1) * 64;
    auto tr_base = [&](int cb) { const int ch = cb / 8 + 2 * (g & 1) + (p4 >> 1); return (8 * h + q4) * RB + 16 * (ch ^ sw(q4)) + 8 * (p4 & 1); };
    const int xb_o = tr_base(wr), yb_i = tr_base(wc);
    typedef s16x4 __attribute__((address_space(3))) lds_s16x4;
    auto tr_frag = [&](const unsigned char* base, int off, int u) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u) * RB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u + 4) * RB));
        union { struct { s16x4 a, b; } s; bf16x8 v; } o;
        o.s.a = lo;
        o.s.b = hi;
        return o.v;
    };
    const int sch = tid & 7, srow = tid >> 3;
    uint4 za[4], zb[4], pr[4];
    auto fetch_tile = [&](const unsigned short* src, long long ld, long long r0, uint4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long r = r0 + srow + 32 * e;
            dst[e] = make_uint4(0u, 0u, 0u, 0u);
            if (r < a.M) dst[e] = *(const uint4*)(src + r * ld + 8 * sch);
        }
    };
    auto stash_tile = [&](unsigned char* img, const uint4 (&src)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = srow + 32 * e;
            *(uint4*)(img + row * RB + 16 * (sch ^ sw(row))) = src[e];
        }
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = srow + 32 * e;
        *(uint4*)(Ws + row * RB + 16 * (sch ^ sw(row))) = *(const uint4*)(a.wa + row * W + 8 * sch);
        *(uint4*)(Ws + WIMG + row * RB + 16 * (sch ^ sw(row))) = *(const uint4*)(a.wb + row * W + 8 * sch);
    }
    f32x16 dwa, dwb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dwa[r] = 0.f; dwb[r] = 0.f; }
    float dbs = 0.f;

    const long long ntiles = (a.M + RT - 1) / RT;
    long long t = blockIdx.x;
    if (t < ntiles) {
        fetch_tile(a.dha, a.ld_dh, t * RT, za);
        fetch_tile(a.dhb, a.ld_dh, t * RT, zb);
        fetch_tile(a.pe, a.ld_pe, t * RT, pr);
    }
    for (; t < ntiles; t += gridDim.x) {
        const long long r0 = t * RT;
        __syncthreads();                      // the previous tile's d_pe has been read
        stash_tile(Za, za);
        stash_tile(Zb, zb);
        stash_tile(Ps, pr);
        __syncthreads();
        if (t + gridDim.x < ntiles) {
            fetch_tile(a.dha, a.ld_dh, (t + gridDim.x) * RT, za);
            fetch_tile(a.dhb, a.ld_dh, (t + gridDim.x) * RT, zb);
            fetch_tile(a.pe, a.ld_pe, (t + gridDim.x) * RT, pr);
        }
        {   // db_b: column sums of the dh_b image
            const int col = tid & 63, rg = tid >> 6;
            float s = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) {
                const int row = 32 * rg + rr;
                const unsigned short v = *(const unsigned short*)(Zb + row * RB + 16 * ((col >> 3) ^ sw(row)) + 2 * (col & 7));
                s += __builtin_bit_cast(float, (unsigned)v << 16);
            }
            dbs += s;
        }
#pragma unroll
        for (int u = 0; u < RT / 16; ++u) {
            const bf16x8 yv = tr_frag(Ps, yb_i, u);
            dwa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Za, xb_o, u), yv, dwa, 0, 0, 0);
            dwb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Zb, xb_o, u), yv, dwb, 0, 0, 0);
        }
        // d_pe: T[pe][m] = sum_o Wa[o][pe] dh_a[m][o] + Wb[o][pe] dh_b[m][o]; wave: pe block wr, m blocks wc2, wc2 + 32
        f32x16 dx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[j][r] = 0.f;
#pragma unroll
        for (int u = 0; u < W / 16; ++u) {
            const bf16x8 xa = tr_frag(Ws, xb_o, u), xb = tr_frag(Ws + WIMG, xb_o, u);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc2 + 32 * j + (lane & 31);
                const int off = row * RB + 16 * ((2 * u + h) ^ sw(row));
                dx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, *(const bf16x8*)(Za + off), dx[j], 0, 0, 0);
                dx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, *(const bf16x8*)(Zb + off), dx[j], 0, 0, 0);
            }
        }
        __syncthreads();                      // every wave has read Za / Zb: the fp32 d_pe tile takes their place
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wc2 + 32 * j + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q)       // registers 4q .. 4q+3 = PE columns wr + 8q + 4h .. + 3 of sample `row`
                *(float4*)(dpe + row * W + wr + 8 * q + 4 * h) = make_float4(dx[j][4 * q], dx[j][4 * q + 1], dx[j][4 * q + 2], dx[j][4 * q + 3]);
        }
        __syncthreads();
        // the embedding backward of the tile: one (sample, coordinate) per thread, 384 of them
        for (int i = tid; i < RT * 3; i += 256) {
            const int m = i / 3, c = i - 3 * m;
            if (r0 + m >= a.M) continue;
            const float* gm = dpe + m * W;
            const float x = a.xyz[(r0 + m) * 3 + c];
            float d = gm[c];
            for (int k = 0; k < a.n_freq; ++k) {
                float sn, cs;
                sincos_rr(ldexpf(x, k), sn, cs);
                d += ldexpf(a.win[k], k) * (cs * gm[3 + 6 * k + c] - sn * gm[6 + 6 * k + c]);
            }
            a.d_xyz[(r0 + m) * 3 + c] = d;
        }
    }
    float* part = a.part + (long long)blockIdx.x * 2 * PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = wr + (r & 3) + 8 * (r >> 2) + 4 * h;
        part[o * W + wc + (lane & 31)] = dwa[r];
        part[PART + o * W + wc + (lane & 31)] = dwb[r];
    }
    __syncthreads();
    red[(tid >> 6) * W + (tid & 63)] = dbs;
    __syncthreads();
    if (tid < W) {
        part[W * W + tid] = 0.f;
        part[PART + W * W + tid] = red[tid] + red[W + tid] + red[2 * W + tid] + red[3 * W + tid];
    }
}

struct ReduceArgs {
    const float* part; int nwg, n;
    float* gW[MAXL]; long long ldw[MAXL];     // gW[j] (64 x 64 block, leading dimension ldw[j]) += sum over workgroups
    float* gb[MAXL];                          // gb[j] (64) +=, or null
    int ncol[MAXL], nrow[MAXL];               // rows / columns of the 64-wide partial tile that exist in gW[j] (63 columns for the PE blocks)
};

// 64 elements x 4 phases of the workgroup list per block (the first version walked all 512 partials per thread: 30 us of latency)
__global__ __launch_bounds__(256) void chain64_reduce_kernel(ReduceArgs a) {
    __shared__ float red[4][64];
    const int l = blockIdx.y;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < PART) {
        const float* p = a.part + (long long)l * PART + e;
        const long long st = (long long)a.n * PART;
        int w = ph;
        for (; w + 12 < a.nwg; w += 16) {
            s0 += p[w * st]; s1 += p[(w + 4) * st]; s2 += p[(w + 8) * st]; s3 += p[(w + 12) * st];
        }
        for (; w < a.nwg; w += 4) s0 += p[w * st];
    }
    red[ph][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ph != 0 || e >= PART) return;
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (e < W * W) { if (e % W < a.ncol[l] && e / W < a.nrow[l]) a.gW[l][(long long)(e / W) * a.ldw[l] + (e % W)] += s; }
    else if (a.gb[l] != nullptr && e - W * W < a.nrow[l]) a.gb[l][e - W * W] += s;
}

}   // namespace

// dh_in (M, 64) bf16, ld_in; h[j] / wb[j] / gW[j] / ldw[j] / gb[j] for the n = 4 layers L, L-1, ...; dh_out (M, 64) bf16;
// part: moda_chain64_part_floats(M) floats of scratch.  Returns the launch status.
long long moda_chain64_part_floats(long long M) {
    const long long tiles = (M + RT - 1) / RT;
    const long long nwg = tiles < 512 ? tiles : 512;
    return nwg * MAXL * PART;
}

int moda_chain64_bwd(const void* dh_in, long long ld_in, const void* const* h, long long ld_h, const void* const* wb, void* dh_out,
                     long long ld_out, float* const* gW, const long long* ldw, float* const* gb, int n, long long M, float* part,
                     void* stream) {
    if (M <= 0 || n <= 0) return 0;
    if (n != 4 || !dh_in || !h || !wb || !dh_out || !gW || !ldw || !gb || !part || ld_in % 8 || ld_h % 8 || ld_out % 8) return MODA_EINVAL;
    ChainArgs a;
    a.dh_in = (const unsigned short*)dh_in; a.ld_in = ld_in; a.dh_out = (unsigned short*)dh_out; a.ld_out = ld_out;
    a.part = part; a.M = M; a.n = n;
    ReduceArgs r;
    a.in_cols = W;
    for (int j = 0; j < MAXL; ++j) {
        const int k = j < n ? j : 0;
        a.h[j] = (const unsigned short*)h[k]; a.wb[j] = (const unsigned short*)wb[k];
        a.ld_h[j] = ld_h; a.h_cols[j] = W; a.w_rows[j] = W; a.w_cols[j] = W; a.w_ld[j] = W;
        r.gW[j] = gW[k]; r.ldw[j] = ldw[k]; r.gb[j] = gb[k]; r.ncol[j] = W; r.nrow[j] = W;
        if (!a.h[j] || !a.wb[j] || !r.gW[j] || (((uintptr_t)a.h[j] | (uintptr_t)a.wb[j]) & 15)) return MODA_EINVAL;
    }
    if ((((uintptr_t)dh_in | (uintptr_t)dh_out) & 15)) return MODA_EINVAL;
    const long long tiles = (M + RT - 1) / RT;
    const int nwg = (int)(tiles < 512 ? tiles : 512);
    hipLaunchKernelGGL(chain64_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, a);
    r.part = part; r.nwg = nwg; r.n = n;
    hipLaunchKernelGGL(chain64_reduce_kernel, dim3((PART + 63) / 64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, r);
    return (int)hipGetLastError();
}

// The positional-encoding ends of a 64-wide network's backward (see pe_ends64_kernel): dha / dhb / pe (M, 64) bf16 with leading
// dimensions ld_dh / ld_pe; wa / wb bf16 [64][64]; gWa / gWb the fp32 gradient blocks (64 x 63 used, leading dimensions lda / ldb),
// gb_b (64) or NULL; d_xyz (M, 3) written; part: moda_chain64_part_floats(M) floats of scratch.
int moda_pe_ends64_bwd(const void* dha, const void* dhb, long long ld_dh, const void* pe, long long ld_pe, const void* wa, const void* wb,
                       const float* xyz, int n_freq, const float* window, float* gWa, long long lda, float* gWb, long long ldb,
                       float* gb_b, float* d_xyz, long long M, float* part, void* stream) {
    if (M <= 0) return 0;
    if (!dha || !dhb || !pe || !wa || !wb || !xyz || !gWa || !gWb || !d_xyz || !part || n_freq < 0 || n_freq > 10 || ld_dh % 8 || ld_pe % 8)
        return MODA_EINVAL;
    if ((((uintptr_t)dha | (uintptr_t)dhb | (uintptr_t)pe | (uintptr_t)wa | (uintptr_t)wb) & 15)) return MODA_EINVAL;
    PeEndsArgs a;
    a.dha = (const unsigned short*)dha; a.dhb = (const unsigned short*)dhb; a.ld_dh = ld_dh; a.pe = (const unsigned short*)pe; a.ld_pe = ld_pe;
    a.wa = (const unsigned short*)wa; a.wb = (const unsigned short*)wb; a.xyz = xyz; a.d_xyz = d_xyz; a.part = part; a.M = M; a.n_freq = n_freq;
    for (int i = 0; i < 16; ++i) a.win[i] = (i < n_freq && window) ? window[i] : 0.f;
    const long long tiles = (M + RT - 1) / RT;
    const int nwg = (int)(tiles < 512 ? tiles : 512);
    hipLaunchKernelGGL(pe_ends64_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, a);
    ReduceArgs r;
    r.part = part; r.nwg = nwg; r.n = 2;
    for (int j = 0; j < MAXL; ++j) { r.gW[j] = j == 0 ? gWa : gWb; r.ldw[j] = j == 0 ? lda : ldb; r.gb[j] = j == 1 ? gb_b : nullptr; r.ncol[j] = W - 1; r.nrow[j] = W; }
    hipLaunchKernelGGL(chain64_reduce_kernel, dim3((PART + 63) / 64, 2), dim3(256), 0, (hipStream_t)stream, r);
    return (int)hipGetLastError();
}

// The heads of a 64-wide raw_feat network without a direction input, as a two-layer chain of the same kernel (operands padded to
// 64 columns as they are staged):   dz_rgb (M, 32) bf16 [columns >= n_out zero]  ->  dzd = (dz_rgb Wrgb) (.) [dd > 0]  ->
// dh = (dzd Wext) (.) [hD > 0]   with   g_rgb (n_out x 32) += dz_rgb^T dd,  g_brgb += 1^T dz_rgb,  Tm (32 x 64) += dzd^T hD,
// svec (32) += 1^T dzd  (the products the folded heads of moda_nerf_train_bwd need, see head_finish_kernel there).
// wrgb: bf16 [32][32] (rows >= n_out zero); wext: bf16 [>= 32][64]; dd (M, 32) bf16, hD (M, 64) bf16; dh (M, 64) bf16 out.
int moda_heads64_bwd(const void* dzb, long long ld_dzb, const void* dd, long long ld_dd, const void* hD, long long ld_hD, const void* wrgb,
                     const void* wext, void* dh, long long ld_dh, float* g_rgb, long long ld_grgb, float* g_brgb, int n_out, float* Tm,
                     float* svec, long long M, float* part, void* stream) {
    if (M <= 0) return 0;
    if (!dzb || !dd || !hD || !wrgb || !wext || !dh || !g_rgb || !Tm || !svec || !part || n_out < 1 || n_out > 32) return MODA_EINVAL;
    if (ld_dzb % 8 || ld_dd % 8 || ld_hD % 8 || ld_dh % 8) return MODA_EINVAL;
    if ((((uintptr_t)dzb | (uintptr_t)dd | (uintptr_t)hD | (uintptr_t)wrgb | (uintptr_t)wext | (uintptr_t)dh) & 15)) return MODA_EINVAL;
    ChainArgs a;
    a.dh_in = (const unsigned short*)dzb; a.ld_in = ld_dzb; a.in_cols = 32;
    a.dh_out = (unsigned short*)dh; a.ld_out = ld_dh; a.part = part; a.M = M; a.n = 2;
    for (int j = 0; j < MAXL; ++j) {
        const bool first = (j & 1) == 0;
        a.h[j] = (const unsigned short*)(first ? dd : hD); a.ld_h[j] = first ? ld_dd : ld_hD; a.h_cols[j] = first ? 32 : W;
        a.wb[j] = (const unsigned short*)(first ? wrgb : wext); a.w_rows[j] = 32; a.w_cols[j] = first ? 32 : W; a.w_ld[j] = first ? 32 : W;
    }
    const long long tiles = (M + RT - 1) / RT;
    const int nwg = (int)(tiles < 512 ? tiles : 512);
    hipLaunchKernelGGL(chain64_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, a);
    ReduceArgs r;
    r.part = part; r.nwg = nwg; r.n = 2;
    for (int j = 0; j < MAXL; ++j) {
        const bool first = (j & 1) == 0;
        r.gW[j] = first ? g_rgb : Tm; r.ldw[j] = first ? ld_grgb : W; r.gb[j] = first ? g_brgb : svec;
        r.ncol[j] = first ? 32 : W; r.nrow[j] = first ? n_out : 32;
    }
    hipLaunchKernelGGL(chain64_reduce_kernel, dim3((PART + 63) / 64, 2), dim3(256), 0, (hipStream_t)stream, r);
    return (int)hipGetLastError();
}
