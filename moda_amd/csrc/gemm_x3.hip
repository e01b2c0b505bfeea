// The three large GEMM forms of a Linear layer in the split-bf16 training modes, gfx950: every operand is fp32 in memory and
// leaves the staging registers as NS bf16 LDS images, each the bf16 rounding of what the previous ones left behind:
//   NS = 2 ("bf16x3"): hi + lo, 16 significand bits; a product is three v_mfma_f32_32x32x16_bf16 -- lo*hi + hi*lo + hi*hi, small
//          terms first, fp32 sums.  Operand error 2^-17 (2^-9 in the bf16 mode): results within ~1e-6 of the exact-fp32 GEMM.
//   NS = 3 ("bf16x6"): hi + mid + lo = the fp32 value EXACTLY (3 x 8 significand bits); six MFMAs keep every product term down to
//          2^-18 (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi), the dropped ones are <= 2^-26 of the product: the accuracy class
//          of the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) at 16/6 of its matrix rate -- and these forms are bound by HBM anyway.
//
// Reference: nn.Linear forward and its autograd backward in NeRF.forward (nnutils/nerf.py:147-198) --
//   forward  h' = act(h W^T + b)                  X = W as it lies ([o][i]: k-FAST), Y = h rows (k-fast), T^T stored
//   dX       dh = (dZ W) (.) [h > 0]              X = W as it lies ([o][i] = [k][r]: k-SLOW), Y = dZ rows (k-fast), T^T stored
//   dW       dW += dZ^T h,  db += 1^T dZ          X = dZ ([k][r], k-slow), Y = h ([k][c], k-slow), atomics (split over k)
// All three are thin (K or N is the layer width, M = rays x samples is long): bound by streaming the long operands through
// HBM once.  The generic kernel (train_kernels.hip gemm2: fp32 LDS image, operands split as fragments are read, eight scalar
// LDS reads per fragment of an operand whose k is the slow index) reaches 1.8-2.7 TB/s on them; here the split is done ONCE per
// element at staging time and every MFMA operand is one or two LDS reads, the design of gemm_bf16.hip:
//   * k-slow operand: image [k][row], 16-byte chunks swizzled, read with ds_read_b64_tr_b16 (the transposing read);
//   * k-fast operand: image [row][k] with padded rows, read with ds_read_b128.
// A k-tile is 32 deep (the same bytes in flight and the same LDS footprint as the 64-deep bf16 tiles of gemm_bf16.hip).
// moda_gemm_f32_ex routes a MODA_GEMM_BF16X3 call here when strides and alignment fit (moda_x3_try); the rest stays on gemm2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 32;                    // k per tile: two 32x32x16 steps
#ifndef MODA_X3_PF
#define MODA_X3_PF 1
#endif
constexpr int PF = MODA_X3_PF;            // k-tiles of global loads in flight per engine (1 or 2)
static_assert(PF == 1 || PF == 2, "one or two register stages");
constexpr int KF_STRIDE = KT * 2 + 16;    // bytes per row of a k-fast image (80: sixteen consecutive rows cover all 64 banks)
enum { EPI_ATOMIC = 0, EPI_T = 1 };

struct X3Args {
    const float* X; long long ldx;        // k-slow: (k, r) at k * ldx + r;  k-fast: (r, k) at r * ldx + k
    const float* Y; long long ldy;        // k-slow: (k, c) at k * ldy + c;  k-fast: (c, k) at c * ldy + k
    float* C; long long ldc;              // EPI_ATOMIC: (r, c) at r * ldc + c;  EPI_T: (c, r) at c * ldc + r
    const float* mask; long long ldm;     // EPI_T: (c, r) at c * ldm + r; result zeroed where mask <= 0; or null
    const float* bias;                    // EPI_T: bias[r] added; or null
    float* xsum;                          // EPI_ATOMIC: xsum[r] += sum_k X(k, r), or null
    unsigned char* bits_out;              // EPI_ATOMIC: sign map of Y, bit (c & 7) of byte k * ldb + c / 8 = [Y(k, c) > 0]; or null
    const unsigned char* bits_in;         // EPI_T: such a map as the mask, (c, r) at byte c * ldb + r / 8; or null
    long long ldb;
    int R, Cn, K;
    int splits;                           // EPI_ATOMIC: slices over k (slice s takes the k-tiles s, s + splits, ...); else 1
    int accumulate;                       // EPI_T: 2 = C += T
    int relu;                             // EPI_T: max(., 0) after the bias
    unsigned gr, gc;                      // tiles along r / c
};

// chunk swizzle of a [k][T] bf16 image (rows of 2T bytes, 16-byte chunks): the four rows one transposing read takes
// (4n .. 4n+3, 64 bytes each per 32-lane half) land in four different 64-byte bank groups
template <int T>
DEVINL int ks_sw(int row) { return T == 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }

DEVINL unsigned pk2(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}
DEVINL float bflo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
DEVINL float bfhi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// eight fp32 values -> NS bf16 images: the rounding of the value, then of what each rounding left behind (the differences are
// exact in fp32: a bf16 keeps the top 8 significand bits of its argument)
template <int NS>
DEVINL void split8(float4 a, float4 b, uint4 (&o)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint4 w = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
        o[s] = w;
        if (s + 1 < NS) {
            a.x -= bflo(w.x); a.y -= bfhi(w.x); a.z -= bflo(w.y); a.w -= bfhi(w.y);
            b.x -= bflo(w.z); b.y -= bfhi(w.z); b.z -= bflo(w.w); b.w -= bfhi(w.w);
        }
    }
}

// NG (dW form): wave groups per workgroup, each a complete 2 x 2 tile engine with its own LDS stage and its own k-tiles; the
// groups' accumulators are summed through LDS before ONE set of atomics leaves the workgroup (gemm_bf16.hip has the numbers).
// (Round 4 tried NG = 2 on the row forms as a PING-PONG: two wave groups on adjacent c tiles, one barrier apart, so that on every
// SIMD one wave stages while the other multiplies.  Correct, and 8-30 % SLOWER than two independent workgroups per CU (256-wide
// bf16x6 forward 326 vs 302 us, masked dX 401 vs 306 us on random operands): the workgroup-wide barrier ties each group to the
// slower phase of the other.  PMC of these forms: MFMA busy 38 %, VALU 28 %, waves stalled in s_waitcnt 46 % of their cycles (8 % of them on LDS).)
template <int NS, int TR, int TC, bool XKF, bool YKF, int EPI, int NG = 1>
__global__ __launch_bounds__(256 * NG, NG == 1 ? (NS == 2 && PF == 1 ? 3 : 2) : 1) void gemm_x3_kernel(X3Args a) {
    static_assert(NS == 2 || NS == 3, "two or three bf16 images per operand");
    static_assert(NG == 1 || EPI == EPI_ATOMIC, "wave groups split k: the dW form");
    static_assert(!XKF || EPI == EPI_T, "a k-fast X is the weight of the forward form");
    constexpr int RBX = XKF ? KF_STRIDE : TR * 2;                 // bytes per row of an X image
    constexpr int RBY = YKF ? KF_STRIDE : TC * 2;
    constexpr int XS_BYTES = XKF ? TR * KF_STRIDE : KT * RBX;     // one image (hi or lo)
    constexpr int YS_BYTES = YKF ? TC * KF_STRIDE : KT * RBY;
    constexpr int IMG_BYTES = XS_BYTES + YS_BYTES;                // image s of an operand lies s * IMG_BYTES after its first
    constexpr int WR = TR / 2, WC = TC / 2, NI = WR / 32, NJ = WC / 32;
    constexpr int EPI_ROW = WR * 4 + 16;                          // bytes per row of a wave's transpose buffer
    constexpr int EPI_BYTES = (EPI == EPI_ATOMIC) ? 256 * 8 * 4 : 4 * 32 * EPI_ROW;
    constexpr int STAGE_BYTES = (NS * IMG_BYTES > EPI_BYTES) ? NS * IMG_BYTES : EPI_BYTES;
    constexpr int NREG = NI * NJ * 16;                            // accumulator registers per lane
    static_assert(NG == 1 || (NG - 1) * NREG * 256 * 4 <= NG * STAGE_BYTES, "the partial tiles are summed through the stage LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NG * STAGE_BYTES];
    const int grp = NG == 1 ? 0 : (int)(threadIdx.x >> 8);
    unsigned char* Xs = lds + grp * STAGE_BYTES;
    unsigned char* Ys = Xs + XS_BYTES;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int wr = (wave >> 1) * WR, wc = (wave & 1) * WC;
    // tiles that read the same rows of a long operand get dispatch ids 8 apart: one XCD, one L2 (workgroups are dealt
    // round-robin over the 8 XCDs, each with its own L2: what one XCD has fetched, another fetches again)
    const unsigned lid = blockIdx.x;
    unsigned tr_idx, tc_idx, slice = 0;
    if (EPI == EPI_ATOMIC) {
        const unsigned nt = a.gr * a.gc;
        unsigned t = lid % nt;
        slice = lid / nt;
        if ((a.splits & 7) == 0) {
            t = (lid >> 3) % nt;
            slice = (lid / (8 * nt)) * 8 + (lid & 7);
        }
        tr_idx = t % a.gr;
        tc_idx = t / a.gr;
    } else {
        tr_idx = lid % a.gr;
        tc_idx = lid / a.gr;
        if (a.gr > 1 && (a.gc & 7) == 0) {
            const unsigned span = 8 * a.gr, r = lid % span;
            tr_idx = r >> 3;
            tc_idx = (lid / span) * 8 + (r & 7);
        }
    }
    const long long r0 = (long long)tr_idx * TR, c0 = (long long)tc_idx * TC;
    // split over k: engine e of E = splits x NG (slice s, group g: e = s NG + g) takes the k-tiles e, e + E, e + 2E, ...
    const int kstep = KT * a.splits * NG;
    const int kbeg = ((int)slice * NG + grp) * KT;
    const int kend = a.K;
    const int nit = (kend - (int)slice * NG * KT + kstep - 1) / kstep;    // trips of group 0, the longest: every group runs as many

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging: a thread takes eight consecutive fp32 (two 16-byte loads) = one 16-byte chunk of each bf16 image ------------
    //      k-slow: (row k, chunk of eight tile columns);  k-fast: (tile row, chunk of eight k: KT / 8 = 4 chunks per row)
    constexpr int XCPR = XKF ? KT / 8 : TR / 8, XRPP = 256 / XCPR, XNP = (XKF ? TR : KT) / XRPP;
    const int xch = tid % XCPR, xrow = tid / XCPR;
    float4 xf[PF][XNP][2];          // PF register stages: k-tile t + PF is in flight while tile t is multiplied
    constexpr int YCPR = YKF ? KT / 8 : TC / 8, YRPP = 256 / YCPR, YNP = (YKF ? TC : KT) / YRPP;
    const int ych = tid % YCPR, yrow = tid / YCPR;
    float4 yf[PF][YNP][2];
    const bool do_xsum = (EPI == EPI_ATOMIC) && a.xsum != nullptr && tc_idx == 0;
    float xsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto fetch = [&](auto bsel, int k0) __attribute__((always_inline)) {
        constexpr int B = decltype(bsel)::value;
#pragma unroll
        for (int e = 0; e < XNP; ++e) {
            xf[B][e][0] = xf[B][e][1] = z4;
            if (XKF) {
                const long long r = r0 + xrow + XRPP * e, k = k0 + 8 * xch;
                if (r < a.R && k < kend) {
                    const float* p = a.X + r * a.ldx + k;
                    xf[B][e][0] = *(const float4*)p;
                    xf[B][e][1] = *(const float4*)(p + 4);
                }
            } else {
                const long long k = k0 + xrow + XRPP * e, col = r0 + 8 * xch;
                if (k < kend && col < a.R) {
                    const float* p = a.X + k * a.ldx + col;
                    xf[B][e][0] = *(const float4*)p;
                    xf[B][e][1] = *(const float4*)(p + 4);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < YNP; ++e) {
            yf[B][e][0] = yf[B][e][1] = z4;
            if (YKF) {
                const long long c = c0 + yrow + YRPP * e, k = k0 + 8 * ych;
                if (c < a.Cn && k < kend) {
                    const float* p = a.Y + c * a.ldy + k;
                    yf[B][e][0] = *(const float4*)p;
                    yf[B][e][1] = *(const float4*)(p + 4);
                }
            } else {
                const long long k = k0 + yrow + YRPP * e, col = c0 + 8 * ych;
                if (k < kend && col < a.Cn) {
                    const float* p = a.Y + k * a.ldy + col;
                    yf[B][e][0] = *(const float4*)p;
                    yf[B][e][1] = *(const float4*)(p + 4);
                }
            }
        }
    };
    // (the dW form leaves the sign map of its Y operand -- the saved activations h -- behind for the dX launch that follows: one
    //  byte per staged chunk of eight columns, written by the workgroups of the first tile row only; 1 bit instead of 32 per
    //  element of the mask the dX epilogue reads)
    const bool do_bits = (EPI == EPI_ATOMIC) && !YKF && a.bits_out != nullptr && tr_idx == 0;
    auto stash = [&](auto bsel, int k0) __attribute__((always_inline)) {
        constexpr int B = decltype(bsel)::value;
#pragma unroll
        for (int e = 0; e < XNP; ++e) {
            const int row = xrow + XRPP * e;
            uint4 im[NS];
            split8<NS>(xf[B][e][0], xf[B][e][1], im);
            unsigned char* p = XKF ? Xs + row * KF_STRIDE + 16 * xch : Xs + row * RBX + 16 * (xch ^ ks_sw<TR>(row));
#pragma unroll
            for (int s = 0; s < NS; ++s) *(uint4*)(p + s * IMG_BYTES) = im[s];
            if (do_xsum) {
                xsum8[0] += xf[B][e][0].x; xsum8[1] += xf[B][e][0].y; xsum8[2] += xf[B][e][0].z; xsum8[3] += xf[B][e][0].w;
                xsum8[4] += xf[B][e][1].x; xsum8[5] += xf[B][e][1].y; xsum8[6] += xf[B][e][1].z; xsum8[7] += xf[B][e][1].w;
            }
        }
#pragma unroll
        for (int e = 0; e < YNP; ++e) {
            const int row = yrow + YRPP * e;
            uint4 im[NS];
            if (do_bits) {
                const long long k = (long long)k0 + row, col = c0 + 8 * ych;
                if (k < kend && col < a.Cn) {
                    const float4 u = yf[B][e][0], v = yf[B][e][1];
                    const unsigned b = (u.x > 0.f ? 1u : 0u) | (u.y > 0.f ? 2u : 0u) | (u.z > 0.f ? 4u : 0u) | (u.w > 0.f ? 8u : 0u) |
                                       (v.x > 0.f ? 16u : 0u) | (v.y > 0.f ? 32u : 0u) | (v.z > 0.f ? 64u : 0u) | (v.w > 0.f ? 128u : 0u);
                    a.bits_out[k * a.ldb + (col >> 3)] = (unsigned char)b;
                }
            }
            split8<NS>(yf[B][e][0], yf[B][e][1], im);
            unsigned char* p = YKF ? Ys + row * KF_STRIDE + 16 * ych : Ys + row * RBY + 16 * (ych ^ ks_sw<TC>(row));
#pragma unroll
            for (int s = 0; s < NS; ++s) *(uint4*)(p + s * IMG_BYTES) = im[s];
        }
    };

    // ---- per-lane bases of the fragment reads.  Transposing read of a [k][T] image, 16-lane group gq = g & 1 of lane half
    //      h: lane 4q+p supplies row (16u + 8h + 4e) + q, elements 16 gq + 4p .. +3 of the 32-wide tile; it receives, for
    //      its own column (lane & 31), the k = 8h + 4e + (0..3) of k-step u -- elements 4e .. 4e+3 of the MFMA operand.
    //      k-fast image: the lane reads its own row's eight k = 16u + 8h .. +7 with one 16-byte read.
    int xb[NI], yb[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (XKF) {
            xb[i] = (wr + 32 * i + (lane & 31)) * KF_STRIDE + 16 * h;
        } else {
            const int ch = (wr + 32 * i) / 8 + 2 * (g & 1) + (p4 >> 1);
            xb[i] = (8 * h + q4) * RBX + 16 * (ch ^ ks_sw<TR>(q4)) + 8 * (p4 & 1);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (YKF) {
            yb[j] = (wc + 32 * j + (lane & 31)) * KF_STRIDE + 16 * h;
        } else {
            const int ch = (wc + 32 * j) / 8 + 2 * (g & 1) + (p4 >> 1);
            yb[j] = (8 * h + q4) * RBY + 16 * (ch ^ ks_sw<TC>(q4)) + 8 * (p4 & 1);
        }
    }
    typedef s16x4 __attribute__((address_space(3))) lds_s16x4;
    auto tr_frag = [&](const unsigned char* base, int off, int rb, int u) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u) * rb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + (16 * u + 4) * rb));
        union { struct { s16x4 a, b; } s; bf16x8 v; } o;
        o.s.a = lo;
        o.s.b = hi;
        return o.v;
    };

    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KT / 16; ++u) {
            bf16x8 xv[NS][NI], yv[NS][NJ];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (XKF) xv[s][i] = *(const bf16x8*)(Xs + s * IMG_BYTES + xb[i] + 32 * u);
                    else xv[s][i] = tr_frag(Xs + s * IMG_BYTES, xb[i], RBX, u);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (YKF) yv[s][j] = *(const bf16x8*)(Ys + s * IMG_BYTES + yb[j] + 32 * u);
                    else yv[s][j] = tr_frag(Ys + s * IMG_BYTES, yb[j], RBY, u);
                }
            }
            // product terms, smallest first (image 0 = hi, 1 = the next 8 bits, 2 = the last 8)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (NS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[1][i], yv[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[0][i], yv[NS - 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[NS - 1][i], yv[0][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[0][i], yv[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[1][i], yv[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[0][i], yv[0][j], acc[i][j], 0, 0, 0);
                }
        }
    };
    // k-tile t of this engine is staged (registers -> split -> LDS images) one barrier before it is multiplied; the loads of tile
    // t + PF are issued right after that barrier.  PF = 2 (a load has two tiles' time to land) measured the same as PF = 1 within
    // the box-to-box spread (256-wide forms 3-10 % faster, 64-wide 5-10 % slower, the training step +-2 %): these kernels are not
    // waiting for memory latency but adding up their phases (split VALU + LDS writes | barrier | LDS reads + MFMA | barrier)
    // with only two workgroups per CU to overlap them.  PF = 1 keeps 40-50 registers fewer.
    if (nit > 0) fetch(std::integral_constant<int, 0>{}, kbeg);
    if (PF > 1 && nit > 1) fetch(std::integral_constant<int, PF - 1>{}, kbeg + kstep);
    for (int it = 0; it < nit; it += PF) {      // (a k-tile past the end loads zeros)
        {
            stash(std::integral_constant<int, 0>{}, kbeg + it * kstep);
            __syncthreads();
            if (it + PF < nit) fetch(std::integral_constant<int, 0>{}, kbeg + (it + PF) * kstep);
            multiply();
            __syncthreads();
        }
        if (PF > 1 && it + 1 < nit) {
            stash(std::integral_constant<int, PF - 1>{}, kbeg + (it + 1) * kstep);
            __syncthreads();
            if (it + 1 + PF < nit) fetch(std::integral_constant<int, PF - 1>{}, kbeg + (it + 1 + PF) * kstep);
            multiply();
            __syncthreads();
        }
    }

    // C/D map: lane l register r -> T row (r&3) + 8(r>>2) + 4(l>>5), T column l & 31
    if (EPI == EPI_ATOMIC) {
        if (NG > 1) {       // sum the groups' partial tiles into group 0 (lane-linear fp32 images in the now idle stage LDS)
            float* part = (float*)lds;
            if (grp > 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[(((grp - 1) * NREG) + (i * NJ + j) * 16 + r) * 256 + tid] = acc[i][j][r];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int g2 = 1; g2 < NG; ++g2)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] += part[(((g2 - 1) * NREG) + (i * NJ + j) * 16 + r) * 256 + tid];
            }
            if (do_xsum) {
                __syncthreads();
                float* red = (float*)lds;
#pragma unroll
                for (int e = 0; e < 8; ++e) red[((grp * 256) + xrow * XCPR + xch) * 8 + e] = xsum8[e];
                __syncthreads();
                if (grp == 0 && tid < TR) {
                    float sm = 0.f;
                    for (int rw = 0; rw < NG * XRPP; ++rw) sm += red[(rw * XCPR + (tid >> 3)) * 8 + (tid & 7)];
                    if (r0 + tid < a.R) atomicAdd(a.xsum + r0 + tid, sm);
                }
            }
            if (grp != 0) return;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const long long c = c0 + wc + 32 * j + (lane & 31);
                if (c >= a.Cn) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long rr = r0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (rr < a.R) atomicAdd(a.C + rr * a.ldc + c, acc[i][j][r]);
                }
            }
        if (NG == 1 && do_xsum) {      // threads with the same chunk column hold partial sums of the same eight r
            float* red = (float*)lds;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(xrow * XCPR + xch) * 8 + e] = xsum8[e];
            __syncthreads();
            if (tid < TR) {
                float s = 0.f;
                for (int rw = 0; rw < XRPP; ++rw) s += red[(rw * XCPR + (tid >> 3)) * 8 + (tid & 7)];
                if (r0 + tid < a.R) atomicAdd(a.xsum + r0 + tid, s);
            }
        }
        return;
    }
    // ---- transposed store: the wave writes one 32-column (c) slab of its tile at a time into its LDS buffer as [c][r] rows,
    //      then every lane takes 16 bytes (four r) of one row: coalesced bias / mask / C reads and C stores -------------------
    unsigned char* tb = lds + wave * (32 * EPI_ROW);
    constexpr int NCH = WR * 4 / 16;                          // 16-byte pieces per row
    constexpr int ROWS_PP = 64 / NCH;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(tb + (lane & 31) * EPI_ROW + (32 * i + 8 * q + 4 * h) * 4) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 32 / ROWS_PP; ++ps) {
            const int row = lane / NCH + ROWS_PP * ps, piece = lane % NCH;
            const long long c = c0 + wc + 32 * j + row;
            const long long rr = r0 + wr + piece * 4;
            float4 v = *(const float4*)(tb + row * EPI_ROW + 16 * piece);
            if (c >= a.Cn || rr >= a.R) continue;
            if (a.bias != nullptr) {
                const float4 b = *(const float4*)(a.bias + rr);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* cp = a.C + c * a.ldc + rr;
            if (a.accumulate == 2) {
                const float4 o = *(const float4*)cp;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.bits_in != nullptr) {
                const unsigned nib = (unsigned)a.bits_in[c * a.ldb + (rr >> 3)] >> (rr & 4);
                if (!(nib & 1u)) v.x = 0.f;
                if (!(nib & 2u)) v.y = 0.f;
                if (!(nib & 4u)) v.z = 0.f;
                if (!(nib & 8u)) v.w = 0.f;
            }
            if (a.mask != nullptr) {
                const float4 m = *(const float4*)(a.mask + c * a.ldm + rr);
                if (!(m.x > 0.f)) v.x = 0.f;
                if (!(m.y > 0.f)) v.y = 0.f;
                if (!(m.z > 0.f)) v.z = 0.f;
                if (!(m.w > 0.f)) v.w = 0.f;
            }
            *(float4*)cp = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int TR, int TC, bool XKF, bool YKF, int EPI, int NG = 1>
int x3_launch(const X3Args& a, int ns, unsigned splits, hipStream_t st) {
    const dim3 grid(a.gr * a.gc * splits), block(256 * NG);
    if (ns == 3) hipLaunchKernelGGL((gemm_x3_kernel<3, TR, TC, XKF, YKF, EPI, NG>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_x3_kernel<2, TR, TC, XKF, YKF, EPI, NG>), grid, block, 0, st, a);
    return (int)hipGetLastError();
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}   // namespace

// Route of moda_gemm_f32_ex for MODA_GEMM_BF16X3 / MODA_GEMM_BF16X6 calls (ns = 2 / 3 images per operand): returns true (and the launch status in *rc) when the call is one of the
// three forms above with operands these kernels can take as they lie; false leaves it to the generic kernel (same arithmetic:
// the same split and product terms in the same order inside a 16-deep step; the summation order over k differs).
bool moda_x3_try(const moda_gemm_desc* d, int ns, void* stream, int* rc) {
    static const bool off = [] { const char* e = getenv("MODA_GEMM_X3"); return e && e[0] == '0'; }();
    if (off) return false;
    if (d->A2 || d->rowbias || d->K <= 0) return false;
    const int64_t lim = 0x7fffffff;
    if (d->M > lim || d->N > lim || d->K > lim) return false;
    hipStream_t st = (hipStream_t)stream;
    X3Args a;
    a.mask = nullptr; a.ldm = 0; a.bias = nullptr; a.xsum = nullptr; a.accumulate = 0; a.relu = 0; a.splits = 1;
    a.bits_out = nullptr; a.bits_in = nullptr; a.ldb = d->ld_bits;
    if (d->sam == 1 && d->sak != 1) {
        // ---- dW form: A(m, k) = dZ[k * sak + m] (m-fast), B(k, n) = X[k * sbk + n] (n-fast); C += A B with atomics -----------
        if (d->sbn != 1 || d->accumulate != 1 || d->mask_src || d->bias || d->act != 0) return false;
        const int64_t R = d->M, Cn = d->N, K = d->K;
        if (!al16(d->A) || d->sak % 4 || (R + 7) / 8 * 8 > d->sak) return false;
        if (!al16(d->B) || d->sbk % 4 || (Cn + 7) / 8 * 8 > d->sbk) return false;
        a.X = d->A; a.ldx = d->sak; a.Y = d->B; a.ldy = d->sbk; a.C = d->C; a.ldc = d->ldc;
        a.R = (int)R; a.Cn = (int)Cn; a.K = (int)K; a.xsum = d->a_sum;
        if (d->mask_bits && d->ld_bits * 8 < Cn) return false;
        a.bits_out = (unsigned char*)d->mask_bits;       // written: the sign map of B (moda_hip.h, moda_gemm_desc.mask_bits)
        const bool big = R > 64 || Cn > 64;                            // 128 x 128 tiles, or 64 x 64 for the 64-wide nets
        const int T = big ? 128 : 64;
        a.gr = (unsigned)((R + T - 1) / T); a.gc = (unsigned)((Cn + T - 1) / T);
        // split over k: one workgroup per CU (NG wave groups each), every engine with at least two k-tiles
        static const long long target_env = [] { const char* e = getenv("MODA_GEMM_X3_BLOCKS"); return e ? atoll(e) : 0LL; }();
        const long long target = target_env > 0 ? target_env : (big ? 256 : 128);
        const int ng = big ? 2 : 4;
        const long long nkt = (K + KT - 1) / KT;
        long long splits = target / ((long long)a.gr * a.gc);
        if (splits > nkt / (2 * ng)) splits = nkt / (2 * ng);
        if (splits < 1) splits = 1;
        if (splits >= 8) splits &= ~7LL;              // whole groups of 8 slices: the XCD placement in the kernel
        a.splits = (int)splits;
        if (big) *rc = x3_launch<128, 128, false, false, EPI_ATOMIC, 2>(a, ns, (unsigned)splits, st);
        else *rc = x3_launch<64, 64, false, false, EPI_ATOMIC, 4>(a, ns, (unsigned)splits, st);
        return true;
    }
    if (d->sak == 1) {
        // ---- row forms: A(m, k) = rows of h or dZ (k-fast); B(k, n) = the weight, n-fast (dX: W as it lies) or k-fast (forward:
        //      W as it lies); C(m, n) row-major = act(A B + bias) (.) [mask > 0] (+ C) ------------------------------------------
        if (d->a_sum || (d->accumulate != 0 && d->accumulate != 2) || d->split_k > 1 || d->act > 1) return false;
        const int64_t N = d->N, K = d->K, M = d->M;
        if (K % 8 || N % 4) return false;
        if (!al16(d->A) || d->sam % 4) return false;
        if (!al16(d->C) || d->ldc % 4) return false;
        if (d->mask_src && (!al16(d->mask_src) || d->ld_mask % 4)) return false;
        if (d->mask_bits && (d->mask_src || d->ld_bits * 8 < N || N % 8)) return false;
        if (d->bias && !al16(d->bias)) return false;
        const bool xkf = d->sbk == 1 && d->sbn != 1;
        if (xkf) {
            if (!al16(d->B) || d->sbn % 4) return false;
        } else {
            if (d->sbn != 1 || !al16(d->B) || d->sbk % 4 || (N + 7) / 8 * 8 > d->sbk) return false;
        }
        a.X = d->B; a.ldx = xkf ? d->sbn : d->sbk; a.Y = d->A; a.ldy = d->sam; a.C = d->C; a.ldc = d->ldc;
        a.mask = d->mask_src; a.ldm = d->ld_mask; a.bias = d->bias; a.relu = d->act == 1;
        a.bits_in = (const unsigned char*)d->mask_bits;
        a.R = (int)N; a.Cn = (int)M; a.K = (int)K; a.accumulate = d->accumulate;
        a.gc = (unsigned)((M + 127) / 128);
        if (N > 64) {
            a.gr = (unsigned)((N + 127) / 128);
            *rc = xkf ? x3_launch<128, 128, true, true, EPI_T>(a, ns, 1, st) : x3_launch<128, 128, false, true, EPI_T>(a, ns, 1, st);
        } else {
            a.gr = 1;
            *rc = xkf ? x3_launch<64, 128, true, true, EPI_T>(a, ns, 1, st) : x3_launch<64, 128, false, true, EPI_T>(a, ns, 1, st);
        }
        return true;
    }
    return false;
}
