// Training-path kernels (exact fp32): a strided MFMA GEMM with fused epilogues, column sums, and the backward of
// the positional encoding.  The reference trains in fp32 through torch autograd (nn.Linear, ReLU, sigmoid:
// nnutils/nerf.py:147-198); these kernels are what moda_amd's autograd Functions call instead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DEVINL __device__ __forceinline__

constexpr int BM = 128, BK = 16;

struct GemmArgs {
    const float* A; long long sam, sak;   // A(m,k) = A[m*sam + k*sak]
    const float* B; long long sbk, sbn;   // B(k,n) = B[k*sbk + n*sbn]
    float* C; long long ldc;
    const float* bias;       // per column n, or null
    const float* mask_src;   // same layout as C; when given the result is zeroed where mask_src <= 0 (ReLU backward)
    int M, N, K;
    int act;                 // 0 none, 1 relu, 2 sigmoid
    int accumulate;          // 0: C = result; 1: C += result (atomic; used with split-K)
    int ksplit;              // K range per blockIdx.z
};

// C tile 128xBN per workgroup (BN = 128: 4 waves as 2x2 of 64x64; BN = 64: 4 waves stacked, 32x64 each);
// v_mfma_f32_32x32x2_f32 tiles.
template <int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
    constexpr int TM = (BN == 128) ? 2 : 1;   // 32-row MFMA tiles per wave
    __shared__ float As[BK][BM + 4];   // k-major: MFMA A operand lane l reads As[k + (l>>5)][m + (l&31)]
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (BN == 128) ? (wave >> 1) * 64 : wave * 32;
    const int wn = (BN == 128) ? (wave & 1) * 64 : 0;
    const long long m0 = (long long)blockIdx.y * BM, n0 = (long long)blockIdx.x * BN;
    const int kbeg = blockIdx.z * a.ksplit;
    const int kend = min(a.K, kbeg + a.ksplit);
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool a_kfast = a.sak == 1;   // pick the thread->element map that walks the unit stride
    const bool b_nfast = a.sbn == 1;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int e = 0; e < (BM * BK) / 256; ++e) {
            const int idx = tid + e * 256;
            int mm, kk;
            if (a_kfast) { mm = idx / BK; kk = idx % BK; } else { kk = idx / BM; mm = idx % BM; }
            const long long gm = m0 + mm;
            const int gk = k0 + kk;
            As[kk][mm] = (gm < a.M && gk < kend) ? a.A[gm * a.sam + gk * a.sak] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < (BN * BK) / 256; ++e) {
            const int idx = tid + e * 256;
            int nn, kk;
            if (b_nfast) { kk = idx / BN; nn = idx % BN; } else { nn = idx / BK; kk = idx % BK; }
            const long long gn = n0 + nn;
            const int gk = k0 + kk;
            Bs[kk][nn] = (gn < a.N && gk < kend) ? a.B[gk * a.sbk + gn * a.sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[kk + (lane >> 5)][wm + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[kk + (lane >> 5)][wn + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map: lane l register r -> row (r&3) + 8(r>>2) + 4(l>>5), column l & 31
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + wn + 32 * j + (lane & 31);
            if (n >= a.N) continue;
            const float bz = (a.bias && blockIdx.z == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bz;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = 1.f / (1.f + expf(-v));
                if (a.mask_src && !(a.mask_src[m * a.ldc + n] > 0.f)) v = 0.f;
                float* c = a.C + m * a.ldc + n;
                if (a.accumulate) atomicAdd(c, v); else *c = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined fp32 MFMA GEMM for the training route (moda_gemm_f32_ex).  128 x BN x 32 tiles, v_mfma_f32_32x32x2_f32,
// 4 waves (BN = 128: 2x2 of 64x64; BN = 64: 4 stacked 32x64).  Each operand is either "k-fast" (unit stride along the
// reduction index: LDS image [row][32 k + 4 pad], fragments by ds_read_b128, four k per read) or "k-slow" (unit
// stride along m / n: LDS image [k][rows + 4], fragments by ds_read_b32); the tile of step t+1 is fetched into
// registers (16-byte loads where the address allows) while step t computes.  Inside a k-tile, MFMA step s pairs
// k = s (lanes 0-31) with k = 16 + s (lanes 32-63) for both operands -- any fixed order of the k sum is a valid
// fp32 result.  Epilogue: column bias, per-row-group bias (the folded per-ray codes), C += (accumulate 2), ReLU /
// sigmoid, ReLU-backward mask, atomic accumulation for split-K.
struct Gemm2Args {
    const float* A; long long sam, sak;
    const float* A2; long long sam2; int K1;        // k >= K1 reads A2(m, k - K1) (k-fast only); null: none
    const float* B; long long sbk, sbn;
    float* C; long long ldc;
    const float* bias;
    const float* rowbias; long long ld_rb; int rb_div;
    const float* mask_src; long long ld_mask;
    int M, N, K;
    int act, accumulate, ksplit;
    float* asum;                // k-slow A only: asum[m] += sum_k A(m,k) (the bias gradient next to a weight gradient), or null
    int vec_c;                  // C rows allow 16-byte accesses (base aligned, ldc % 4 == 0)
    int vec_a, vec_a2, vec_b;   // 16-byte loads are legal for that operand (base aligned, leading dimension % 4 == 0)
    unsigned gx, gy;            // column / row tiles: the grid is launched linear on x (gx * gy blocks), no grid.y limit on M
    // storage types (MODA_GEMM_*_BF16): that operand's elements are bf16 in memory (pointer and strides still count elements);
    // everything is fp32 once it is in registers / LDS
    int a_bf, b_bf, c_bf, m_bf;
};

// ---- element access by storage type: `off` counts elements from `base` -------------------------------------------------
DEVINL float g2_bf2f(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
DEVINL float g2_ld1(const float* __restrict__ base, long long off, int bf) {
    return bf ? g2_bf2f(((const unsigned short*)base)[off]) : base[off];
}
DEVINL float4 g2_ld4(const float* __restrict__ base, long long off, int bf) {     // 4 consecutive elements, vector-aligned
    if (bf) {
        const uint2 u = *(const uint2*)((const unsigned short*)base + off);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
    return *(const float4*)(base + off);
}
DEVINL unsigned g2_pack2(float lo, float hi) {       // two floats -> two bf16 (round to nearest even), lo in the low half
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}
DEVINL void g2_st1(float* __restrict__ base, long long off, float v, int bf) {
    if (bf) ((unsigned short*)base)[off] = (unsigned short)(g2_pack2(v, 0.f) & 0xffffu);
    else base[off] = v;
}
DEVINL void g2_st4(float* __restrict__ base, long long off, const float v[4], int bf) {
    if (bf) *(uint2*)((unsigned short*)base + off) = make_uint2(g2_pack2(v[0], v[1]), g2_pack2(v[2], v[3]));
    else *(float4*)(base + off) = make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int G2_BM = 128, G2_BK = 32, G2_KP = G2_BK + 4;

// Loaders.  No load sits behind a branch (hipcc waits at every join for the loads of both sides, one load at a time, and
// the prefetch serialises -- measured 1.3-3 TB/s on an HBM-bound GEMM): addresses are clamped into the matrix and
// out-of-range elements are zeroed by selects afterwards.  VEC (kernel-uniform: base 16-byte aligned, leading dimension a
// multiple of 4; for the k-fast form also tile-uniform: the whole 32-deep k-tile in range) picks the one-instruction form
// and is a template parameter: the caller branches ONCE around all the loads of a tile, not once per load.
DEVINL float4 g2_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// k-fast: element (row, k) at base[row*ld + k]; four consecutive k
// (the VEC forms return the loaded vector untouched and whether it counts in `ok`: the select that zeroes it is applied
// when the tile is written to LDS, so that nothing reads the registers -- and waits for the load -- before then)
template <bool VEC>
DEVINL float4 g2_load_kfast(const float* __restrict__ base, long long ld, long long row, int rows, int k, int kend,
                            bool& ok, int bf = 0) {
    constexpr bool vec_full = VEC;
    const bool rok = row < rows;
    const long long p = (rok ? row : (long long)rows - 1) * ld;
    float4 v;
    if (vec_full) {
        ok = rok;
        return g2_ld4(base, p + k, bf);
    } else {
        ok = true;
        const int kl = kend - 1;
        v.x = g2_ld1(base, p + min(k, kl), bf); v.y = g2_ld1(base, p + min(k + 1, kl), bf);
        v.z = g2_ld1(base, p + min(k + 2, kl), bf); v.w = g2_ld1(base, p + min(k + 3, kl), bf);
        if (k >= kend) v.x = 0.f;
        if (k + 1 >= kend) v.y = 0.f;
        if (k + 2 >= kend) v.z = 0.f;
        if (k + 3 >= kend) v.w = 0.f;
    }
    return rok ? v : g2_zero4();
}

// k-slow: element (k, c) at base[k*ld + c]; four consecutive c
template <bool VEC>
DEVINL float4 g2_load_kslow(const float* __restrict__ base, long long ld, int k, int kend, long long c, int cols,
                            bool& ok, int bf = 0) {
    constexpr bool vec = VEC;
    const bool kok = k < kend;
    const long long p = (long long)(kok ? k : kend - 1) * ld;
    float4 v;
    if (vec) {                           // cols % 4 == 0 and c % 4 == 0: the float4 is wholly inside or wholly outside
        const long long cc = c < cols ? c : (long long)cols - 4;
        ok = kok && c < cols;
        return g2_ld4(base, p + cc, bf);
    } else {
        ok = true;
        const long long cl = cols - 1;
        v.x = g2_ld1(base, p + (c < cl ? c : cl), bf); v.y = g2_ld1(base, p + (c + 1 < cl ? c + 1 : cl), bf);
        v.z = g2_ld1(base, p + (c + 2 < cl ? c + 2 : cl), bf); v.w = g2_ld1(base, p + (c + 3 < cl ? c + 3 : cl), bf);
        if (c >= cols) v.x = 0.f;
        if (c + 1 >= cols) v.y = 0.f;
        if (c + 2 >= cols) v.z = 0.f;
        if (c + 3 >= cols) v.w = 0.f;
    }
    return kok ? v : g2_zero4();
}

// BF: the throughput mode of the training route -- operands rounded to bf16 (round-to-nearest-even) as they leave the LDS
// image, products and sums in fp32 on v_mfma_f32_32x32x16_bf16 (16x the matrix rate of the exact-fp32 form); lane half h of
// MFMA u of a 32-deep k-tile takes k = 16 h + 8 u + (0..7) from both operands.  Same loaders, tiles and epilogue.
// BF == 2 (MODA_GEMM_BF16X3): split-bf16 on the same matrix cores -- each fp32 operand value v leaves the LDS image as the pair
// hi = bf16(v), lo = bf16(v - hi) (16 significand bits together) and every product is three MFMAs, lo*hi + hi*lo + hi*hi in
// that order (small terms first): 2^-17 relative per operand, ~1e-6 of a sum.
// BF == 3 (MODA_GEMM_BF16X6): three images, hi + mid + lo = v exactly, six MFMAs (every term down to 2^-18 of the product): the
// accuracy class of the exact-fp32 form.  The large aligned forms of both never get here (gemm_x3.hip); this is the path of the
// ragged ones (K not a multiple of 8, second A source, row bias, sigmoid).
// ST: the storage-type flags of Gemm2Args are honoured (bf16 mode only); without it every operand is fp32 in memory and the
// element accessors fold to plain loads / stores.
#ifndef MODA_G2_OCC128
#define MODA_G2_OCC128 2
#endif
template <int BN, bool AK, bool BK, int BF = 0, bool ST = false>
__global__ __launch_bounds__(256, (BN == 128 && BF >= 2) ? MODA_G2_OCC128 : 3) void gemm2_kernel(Gemm2Args a) {
    static_assert(BF == 1 || !ST, "storage types belong to the bf16 mode");
    const int a_bf = ST ? a.a_bf : 0, b_bf = ST ? a.b_bf : 0, c_bf = ST ? a.c_bf : 0, m_bf = ST ? a.m_bf : 0;
    constexpr int TM = (BN == 128) ? 2 : 1;
    constexpr int A_FLOATS = AK ? G2_BM * G2_KP : G2_BK * (G2_BM + 4);
    constexpr int B_FLOATS = BK ? BN * G2_KP : G2_BK * (BN + 4);
    constexpr int NA = G2_BM * G2_BK / 4 / 256;     // float4 per thread per A tile (4)
    constexpr int NB = BN * G2_BK / 4 / 256;        // 4 or 2
    constexpr int EPI_FLOATS = 4 * 32 * 36;         // epilogue: one 32 x (32 + 4) transpose buffer per wave
    constexpr int LDS_FLOATS = (A_FLOATS + B_FLOATS > EPI_FLOATS) ? A_FLOATS + B_FLOATS : EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* As = lds;
    float* Bs = lds + A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = (BN == 128) ? (wave >> 1) * 64 : wave * 32;
    const int wn = (BN == 128) ? (wave & 1) * 64 : 0;
    // The column tiles of one row tile read the same A rows: give them dispatch ids that differ by 8 so that they land on
    // one XCD (blocks are dealt round-robin over the 8 XCDs) and the second read of the A tile hits that XCD's L2.
    // Placement is a speed hint only; any block -> tile bijection is correct.
    const unsigned lid = blockIdx.x;
    unsigned bx = lid % a.gx, by = lid / a.gx;
#ifndef MODA_ABL_NO_XCD
    if (a.gx > 1 && (a.gy & 7) == 0) {
        const unsigned span = 8 * a.gx;
        const unsigned r = lid % span;
        bx = r >> 3;
        by = (lid / span) * 8 + (r & 7);
    }
#endif
    const long long m0 = (long long)by * G2_BM, n0 = (long long)bx * BN;
    const int kbeg = blockIdx.z * a.ksplit;
    const int kend = min(a.K, kbeg + a.ksplit);

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NA], rb[NB];
    bool oka[NA], okb[NB];      // whether a prefetched vector counts (applied in stash)
    const bool vecA = a.vec_a != 0, vecB = a.vec_b != 0;
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        const bool full = k0 + G2_BK <= kend;          // tile-uniform
        if (AK) {
            const int kq = tid & 7, r0 = tid >> 3;
            const int k = k0 + 4 * kq;
            if (a.A2 == nullptr || k0 + G2_BK <= a.K1) {           // entirely in the first source
                const int ke = min(kend, a.K1);
                if (vecA && k0 + G2_BK <= ke) {
#pragma unroll
                    for (int e = 0; e < NA; ++e) ra[e] = g2_load_kfast<true>(a.A, a.sam, m0 + r0 + 32 * e, a.M, k, ke, oka[e], a_bf);
                } else {
#pragma unroll
                    for (int e = 0; e < NA; ++e) ra[e] = g2_load_kfast<false>(a.A, a.sam, m0 + r0 + 32 * e, a.M, k, ke, oka[e], a_bf);
                }
            } else if (k0 >= a.K1) {                               // entirely in the second source
                if (a.vec_a2 != 0 && full) {
#pragma unroll
                    for (int e = 0; e < NA; ++e) ra[e] = g2_load_kfast<true>(a.A2, a.sam2, m0 + r0 + 32 * e, a.M, k - a.K1, kend - a.K1, oka[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < NA; ++e) ra[e] = g2_load_kfast<false>(a.A2, a.sam2, m0 + r0 + 32 * e, a.M, k - a.K1, kend - a.K1, oka[e]);
                }
            } else {                                               // the tile holds the seam: per-element source
#pragma unroll
                for (int e = 0; e < NA; ++e) {
                    const long long row = m0 + r0 + 32 * e;
                    const bool rok = row < a.M;
                    const long long rr = rok ? row : (long long)a.M - 1;
                    float t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int kk = min(k + q, kend - 1);
                        const float* src = kk < a.K1 ? a.A + rr * a.sam + kk : a.A2 + rr * a.sam2 + (kk - a.K1);
                        const float tv = *src;
                        t[q] = (rok && k + q < kend) ? tv : 0.f;
                    }
                    ra[e] = make_float4(t[0], t[1], t[2], t[3]);
                    oka[e] = true;
                }
            }
        } else {
            const int cq = tid & 31, kr = tid >> 5;
            if (vecA) {
#pragma unroll
                for (int e = 0; e < NA; ++e) ra[e] = g2_load_kslow<true>(a.A, a.sak, k0 + kr + 8 * e, kend, m0 + 4 * cq, a.M, oka[e], a_bf);
            } else {
#pragma unroll
                for (int e = 0; e < NA; ++e) ra[e] = g2_load_kslow<false>(a.A, a.sak, k0 + kr + 8 * e, kend, m0 + 4 * cq, a.M, oka[e], a_bf);
            }
        }
        if (BK) {
            const int kq = tid & 7, r0 = tid >> 3;
            if (vecB && full) {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kfast<true>(a.B, a.sbn, n0 + r0 + 32 * e, a.N, k0 + 4 * kq, kend, okb[e], b_bf);
            } else {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kfast<false>(a.B, a.sbn, n0 + r0 + 32 * e, a.N, k0 + 4 * kq, kend, okb[e], b_bf);
            }
        } else if (BN == 128) {
            const int cq = tid & 31, kr = tid >> 5;
            if (vecB) {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kslow<true>(a.B, a.sbk, k0 + kr + 8 * e, kend, n0 + 4 * cq, a.N, okb[e], b_bf);
            } else {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kslow<false>(a.B, a.sbk, k0 + kr + 8 * e, kend, n0 + 4 * cq, a.N, okb[e], b_bf);
            }
        } else {
            const int cq = tid & 15, kr = tid >> 4;
            if (vecB) {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kslow<true>(a.B, a.sbk, k0 + kr + 16 * e, kend, n0 + 4 * cq, a.N, okb[e], b_bf);
            } else {
#pragma unroll
                for (int e = 0; e < NB; ++e) rb[e] = g2_load_kslow<false>(a.B, a.sbk, k0 + kr + 16 * e, kend, n0 + 4 * cq, a.N, okb[e], b_bf);
            }
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
        if (AK) {
            const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int e = 0; e < NA; ++e) *(float4*)(As + (r0 + 32 * e) * G2_KP + 4 * kq) = oka[e] ? ra[e] : g2_zero4();
        } else {
            const int cq = tid & 31, kr = tid >> 5;
#pragma unroll
            for (int e = 0; e < NA; ++e) *(float4*)(As + (kr + 8 * e) * (G2_BM + 4) + 4 * cq) = oka[e] ? ra[e] : g2_zero4();
        }
        if (BK) {
            const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int e = 0; e < NB; ++e) *(float4*)(Bs + (r0 + 32 * e) * G2_KP + 4 * kq) = okb[e] ? rb[e] : g2_zero4();
        } else if (BN == 128) {
            const int cq = tid & 31, kr = tid >> 5;
#pragma unroll
            for (int e = 0; e < NB; ++e) *(float4*)(Bs + (kr + 8 * e) * (BN + 4) + 4 * cq) = okb[e] ? rb[e] : g2_zero4();
        } else {
            const int cq = tid & 15, kr = tid >> 4;
#pragma unroll
            for (int e = 0; e < NB; ++e) *(float4*)(Bs + (kr + 16 * e) * (BN + 4) + 4 * cq) = okb[e] ? rb[e] : g2_zero4();
        }
    };

    float arow = 0.f;           // this thread's running sum over k of A(m0 + tid, k) (threads < 128, first column block only)
    const bool do_asum = !AK && a.asum != nullptr && bx == 0 && tid < G2_BM;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += G2_BK) {
        stash();
        __syncthreads();
        if (k0 + G2_BK < kend) fetch(k0 + G2_BK);
        if (!AK && do_asum) {   // As is [k][m + pad]: consecutive threads read consecutive words (out-of-range k / m are zeros)
#pragma unroll
            for (int kk = 0; kk < G2_BK; ++kk) arow += As[kk * (G2_BM + 4) + tid];
        }
        if constexpr (BF != 0) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto pack8 = [](const float* v) __attribute__((always_inline)) {
                union { u32x4 w; bf16x8 b; } o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    union { bf16x2 b; unsigned u; } c;
                    const f32x2 t = {v[2 * q], v[2 * q + 1]};
                    c.b = __builtin_convertvector(t, bf16x2);
                    o.w[q] = c.u;
                }
                return o.b;
            };
            // v[] <- v[] - bf16(v[]), returning the rounding: applied once (BF 2) or twice (BF 3), each remainder exact in fp32
            auto peel8 = [&](float* v) __attribute__((always_inline)) {
                union { u32x4 w; bf16x8 b; } o;
                o.b = pack8(v);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] -= __uint_as_float(o.w[q] << 16);
                    v[2 * q + 1] -= __uint_as_float(o.w[q] & 0xffff0000u);
                }
                return o.b;
            };
            constexpr int NS = BF == 1 ? 1 : BF;       // bf16 images per operand
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 af[NS][TM], bfr[NS][2];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[8];
                    if (AK) {
                        const float* p = As + (wm + 32 * i + li) * G2_KP + 16 * h + 8 * u;
                        const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 4);
                        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = As[(16 * h + 8 * u + e) * (G2_BM + 4) + wm + 32 * i + li];
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) af[s][i] = (s + 1 < NS) ? peel8(v) : pack8(v);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
                    if (BK) {
                        const float* p = Bs + (wn + 32 * j + li) * G2_KP + 16 * h + 8 * u;
                        const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 4);
                        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = Bs[(16 * h + 8 * u + e) * (BN + 4) + wn + 32 * j + li];
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) bfr[s][j] = (s + 1 < NS) ? peel8(v) : pack8(v);
                }
                // product terms, smallest first (image 0 = hi, 1 = the next 8 significand bits, 2 = the last 8)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (NS == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[NS - 1][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NS - 1][i], bfr[0][j], acc[i][j], 0, 0, 0);
                        }
                        if constexpr (NS >= 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[NS > 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NS > 1 ? 1 : 0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            float av[TM][4], bv[2][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (AK) {
                    const float4 v = *(const float4*)(As + (wm + 32 * i + li) * G2_KP + 16 * h + 4 * t4);
                    av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[i][e] = As[(16 * h + 4 * t4 + e) * (G2_BM + 4) + wm + 32 * i + li];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BK) {
                    const float4 v = *(const float4*)(Bs + (wn + 32 * j + li) * G2_KP + 16 * h + 4 * t4);
                    bv[j][0] = v.x; bv[j][1] = v.y; bv[j][2] = v.z; bv[j][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[j][e] = Bs[(16 * h + 4 * t4 + e) * (BN + 4) + wn + 32 * j + li];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
        }
        __syncthreads();
    }
    if (!AK && do_asum && m0 + tid < a.M) atomicAdd(a.asum + m0 + tid, arow);
    // C/D map: lane l register r -> row (r&3) + 8(r>>2) + 4(l>>5), column l & 31
    const bool first = blockIdx.z == 0;
    if (a.accumulate == 1) {   // split-K partial sums: atomics straight from the accumulators (two 128-B segments per instruction)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long long n = n0 + wn + 32 * j + li;
                if (n >= a.N) continue;
                const float bz = (a.bias && first) ? a.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m >= a.M) continue;
                    float v = acc[i][j][r] + bz;
                    if (a.rowbias && first) v += a.rowbias[(m / a.rb_div) * a.ld_rb + n];
                    atomicAdd(a.C + m * a.ldc + n, v);
                }
            }
        return;
    }
    // Plain stores: each 32x32 accumulator tile goes through a per-wave LDS buffer so that a lane ends up with four
    // consecutive columns of one row -- 16-byte stores (and 16-byte reads of the bias / mask / C operands), eight lanes
    // per 128-byte row segment, instead of 16 scattered dword stores per tile.
    float* tb = lds + wave * (32 * 36);
    const int er = lane >> 3, ec = (lane & 7) * 4;        // this lane's row (+8 per pass) and first column inside a tile
    const bool vecC = a.vec_c != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + li] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const long long nb = n0 + wn + 32 * j + ec;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = er + 8 * ps;
                const float4 t = *(const float4*)(tb + row * 36 + ec);
                float v[4] = {t.x, t.y, t.z, t.w};
                const long long m = m0 + wm + 32 * i + row;
                if (m >= a.M || nb >= a.N) continue;
                const long long co = m * a.ldc + nb;         // element offset of this lane's four columns in C
                const bool v4 = vecC && nb + 3 < a.N;
                if (a.bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < a.N) v[q] += a.bias[nb + q];
                }
                if (a.rowbias) {
                    const float* rbp = a.rowbias + (m / a.rb_div) * a.ld_rb + nb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < a.N) v[q] += rbp[q];
                }
                if (a.accumulate == 2) {
                    if (v4) { const float4 o = g2_ld4(a.C, co, c_bf); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (nb + q < a.N) v[q] += g2_ld1(a.C, co + q, c_bf);
                    }
                }
                if (a.act == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (a.act == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = 1.f / (1.f + expf(-v[q]));
                }
                if (a.mask_src) {
                    const long long mo = m * a.ld_mask + nb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < a.N && !(g2_ld1(a.mask_src, mo + q, m_bf) > 0.f)) v[q] = 0.f;
                }
                if (v4) g2_st4(a.C, co, v, c_bf);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < a.N) g2_st1(a.C, co + q, v[q], c_bf);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// out[r, n] = sum_{s < S} X[(r*S + s)*ld + n]   (segment sums over the samples of a ray: gradient of a per-ray row bias)
__global__ __launch_bounds__(256) void segsum_kernel(const float* __restrict__ X, long long R, int S, int N, long long ld,
                                                    float* __restrict__ out, long long ldo, int bf) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const long long r = blockIdx.y;
    float s = 0.f;
    if (n < N)
        for (int q = sub; q < S; q += 4) s += g2_ld1(X, (r * S + q) * ld + n, bf);
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && n < N) out[r * ldo + n] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[n] (+)= sum_m X[m*ld + n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long long M, int N, long long ld,
                                                    float* __restrict__ out, int rows_per_block, int bf) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;   // 4 row phases per block
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (long long r = r0 + sub; r < r1; r += 4) s += g2_ld1(X, r * ld + n, bf);
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && n < N) atomicAdd(out + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// The same for a tall matrix of at most 64 columns whose rows are 16-byte vectors (N a multiple of VW = 8 bf16 / 4 fp32, base and
// ld aligned): a thread reads VW consecutive columns, N / VW threads cover a row, 256 * VW / N rows per pass -- whole rows per
// wave instruction where colsum_kernel reads one element per lane (the 262144 x 64 bf16 gradient of a row-bias fold: 41 -> ~8 us).
template <bool BF>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ X, long long M, int N, long long ld,
                                                        float* __restrict__ out, int rows_per_block) {
    constexpr int VW = BF ? 8 : 4;
    const int tpr = N / VW, rpp = 256 / tpr;                 // threads per row, rows per pass (tpr divides 256: N in {8..64} x VW)
    const int ch = threadIdx.x % tpr, rr = threadIdx.x / tpr;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(M, r0 + rows_per_block);
    float acc[VW];
#pragma unroll
    for (int q = 0; q < VW; ++q) acc[q] = 0.f;
    if (rr < rpp) {
        for (long long r = r0 + rr; r < r1; r += rpp) {
            if (BF) {
                const uint4 v = *(const uint4*)((const unsigned short*)X + r * ld + 8 * ch);
                acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
                acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
                acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
                acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
            } else {
                const float4 v = *(const float4*)(X + r * ld + 4 * ch);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            }
        }
    }
    __shared__ float red[256 * VW];
#pragma unroll
    for (int q = 0; q < VW; ++q) red[(rr * tpr + ch) * VW + q] = acc[q];
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float sum = 0.f;
        for (int g = 0; g < rpp; ++g) sum += red[g * N + threadIdx.x];     // row group g holds its N column sums contiguously
        atomicAdd(out + threadIdx.x, sum);
    }
}

struct Window { float w[16]; };

// Backward of Embedding.forward (nerf.py:35-75): dx[m,c] = g[m,c] + sum_k w_k 2^k (cos(2^k x) g_sin - sin(2^k x) g_cos),
// and through the optional row normalisation x/|x| (rendering.py:64).
__global__ void embed_bwd_kernel(const float* __restrict__ x, long long M, int C, int F, Window win, int normalize,
                                 const float* __restrict__ g, long long ldg, float* __restrict__ dx) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float nrm = 1.f;
    if (normalize) {
        float n2 = 0.f;
        for (int c = 0; c < C; ++c) n2 += x[m * C + c] * x[m * C + c];
        nrm = sqrtf(n2);
    }
    const float* gm = g + m * ldg;
    float dot = 0.f;      // <du, u> for the normalisation backward
    for (int c = 0; c < C; ++c) {
        const float u = x[m * C + c] / nrm;
        float d = gm[c];
        for (int k = 0; k < F; ++k) {
            float sn, cs;
            sincosf(ldexpf(u, k), &sn, &cs);
            const float f = ldexpf(win.w[k], k);
            d += f * (cs * gm[C + (2 * k) * C + c] - sn * gm[C + (2 * k + 1) * C + c]);
        }
        dx[m * C + c] = d;
        dot += d * u;
    }
    if (normalize)   // d(x/|x|) = (du - u <du,u>) / |x|
        for (int c = 0; c < C; ++c) dx[m * C + c] = (dx[m * C + c] - (x[m * C + c] / nrm) * dot) / nrm;
}

// The same without the row normalisation (sample positions: every network's d_xyz), one thread per (row, channel): three
// times the threads of the row form, and the three channels of a row read neighbouring floats of each frequency pair.
// Same operations in the same order per element as embed_bwd_kernel.
__global__ __launch_bounds__(256) void embed_bwd_elem_kernel(const float* __restrict__ x, long long MC, int C, int F, Window win,
                                                            const float* __restrict__ g, long long ldg, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MC) return;
    const long long m = i / C;
    const int c = (int)(i - m * C);
    const float* gm = g + m * ldg;
    const float u = x[i];
    float d = gm[c];
    for (int k = 0; k < F; ++k) {
        float sn, cs;
        sincos_rr(ldexpf(u, k), sn, cs);           // <= 1.3e-7 absolute, a third of sincosf's instructions (this kernel was bound by them)
        const float f = ldexpf(win.w[k], k);
        d += f * (cs * gm[C + (2 * k) * C + c] - sn * gm[C + (2 * k + 1) * C + c]);
    }
    dx[i] = d;
}

// The tangent of Embedding.forward at constant positions: out[m] = J(x[m]) u[m], i.e. [u_c, w_k 2^k cos(2^k x_c) u_c,
// -w_k 2^k sin(2^k x_c) u_c] in the encoding's column order -- the transpose of embed_bwd_elem_kernel's product, same sincos.
// It is the backward of the eikonal term's d sigma / d x = J^T g w.r.t. g (loss_utils.nerf_gradient: x is a constant there).
__global__ __launch_bounds__(256) void embed_jvp_kernel(const float* __restrict__ x, long long MC, int C, int F, Window win,
                                                       const float* __restrict__ u, float* __restrict__ out, long long ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MC) return;
    const long long m = i / C;
    const int c = (int)(i - m * C);
    const float xv = x[i], uv = u[i];
    float* om = out + m * ldo;
    om[c] = uv;
    for (int k = 0; k < F; ++k) {
        float sn, cs;
        sincos_rr(ldexpf(xv, k), sn, cs);
        const float f = ldexpf(win.w[k], k);
        om[C + (2 * k) * C + c] = f * cs * uv;
        om[C + (2 * k + 1) * C + c] = -f * sn * uv;
    }
}

}   // namespace

extern "C" int moda_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                             int64_t ldc, int64_t M, int64_t N, int64_t K, const float* bias, int32_t act,
                             const float* mask_src, int32_t accumulate, int32_t split_k, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A || !B || !C || K < 0 || M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MODA_EINVAL;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && !accumulate) return MODA_EINVAL;   // partial sums need C += (C zeroed or holding the addend)
    GemmArgs a;
    a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.ldc = ldc;
    a.bias = bias; a.mask_src = mask_src; a.M = (int)M; a.N = (int)N; a.K = (int)K; a.act = act; a.accumulate = accumulate;
    const int per = (int)((K + split_k - 1) / split_k);
    a.ksplit = ((per + BK - 1) / BK) * BK;
    if (a.ksplit < BK) a.ksplit = BK;
    const int zs = K > 0 ? (int)((K + a.ksplit - 1) / a.ksplit) : 1;
    if ((M + BM - 1) / BM > 65535 || zs > 65535) return MODA_ESHAPE;   // grid.y / grid.z limit (moda_gemm_f32_ex has none on M)
    if (N <= 64) {
        dim3 grid(1, (unsigned)((M + BM - 1) / BM), (unsigned)zs);
        hipLaunchKernelGGL(gemm_f32_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + BM - 1) / BM), (unsigned)zs);
        hipLaunchKernelGGL(gemm_f32_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

template <int BN, int BF, bool ST>
static void gemm2_launch_p(const Gemm2Args& a, bool ak, bool bk, dim3 grid, hipStream_t st) {
    if (ak && bk) hipLaunchKernelGGL((gemm2_kernel<BN, true, true, BF, ST>), grid, dim3(256), 0, st, a);
    else if (ak) hipLaunchKernelGGL((gemm2_kernel<BN, true, false, BF, ST>), grid, dim3(256), 0, st, a);
    else if (bk) hipLaunchKernelGGL((gemm2_kernel<BN, false, true, BF, ST>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm2_kernel<BN, false, false, BF, ST>), grid, dim3(256), 0, st, a);
}

template <int BN>
static void gemm2_launch(const Gemm2Args& a, bool ak, bool bk, bool bf16, int x3, dim3 grid, hipStream_t st) {
    const bool typed = a.a_bf || a.b_bf || a.c_bf || a.m_bf;
    if (x3 == 3) gemm2_launch_p<BN, 3, false>(a, ak, bk, grid, st);
    else if (x3) gemm2_launch_p<BN, 2, false>(a, ak, bk, grid, st);
    else if (bf16 && typed) gemm2_launch_p<BN, 1, true>(a, ak, bk, grid, st);
    else if (bf16) gemm2_launch_p<BN, 1, false>(a, ak, bk, grid, st);
    else gemm2_launch_p<BN, 0, false>(a, ak, bk, grid, st);
}

extern "C" int moda_gemm_f32_ex(const moda_gemm_desc* d, void* stream) {
    if (!d) return MODA_EINVAL;
    if (d->M <= 0 || d->N <= 0) return 0;
    const int64_t lim = 0x7fffffff;
    if (!d->A || !d->B || !d->C || d->K < 0 || d->M > lim || d->N > lim || d->K > lim) return MODA_EINVAL;
    // a single-row / single-column operand has both strides 1: with a_sum requested it must take the m-fast (k-slow) form
    const bool ak = d->sak == 1 && !(d->a_sum && d->sam == 1), bk = d->sbk == 1;
    if ((!ak && d->sam != 1) || (!bk && d->sbn != 1)) return MODA_ESHAPE;     // each operand needs one unit stride
    if (d->A2 && (!ak || d->K1 < 0 || d->K1 > d->K)) return MODA_EINVAL;
    int split_k = d->split_k < 1 ? 1 : d->split_k;
    if (split_k > 1 && d->accumulate != 1) return MODA_EINVAL;
    if (d->rowbias && d->rows_per_bias < 1) return MODA_EINVAL;
    Gemm2Args a;
    a.A = d->A; a.sam = ak ? d->sam : 1; a.sak = ak ? 1 : d->sak;
    a.A2 = d->A2; a.sam2 = d->sam2; a.K1 = d->A2 ? (int)d->K1 : (int)d->K;
    a.B = d->B; a.sbk = d->sbk; a.sbn = d->sbn;
    a.C = d->C; a.ldc = d->ldc; a.bias = d->bias;
    a.rowbias = d->rowbias; a.ld_rb = d->ld_rowbias; a.rb_div = d->rowbias ? (int)(d->rows_per_bias > lim ? lim : d->rows_per_bias) : 1;
    a.mask_src = d->mask_src; a.ld_mask = d->ld_mask;
    a.asum = ak ? nullptr : d->a_sum;
    if (d->a_sum && ak) return MODA_EINVAL;
    a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K;
    a.act = d->act; a.accumulate = d->accumulate;
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    // k-fast operands step along k in units of 4 from a k0 that is a multiple of 32 (offset by K1 for A2);
    // k-slow operands step along m / n in units of 4 from multiples of 128 / 64
    a.vec_a = al16(d->A) && ((ak ? d->sam : d->sak) % 4 == 0) && (ak || d->M % 4 == 0);
    a.vec_a2 = d->A2 && al16(d->A2) && d->sam2 % 4 == 0 && d->K1 % 4 == 0;
    a.vec_c = al16(d->C) && d->ldc % 4 == 0;
    a.vec_b = al16(d->B) && ((bk ? d->sbn : d->sbk) % 4 == 0) && (bk || d->N % 4 == 0);
    const int per = (int)((d->K + split_k - 1) / split_k);
    a.ksplit = ((per + G2_BK - 1) / G2_BK) * G2_BK;
    if (a.ksplit < G2_BK) a.ksplit = G2_BK;
    const unsigned zs = d->K > 0 ? (unsigned)((d->K + a.ksplit - 1) / a.ksplit) : 1u;
    a.gy = (unsigned)((d->M + G2_BM - 1) / G2_BM);
    a.gx = d->N <= 64 ? 1u : (unsigned)((d->N + 127) / 128);
    if ((uint64_t)a.gx * a.gy > 0x7fffffffull || zs > 65535u) return MODA_ESHAPE;
    const bool bf16 = (d->reserved & MODA_GEMM_BF16) != 0;
    const int x3 = (d->reserved & MODA_GEMM_BF16X6) ? 3 : ((d->reserved & MODA_GEMM_BF16X3) ? 2 : 0);    // bf16 images per operand
    if ((bf16 && x3) || ((d->reserved & MODA_GEMM_BF16X6) && (d->reserved & MODA_GEMM_BF16X3))) return MODA_EINVAL;
    a.a_bf = (d->reserved & MODA_GEMM_A_BF16) != 0; a.b_bf = (d->reserved & MODA_GEMM_B_BF16) != 0;
    a.c_bf = (d->reserved & MODA_GEMM_C_BF16) != 0; a.m_bf = (d->reserved & MODA_GEMM_MASK_BF16) != 0;
    if (a.c_bf && d->accumulate == 1) return MODA_EINVAL;            // the atomics of the split-K form are fp32
    if ((a.a_bf || a.b_bf || a.c_bf || a.m_bf) && !bf16) return MODA_EINVAL;   // storage types belong to the bf16 mode
    {   // the large forms of the bf16-storage backward and of the split-bf16 mode have their own kernels (gemm_bf16.hip, gemm_x3.hip)
        int rc3 = 0;
        if (x3 ? moda_x3_try(d, x3, stream, &rc3) : moda_g3_try(d, stream, &rc3)) return rc3;
    }
    if (d->mask_bits) return MODA_ESHAPE;        // sign-bit maps exist for the bf16-native and split-bf16 forms only
    // 8-byte vectors of bf16 need 8-byte alignment; the 16-byte test above already covers it
    if (d->N <= 64) gemm2_launch<64>(a, ak, bk, bf16, x3, dim3(a.gx * a.gy, 1, zs), (hipStream_t)stream);
    else gemm2_launch<128>(a, ak, bk, bf16, x3, dim3(a.gx * a.gy, 1, zs), (hipStream_t)stream);
    return (int)hipGetLastError();
}

// X may hold bf16 elements (bf != 0; same element offsets) -- the storage mode of the training route
static int segsum_any(const float* X, int64_t R, int64_t S, int64_t N, int64_t ld, float* out, int64_t ldo, int bf, void* stream) {
    if (R <= 0 || N <= 0) return 0;
    if (!X || !out || S < 1 || R > 0x7fffffff) return MODA_EINVAL;
    if (R > 65535) {   // grid.y limit: fold rows into chunks
        for (int64_t r0 = 0; r0 < R; r0 += 65535) {
            const int64_t rr = R - r0 < 65535 ? R - r0 : 65535;
            hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)rr), dim3(256), 0, (hipStream_t)stream,
                               bf ? (const float*)((const unsigned short*)X + r0 * S * ld) : X + r0 * S * ld, (long long)rr, (int)S,
                               (int)N, (long long)ld, out + r0 * ldo, (long long)ldo, bf);
        }
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)R), dim3(256), 0, (hipStream_t)stream, X,
                       (long long)R, (int)S, (int)N, (long long)ld, out, (long long)ldo, bf);
    return (int)hipGetLastError();
}

extern "C" int moda_segsum_f32(const float* X, int64_t R, int64_t S, int64_t N, int64_t ld, float* out, int64_t ldo, void* stream) {
    return segsum_any(X, R, S, N, ld, out, ldo, 0, stream);
}

static int colsum_any(const float* X, int64_t M, int64_t N, int64_t ld, float* out, int bf, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!X || !out) return MODA_EINVAL;
    // >= 1024 workgroups at training sizes; one atomicAdd per column per workgroup.  Short matrices (the per-ray sums: a few
    // thousand rows) take 32 rows per workgroup: with 256 the handful of workgroups walked their rows one after the other
    {   // tall and narrow with vector-aligned rows: the wide-load form
        const int vw = bf ? 8 : 4;
        const bool al = (((uintptr_t)X) & 15) == 0 && ld % vw == 0;
        if (M >= 4096 && N <= 64 && N % vw == 0 && 256 % (N / vw) == 0 && al) {
            const int rows = 1024;
            const dim3 grid((unsigned)((M + rows - 1) / rows));
            if (bf) hipLaunchKernelGGL(colsum_vec_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, (long long)M, (int)N, (long long)ld, out, rows);
            else hipLaunchKernelGGL(colsum_vec_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, (long long)M, (int)N, (long long)ld, out, rows);
            return (int)hipGetLastError();
        }
    }
    int rows = M <= 16384 ? 32 : 256;
    while ((M + rows - 1) / rows > 65535) rows *= 2;   // grid.y limit
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + rows - 1) / rows));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, (long long)M, (int)N, (long long)ld, out, rows, bf);
    return (int)hipGetLastError();
}

extern "C" int moda_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* stream) {
    return colsum_any(X, M, N, ld, out, 0, stream);
}

extern "C" int moda_embed_bwd(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window, int32_t normalize,
                              const float* grad_out, int64_t ldg, float* grad_x, void* stream) {
    if (M <= 0) return 0;
    if (!x || !grad_out || !grad_x || C < 1 || n_freq < 0 || n_freq > 16) return MODA_EINVAL;
    if (ldg < (int64_t)C * (1 + 2 * n_freq)) return MODA_EINVAL;
    Window w;
    for (int i = 0; i < 16; ++i) w.w[i] = (i < n_freq && window) ? window[i] : 0.f;
    if (!normalize)
        hipLaunchKernelGGL(embed_bwd_elem_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)M * C, C, n_freq, w, grad_out, (long long)ldg, grad_x);
    else
        hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)M, C,
                           n_freq, w, normalize, grad_out, (long long)ldg, grad_x);
    return (int)hipGetLastError();
}

extern "C" int moda_embed_jvp(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window, const float* u,
                              float* out, int64_t ldo, void* stream) {
    if (M <= 0) return 0;
    if (!x || !u || !out || C < 1 || n_freq < 0 || n_freq > 16 || ldo < (int64_t)C * (1 + 2 * n_freq)) return MODA_EINVAL;
    Window w;
    for (int i = 0; i < 16; ++i) w.w[i] = (i < n_freq && window) ? window[i] : 0.f;
    hipLaunchKernelGGL(embed_jvp_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)M * C, C,
                       n_freq, w, u, out, (long long)ldo);
    return (int)hipGetLastError();
}

// ================================================================================================
// Fan-in of gradients and small affine maps (round 6): launches that replace chains of PyTorch-native elementwise kernels
// ================================================================================================
namespace {

// out = x[0] + x[1] + ... + x[n-1] (n <= 8), summed in argument order: the ONE launch that replaces autograd's n - 1 accumulation
// adds for a tensor that feeds n nodes (autograd.FanOutFn)
struct SumArgs { const float* x[8]; int n; long long numel; float* out; };
__global__ __launch_bounds__(256) void sum_tensors_kernel(SumArgs a) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= a.numel) return;
    if (i4 + 3 < a.numel) {
        float4 s = *(const float4*)(a.x[0] + i4);
        for (int k = 1; k < a.n; ++k) {
            const float4 v = *(const float4*)(a.x[k] + i4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *(float4*)(a.out + i4) = s;
    } else {
        for (long long i = i4; i < a.numel; ++i) {
            float s = a.x[0][i];
            for (int k = 1; k < a.n; ++k) s += a.x[k][i];
            a.out[i] = s;
        }
    }
}

// out[i][c] = y[i][c] + (x[i][c] * scale[c]) * post + shift[c] for rows of three (y, shift optional), every operation rounded on
// its own in THIS order (no fma contraction): the lattice jitter `query + randn * bound * 0.05` of feat_match
// (loss_utils.py:304-306) and the negatives `rand * 2 * bound - bound` of the visibility loss (loss_utils.py:137-138) bit for bit
// as the eager expressions evaluate them -- one float32 ulp of a lattice node is 8e-4 of nerf_feat's first-layer gradient behind
// the 2^9 frequency of the encoding (round 5's float64-truth finding) -- as one launch each
__global__ __launch_bounds__(256) void affine3_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ scale, float post, const float* __restrict__ shift,
                                                      long long n3, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const int c = (int)(i % 3);
    float v = __fmul_rn(x[i], scale[c]);
    if (post != 1.f) v = __fmul_rn(v, post);
    if (y) v = __fadd_rn(y[i], v);
    if (shift) v = __fadd_rn(v, shift[c]);
    out[i] = v;
}
}   // namespace

extern "C" int moda_sum_tensors(const float* const* xs, int32_t n, int64_t numel, float* out, void* stream) {
    if (numel <= 0) return 0;
    if (!xs || !out || n < 1 || n > 8) return MODA_EINVAL;
    SumArgs a;
    for (int k = 0; k < 8; ++k) {
        a.x[k] = k < n ? xs[k] : nullptr;
        if (k < n && (!xs[k] || (((uintptr_t)xs[k]) & 15))) return MODA_EINVAL;       // 16-byte aligned operands (float4 loads)
    }
    if (((uintptr_t)out) & 15) return MODA_EINVAL;
    a.n = n; a.numel = numel; a.out = out;
    hipLaunchKernelGGL(sum_tensors_kernel, dim3((unsigned)(((numel + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int moda_affine3(const float* x, const float* y, const float* scale, float post, const float* shift, int64_t rows,
                            float* out, void* stream) {
    if (rows <= 0) return 0;
    if (!x || !scale || !out) return MODA_EINVAL;
    hipLaunchKernelGGL(affine3_kernel, dim3((unsigned)((rows * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, scale, post,
                       shift, (long long)rows * 3, out);
    return (int)hipGetLastError();
}

// ================================================================================================
// Backward of the per-ray kernels
// ================================================================================================
namespace {

// dz = dy * act'(y): act 1 relu (y > 0), 2 sigmoid (y (1 - y))
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n, int act,
                               float* __restrict__ dz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    dz[i] = act == 1 ? (v > 0.f ? dy[i] : 0.f) : (act == 2 ? dy[i] * v * (1.f - v) : dy[i]);
}

DEVINL float wave_sum_f(float v) { return comp_wave_sum(v); }      // (DPP / lane-swap form, bitwise the shuffle butterfly: moda_dev.h)

// Backward of composite_kernel (rendering.py:183-237).  One wave per ray, 64-sample blocks in REVERSE order with a
// carried suffix sum  suffix_i = sum_{k>i} v_k w_k  (v_k = dL/dw_k):  dL/dalpha_i = v_i T_i - suffix_i / t_i.
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ rgbsigma, const float* __restrict__ feat, int F, const float* __restrict__ zv,
    const float* __restrict__ rd, const float* __restrict__ beta, const float* __restrict__ noise,
    const float* __restrict__ xyz, const float* __restrict__ clip, const float* __restrict__ vis_pred,
    const float* __restrict__ cyc, const float* __restrict__ weights, const float* __restrict__ visibility,
    float rgb_filter_scale, long long N, long long S, const float* __restrict__ g_rgb, const float* __restrict__ g_feat, const float* __restrict__ g_depth,
    const float* __restrict__ g_sil, const float* __restrict__ g_w, const float* __restrict__ g_cyc,
    float* __restrict__ d_rgbsigma, float* __restrict__ d_feat, float* __restrict__ d_z, float* __restrict__ d_rd,
    float* __restrict__ d_beta, float* __restrict__ d_cyc) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float dx = rd[n * 3], dy = rd[n * 3 + 1], dzc = rd[n * 3 + 2];
    const float dnorm = comp_dnorm(rd, n);            // (the forward's own routines: alpha and t below are bit-for-bit what the
    const float b = beta[0];                          //  forward multiplied into the saved transmittances)
    const float ib = comp_ibeta(beta);
    float cbx = 0.f, cby = 0.f, cbz = 0.f;
    if (clip) { cbx = clip[0]; cby = clip[1]; cbz = clip[2]; }
    const float gr = g_rgb ? g_rgb[n * 3] : 0.f, gg = g_rgb ? g_rgb[n * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[n * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[n] : 0.f, gs = g_sil ? g_sil[n] : 0.f, gc = g_cyc ? g_cyc[n] : 0.f;
    const bool fvec = feat && g_feat && (F & 3) == 0 && F <= 16 &&
                      ((((uintptr_t)feat) | ((uintptr_t)g_feat) | ((uintptr_t)d_feat)) & 15) == 0;
    float suffix = 0.f;      // sum over samples after the current block of v w
    float a_dnorm = 0.f, a_ib = 0.f;
    float dz_from_next = 0.f;   // contribution to d z_i from delta_{i-1} is handled by writing both ends
    const long long nblk = (S + 63) / 64;
    for (long long blk = nblk - 1; blk >= 0; --blk) {
        const long long s = blk * 64 + lane;
        const bool in = s < S;
        const long long i = n * S + (in ? s : S - 1);
        float v = 0.f, w = 0.f, T = 1.f, alpha = 0.f, t = 1.f, delta = 0.f, dens = 0.f, e = 0.f, sdf = 0.f, zdiff = 0.f;
        float sem = 1.f, sg10 = 0.f, grgb = 0.f;
        bool masked = false;
        float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
            rs = *(const float4*)(rgbsigma + i * 4);
            const float z = zv[i];
            zdiff = (s + 1 < S ? zv[i + 1] - z : 1e10f);
            delta = zdiff * dnorm;
            float sg = rs.w;
            if (noise) sg += noise[i];
            sdf = -sg;
            e = expf(-fabsf(sdf) * ib);
            const float sgn = sdf > 0.f ? 1.f : (sdf < 0.f ? -1.f : 0.f);
            dens = (0.5f + 0.5f * sgn * (e - 1.f)) * ib;
            alpha = comp_alpha(rs.w, noise != nullptr, noise ? noise[i] : 0.f, delta, ib);
            if (clip) {
                const float* p = xyz + i * 3;
                if (fabsf(p[0]) > cbx || fabsf(p[1]) > cby || fabsf(p[2]) > cbz) masked = true;
            }
            if (vis_pred && vis_pred[i] < 0.5f) masked = true;
            if (masked) alpha = 0.f;
            t = 1.f - alpha + 1e-10f;
            T = visibility[i];
            w = weights[i];
            // rgb_filter: rgb = sum_{s < S-1} w sem rgb_s, sem = scale sigmoid(-10 sigma_raw) (rendering.py:171, 225-230)
            sem = 1.f;
            if (rgb_filter_scale > 0.f) {
                sg10 = 1.f / (1.f + expf(10.f * rs.w));
                sem = s + 1 < S ? rgb_filter_scale * sg10 : 0.f;
            }
            grgb = gr * rs.x + gg * rs.y + gb * rs.z;
            v = sem * grgb + gd * z + (s + 1 < S ? gs : 0.f) + (g_w ? g_w[i] : 0.f);
            if (feat && g_feat) {
                const float* fp = feat + i * F;
                if (fvec) {      // 16-byte loads issued together, the sum in the same order
#pragma unroll
                    for (int f4 = 0; f4 < 4; ++f4)
                        if (4 * f4 < F) {
                            const float4 q = ((const float4*)fp)[f4], gq = ((const float4*)(g_feat + n * F))[f4];
                            v += gq.x * q.x; v += gq.y * q.y; v += gq.z * q.z; v += gq.w * q.w;
                        }
                } else {
                    for (int f = 0; f < F; ++f) v += g_feat[n * F + f] * fp[f];
                }
            }
        }
        // exclusive suffix within the block: sum of v w over lanes > this lane
        const float vw = in ? v * w : 0.f;
        float p = vw;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float q = __shfl_down(p, o, 64);
            if (lane + o < 64) p += q;
        }
        const float excl = p - vw + suffix;
        suffix += __shfl(p, 0, 64);
        if (in) {
            const float dalpha = masked ? 0.f : (v * T - excl / t);
            const float one_m_a = expf(-delta * dens);              // 1 - alpha (unmasked value)
            const float ddens = dalpha * delta * one_m_a;
            const float ddelta = dalpha * dens * one_m_a;
            // sigma_raw: dens' wrt sdf is -0.5 ib^2 e (both signs), sdf = -sigma
            float dsig = (sdf == 0.f) ? 0.f : ddens * 0.5f * ib * ib * e;
            if (rgb_filter_scale > 0.f && s + 1 < S)      // d sem / d sigma_raw = -10 scale sg (1 - sg)
                dsig += w * grgb * (-10.f * rgb_filter_scale * sg10 * (1.f - sg10));
            float4 o4;
            o4.x = w * sem * gr; o4.y = w * sem * gg; o4.z = w * sem * gb; o4.w = dsig;
            *(float4*)(d_rgbsigma + i * 4) = o4;
            if (feat && d_feat && g_feat) {
                if (fvec) {
#pragma unroll
                    for (int f4 = 0; f4 < 4; ++f4)
                        if (4 * f4 < F) {
                            const float4 gq = ((const float4*)(g_feat + n * F))[f4];
                            ((float4*)(d_feat + i * F))[f4] = make_float4(w * gq.x, w * gq.y, w * gq.z, w * gq.w);
                        }
                } else {
                    for (int f = 0; f < F; ++f) d_feat[i * F + f] = w * g_feat[n * F + f];
                }
            }
            if (d_cyc && cyc) d_cyc[i] = gc * w;
            // d dens / d ib at fixed sdf
            const float x = fabsf(sdf) * ib;
            const float ddib = sdf > 0.f ? 0.5f * e * (1.f - x) : (sdf < 0.f ? (1.f - 0.5f * e) + 0.5f * x * e : 0.5f);
            a_ib += ddens * ddib;
            a_dnorm += ddelta * zdiff;
            // z: depth term + the two deltas it bounds (the last delta is the constant 1e10)
            if (d_z) {
                float dzv = w * gd;
                if (s + 1 < S) dzv -= ddelta * dnorm;
                atomicAdd(d_z + i, dzv);
                if (s + 1 < S) atomicAdd(d_z + i + 1, ddelta * dnorm);
            }
        }
    }
    (void)dz_from_next;
    a_dnorm = wave_sum_f(a_dnorm);
    a_ib = wave_sum_f(a_ib);
    if (lane == 0) {
        if (d_rd) {   // |d| = sqrt(d.d): grad = a_dnorm * d / |d|
            atomicAdd(d_rd + n * 3 + 0, a_dnorm * dx / dnorm);
            atomicAdd(d_rd + n * 3 + 1, a_dnorm * dy / dnorm);
            atomicAdd(d_rd + n * 3 + 2, a_dnorm * dzc / dnorm);
        }
        if (d_beta) atomicAdd(d_beta, a_ib * (-(b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * ib * ib));   // ib = 1/(|b|+eps)
    }
}

// xyz = o + d z: d_o[n] += sum_s dxyz, d_d[n] += sum_s z dxyz, d_z[n,s] += d . dxyz   (rendering.py:88-89)
__global__ __launch_bounds__(256) void points_bwd_kernel(const float* __restrict__ dxyz, const float* __restrict__ zv,
                                                        const float* __restrict__ rd, long long N, long long S,
                                                        float* __restrict__ d_o, float* __restrict__ d_d, float* __restrict__ d_z) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float ox = 0.f, oy = 0.f, oz = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        const float gx = dxyz[i * 3], gy = dxyz[i * 3 + 1], gz = dxyz[i * 3 + 2];
        const float z = zv[i];
        ox += gx; oy += gy; oz += gz;
        ax += z * gx; ay += z * gy; az += z * gz;
        if (d_z) atomicAdd(d_z + i, rd[n * 3] * gx + rd[n * 3 + 1] * gy + rd[n * 3 + 2] * gz);
    }
    ox = wave_sum_f(ox); oy = wave_sum_f(oy); oz = wave_sum_f(oz);
    ax = wave_sum_f(ax); ay = wave_sum_f(ay); az = wave_sum_f(az);
    if (lane == 0) {
        if (d_o) { atomicAdd(d_o + n * 3, ox); atomicAdd(d_o + n * 3 + 1, oy); atomicAdd(d_o + n * 3 + 2, oz); }
        if (d_d) { atomicAdd(d_d + n * 3, ax); atomicAdd(d_d + n * 3 + 1, ay); atomicAdd(d_d + n * 3 + 2, az); }
    }
}

// Backward of the skinning + DQS warp given prepared per-bone data (see warp_kernel in render_kernels.hip):
//   prep (nsets,B,16) = [c(3) | R row-major (9) | s(3) | -],  q (N,B,8) the dual quaternions actually blended,
//   skin (N,S,B) saved by the forward.  Produces d_pts, d_dskin (= d logits), and accumulates d_prep, d_q, d_eaux.
// STAGE: the workgroup's 256 rows of B skinning weights (one contiguous piece of `skin`), the direct gradient g_skin and
// the logit gradients it produces go through LDS tiles with coalesced 16-byte global accesses; a thread's row is B words
// from its neighbour's (B odd: conflict-free).  Without it every lane walks its own row in global memory: 50 cache lines
// per wave-instruction, three passes.
typedef float __attribute__((address_space(3))) lds_float_t;
DEVINL void tile_copy_in(float* tile, const float* src, int cnt) {
    if ((((uintptr_t)src) & 15) == 0) {
        for (int j = threadIdx.x; j < cnt / 4; j += 256) ((float4*)tile)[j] = ((const float4*)src)[j];
        for (int j = (cnt / 4) * 4 + threadIdx.x; j < cnt; j += 256) tile[j] = src[j];
    } else {
        for (int j = threadIdx.x; j < cnt; j += 256) tile[j] = src[j];
    }
}
template <bool STAGE>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ prep, int per_ray, const float* __restrict__ q,
                                                      const float* __restrict__ pts, const float* __restrict__ pts_tf,
                                                      float* __restrict__ d_pts_tf, const float* __restrict__ skin,
                                                      const float* __restrict__ e_aux_p, const float* __restrict__ cyc_ref,
                                                      const float* __restrict__ g_out, const float* __restrict__ g_cyc,
                                                      const float* __restrict__ g_skin, long long N, long long S, int B,
                                                      float* __restrict__ d_pts, float* __restrict__ d_dskin,
                                                      float* __restrict__ d_bl, float* __restrict__ d_ref) {
    extern __shared__ __attribute__((aligned(16))) float wt[];      // STAGE: [skin rows -> logit gradients][g_skin rows]
    const long long i_base = (long long)blockIdx.x * 256;
    const int rows_here = (int)(N * S - i_base < 256 ? N * S - i_base : 256);
    if (STAGE) {
        tile_copy_in(wt, skin + i_base * B, rows_here * B);
        if (g_skin) tile_copy_in(wt + 256 * B, g_skin + i_base * B, rows_here * B);
        __syncthreads();
    }
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N * S;   // dead tail lanes run on the last sample with zero upstream gradients (the wave
    if (!live) i = N * S - 1;      // stays whole for the shuffle reduction); their direct stores are suppressed
    const long long n = i / S;
    const float e_aux = expf(e_aux_p[0]);   // skin_aux[0] is the log scale (geom_utils.py:244,265)
    lds_float_t* const srow = (lds_float_t*)wt + (int)(i - i_base) * B;
    lds_float_t* const grow = srow + 256 * B;
    // lx: the point the skinning weights are evaluated at; px: the point the blended transform is applied to (the same
    // unless pts_tf is given: x + nerf_dis(x) in neu_dbs' forward direction, geom_utils.py:420-425)
    const float lx = pts[i * 3], ly = pts[i * 3 + 1], lz = pts[i * 3 + 2];
    const float px = pts_tf ? pts_tf[i * 3] : lx, py = pts_tf ? pts_tf[i * 3 + 1] : ly, pz = pts_tf ? pts_tf[i * 3 + 2] : lz;
    const float* P0 = prep + (per_ray ? n * B * 16 : 0);
    const float* Q0 = q + n * B * 8;
    const float* sk = skin + i * B;
    // forward recompute of the blend
    float bl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 8; ++k) bl[k] = fmaf(STAGE ? srow[b] : sk[b], Q0[b * 8 + k], bl[k]);
    const float nrm = sqrtf(bl[0] * bl[0] + bl[1] * bl[1] + bl[2] * bl[2] + bl[3] * bl[3]);
    float c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = bl[k] / nrm;
    const float a0 = c[0], d0x = c[1], d0y = c[2], d0z = c[3], ae = c[4], dex = c[5], dey = c[6], dez = c[7];
    // u = d0 x p + a0 p ; out = p + 2 d0 x u + 2 (a0 de - ae d0 + d0 x de)
    const float ux = d0y * pz - d0z * py + a0 * px, uy = d0z * px - d0x * pz + a0 * py, uz = d0x * py - d0y * px + a0 * pz;
    float gx = (g_out && live) ? g_out[i * 3] : 0.f, gy = (g_out && live) ? g_out[i * 3 + 1] : 0.f,
          gz = (g_out && live) ? g_out[i * 3 + 2] : 0.f;
    if (cyc_ref && g_cyc && d_ref && live) {   // cyc = |ref - out| (rendering.py:341)
        const float tx = 2.f * (a0 * dex - ae * d0x + (d0y * dez - d0z * dey));
        const float ty = 2.f * (a0 * dey - ae * d0y + (d0z * dex - d0x * dez));
        const float tz = 2.f * (a0 * dez - ae * d0z + (d0x * dey - d0y * dex));
        const float ox = px + 2.f * (d0y * uz - d0z * uy) + tx, oy = py + 2.f * (d0z * ux - d0x * uz) + ty,
                    oz = pz + 2.f * (d0x * uy - d0y * ux) + tz;
        const float rx = cyc_ref[i * 3] - ox, ry = cyc_ref[i * 3 + 1] - oy, rz = cyc_ref[i * 3 + 2] - oz;
        const float len = sqrtf(rx * rx + ry * ry + rz * rz);
        const float gcv = g_cyc[i];
        const float sx = len > 0.f ? gcv * rx / len : 0.f, sy = len > 0.f ? gcv * ry / len : 0.f, sz = len > 0.f ? gcv * rz / len : 0.f;
        d_ref[i * 3] = sx; d_ref[i * 3 + 1] = sy; d_ref[i * 3 + 2] = sz;
        gx -= sx; gy -= sy; gz -= sz;
    }
    // y1 = d0 x u (weight 2), t (weight 2)
    const float g2x = 2.f * gx, g2y = 2.f * gy, g2z = 2.f * gz;
    // dL/du = g2 x d0 ; dL/dd0 += u x g2
    const float gux = g2y * d0z - g2z * d0y, guy = g2z * d0x - g2x * d0z, guz = g2x * d0y - g2y * d0x;
    float dd0x = uy * g2z - uz * g2y, dd0y = uz * g2x - ux * g2z, dd0z = ux * g2y - uy * g2x;
    // u = d0 x p + a0 p : dL/dp += gu x d0 + a0 gu ; dL/dd0 += p x gu ; dL/da0 += gu . p
    float dpx = gx + (guy * d0z - guz * d0y) + a0 * gux;
    float dpy = gy + (guz * d0x - gux * d0z) + a0 * guy;
    float dpz = gz + (gux * d0y - guy * d0x) + a0 * guz;
    dd0x += py * guz - pz * guy; dd0y += pz * gux - px * guz; dd0z += px * guy - py * gux;
    float da0 = gux * px + guy * py + guz * pz;
    // t = a0 de - ae d0 + d0 x de (gradient g2)
    da0 += g2x * dex + g2y * dey + g2z * dez;
    const float ddex = a0 * g2x + (g2y * d0z - g2z * d0y), ddey = a0 * g2y + (g2z * d0x - g2x * d0z),
                ddez = a0 * g2z + (g2x * d0y - g2y * d0x);
    const float dae = -(g2x * d0x + g2y * d0y + g2z * d0z);
    dd0x += -ae * g2x + (dey * g2z - dez * g2y);
    dd0y += -ae * g2y + (dez * g2x - dex * g2z);
    dd0z += -ae * g2z + (dex * g2y - dey * g2x);
    const float dc[8] = {da0, dd0x, dd0y, dd0z, dae, ddex, ddey, ddez};
    // c = bl / n, n = |bl[:4]|
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot += dc[k] * bl[k];
    const float dn = -dot / (nrm * nrm);
    float dbl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dbl[k] = dc[k] / nrm + (k < 4 ? dn * bl[k] / nrm : 0.f);
    // blend: d skin_b = dbl . q_b (+ direct g_skin) ; d q_b += skin_b dbl
    float sdot = 0.f;   // sum_j skin_j dskin_j for the softmax backward
    for (int b = 0; b < B; ++b) {
        float ds = (g_skin && live) ? (STAGE ? grow[b] : g_skin[i * B + b]) : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ds = fmaf(dbl[k], Q0[b * 8 + k], ds);
        sdot = fmaf(STAGE ? srow[b] : sk[b], ds, sdot);
    }
    if (live) {   // d q_b = sum_s skin_b dbl is reduced per (ray, bone) by warp_bwd_reduce_kernel
#pragma unroll
        for (int k = 0; k < 8; ++k) d_bl[i * 8 + k] = dbl[k];
    }
    const float G = -10.f * 100.f * e_aux;   // logit = G * sum_k s_k m_k^2 + dskin
    float dlx = 0.f, dly = 0.f, dlz = 0.f;
    for (int b = 0; b < B; ++b) {
        float ds = (g_skin && live) ? (STAGE ? grow[b] : g_skin[i * B + b]) : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ds = fmaf(dbl[k], Q0[b * 8 + k], ds);
        const float dl = (STAGE ? srow[b] : sk[b]) * (ds - sdot);   // softmax backward
        if (STAGE) {
            if (live) srow[b] = dl;             // in place: this thread's own row, its weight just read
        } else if (d_dskin && live) {
            d_dskin[i * B + b] = dl;
        }
        const float* P = P0 + b * 16;
        const float ex = P[0] - lx, ey = P[1] - ly, ez = P[2] - lz;
        const float m0 = P[3] * ex + P[6] * ey + P[9] * ez;
        const float m1 = P[4] * ex + P[7] * ey + P[10] * ez;
        const float m2 = P[5] * ex + P[8] * ey + P[11] * ez;
        const float dm0 = dl * G * 2.f * P[12] * m0, dm1 = dl * G * 2.f * P[13] * m1, dm2 = dl * G * 2.f * P[14] * m2;
        // m_k = sum_j R[j][k] e_j  (R row-major at P[3 + 3j + k])
        const float dex_ = P[3] * dm0 + P[4] * dm1 + P[5] * dm2;
        const float dey_ = P[6] * dm0 + P[7] * dm1 + P[8] * dm2;
        const float dez_ = P[9] * dm0 + P[10] * dm1 + P[11] * dm2;
        dlx -= dex_; dly -= dey_; dlz -= dez_;
    }
    if (pts_tf) {   // two points: the transform's gradient and the weights' gradient go to their own inputs
        if (d_pts_tf && live) { d_pts_tf[i * 3] = dpx; d_pts_tf[i * 3 + 1] = dpy; d_pts_tf[i * 3 + 2] = dpz; }
        if (d_pts && live) { d_pts[i * 3] = dlx; d_pts[i * 3 + 1] = dly; d_pts[i * 3 + 2] = dlz; }
    } else if (d_pts && live) {
        d_pts[i * 3] = dpx + dlx; d_pts[i * 3 + 1] = dpy + dly; d_pts[i * 3 + 2] = dpz + dlz;
    }
    if (STAGE && d_dskin) {
        __syncthreads();
        float* dst = d_dskin + i_base * B;
        const int cnt = rows_here * B;
        if ((((uintptr_t)dst) & 15) == 0) {
            for (int j = threadIdx.x; j < cnt / 4; j += 256) ((float4*)dst)[j] = ((const float4*)wt)[j];
            for (int j = (cnt / 4) * 4 + threadIdx.x; j < cnt; j += 256) dst[j] = wt[j];
        } else {
            for (int j = threadIdx.x; j < cnt; j += 256) dst[j] = wt[j];
        }
    }
}

// Second stage: everything that is summed over the samples of a ray, per (ray, bone), without atomics.
// One workgroup per ray; thread t handles bone t & 31 on sample slice t >> 5 (8 slices), LDS tree over the slices.
//   d_q[n,b,:]    = sum_s skin[n,s,b] * d_bl[n,s,:]
//   d_prep[n,b,:] = sum_s dlogit[n,s,b] * d(logit)/d(prep)         (written per ray; summed over rays by the caller
//                                                                    when the bones are shared)
//   d_aux0       += sum dlogit * gaussian logit
__global__ __launch_bounds__(256) void warp_bwd_reduce_kernel(const float* __restrict__ prep, int per_ray,
                                                             const float* __restrict__ pts, const float* __restrict__ skin,
                                                             const float* __restrict__ dl, const float* __restrict__ d_bl,
                                                             const float* __restrict__ skin_aux, long long S, int B,
                                                             float* __restrict__ d_prep_ray, float* __restrict__ d_q,
                                                             float* __restrict__ d_aux0) {
    __shared__ float red[8][32][25];
    const long long n = blockIdx.x;
    const int b = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float G = -10.f * 100.f * expf(skin_aux[0]);
    float acc[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) acc[k] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int bb = b0 + b;
        const bool ok = bb < B;
        float P[16];
        if (ok) {
            const float* Pp = prep + ((per_ray ? n * B : 0) + bb) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) P[k] = Pp[k];
        }
#pragma unroll
        for (int k = 0; k < 25; ++k) acc[k] = 0.f;
        if (ok) {
            // (four samples per trip: their ~14 loads each are issued together -- one sample per trip was one exposed memory
            //  round trip per sample, 16 in a row for S = 128; the sums keep their order)
#pragma unroll 4
            for (long long s = g; s < S; s += 8) {
                const long long i = n * S + s;
                const float w = skin[i * B + bb], d = dl[i * B + bb];
                const float* v = d_bl + i * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, v[k], acc[k]);
                const float ex = P[0] - pts[i * 3], ey = P[1] - pts[i * 3 + 1], ez = P[2] - pts[i * 3 + 2];
                const float m0 = P[3] * ex + P[6] * ey + P[9] * ez;
                const float m1 = P[4] * ex + P[7] * ey + P[10] * ez;
                const float m2 = P[5] * ex + P[8] * ey + P[11] * ez;
                const float dm0 = d * G * 2.f * P[12] * m0, dm1 = d * G * 2.f * P[13] * m1, dm2 = d * G * 2.f * P[14] * m2;
                acc[8] += P[3] * dm0 + P[4] * dm1 + P[5] * dm2;      // d c
                acc[9] += P[6] * dm0 + P[7] * dm1 + P[8] * dm2;
                acc[10] += P[9] * dm0 + P[10] * dm1 + P[11] * dm2;
                acc[11] += ex * dm0; acc[12] += ex * dm1; acc[13] += ex * dm2;   // d R (row-major)
                acc[14] += ey * dm0; acc[15] += ey * dm1; acc[16] += ey * dm2;
                acc[17] += ez * dm0; acc[18] += ez * dm1; acc[19] += ez * dm2;
                acc[20] += d * G * m0 * m0; acc[21] += d * G * m1 * m1; acc[22] += d * G * m2 * m2;   // d s
                acc[23] += d * G * (P[12] * m0 * m0 + P[13] * m1 * m1 + P[14] * m2 * m2);             // d aux0
            }
        }
#pragma unroll
        for (int k = 0; k < 25; ++k) red[g][b][k] = acc[k];
        __syncthreads();
        if (g == 0 && ok) {
            float t[25];
#pragma unroll
            for (int k = 0; k < 25; ++k) {
                t[k] = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) t[k] += red[gg][b][k];
            }
            float* dq = d_q + (n * B + bb) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dq[k] = t[k];
            float* dp = d_prep_ray + (n * B + bb) * 16;
#pragma unroll
            for (int k = 0; k < 15; ++k) dp[k] = t[8 + k];
            dp[15] = 0.f;
            atomicAdd(d_aux0, t[23]);
        }
        __syncthreads();
    }
}

}   // namespace

extern "C" int moda_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float* dz, void* stream) {
    if (n <= 0) return 0;
    if (!dy || !y || !dz) return MODA_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, (long long)n, act, dz);
    return (int)hipGetLastError();
}

extern "C" int moda_composite_bwd(const float* rgbsigma, const float* feat, int32_t F, const float* z_vals, const float* rays_d,
                                  const float* beta, const float* noise, const float* xyz, const float* clip_bound,
                                  const float* vis_pred, const float* cyc, const float* weights, const float* visibility,
                                  float rgb_filter_scale, int64_t N, int64_t S, const float* g_rgb, const float* g_feat,
                                  const float* g_depth, const float* g_sil, const float* g_weights, const float* g_cyc,
                                  float* d_rgbsigma, float* d_feat, float* d_z, float* d_rays_d, float* d_beta, float* d_cyc,
                                  void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!rgbsigma || !z_vals || !rays_d || !beta || !weights || !visibility || !d_rgbsigma) return MODA_EINVAL;
    if (feat && (F < 1 || F > 16)) return MODA_ESHAPE;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rgbsigma, feat, F,
                       z_vals, rays_d, beta, noise, xyz, clip_bound, vis_pred, cyc, weights, visibility, rgb_filter_scale,
                       (long long)N, (long long)S, g_rgb, g_feat, g_depth, g_sil, g_weights, g_cyc, d_rgbsigma, d_feat, d_z,
                       d_rays_d, d_beta, d_cyc);
    return (int)hipGetLastError();
}

extern "C" int moda_points_bwd(const float* d_xyz, const float* z_vals, const float* rays_d, int64_t N, int64_t S, float* d_rays_o,
                               float* d_rays_d, float* d_z, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!d_xyz || !z_vals || !rays_d) return MODA_EINVAL;
    hipLaunchKernelGGL(points_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_xyz, z_vals, rays_d,
                       (long long)N, (long long)S, d_rays_o, d_rays_d, d_z);
    return (int)hipGetLastError();
}

extern "C" int moda_warp_prepped_bwd(const float* prep, int32_t per_ray, const float* q, const float* pts, const float* pts_tf,
                                     float* d_pts_tf, const float* skin,
                                     const float* skin_aux, const float* cyc_ref, const float* g_out, const float* g_cyc,
                                     const float* g_skin, int64_t N, int64_t S, int32_t B, float* d_pts, float* d_dskin,
                                     float* d_prep_ray, float* d_q, float* d_aux0, float* d_ref, float* d_bl, void* stream) {
    if (N <= 0 || S <= 0 || B <= 0) return 0;
    if (!prep || !q || !pts || !skin || !skin_aux || !d_dskin || !d_prep_ray || !d_q || !d_aux0 || !d_bl) return MODA_EINVAL;
    if (pts_tf && !d_pts_tf) return MODA_EINVAL;
    const size_t tile_bytes = (size_t)(g_skin ? 2 : 1) * 256 * B * sizeof(float);
    if (tile_bytes <= 64 * 1024)
        hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3((unsigned)((N * S + 255) / 256)), dim3(256), tile_bytes, (hipStream_t)stream, prep,
                           per_ray, q, pts, pts_tf, d_pts_tf, skin, skin_aux, cyc_ref, g_out, g_cyc, g_skin, (long long)N, (long long)S, B,
                           d_pts, d_dskin, d_bl, d_ref);
    else
        hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3((unsigned)((N * S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prep, per_ray,
                           q, pts, pts_tf, d_pts_tf, skin, skin_aux, cyc_ref, g_out, g_cyc, g_skin, (long long)N, (long long)S, B, d_pts,
                           d_dskin, d_bl, d_ref);
    hipLaunchKernelGGL(warp_bwd_reduce_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, prep, per_ray, pts, skin, d_dskin,
                       d_bl, skin_aux, (long long)S, B, d_prep_ray, d_q, d_aux0);
    return (int)hipGetLastError();
}

// ================================================================================================
// Correspondence heads of inference_deform (rendering.py:439-499): camera projection and flow rendering
// ================================================================================================
namespace {

// obj_to_cam (geom_utils.py:567-581) + pinhole_cam (geom_utils.py:654-673) with K = mat2K(Kmatinv(Kinv))
// (rendering.py:443-449).  rtk (N,21) = [R row-major 9 | T 3 | Kinv row-major 9]; out (N,S,3) = (u, v, Z).
struct Cam { float R[9], T[3], fx, fy, px, py; };

DEVINL Cam load_cam(const float* __restrict__ r) {
    Cam c;
#pragma unroll
    for (int k = 0; k < 9; ++k) c.R[k] = r[k];
    c.T[0] = r[9]; c.T[1] = r[10]; c.T[2] = r[11];
    const float k0 = r[12], k1 = r[16], k2 = r[14], k3 = r[17];   // mat2K(Kinv)
    c.fx = 1.f / k0; c.fy = 1.f / k1; c.px = -k2 / k0; c.py = -k3 / k1;   // K2inv
    return c;
}

__global__ void project_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ rtk, long long N, long long S,
                                   float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const Cam c = load_cam(rtk + (i / S) * 21);
    const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    const float X = c.R[0] * x + c.R[1] * y + c.R[2] * z + c.T[0];
    const float Y = c.R[3] * x + c.R[4] * y + c.R[5] * z + c.T[1];
    const float Z = c.R[6] * x + c.R[7] * y + c.R[8] * z + c.T[2];
    const float den = 1e-6f + Z;
    out[i * 3] = (c.fx * X + c.px * Z) / den;
    out[i * 3 + 1] = (c.fy * Y + c.py * Z) / den;
    out[i * 3 + 2] = Z;
}

// one wave per ray; d_rtk (N,21) written, d_xyz (N,S,3) written
__global__ __launch_bounds__(256) void project_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ rtk,
                                                         const float* __restrict__ g, long long N, long long S,
                                                         float* __restrict__ d_xyz, float* __restrict__ d_rtk) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* r = rtk + n * 21;
    const Cam c = load_cam(r);
    float aR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, aT[3] = {0, 0, 0}, afx = 0.f, afy = 0.f, apx = 0.f, apy = 0.f;
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        const float X = c.R[0] * x + c.R[1] * y + c.R[2] * z + c.T[0];
        const float Y = c.R[3] * x + c.R[4] * y + c.R[5] * z + c.T[1];
        const float Z = c.R[6] * x + c.R[7] * y + c.R[8] * z + c.T[2];
        const float den = 1e-6f + Z;
        const float nu = c.fx * X + c.px * Z, nv = c.fy * Y + c.py * Z;
        const float gu = g[i * 3], gv = g[i * 3 + 1], gz = g[i * 3 + 2];
        // u = nu/den, v = nv/den
        const float dnu = gu / den, dnv = gv / den;
        const float dX = dnu * c.fx, dY = dnv * c.fy;
        const float dZ = gz + dnu * c.px + dnv * c.py - (gu * nu + gv * nv) / (den * den);
        afx += dnu * X; apx += dnu * Z; afy += dnv * Y; apy += dnv * Z;
        d_xyz[i * 3] = c.R[0] * dX + c.R[3] * dY + c.R[6] * dZ;
        d_xyz[i * 3 + 1] = c.R[1] * dX + c.R[4] * dY + c.R[7] * dZ;
        d_xyz[i * 3 + 2] = c.R[2] * dX + c.R[5] * dY + c.R[8] * dZ;
        aR[0] += dX * x; aR[1] += dX * y; aR[2] += dX * z;
        aR[3] += dY * x; aR[4] += dY * y; aR[5] += dY * z;
        aR[6] += dZ * x; aR[7] += dZ * y; aR[8] += dZ * z;
        aT[0] += dX; aT[1] += dY; aT[2] += dZ;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) aR[k] = wave_sum_f(aR[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) aT[k] = wave_sum_f(aT[k]);
    afx = wave_sum_f(afx); afy = wave_sum_f(afy); apx = wave_sum_f(apx); apy = wave_sum_f(apy);
    if (lane == 0) {
        float* d = d_rtk + n * 21;
#pragma unroll
        for (int k = 0; k < 21; ++k) d[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = aR[k];
        d[9] = aT[0]; d[10] = aT[1]; d[11] = aT[2];
        // fx = 1/k0, px = -k2/k0 ; fy = 1/k1, py = -k3/k1
        const float k0 = r[12], k1 = r[16], k2 = r[14], k3 = r[17];
        d[12] = -afx / (k0 * k0) + apx * k2 / (k0 * k0);
        d[14] = -apx / k0;
        d[16] = -afy / (k1 * k1) + apy * k3 / (k1 * k1);
        d[17] = -apy / k1;
    }
}

// vrender_flo (geom_utils.py:1704-1743): one wave per ray.
//   invalid_i = Z_i < 1e-5 | |xy_i| > 2 img_size ; w'_i = w_i [valid] ; wn = w' / (1e-9 + sum w')
//   flo = sum wn_i (xy_i [valid] - xys) / img_size * 2 ; valid = no invalid sample on the ray
__global__ __launch_bounds__(256) void flow_render_kernel(const float* __restrict__ w, const float* __restrict__ proj,
                                                         const float* __restrict__ xys, float img_size, long long N, long long S,
                                                         float* __restrict__ flo, float* __restrict__ valid,
                                                         const float* __restrict__ g_flo, float* __restrict__ d_w,
                                                         float* __restrict__ d_proj) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float ox = xys[n * 2], oy = xys[n * 2 + 1];
    float sw = 0.f, sx = 0.f, sy = 0.f, ninv = 0.f;
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        const float u = proj[i * 3], v = proj[i * 3 + 1], Z = proj[i * 3 + 2];
        const bool inv = (Z < 1e-5f) || (sqrtf(u * u + v * v) > 2.f * img_size);
        const float ww = inv ? 0.f : w[i];
        sw += ww;
        sx += ww * ((inv ? 0.f : u) - ox);
        sy += ww * ((inv ? 0.f : v) - oy);
        ninv += inv ? 1.f : 0.f;
    }
    sw = wave_sum_f(sw); sx = wave_sum_f(sx); sy = wave_sum_f(sy); ninv = wave_sum_f(ninv);
    const float den = 1e-9f + sw;
    const float sc = 2.f / img_size;
    if (!g_flo) {
        if (lane == 0) {
            flo[n * 2] = sx / den * sc;
            flo[n * 2 + 1] = sy / den * sc;
            valid[n] = ninv == 0.f ? 1.f : 0.f;
        }
        return;
    }
    // backward: F = sc * A / den, A = sum w'_i (xy_i - o), den = eps + sum w'_i
    const float gx = g_flo[n * 2] * sc, gy = g_flo[n * 2 + 1] * sc;
    const float common = -(gx * sx + gy * sy) / (den * den);
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        const float u = proj[i * 3], v = proj[i * 3 + 1], Z = proj[i * 3 + 2];
        const bool inv = (Z < 1e-5f) || (sqrtf(u * u + v * v) > 2.f * img_size);
        const float ww = inv ? 0.f : w[i];
        d_w[i] = inv ? 0.f : (gx * (u - ox) + gy * (v - oy)) / den + common;
        d_proj[i * 3] = inv ? 0.f : ww * gx / den;
        d_proj[i * 3 + 1] = inv ? 0.f : ww * gy / den;
        d_proj[i * 3 + 2] = 0.f;
    }
}

// compute_pts_exp (loss_utils.py:165-175): out (N,3) = sum_s w_s / (1e-9 + sum w) * pts_s
__global__ __launch_bounds__(256) void pts_exp_kernel(const float* __restrict__ w, const float* __restrict__ pts, long long N,
                                                     long long S, float* __restrict__ out, const float* __restrict__ g,
                                                     float* __restrict__ d_w, float* __restrict__ d_pts) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float sw = 0.f, a[3] = {0.f, 0.f, 0.f};
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        const float ww = w[i];
        sw += ww;
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] += ww * pts[i * 3 + k];
    }
    sw = wave_sum_f(sw);
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = wave_sum_f(a[k]);
    const float den = 1e-9f + sw;
    if (!g) {
        if (lane == 0)
            for (int k = 0; k < 3; ++k) out[n * 3 + k] = a[k] / den;
        return;
    }
    const float g0 = g[n * 3], g1 = g[n * 3 + 1], g2 = g[n * 3 + 2];
    const float common = -(g0 * a[0] + g1 * a[1] + g2 * a[2]) / (den * den);
    for (long long s = lane; s < S; s += 64) {
        const long long i = n * S + s;
        d_w[i] = (g0 * pts[i * 3] + g1 * pts[i * 3 + 1] + g2 * pts[i * 3 + 2]) / den + common;
        const float ww = w[i] / den;
        d_pts[i * 3] = ww * g0; d_pts[i * 3 + 1] = ww * g1; d_pts[i * 3 + 2] = ww * g2;
    }
}

}   // namespace

extern "C" int moda_project_fwd(const float* xyz, const float* rtk_vec, int64_t N, int64_t S, float* out, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!xyz || !rtk_vec || !out) return MODA_EINVAL;
    hipLaunchKernelGGL(project_fwd_kernel, dim3((unsigned)((N * S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, rtk_vec,
                       (long long)N, (long long)S, out);
    return (int)hipGetLastError();
}

extern "C" int moda_project_bwd(const float* xyz, const float* rtk_vec, const float* g_out, int64_t N, int64_t S, float* d_xyz,
                                float* d_rtk_vec, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!xyz || !rtk_vec || !g_out || !d_xyz || !d_rtk_vec) return MODA_EINVAL;
    hipLaunchKernelGGL(project_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xyz, rtk_vec, g_out,
                       (long long)N, (long long)S, d_xyz, d_rtk_vec);
    return (int)hipGetLastError();
}

extern "C" int moda_flow_render(const float* weights, const float* proj, const float* xys, float img_size, int64_t N, int64_t S,
                                float* flo, float* valid, const float* g_flo, float* d_weights, float* d_proj, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!weights || !proj || !xys) return MODA_EINVAL;
    if (!g_flo && (!flo || !valid)) return MODA_EINVAL;
    if (g_flo && (!d_weights || !d_proj)) return MODA_EINVAL;
    hipLaunchKernelGGL(flow_render_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, weights, proj, xys,
                       img_size, (long long)N, (long long)S, flo, valid, g_flo, d_weights, d_proj);
    return (int)hipGetLastError();
}

extern "C" int moda_pts_exp(const float* weights, const float* pts, int64_t N, int64_t S, float* out, const float* g_out,
                            float* d_weights, float* d_pts, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!weights || !pts) return MODA_EINVAL;
    if (!g_out && !out) return MODA_EINVAL;
    if (g_out && (!d_weights || !d_pts)) return MODA_EINVAL;
    hipLaunchKernelGGL(pts_exp_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, weights, pts, (long long)N,
                       (long long)S, out, g_out, d_weights, d_pts);
    return (int)hipGetLastError();
}

// ================================================================================================
// Whole-network training schedule (autograd.NerfFn): every launch of one NeRF's forward or backward from ONE call,
// so that a training step spends its host time on ~16 entry points instead of ~1000 Python-level launches.
// Same kernels and the same order of operations as the per-layer calls they replace.
// ================================================================================================
namespace {

__global__ void sigmoid_bwd_strided_kernel(const float* __restrict__ g, long long ldg, const float* __restrict__ y, long long ldy,
                                           long long M, int n, float* __restrict__ dz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * n) return;
    const long long m = i / n;
    const int c = (int)(i - m * n);
    const float v = y[m * ldy + c];
    dz[i] = g[m * ldg + c] * v * (1.f - v);
}

// fp32 matrices -> zero-padded bf16 copies, every matrix of a network's backward in one launch (blockIdx.y = entry)
struct WPrepEntry { const float* src; unsigned short* dst; int rows, cols, ld, rows_pad, ld_dst; };
struct WPrepArgs { WPrepEntry e[14]; };
__global__ __launch_bounds__(256) void wprep_kernel(WPrepArgs a) {
    const WPrepEntry e = a.e[blockIdx.y];
    const int n = e.rows_pad * e.ld_dst;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int r = i / e.ld_dst, c = i - r * e.ld_dst;
        const float v = (r < e.rows && c < e.cols) ? e.src[(long long)r * e.ld + c] : 0.f;
        e.dst[i] = (unsigned short)(g2_pack2(v, 0.f) & 0xffffu);
    }
}

// Head of the bf16-storage backward, four threads per sample: dzb[m] = 32 bf16 = d(loss)/d(rgb pre-activation) (through the
// sigmoid unless raw_feat) zero-padded, and the 8 bf16 at dzd_tail[m] = (d_sigma | 0, 0 x 7) -- the extra k columns of
// [d_dir_encoding | d_sigma] @ [W_dir W_final ; W_sigma].  Thread 4m + q writes the 16 bytes of columns 8q .. 8q+7: a wave stores
// 1 KB contiguous per instruction (one thread per sample stored 64 lanes x 16 B at a 64-byte stride, four times: 17 us at
// M = 262 144 for 17 MB).
__global__ __launch_bounds__(256) void head_prep_kernel(const float* __restrict__ g, const float* __restrict__ y, long long ldo, long long M,
                                                       int n_out, int raw_feat, unsigned short* __restrict__ dzb,
                                                       unsigned short* __restrict__ dzd_tail, long long ld_tail,
                                                       float* __restrict__ zero_p, int zero_n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < zero_n) zero_p[t] = 0.f;        // T and s of the head products (they accumulate): one memset launch less per network
    const long long m = t >> 2;
    const int q = (int)(t & 3);
    if (m >= M) return;
    unsigned w[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 8 * q + 2 * c2 + e;
            v[e] = 0.f;
            if (c < n_out) {
                v[e] = g[m * ldo + c];
                if (!raw_feat) {
                    const float o = y[m * ldo + c];
                    v[e] *= o * (1.f - o);
                }
            }
        }
        w[c2] = g2_pack2(v[0], v[1]);
    }
    *(uint4*)(dzb + m * 32 + 8 * q) = make_uint4(w[0], w[1], w[2], w[3]);
    if (q == 0) {
        const float ds = raw_feat ? 0.f : g[m * ldo + n_out];
        *(uint4*)(dzd_tail + m * ld_tail) = make_uint4(g2_pack2(ds, 0.f), 0u, 0u, 0u);
    }
}

// The small products that finish the folded heads of moda_nerf_train_bwd (W2 = W / 2; T = dzd^T h (W2, W), s = 1^T dzd (W2)):
//   g_dir[:, :W] += T Wfin^T + s bfin^T      g_fin += Wdh^T T      g_bfin += Wdh^T s      g_bdir += s (when asked)
// as ONE launch of exact-fp32 FMA chains (they were four generic GEMM launches and a column sum, 12-15 us each for 17 M MACs
// at W = 256).  One thread per output element, k ascending.
__global__ __launch_bounds__(256) void head_finish_kernel(const float* __restrict__ Tm, const float* __restrict__ svec,
                                                          const float* __restrict__ Wfin, const float* __restrict__ bfin,
                                                          const float* __restrict__ Wdir, long long ldd, int W,
                                                          float* __restrict__ g_dir, float* __restrict__ g_fin,
                                                          float* __restrict__ g_bfin, float* __restrict__ g_bdir, int vec) {
    const int W2 = W / 2;
    const long long n1 = (long long)W2 * W, n2 = (long long)W * W;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n1) {                                  // g_dir[i, n] += sum_k T[i, k] Wfin[n, k] + s[i] bfin[n]
        const int i = (int)(e / W), n = (int)(e % W);
        const float4* t4 = (const float4*)(Tm + (long long)i * W);
        const float4* w4 = (const float4*)(Wfin + (long long)n * W);
        float acc = 0.f;
        if (vec) {                                 // both rows 16-byte aligned
            for (int k = 0; k < W / 4; ++k) {
                const float4 a = t4[k], b = w4[k];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        } else {
            for (int k = 0; k < W; ++k) acc = fmaf(Tm[(long long)i * W + k], Wfin[(long long)n * W + k], acc);
        }
        acc = fmaf(svec[i], bfin[n], acc);
        g_dir[(long long)i * ldd + n] += acc;
        return;
    }
    e -= n1;
    if (e < n2) {                                  // g_fin[m, n] += sum_k Wdir[k, m] T[k, n]
        const int m = (int)(e / W), n = (int)(e % W);
        float acc = 0.f;
        for (int k = 0; k < W2; ++k) acc = fmaf(Wdir[(long long)k * ldd + m], Tm[(long long)k * W + n], acc);
        g_fin[(long long)m * W + n] += acc;
        return;
    }
    e -= n2;
    if (e < W) {                                   // g_bfin[m] += sum_k Wdir[k, m] s[k]
        float acc = 0.f;
        for (int k = 0; k < W2; ++k) acc = fmaf(Wdir[(long long)k * ldd + e], svec[k], acc);
        g_bfin[e] += acc;
        return;
    }
    e -= W;
    if (g_bdir != nullptr && e < W2) g_bdir[e] += svec[e];
}

struct Net {
    const moda_nerf_train_desc* d;
    hipStream_t st;
    int rc = 0;
    int dt = 0;                                   // storage-type flags of the NEXT gemm / gemm_tn call (consumed by it)
    bool ex = false;                              // the NEXT call is exact fp32 whatever the network's precision mode
    void* bits = nullptr;                         // sign-bit map of the NEXT gemm / gemm_tn call (written by the dW form, read by dX)
    long long ldbits = 0;
    Net& with(int f) { dt = f; return *this; }
    Net& signs(void* p, long long ld) { bits = p; ldbits = ld; return *this; }
    Net& exact() { ex = true; return *this; }
    // the NEXT call keeps its fp32 operands at split-bf16 accuracy (MODA_GEMM_BF16X3) when the network runs in the bf16 mode: the
    // small per-ray products on fp32 operands -- 21-33 us on the generic kernel's bf16 path, 11 us on gemm_x3.hip's dW form
    Net& fine() { x3next = true; return *this; }
    bool x3next = false;
    void gemm(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C, long long ldc,
              long long M, long long N, long long K, const float* bias = nullptr, int act = 0, const float* mask = nullptr,
              long long ldm = 0, int acc = 0, int split = 1, const float* A2 = nullptr, long long sam2 = 0, long long K1 = 0,
              const float* rb = nullptr, long long ldrb = 0, long long rpb = 1, float* asum = nullptr) {
        if (rc) return;
        moda_gemm_desc g;
        g.a_sum = asum;
        g.A = A; g.sam = sam; g.sak = sak; g.A2 = A2; g.sam2 = sam2; g.K1 = A2 ? K1 : K;
        g.B = B; g.sbk = sbk; g.sbn = sbn; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
        g.bias = bias; g.rowbias = rb; g.ld_rowbias = ldrb; g.rows_per_bias = rpb; g.mask_src = mask; g.ld_mask = ldm;
        g.act = act; g.accumulate = acc; g.split_k = split; g.reserved = ex ? 0 : ((d->reserved & (MODA_GEMM_BF16 | MODA_GEMM_BF16X3 | MODA_GEMM_BF16X6)) | dt);
        if (x3next && !ex && dt == 0 && (d->reserved & MODA_GEMM_BF16)) g.reserved = MODA_GEMM_BF16X3;
        x3next = false;
        g.mask_bits = bits; g.ld_bits = ldbits;
        dt = 0;
        ex = false;
        bits = nullptr; ldbits = 0;
        rc = moda_gemm_f32_ex(&g, st);
    }
    static int split_k(long long M, long long rows, long long cols) {
        const long long tiles = ((rows + 127) / 128) * (cols > 64 ? (cols + 127) / 128 : 1);
        static const long long target = [] { const char* e = getenv("MODA_SPLITK_TARGET"); return e ? atoll(e) : 512LL; }();
        long long s = target / tiles;
        if (s > M / 256) s = M / 256;
        return s < 1 ? 1 : (int)s;
    }
    // dW (rows x cols; ldc) += dz^T (rows x M) @ x (M x cols);  db (rows) += column sums of dz when given (same pass over dz)
    void gemm_tn(const float* dz, long long ldz, const float* x, long long ldx, float* dW, long long ldc, long long M,
                 long long rows, long long cols, float* db = nullptr) {
        gemm(dz, 1, ldz, x, ldx, 1, dW, ldc, rows, cols, M, nullptr, 0, nullptr, 0, 1, split_k(M, rows, cols), nullptr, 0, 0,
             nullptr, 0, 1, db);
    }
    void colsum(const float* x, long long M, long long N, long long ld, float* out, int bf = 0) {
        if (!rc) rc = colsum_any(x, M, N, ld, out, bf, st);
    }
    // sums over the S = M / R consecutive rows of each of R groups; R == 1 goes through the atomics-based column sum
    void segsum(const float* x, long long M, long long R, long long N, long long ld, float* out, int bf = 0) {
        if (rc) return;
        if (R == 1) {
            rc = (int)hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st);
            if (!rc) rc = colsum_any(x, M, N, ld, out, bf, st);
        } else {
            rc = segsum_any(x, R, M / R, N, ld, out, N, bf, st);
        }
    }
    void copy2d(float* dst, long long ldd, const float* src, long long lds, long long rows, long long cols) {
        if (!rc) rc = (int)hipMemcpy2DAsync(dst, (size_t)ldd * 4, src, (size_t)lds * 4, (size_t)cols * 4, (size_t)rows,
                                             hipMemcpyDeviceToDevice, st);
    }
    void zero(float* p, long long n) {
        if (!rc) rc = (int)hipMemsetAsync(p, 0, (size_t)n * sizeof(float), st);
    }
};

// workspace layout shared by forward and backward (offsets in floats)
struct WsLayout {
    long long pe, h, fin, dd, W1p, W5p, Wdh, rb1, rb5, rbd, total;
    long long Pp;
    explicit WsLayout(const moda_nerf_train_desc& d) {
        Pp = (d.P + 3) / 4 * 4;
        long long o = 0;
        auto take = [&](long long n) { const long long r = o; o += (n + 3) / 4 * 4; return r; };
        pe = take(d.M * Pp);
        h = take(d.M * d.W * d.D);
        fin = take(d.sigma_only ? 0 : d.M * d.W);
        dd = take(d.sigma_only ? 0 : d.M * (d.W / 2));
        W1p = take((long long)d.W * Pp);
        W5p = take((long long)d.W * (Pp + d.W));
        Wdh = take((long long)(d.W / 2) * d.W);
        rb1 = take(d.C1 ? d.R1 * d.W : 0);
        rb5 = take(d.C1 ? d.R1 * d.W : 0);
        rbd = take(d.Cd ? d.Rd * (d.W / 2) : 0);
        total = o;
    }
};

// xyz_encoding_final folded into dir_encoding (nerf.py:184-187, mlp_pack.fold_final): prod (W/2, W) = Wd[:, :W] Wf and
// bd'[o] = bd[o] + sum_k Wd[o, k] bf[k], fp32 fmaf chains in k order.  One workgroup per output row; the generic GEMM took
// 13-54 us for this 128 x 256 x 256 product (four tiles, K walked serially) plus a second launch for the bias, at every call.
__global__ __launch_bounds__(256) void fold_final_kernel(const float* __restrict__ Wd, int ldd, const float* __restrict__ Wf,
                                                         const float* __restrict__ bf, const float* __restrict__ bd, int W,
                                                         float* __restrict__ prod, float* __restrict__ bd_out) {
    __shared__ float wrow[256];
    const int o = blockIdx.x;
    const float* wd = Wd + (long long)o * ldd;
    for (int k = threadIdx.x; k < W; k += 256) wrow[k] = wd[k];
    __syncthreads();
    for (int j = threadIdx.x; j < W; j += 256) {
        float acc = 0.f;
        const float* col = Wf + j;
#pragma unroll 16
        for (int k = 0; k < W; ++k) acc = fmaf(wrow[k], col[(long long)k * W], acc);      // (loads of 16 steps in flight)
        prod[(long long)o * W + j] = acc;
    }
    if (bd_out != nullptr && threadIdx.x < 64) {          // one wavefront: the bias row (k order within a lane, lanes summed in order)
        float acc = 0.f;
        for (int k = threadIdx.x; k < W; k += 64) acc = fmaf(wrow[k], bf[k], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (threadIdx.x == 0) bd_out[o] = bd[o] + acc;
    }
}

static int fold_final_launch(const float* Wd, long long ldd, const float* Wf, const float* bf, const float* bd, long long W,
                             float* prod, float* bd_out, hipStream_t st) {
    hipLaunchKernelGGL(fold_final_kernel, dim3((unsigned)(W / 2)), dim3(256), 0, st, Wd, (int)ldd, Wf, bf, bd, (int)W, prod, bd_out);
    return (int)hipGetLastError();
}

// The positional encoding of the training forward as rows of PP = 64 columns (63 features of nerf.py:58-72 + a zero pad), eight
// consecutive features per thread: a wavefront writes eight whole rows (1 KiB as bf16, 2 KiB as fp32) where the general
// embed_kernel scatters 4-byte stores at a 12-byte pitch.  bf16 rows are what the bf16-storage backward reads as the X of its
// two PE weight-gradient GEMMs.
template <bool BF>
__global__ __launch_bounds__(256) void embed_rows64_kernel(const float* __restrict__ xyz, long long M, int F, Window win,
                                                           void* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = i >> 3;
    if (m >= M) return;
    const int g = (int)(i & 7);
    const float p[3] = {xyz[m * 3 + 0], xyz[m * 3 + 1], xyz[m * 3 + 2]};
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = 8 * g + e;
        float r = 0.f;
        if (f < 3) {
            r = p[f];
        } else if (f < 3 + 6 * F) {
            const int q = f - 3, k = q / 6, t = q - 6 * k, c = t >= 3 ? t - 3 : t;
            if (BF) {   // the encoding of the fused bf16 forward itself (PrecBF16::encode): hardware sine of the exactly scaled revolutions
                const float rev = __builtin_amdgcn_fractf(__builtin_fmaf(p[c] * 0.15915494309189535f, (float)(1 << k), t >= 3 ? 0.25f : 0.f));
                r = win.w[k] * __builtin_amdgcn_sinf(rev);
            } else {
                float sn, cs;
                sincos_rr(ldexpf(p[c], k), sn, cs);
                r = win.w[k] * (t >= 3 ? cs : sn);
            }
        }
        v[e] = r;
    }
    if (BF) {
        ((uint4*)out)[i] = make_uint4(g2_pack2(v[0], v[1]), g2_pack2(v[2], v[3]), g2_pack2(v[4], v[5]), g2_pack2(v[6], v[7]));
    } else {
        ((float4*)out)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
        ((float4*)out)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}   // namespace

extern "C" int64_t moda_nerf_train_ws_floats(const moda_nerf_train_desc* d) {
    if (!d || d->D < 5 || d->D > 8 || d->W < 32 || d->W % 4 || d->M <= 0) return -1;
    return WsLayout(*d).total;
}

// scratch of the backward: dh ping-pong, dzd, d_final, d_pe, dz_rgb, d_rb
extern "C" int64_t moda_nerf_train_scratch_floats(const moda_nerf_train_desc* d) {
    if (!d) return -1;
    const long long Pp = (d->P + 3) / 4 * 4;
    long long R = d->R1 > d->Rd ? d->R1 : d->Rd;
    // (+ the M-independent pieces of the bf16-storage backward: bf16 weight copies, the folded W_dir W_final, dzd^T h)
    // (+ the per-workgroup partial weight-gradient tiles of the 64-wide networks' chained backward, bwd64_chain.hip)
    return d->M * (2LL * d->W + d->W / 2 + d->W + Pp + d->n_out + 4 + 24) + R * d->W + 64 +
           (d->D + 4LL) * d->W * d->W + 4LL * d->W * Pp + 8192 + ((d->W == 64 && (d->reserved & MODA_TRAIN_BF16_STORE)) ? moda_chain64_part_floats(d->M) : 0);
}

// params: 2D + 8 device pointers in NeRF order: (W_i, b_i) for i < D, sigma (W,b), xyz_encoding_final (W,b), dir_encoding (W,b), rgb (W,b)
extern "C" int moda_nerf_train_fwd(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                                   const float* const* params, float* ws, float* out, void* stream) {
    if (!d || !xyz || !params || !ws || !out) return MODA_EINVAL;
    if ((d->C1 > 0) != (code != nullptr) || (d->Cd > 0 && !d->sigma_only) != (dir_src != nullptr)) return MODA_EINVAL;
    if (d->reserved & MODA_TRAIN_BF16_STORE) return MODA_EINVAL;      // bf16 storage belongs to the fused forward
    const WsLayout L(*d);
    const long long M = d->M, W = d->W, P = d->P, Pp = L.Pp, C1 = d->C1, Cd = d->sigma_only ? 0 : d->Cd, D = d->D;
    if (D != 8 && D != 5 && D != 6 && D != 7) return MODA_ESHAPE;
    Net n{d, (hipStream_t)stream};
    const float* const* Wt = params;
    auto Wl = [&](int l) { return Wt[2 * l]; };
    auto bl = [&](int l) { return Wt[2 * l + 1]; };
    const float *Wsig = Wt[2 * D], *bsig = Wt[2 * D + 1], *Wfin = Wt[2 * D + 2], *bfin = Wt[2 * D + 3], *Wdir = Wt[2 * D + 4],
                *bdir = Wt[2 * D + 5], *Wrgb = Wt[2 * D + 6], *brgb = Wt[2 * D + 7];
    const long long ld1 = P + C1, ld5 = P + C1 + W, ldd = W + d->Cd;
    float* pe = ws + L.pe;
    float* hs = ws + L.h;
    // positional encoding into the padded buffer (pad column zeroed), packed weight views
    n.zero(pe, M * Pp);
    if (!n.rc) n.rc = moda_embed_fwd(xyz, M, 3, d->n_freq, d->window, 0, pe, Pp, n.st);
    n.zero(ws + L.W1p, W * Pp);
    n.copy2d(ws + L.W1p, Pp, Wl(0), ld1, W, P);
    n.zero(ws + L.W5p, W * (Pp + W));
    n.copy2d(ws + L.W5p, Pp + W, Wl(4), ld5, W, P);
    n.copy2d(ws + L.W5p + Pp, Pp + W, Wl(4) + P + C1, ld5, W, W);
    const long long rpb1 = C1 ? M / d->R1 : 1, rpbd = Cd ? M / d->Rd : 1;
    if (C1) n.gemm(code, C1, 1, Wl(0) + P, 1, ld1, ws + L.rb1, W, d->R1, W, C1, bl(0));
    n.gemm(pe, Pp, 1, ws + L.W1p, 1, Pp, hs, W, M, W, Pp, C1 ? nullptr : bl(0), 1, nullptr, 0, 0, 1, nullptr, 0, 0,
           C1 ? ws + L.rb1 : nullptr, W, rpb1);
    for (int l = 1; l < D; ++l) {
        float* hin = hs + (long long)(l - 1) * M * W;
        float* hout = hs + (long long)l * M * W;
        if (l == 4) {
            if (C1) n.gemm(code, C1, 1, Wl(4) + P, 1, ld5, ws + L.rb5, W, d->R1, W, C1, bl(4));
            n.gemm(pe, Pp, 1, ws + L.W5p, 1, Pp + W, hout, W, M, W, Pp + W, C1 ? nullptr : bl(4), 1, nullptr, 0, 0, 1, hin, W, Pp,
                   C1 ? ws + L.rb5 : nullptr, W, rpb1);
        } else {
            n.gemm(hin, W, 1, Wl(l), 1, W, hout, W, M, W, W, bl(l), 1);
        }
    }
    const float* hD = hs + (long long)(D - 1) * M * W;
    const long long ldo = d->sigma_only ? 1 : (d->raw_feat ? d->n_out : d->n_out + 1);
    if (d->sigma_only || !d->raw_feat)
        n.gemm(hD, W, 1, Wsig, 1, W, out + (d->sigma_only ? 0 : d->n_out), ldo, M, 1, W, bsig);
    if (!d->sigma_only) {
        float* fin = ws + L.fin;
        float* dd = ws + L.dd;
        n.gemm(hD, W, 1, Wfin, 1, W, fin, W, M, W, W, bfin);
        n.copy2d(ws + L.Wdh, W, Wdir, ldd, W / 2, W);
        if (Cd) n.gemm(dir_src, Cd, 1, Wdir + W, 1, ldd, ws + L.rbd, W / 2, d->Rd, W / 2, Cd, bdir);
        n.gemm(fin, W, 1, ws + L.Wdh, 1, W, dd, W / 2, M, W / 2, W, Cd ? nullptr : bdir, 1, nullptr, 0, 0, 1, nullptr, 0, 0,
               Cd ? ws + L.rbd : nullptr, W / 2, rpbd);
        n.gemm(dd, W / 2, 1, Wrgb, 1, W / 2, out, ldo, M, d->n_out, W / 2, brgb, d->raw_feat ? 0 : 2);
    }
    return n.rc;
}

// The forward of the throughput mode: the SAME workspace contents as moda_nerf_train_fwd, produced by ONE launch of the fused
// bf16 PE+MLP kernel (moda_mlp_dump_fwd: every hidden layer's post-ReLU activations and the dir activations are written
// straight from the accumulators) instead of one GEMM per layer.  What stays outside: the positional encoding the backward
// reads (`pe`), the per-ray code folds, the packed weight views of the backward, and xyz_encoding_final's output (one GEMM:
// the fused kernel folds that layer into dir_encoding and never forms it, but dW of dir_encoding needs it).
// wstream / bias_block / bd_folded: NeRF._packed's bf16 stream, bias block and folded dir bias (bd + Wd[:, :W] bf).
extern "C" int moda_nerf_train_fwd_fused(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                                         const float* const* params, const void* wstream, const float* bias_block,
                                         const float* bd_folded, float* ws, float* out, void* stream) {
    if (!d || !xyz || !params || !ws || !out || !wstream || !bias_block || !bd_folded) return MODA_EINVAL;
    if ((d->C1 > 0) != (code != nullptr) || (d->Cd > 0) != (dir_src != nullptr)) return MODA_EINVAL;
    if (d->sigma_only || (d->W != 64 && d->W != 128 && d->W != 256) || d->D < 5 || d->D > 8 || d->n_freq > 10) return MODA_ESHAPE;
    const WsLayout L(*d);
    const long long M = d->M, W = d->W, P = d->P, Pp = L.Pp, C1 = d->C1, Cd = d->Cd, D = d->D;
    Net n{d, (hipStream_t)stream};
    const float* const* Wt = params;
    auto Wl = [&](int l) { return Wt[2 * l]; };
    auto bl = [&](int l) { return Wt[2 * l + 1]; };
    const float *Wfin = Wt[2 * D + 2], *bfin = Wt[2 * D + 3], *Wdir = Wt[2 * D + 4];
    const long long ld1 = P + C1, ld5 = P + C1 + W, ldd = W + d->Cd;
    float* pe = ws + L.pe;
    float* hs = ws + L.h;
    // the folded bf16-storage backward makes its own (bf16) copies of the weights straight from the parameters and reads the
    // positional encoding only as the X of dW products, where the pad column feeds an output column that is never stored:
    // no packed fp32 views and no zero fill of `pe` on that route (seven small launches and a 64 MB memset per network)
    const bool lean = (d->reserved & MODA_TRAIN_BF16_STORE) != 0 && d->n_out <= 32;
    if (lean && Pp == 64) {            // bf16 rows, pad column written as zero: read as such by the folded backward
        Window w;
        for (int i = 0; i < 16; ++i) w.w[i] = i < d->n_freq ? d->window[i] : 0.f;
        hipLaunchKernelGGL(embed_rows64_kernel<true>, dim3((unsigned)((M * 8 + 255) / 256)), dim3(256), 0, n.st, xyz, M, (int)d->n_freq, w,
                           (void*)pe);
        n.rc = (int)hipGetLastError();
    } else {
        if (!lean) n.zero(pe, M * Pp);
        if (!n.rc) n.rc = moda_embed_fwd(xyz, M, 3, d->n_freq, d->window, 0, pe, Pp, n.st);
    }
    if (!lean) {
        n.zero(ws + L.W1p, W * Pp);
        n.copy2d(ws + L.W1p, Pp, Wl(0), ld1, W, P);
        n.zero(ws + L.W5p, W * (Pp + W));
        n.copy2d(ws + L.W5p, Pp + W, Wl(4), ld5, W, P);
        n.copy2d(ws + L.W5p + Pp, Pp + W, Wl(4) + P + C1, ld5, W, W);
        n.copy2d(ws + L.Wdh, W, Wdir, ldd, W / 2, W);
    }
    if (C1) {
        n.gemm(code, C1, 1, Wl(0) + P, 1, ld1, ws + L.rb1, W, d->R1, W, C1, bl(0));
        n.gemm(code, C1, 1, Wl(4) + P, 1, ld5, ws + L.rb5, W, d->R1, W, C1, bl(4));
    }
    if (Cd) n.gemm(dir_src, Cd, 1, Wdir + W, 1, ldd, ws + L.rbd, W / 2, d->Rd, W / 2, Cd, bd_folded);
    if (n.rc) return n.rc;
    moda_mlp_desc md;
    const bool bst = (d->reserved & MODA_TRAIN_BF16_STORE) != 0;       // h / dd / fin held as bf16 (same element offsets)
    md.W = (int32_t)W; md.D = (int32_t)D; md.n_out = d->n_out; md.n_freq = d->n_freq; md.reserved = bst ? MODA_MLP_DUMP_BF16 : 0;
    md.overflow = nullptr;
    md.flags = MODA_MLP_BF16 | (d->raw_feat ? 0 : (MODA_MLP_SIGMOID | MODA_MLP_WITH_SIGMA));
    for (int i = 0; i < 16; ++i) md.window[i] = d->window[i];
    const long long R1 = C1 ? d->R1 : 1, Rd = Cd ? d->Rd : 1;
    const long long ldo = d->raw_feat ? d->n_out : d->n_out + 1;
    n.rc = moda_mlp_dump_fwd(&md, wstream, bias_block, xyz, nullptr, C1 ? ws + L.rb1 : bl(0), C1 ? ws + L.rb5 : bl(4), R1, M / R1,
                             Cd ? ws + L.rbd : bd_folded, Rd, M / Rd, out, ldo, hs, ws + L.dd, M, stream);
    const float* hD = hs + (long long)(D - 1) * M * W;
    // xyz_encoding_final's output: the backward of the bf16-storage mode does without it (store_heads_folded below)
    if (!(bst && d->n_out <= 32))
        n.with(bst ? MODA_GEMM_A_BF16 | MODA_GEMM_C_BF16 : 0).gemm(hD, W, 1, Wfin, 1, W, ws + L.fin, W, M, W, W, bfin);
    return n.rc;
}

// grads: 2D + 8 device pointers (same order and shapes as params); every parameter gradient is ADDED to what the buffer
// holds (zero it for a fresh gradient; several calls may share one buffer); g_out (M, ldo); d_xyz (M,3)|NULL;
// d_code (R1,C1)|NULL zeroed; d_dir (Rd,Cd)|NULL.
extern "C" int moda_nerf_train_bwd(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                                   const float* const* params, const float* ws, const float* out, const float* g_out,
                                   float* scratch, float* const* grads, float* d_xyz, float* d_code, float* d_dir,
                                   void* stream) {
    if (!d || !xyz || !params || !ws || !g_out || !scratch || !grads) return MODA_EINVAL;
    if (!d->sigma_only && !d->raw_feat && !out) return MODA_EINVAL;     // the sigmoid of the colour head differentiates through its output
    const WsLayout L(*d);
    const long long M = d->M, W = d->W, P = d->P, Pp = L.Pp, C1 = d->C1, Cd = d->sigma_only ? 0 : d->Cd, D = d->D;
    Net n{d, (hipStream_t)stream};
    const float* const* Wt = params;
    auto Wl = [&](int l) { return Wt[2 * l]; };
    const float *Wsig = Wt[2 * D], *Wfin = Wt[2 * D + 2], *Wdir = Wt[2 * D + 4], *Wrgb = Wt[2 * D + 6];
    auto gW = [&](int l) { return grads[2 * l]; };
    auto gb = [&](int l) { return grads[2 * l + 1]; };
    float *g_sig = grads[2 * D], *g_bsig = grads[2 * D + 1], *g_fin = grads[2 * D + 2], *g_bfin = grads[2 * D + 3],
          *g_dir = grads[2 * D + 4], *g_bdir = grads[2 * D + 5], *g_rgb = grads[2 * D + 6], *g_brgb = grads[2 * D + 7];
    const long long ld1 = P + C1, ld5 = P + C1 + W, ldd = W + d->Cd;
    const float* pe = ws + L.pe;
    const float* hs = ws + L.h;
    const float* hD = hs + (long long)(D - 1) * M * W;
    const long long ldo = d->sigma_only ? 1 : (d->raw_feat ? d->n_out : d->n_out + 1);
    // scratch carve-up
    float* dhA = scratch;
    float* dhB = dhA + M * W;
    float* dzd = dhB + M * W;
    float* dfin = dzd + M * (W / 2);
    float* dpe = dfin + M * W;
    float* dzrgb = dpe + M * Pp;
    float* drb = dzrgb + M * d->n_out + 4;
    float* dh = dhA;
    const long long R1 = C1 ? d->R1 : 1, Rd = Cd ? d->Rd : 1;
    // bf16 storage (MODA_TRAIN_BF16_STORE, set with the fused forward): the saved activations h / dd / fin and the big
    // gradient tensors dh / dnext / dzd / dfin are bf16 in memory -- the operand rounding of the bf16 GEMMs applied once at
    // the store instead of at every load, half the bytes.  dpe, dz_rgb, the per-ray sums and every parameter gradient stay fp32.
    const bool bst = (d->reserved & MODA_TRAIN_BF16_STORE) != 0 && !d->sigma_only;
    const int fA = bst ? MODA_GEMM_A_BF16 : 0, fB = bst ? MODA_GEMM_B_BF16 : 0, fC = bst ? MODA_GEMM_C_BF16 : 0,
              fM = bst ? MODA_GEMM_MASK_BF16 : 0, bfi = bst ? 1 : 0;
    // store_heads_folded: the bf16-storage route without xyz_encoding_final's output or its gradient as tensors, and with
    // bf16 copies of every weight the long GEMMs read (see the head section below)
    const bool folded = bst && d->n_out <= 32;
    const int fPE = (folded && L.Pp == 64) ? fB : 0;        // the fused forward of that route wrote the positional encoding as bf16 rows
    // 1-bit ReLU masks (folded route): the dW GEMM of a layer reads the layer below's saved activations h anyway and writes
    // their sign map as a by-product (W/8 bytes per sample, into the unused upper half of that layer's bf16 slot); the dX GEMM's
    // epilogue then reads 1 bit per element instead of 16 -- its traffic per 256-wide layer 0.40 -> 0.27 GB.
    static const bool g3_off = [] { const char* e = getenv("MODA_GEMM3"); return e && e[0] == '0'; }();
    static const bool bits_off = [] { const char* e = getenv("MODA_MASK_BITS"); return e && e[0] == '0'; }();
    const bool use_bits = folded && !g3_off && !bits_off && W % 64 == 0;
    // split-bf16 modes (fp32 storage): the same maps through gemm_x3.hip's dW / dX forms -- the dX epilogue of a 256-wide layer read
    // the 268 MB of fp32 activations as its mask (0.80 GB per launch, 200 us); with the map 0.54 GB.  ONE map buffer serves every
    // layer (a layer's dX launch follows its dW launch), carved from the per-sample spare of the scratch (M * W / 8 bytes).
    const char* xb_env = getenv("MODA_X3_MASK_BITS");          // read per call: an A/B switch for tests and tools
    const bool xbits_off = xb_env && xb_env[0] == '0';
    // (MODA_GEMM_X3=0 sends the split-bf16 forms back to the generic kernel, which has no sign-map epilogue: no maps then -- ADVICE r04)
    static const bool x3_off = [] { const char* e = getenv("MODA_GEMM_X3"); return e && e[0] == '0'; }();
    const bool use_xbits = !bst && !xbits_off && !x3_off && !d->sigma_only && (d->reserved & (MODA_GEMM_BF16X3 | MODA_GEMM_BF16X6)) != 0 &&
                           (W == 64 || W == 128 || W == 256) && ((uintptr_t)scratch & 15) == 0 && ((uintptr_t)ws & 15) == 0;
    void* const xbits = use_xbits ? (void*)(drb + ((R1 > Rd ? R1 : Rd) * W + 3) / 4 * 4) : nullptr;
    auto bits_of = [&](const float* slot, long long width) -> void* {      // upper half of a bf16 slot of M x width elements
        if (use_xbits) return width == W ? xbits : nullptr;
        return use_bits ? (void*)((unsigned char*)slot + M * width * 2) : nullptr;
    };
    const long long ldz2 = W / 2 + 8;              // row of [d_dir_encoding | d_sigma, 0 x 7] (bf16)
    unsigned short *wb_l[8] = {nullptr}, *wb_5pe = nullptr, *wb_1pe = nullptr, *wb_rgb = nullptr, *wb_ext = nullptr, *dzb = nullptr;
    float *Wpp = nullptr, *Tm = nullptr, *svec = nullptr, *chain_part = nullptr;
    if (folded) {
        float* p = scratch;
        auto take = [&](long long nf) { float* r = p; p += (nf + 3) / 4 * 4; return r; };
        dhA = take(M * W / 2); dhB = take(M * W / 2); dzd = take(M * ldz2 / 2);
        dzb = (unsigned short*)take(M * 16);
        dpe = take(M * Pp);
        drb = take((R1 > Rd ? R1 : Rd) * W);
        Wpp = take(W / 2 * W); Tm = take(W / 2 * W); svec = take(W);
        for (int l = 1; l < D; ++l) wb_l[l] = (unsigned short*)take(W * W / 2);
        wb_5pe = (unsigned short*)take(W * Pp / 2);
        wb_1pe = (unsigned short*)take(W * Pp / 2);
        wb_rgb = (unsigned short*)take(32 * (W / 2) / 2);
        wb_ext = (unsigned short*)take(ldz2 * W / 2);
        if (W == 64) chain_part = take(moda_chain64_part_floats(M));
        dh = dhA;
    }
    // per-ray column blocks of a weight gradient: a reduction over the R rays only -- split over k like the long ones
    auto ray_split = [](long long R) { return R >= 128 ? (int)(R / 32) : 1; };   // one 32-deep k-tile per workgroup: the output is a single tile
    if (d->sigma_only) {
        n.gemm(g_out, 1, 1, Wsig, W, 1, dh, W, M, W, 1, nullptr, 0, hD, W);
        n.gemm_tn(g_out, 1, hD, W, g_sig, W, M, 1, W);
        n.colsum(g_out, M, 1, 1, g_bsig);
    } else if (folded) {
        // ---- heads of the bf16-storage route.  With z_d the dir_encoding pre-activation gradient (dzd), h the last hidden
        //      layer's output, fin = h Wfin^T + bfin (never formed):
        //        T = dzd^T h,  s = 1^T dzd
        //        d Wdir[:, :W] = dzd^T fin = T Wfin^T + s bfin^T        d Wfin = (dzd Wdh)^T h = Wdh^T T      d bfin = Wdh^T s
        //        d h = ([dzd | d_sigma] @ [Wdh Wfin ; Wsig]) . [h > 0]
        //      one M-long GEMM (T) where the unfolded route has four (fin, d_fin, d Wfin, d Wdir), and K = W/2 + 8 instead
        //      of W for d h; the products of the small matrices are exact fp32.
        const float* dd = ws + L.dd;
        const float* bfin = Wt[2 * D + 3];
        if (!n.rc) n.rc = fold_final_launch(Wdir, ldd, Wfin, nullptr, nullptr, W, Wpp, nullptr, n.st);   // Wdh Wfin, Wdh = Wdir[:, :W]
        if (!n.rc) {
            WPrepArgs wa;
            int ne = 0;
            auto add = [&](const float* src, unsigned short* dst, long long rows, long long cols, long long ld, long long rows_pad, long long ld_dst) {
                wa.e[ne++] = WPrepEntry{src, dst, (int)rows, (int)cols, (int)ld, (int)rows_pad, (int)ld_dst};
            };
            for (int l = 1; l < D; ++l) {
                if (l == 4) add(Wl(4) + P + C1, wb_l[l], W, W, ld5, W, W);          // the h columns of the skip layer
                else add(Wl(l), wb_l[l], W, W, W, W, W);
            }
            add(Wl(4), wb_5pe, W, P, ld5, W, Pp);                                   // its PE columns, pad column zero
            add(Wl(0), wb_1pe, W, P, ld1, W, Pp);
            add(Wrgb, wb_rgb, d->n_out, W / 2, W / 2, 32, W / 2);
            add(Wpp, wb_ext, W / 2, W, W, W / 2, W);
            add(Wsig, wb_ext + (W / 2) * W, d->raw_feat ? 0 : 1, W, W, 8, W);
            hipLaunchKernelGGL(wprep_kernel, dim3(32, (unsigned)ne), dim3(256), 0, n.st, wa);
            hipLaunchKernelGGL(head_prep_kernel, dim3((unsigned)((4 * M + 255) / 256)), dim3(256), 0, n.st, g_out, out, ldo, M,
                               (int)d->n_out, (int)d->raw_feat, dzb, (unsigned short*)dzd + W / 2, ldz2, Tm, (int)((svec + W) - Tm));
            n.rc = (int)hipGetLastError();          // (it also zeroes T and s, adjacent in the scratch)
        }
        if ((svec + W) - Tm > 4 * M) n.zero(Tm, (svec + W) - Tm);      // fewer samples than words to zero: the memset after all
        const int ALL = fA | fB | fC | fM;
        // 64-wide raw_feat networks without a direction input (nerf_skin, nerf_vis): the four head products below are one
        // two-layer launch of the chain kernel (bwd64_chain.hip, operands padded to 64 columns as they are staged)
        const char* hc_env = getenv("MODA_CHAIN64");
        const bool heads_chain = W == 64 && chain_part != nullptr && d->raw_feat && !Cd && D == 5 && !(hc_env && (hc_env[0] == '0' || hc_env[0] == '3'));
        if (heads_chain) {
            if (!n.rc) n.rc = moda_heads64_bwd(dzb, 32, dd, W / 2, hD, W, wb_rgb, wb_ext, dh, W, g_rgb, W / 2, g_brgb, (int)d->n_out, Tm, svec, M,
                                               chain_part, n.st);
        } else {
        n.with(fA | fB).signs(bits_of(dd, W / 2), W / 16).gemm_tn((const float*)dzb, 32, dd, W / 2, g_rgb, W / 2, M, d->n_out, W / 2, g_brgb);
        if (use_bits) n.with(fA | fB | fC).signs(bits_of(dd, W / 2), W / 16).gemm((const float*)dzb, 32, 1, (const float*)wb_rgb, W / 2, 1, dzd, ldz2, M, W / 2, 32);
        else n.with(ALL).gemm((const float*)dzb, 32, 1, (const float*)wb_rgb, W / 2, 1, dzd, ldz2, M, W / 2, 32, nullptr, 0, dd, W / 2);
        n.with(fA | fB).signs(bits_of(hD, W), W / 8).gemm_tn(dzd, ldz2, hD, W, Tm, W, M, W / 2, W, svec);
        }
        if (Cd) {
            n.segsum(dzd, M, Rd, W / 2, ldz2, drb, 1);
            n.fine().gemm(drb, 1, W / 2, dir_src, Cd, 1, g_dir + W, ldd, W / 2, Cd, Rd, nullptr, 0, nullptr, 0, 1, ray_split(Rd));
            if (d_dir) n.gemm(drb, W / 2, 1, Wdir + W, ldd, 1, d_dir, Cd, Rd, Cd, W / 2);
            n.colsum(drb, Rd, W / 2, W / 2, g_bdir);
        }
        // every write into a parameter gradient ADDS: the caller's buffer may already hold other calls' contributions (moda_hip.h)
        if (!n.rc) {
            const long long outs = (long long)(W / 2) * W + (long long)W * W + W + W / 2;
            hipLaunchKernelGGL(head_finish_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, n.st, Tm, svec, Wfin, bfin, Wdir,
                               ldd, (int)W, g_dir, g_fin, g_bfin, Cd ? (float*)nullptr : g_bdir,
                               (int)(W % 4 == 0 && (((uintptr_t)Tm | (uintptr_t)Wfin) & 15) == 0));
            n.rc = (int)hipGetLastError();
        }
        if (!d->raw_feat)
            n.with(fA | fB).gemm_tn((const float*)((const unsigned short*)dzd + W / 2), ldz2, hD, W, g_sig, W, M, 1, W, g_bsig);
        if (heads_chain) { /* dh already holds the gradient at layer D-1 */ }
        else if (use_bits) n.with(fA | fB | fC).signs(bits_of(hD, W), W / 8).gemm(dzd, ldz2, 1, (const float*)wb_ext, W, 1, dh, W, M, W, d->raw_feat ? W / 2 : ldz2);
        else n.with(ALL).gemm(dzd, ldz2, 1, (const float*)wb_ext, W, 1, dh, W, M, W, d->raw_feat ? W / 2 : ldz2, nullptr, 0, hD, W);
    } else {
        const float* fin = ws + L.fin;
        const float* dd = ws + L.dd;
        const float* dz_rgb = g_out;
        long long ldz = ldo;
        const float* d_sigma = nullptr;
        if (!d->raw_feat) {
            hipLaunchKernelGGL(sigmoid_bwd_strided_kernel, dim3((unsigned)((M * d->n_out + 255) / 256)), dim3(256), 0, n.st, g_out,
                               ldo, out, ldo, M, (int)d->n_out, dzrgb);
            dz_rgb = dzrgb;
            ldz = d->n_out;
            d_sigma = g_out + d->n_out;
        }
        n.with(fB).gemm_tn(dz_rgb, ldz, dd, W / 2, g_rgb, W / 2, M, d->n_out, W / 2, g_brgb);
        n.with(fC | fM).gemm(dz_rgb, ldz, 1, Wrgb, W / 2, 1, dzd, W / 2, M, W / 2, d->n_out, nullptr, 0, dd, W / 2);     // ReLU mask of dir_encoding
        n.with(fA | fB).gemm_tn(dzd, W / 2, fin, W, g_dir, ldd, M, W / 2, W, Cd ? nullptr : g_bdir);
        if (Cd) {
            n.segsum(dzd, M, Rd, W / 2, W / 2, drb, bfi);
            n.fine().gemm(drb, 1, W / 2, dir_src, Cd, 1, g_dir + W, ldd, W / 2, Cd, Rd, nullptr, 0, nullptr, 0, 1, ray_split(Rd));
            if (d_dir) n.gemm(drb, W / 2, 1, Wdir + W, ldd, 1, d_dir, Cd, Rd, Cd, W / 2);
            n.colsum(drb, Rd, W / 2, W / 2, g_bdir);
        }
        n.with(fA | fC).gemm(dzd, W / 2, 1, ws + L.Wdh, W, 1, dfin, W, M, W, W / 2);
        n.with(fA | fB).signs(bits_of(hD, W), W / 8).gemm_tn(dfin, W, hD, W, g_fin, W, M, W, W, g_bfin);
        if (d_sigma) {
            n.with(fC).gemm(d_sigma, ldo, 1, Wsig, W, 1, dh, W, M, W, 1);
            if (use_xbits) n.signs(xbits, W / 8).gemm(dfin, W, 1, Wfin, W, 1, dh, W, M, W, W, nullptr, 0, nullptr, 0, 2);
            else n.with(fA | fC | fM).gemm(dfin, W, 1, Wfin, W, 1, dh, W, M, W, W, nullptr, 0, hD, W, 2);
            n.with(fB).gemm_tn(d_sigma, ldo, hD, W, g_sig, W, M, 1, W);
            n.colsum(d_sigma, M, 1, ldo, g_bsig);
        } else {
            if (use_xbits) n.signs(xbits, W / 8).gemm(dfin, W, 1, Wfin, W, 1, dh, W, M, W, W);
            else n.with(fA | fC | fM).gemm(dfin, W, 1, Wfin, W, 1, dh, W, M, W, W, nullptr, 0, hD, W);
        }
    }
    bool have_dpe = false;
    // 64-wide networks, bf16 storage: the hidden-layer products of ALL layers (dW, db, the masked dX chain) are one launch that
    // keeps dh on chip between the layers (bwd64_chain.hip); what the skip layer adds (its PE / code columns, d_pe) is taken from
    // dh_{D-1} before.  MODA_CHAIN64=0 keeps the per-layer launches.
    const char* chain_env = getenv("MODA_CHAIN64");           // read per call: an A/B switch for tests and tools
    const bool chain_off = chain_env && chain_env[0] == '0';
    const bool chain = folded && W == 64 && chain_part != nullptr && D == 5 && !chain_off;
    // ... and everything the positional encoding takes part in at the chain's two ends (its weight-gradient blocks, d_pe, the
    // embedding backward) is a second launch that never stores d_pe (pe_ends64_kernel); MODA_CHAIN64=2 keeps the per-layer forms
    const bool pe_ends = chain && d_xyz != nullptr && fPE != 0 && P == 63 && d->n_freq <= 10 && !(chain_env && chain_env[0] == '2');
    const float* dh_skip = nullptr;             // dh_4, kept for pe_ends
    // 256- and 128-wide networks, bf16 storage: the h-column products of a hidden layer (dW, db, the masked dX) are ONE launch that reads
    // dh and the layer's input once (bwd256_fused.hip) instead of the dW form + the dX form.  MODA_BWD256=0 keeps the two launches.
    const char* f256_env = getenv("MODA_BWD256");             // read per call: an A/B switch for tests and tools
    const bool fused256 = use_bits && (W == 256 || W == 128) && !(f256_env && f256_env[0] == '0') &&
                          !(f256_env && f256_env[0] == '2' && W != 256);        // "256": the 256-wide networks only
    static const bool bwd256_trace = getenv("MODA_BWD256_TRACE") != nullptr;      // (read once: not a getenv per layer per step)
    if (bwd256_trace) fprintf(stderr, "nerf_train_bwd: M %lld W %lld D %lld bst %d folded %d use_bits %d sigma_only %d n_out %d\n", M, W, D, (int)bst, (int)folded, (int)use_bits, (int)d->sigma_only, (int)d->n_out);
    auto layer256 = [&](const float* dz, const float* hin, const unsigned short* wb, float* dxo, float* gWp, long long ldg, float* gbp) {
        if (!fused256 || n.rc) return false;
        const int r = moda_bwd256_layer((int)W, dz, W, hin, W, wb, W, dxo, W, gWp, ldg, gbp, M, n.st);
        if (bwd256_trace) fprintf(stderr, "bwd256 layer: M %lld rc %d dz %p hin %p dx %p\n", M, r, (const void*)dz, (const void*)hin, (void*)dxo);
        if (r == MODA_ESHAPE) return false;
        n.rc = r;
        return true;
    };
    for (int l = (int)D - 1; l >= 1; --l) {     // dh = d(loss)/d(pre-activation of layer l), mask already applied
        const float* hprev = hs + (long long)(l - 1) * M * W;
        float* dnext = (dh == dhA) ? dhB : dhA;
        if (chain) {
            if (l == 4) {                       // the skip layer's other inputs
                dh_skip = dh;
                if (!pe_ends) n.with(fA | fPE).gemm_tn(dh, W, pe, Pp, gW(4), ld5, M, W, P);
                if (C1) {
                    n.segsum(dh, M, R1, W, W, drb, bfi);
                    n.fine().gemm(drb, 1, W, code, C1, 1, gW(4) + P, ld5, W, C1, R1, nullptr, 0, nullptr, 0, 1, ray_split(R1));
                    if (d_code) n.gemm(drb, W, 1, Wl(4) + P, ld5, 1, d_code, C1, R1, C1, W, nullptr, 0, nullptr, 0, 2);
                    n.colsum(drb, R1, W, W, gb(4));
                }
                if (d_xyz && !pe_ends) {
                    n.with(fA | fB).gemm(dh, W, 1, (const float*)wb_5pe, Pp, 1, dpe, Pp, M, Pp, W);
                    have_dpe = true;
                }
            }
            if (l > 1) continue;                // (l == D - 1 may also be 1)
            const int nl = (int)D - 1;
            const void *hj[4], *wj[4];
            float *gWj[4], *gbj[4];
            long long ldwj[4];
            for (int j = 0; j < nl; ++j) {
                const int lj = (int)D - 1 - j;
                hj[j] = hs + (long long)(lj - 1) * M * W;
                wj[j] = wb_l[lj];
                gWj[j] = lj == 4 ? gW(4) + P + C1 : gW(lj);
                ldwj[j] = lj == 4 ? ld5 : W;
                gbj[j] = (lj == 4 && C1) ? nullptr : gb(lj);
            }
            if (!n.rc) n.rc = moda_chain64_bwd(dh, W, hj, W, wj, dnext, W, gWj, ldwj, gbj, nl, M, chain_part, n.st);
            dh = dnext;
            continue;
        }
        if (l == 4) {
            n.with(fA | fPE).gemm_tn(dh, W, pe, Pp, gW(4), ld5, M, W, P);
            const bool one = layer256(dh, hprev, wb_l[4], dnext, gW(4) + P + C1, ld5, C1 ? nullptr : gb(4));
            if (!one) n.with(fA | fB).signs(bits_of(hprev, W), W / 8).gemm_tn(dh, W, hprev, W, gW(4) + P + C1, ld5, M, W, W, C1 ? nullptr : gb(4));
            if (C1) {
                n.segsum(dh, M, R1, W, W, drb, bfi);
                n.fine().gemm(drb, 1, W, code, C1, 1, gW(4) + P, ld5, W, C1, R1, nullptr, 0, nullptr, 0, 1, ray_split(R1));
                if (d_code) n.gemm(drb, W, 1, Wl(4) + P, ld5, 1, d_code, C1, R1, C1, W, nullptr, 0, nullptr, 0, 2);
                n.colsum(drb, R1, W, W, gb(4));
            }
            if (d_xyz) {
                if (folded) n.with(fA | fB).gemm(dh, W, 1, (const float*)wb_5pe, Pp, 1, dpe, Pp, M, Pp, W);
                else n.with(fA).gemm(dh, W, 1, ws + L.W5p, Pp + W, 1, dpe, Pp, M, Pp, W);
                have_dpe = true;
            }
            if (one) { /* dnext written by the fused launch */ }
            else if (use_bits) n.with(fA | fB | fC).signs(bits_of(hprev, W), W / 8).gemm(dh, W, 1, (const float*)wb_l[4], W, 1, dnext, W, M, W, W);
            else if (use_xbits) n.signs(xbits, W / 8).gemm(dh, W, 1, ws + L.W5p + Pp, Pp + W, 1, dnext, W, M, W, W);
            else if (folded) n.with(fA | fB | fC | fM).gemm(dh, W, 1, (const float*)wb_l[4], W, 1, dnext, W, M, W, W, nullptr, 0, hprev, W);
            else n.with(fA | fC | fM).gemm(dh, W, 1, ws + L.W5p + Pp, Pp + W, 1, dnext, W, M, W, W, nullptr, 0, hprev, W);
        } else if (layer256(dh, hprev, wb_l[l], dnext, gW(l), W, gb(l))) {
        } else {
            n.with(fA | fB).signs(bits_of(hprev, W), W / 8).gemm_tn(dh, W, hprev, W, gW(l), W, M, W, W, gb(l));
            if (use_bits) n.with(fA | fB | fC).signs(bits_of(hprev, W), W / 8).gemm(dh, W, 1, (const float*)wb_l[l], W, 1, dnext, W, M, W, W);
            else if (use_xbits) n.signs(xbits, W / 8).gemm(dh, W, 1, Wl(l), W, 1, dnext, W, M, W, W);
            else if (folded) n.with(fA | fB | fC | fM).gemm(dh, W, 1, (const float*)wb_l[l], W, 1, dnext, W, M, W, W, nullptr, 0, hprev, W);
            else n.with(fA | fC | fM).gemm(dh, W, 1, Wl(l), W, 1, dnext, W, M, W, W, nullptr, 0, hprev, W);
        }
        dh = dnext;
    }
    if (pe_ends) {
        if (!n.rc) n.rc = moda_pe_ends64_bwd(dh_skip, dh, W, pe, Pp, wb_5pe, wb_1pe, xyz, d->n_freq, d->window, gW(4), ld5, gW(0), ld1,
                                             C1 ? nullptr : gb(0), d_xyz, M, chain_part, n.st);
    } else {
        n.with(fA | fPE).gemm_tn(dh, W, pe, Pp, gW(0), ld1, M, W, P, C1 ? nullptr : gb(0));
    }
    if (C1) {
        n.segsum(dh, M, R1, W, W, drb, bfi);
        n.fine().gemm(drb, 1, W, code, C1, 1, gW(0) + P, ld1, W, C1, R1, nullptr, 0, nullptr, 0, 1, ray_split(R1));
        if (d_code) n.gemm(drb, W, 1, Wl(0) + P, ld1, 1, d_code, C1, R1, C1, W, nullptr, 0, nullptr, 0, 2);
        n.colsum(drb, R1, W, W, gb(0));
    }
    if (d_xyz && !pe_ends) {
        if (folded) n.with(fA | fB).gemm(dh, W, 1, (const float*)wb_1pe, Pp, 1, dpe, Pp, M, Pp, W, nullptr, 0, nullptr, 0, have_dpe ? 2 : 0);
        else n.with(fA).gemm(dh, W, 1, ws + L.W1p, Pp, 1, dpe, Pp, M, Pp, W, nullptr, 0, nullptr, 0, have_dpe ? 2 : 0);
        if (!n.rc) n.rc = moda_embed_bwd(xyz, M, 3, d->n_freq, d->window, 0, dpe, Pp, d_xyz, n.st);
    }
    return n.rc;
}

extern "C" int moda_fold_final(const float* Wdir, int64_t ldd, const float* Wfin, const float* bfin, const float* bdir, int64_t W,
                               float* prod, float* bd_out, void* stream) {
    if (!Wdir || !Wfin || !prod || W < 2 || W > 256 || (W & 1) || ldd < W || (bd_out && (!bfin || !bdir))) return MODA_EINVAL;
    return fold_final_launch(Wdir, ldd, Wfin, bfin, bdir, W, prod, bd_out, (hipStream_t)stream);
}
