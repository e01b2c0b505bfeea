// The two large GEMM forms of the bf16-storage training backward (moda_nerf_train_bwd with MODA_TRAIN_BF16_STORE), gfx950.
//
// Reference: the autograd backward of the nn.Linear layers of NeRF.forward (nnutils/nerf.py:147-198) --
//   dX = (dZ @ W) (.) [h > 0]     the gradient that flows to the layer below, with the ReLU mask of that layer's output
//   dW = dZ^T @ X,  db = 1^T dZ   the parameter gradients, a reduction over every sample of the step
// Both are thin: K or N is the layer width (64..256) while the long dimension is the M = rays x samples rows, so each is
// bound by streaming dZ / X / the mask / dX through HBM once.  With the activations and gradients held as bf16 the generic
// kernel (train_kernels.hip gemm2: fp32 LDS image, fragments converted as they are read) is bound by its LDS traffic and
// conversions instead; here bf16 goes from memory into a bf16 LDS image unchanged and every MFMA operand is one or two LDS
// reads:
//   * an operand whose k index is the SLOW index in memory ([k][row], rows contiguous: dZ and X for dW, the weight for dX)
//     is kept as it lies, [k][row], and read with ds_read_b64_tr_b16 -- the hardware's transposing read hands each lane
//     the four consecutive k of its row; rows are 16-byte-chunk swizzled so the reads are bank-conflict free;
//   * an operand whose k index is the FAST one ([row][k]: dZ for dX) is read with ds_read_b128 from padded rows.
// T[r][c] = sum_k X[k][r] * Y[k][c] (or Y[c][k]) on v_mfma_f32_32x32x16_bf16, fp32 accumulation; a tile engine is 4 waves as
// 2 x 2, one k-tile of 64 per barrier pair, global loads of the next k-tile in flight during the MFMAs of the current one.
//   dW form: X = dZ, Y = X_act (bf16, or fp32 for the positional encoding), T added to dW with fp32 atomics (split over k:
//            across workgroups, and across the NG tile engines of a workgroup, which are summed through LDS first), db from the
//            dZ chunks as they pass through the registers;
//   dX form: X = W (a bf16 copy made once per backward by wprep_kernel, or fp32 rounded as it is staged), Y = dZ, T^T stored
//            row-major through a per-wave LDS transpose: 16-byte stores of bf16 (or fp32 for d_pe), the bf16 mask read the
//            same way, optional C += .
// moda_gemm_f32_ex (train_kernels.hip) routes a call here when its operand types and strides fit (g3_try); everything
// else stays on the generic kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 64;                    // k per tile: four 32x32x16 steps
constexpr int KF_STRIDE = KT * 2 + 16;    // bytes per row of a k-fast image (144: sixteen consecutive rows cover all 64 banks)
enum { EPI_ATOMIC = 0, EPI_T_BF16 = 1, EPI_T_F32 = 2 };

struct G3Args {
    const void* X; long long ldx;         // [k][r]: element (k, r) at k * ldx + r
    const void* Y; long long ldy;         // k-slow: (k, c) at k * ldy + c;  k-fast: (c, k) at c * ldy + k
    void* C; long long ldc;               // EPI_ATOMIC: (r, c) at r * ldc + c, fp32;  T forms: (c, r) at c * ldc + r
    const unsigned short* mask; long long ldm;   // T forms: bf16, (c, r) at c * ldm + r; result zeroed where mask <= 0; or null
    float* xsum;                          // EPI_ATOMIC: xsum[r] += sum_k X(k, r), or null
    unsigned char* bits; long long ldb;   // sign bits, row at bits + row * ldb: EPI_ATOMIC: WRITTEN for Y (k, c) by the r-tile 0
                                          // workgroups; T forms: READ as the mask of (c, r) instead of `mask`; or null
    int R, Cn, K;
    int splits;                           // EPI_ATOMIC: slices over k (slice s takes the k-tiles s, s + splits, ...); else 1
    int accumulate;                       // T forms: 2 = C += T
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
DEVINL uint4 pack8(const float4& a, const float4& b) { return make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w)); }
DEVINL float bflo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
DEVINL float bfhi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// NG (dW form): wave groups per workgroup.  Each group of 4 waves is a complete 2 x 2 tile engine with its own LDS stage and
// its own k-tiles; the groups' accumulators are summed through LDS before ONE set of atomics leaves the workgroup.  The
// atomics of all workgroups land on the same R x C addresses and serialise there (~37 ns per workgroup whatever the tile:
// 19 of the 23 us of a 64 x 64 product at 512 workgroups), while the stream needs many k-tiles in flight: NG groups give
// the memory parallelism of NG workgroups at the atomic cost of one.
template <int TR, int TC, bool X32, bool YKF, bool Y32, int EPI, int NG = 1>
__global__ __launch_bounds__(256 * NG, NG == 1 ? 2 : 1) void gemm3_kernel(G3Args a) {
    static_assert(!(YKF && Y32), "the k-fast operand is bf16");
    static_assert(NG == 1 || EPI == EPI_ATOMIC, "wave groups split k: the dW form");
    constexpr int RBX = TR * 2;                                   // bytes per row of the X image
    constexpr int RBY = YKF ? KF_STRIDE : TC * 2;
    constexpr int XS_BYTES = KT * RBX;
    constexpr int YS_BYTES = YKF ? TC * KF_STRIDE : KT * RBY;
    constexpr int WR = TR / 2, WC = TC / 2, NI = WR / 32, NJ = WC / 32;
    constexpr int EPI_ROW = (EPI == EPI_T_F32) ? WR * 4 + 16 : WR * 2 + 16;     // bytes per row of a wave's transpose buffer
    constexpr int EPI_BYTES = (EPI == EPI_ATOMIC) ? 256 * 8 * 4 : 4 * 32 * EPI_ROW;
    constexpr int STAGE_BYTES = (XS_BYTES + YS_BYTES > EPI_BYTES) ? XS_BYTES + YS_BYTES : EPI_BYTES;
    constexpr int NREG = NI * NJ * 16;                            // accumulator registers per lane
    static_assert(NG == 1 || (NG - 1) * NREG * 256 * 4 <= NG * STAGE_BYTES, "the partial tiles are summed through the stage LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NG * STAGE_BYTES];
    const int grp = NG == 1 ? 0 : (int)(threadIdx.x >> 8);
    unsigned char* Xs = lds + grp * STAGE_BYTES;
    unsigned char* Ys = Xs + XS_BYTES;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int wr = (wave >> 1) * WR, wc = (wave & 1) * WC;
    // tiles that share their Y rows (the long operand of the dX form) get dispatch ids 8 apart: one XCD, one L2
    // (workgroups are dealt round-robin over the 8 XCDs, each with its own L2: what one XCD has fetched, another fetches again)
    const unsigned lid = blockIdx.x;
    unsigned tr_idx, tc_idx, slice = 0;
    if (EPI == EPI_ATOMIC) {
        // the tiles of one k slice read the same rows of dZ and X: dispatch ids 8 apart, so that one XCD's L2 serves the
        // second reader of every chunk (measured without: every operand byte crosses the fabric once per tile column / row)
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
    // split over k: engine e of E = splits x NG (slice s, group g: e = s NG + g) takes the k-tiles e, e + E, e + 2E, ...:
    // engines that run at the same time read neighbouring 64-row pieces of the operands (measured the same as one
    // contiguous range per engine; kept because it makes the ranges independent of K)
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

    // ---- staging: thread -> (row, 16-byte chunk) of each image, the same chunk column in every pass --------------------
    constexpr int XCPR = TR / 8, XRPP = 256 / XCPR, XNP = KT / XRPP;
    const int xch = tid % XCPR, xrow = tid / XCPR;
    uint4 xs[XNP];
    float4 xf[X32 ? XNP : 1][2];
    constexpr int YCPR = YKF ? KT / 8 : TC / 8, YRPP = 256 / YCPR, YNP = (YKF ? TC : KT) / YRPP;
    const int ych = tid % YCPR, yrow = tid / YCPR;
    uint4 ys[YNP];
    float4 yf[Y32 ? YNP : 1][2];
    const bool do_xsum = (EPI == EPI_ATOMIC) && a.xsum != nullptr && tc_idx == 0;
    float xsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < XNP; ++e) {
            const long long k = k0 + xrow + XRPP * e, col = r0 + 8 * xch;
            const bool ok = k < kend && col < a.R;
            if (X32) {
                xf[e][0] = xf[e][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const float* p = (const float*)a.X + k * a.ldx + col;
                    xf[e][0] = *(const float4*)p;
                    xf[e][1] = *(const float4*)(p + 4);
                }
            } else {
                xs[e] = make_uint4(0u, 0u, 0u, 0u);
                if (ok) xs[e] = *(const uint4*)((const unsigned short*)a.X + k * a.ldx + col);
            }
        }
#pragma unroll
        for (int e = 0; e < YNP; ++e) {
            if (YKF) {
                const long long c = c0 + yrow + YRPP * e, k = k0 + 8 * ych;
                ys[e] = make_uint4(0u, 0u, 0u, 0u);
                if (c < a.Cn && k < kend) ys[e] = *(const uint4*)((const unsigned short*)a.Y + c * a.ldy + k);
            } else {
                const long long k = k0 + yrow + YRPP * e, col = c0 + 8 * ych;
                const bool ok = k < kend && col < a.Cn;
                if (Y32) {
                    yf[e][0] = yf[e][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) {
                        const float* p = (const float*)a.Y + k * a.ldy + col;
                        yf[e][0] = *(const float4*)p;
                        yf[e][1] = *(const float4*)(p + 4);
                    }
                } else {
                    ys[e] = make_uint4(0u, 0u, 0u, 0u);
                    if (ok) ys[e] = *(const uint4*)((const unsigned short*)a.Y + k * a.ldy + col);
                }
            }
        }
    };
    const bool do_bits = (EPI == EPI_ATOMIC) && !Y32 && !YKF && a.bits != nullptr && tr_idx == 0;
    auto stash = [&](int k0s) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < XNP; ++e) {
            const int row = xrow + XRPP * e;
            const uint4 v = X32 ? pack8(xf[e][0], xf[e][1]) : xs[e];
            *(uint4*)(Xs + row * RBX + 16 * (xch ^ ks_sw<TR>(row))) = v;
            if (do_xsum) {
                xsum8[0] += bflo(v.x); xsum8[1] += bfhi(v.x); xsum8[2] += bflo(v.y); xsum8[3] += bfhi(v.y);
                xsum8[4] += bflo(v.z); xsum8[5] += bfhi(v.z); xsum8[6] += bflo(v.w); xsum8[7] += bfhi(v.w);
            }
        }
#pragma unroll
        for (int e = 0; e < YNP; ++e) {
            const int row = yrow + YRPP * e;
            if (YKF) {
                *(uint4*)(Ys + row * KF_STRIDE + 16 * ych) = ys[e];
            } else {
                const uint4 v = Y32 ? pack8(yf[e][0], yf[e][1]) : ys[e];
                *(uint4*)(Ys + row * RBY + 16 * (ych ^ ks_sw<TC>(row))) = v;
                if (do_bits) {          // the sign map of this 8-element chunk of Y: one byte (bf16 > 0 <=> its bits, as int16, > 0)
                    const long long k = k0s + row, col = c0 + 8 * ych;
                    if (k < kend && col < a.Cn) {
                        auto b2 = [](unsigned w) { return (((short)(w & 0xffffu) > 0) ? 1u : 0u) | (((int)w > 0xffff) ? 2u : 0u); };
                        a.bits[k * a.ldb + (col >> 3)] = (unsigned char)(b2(v.x) | (b2(v.y) << 2) | (b2(v.z) << 4) | (b2(v.w) << 6));
                    }
                }
            }
        }
    };

    // ---- per-lane bases of the fragment reads.  Transposing read of a [k][T] image, 16-lane group gq = g & 1 of lane half
    //      h: lane 4q+p supplies row (16u + 8h + 4e) + q, elements 16 gq + 4p .. +3 of the 32-wide tile; it receives, for
    //      its own column (lane & 31), the k = 8h + 4e + (0..3) of k-step u -- elements 4e .. 4e+3 of the MFMA operand.
    int xb[NI], yb[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int ch = (wr + 32 * i) / 8 + 2 * (g & 1) + (p4 >> 1);
        xb[i] = (8 * h + q4) * RBX + 16 * (ch ^ ks_sw<TR>(q4)) + 8 * (p4 & 1);
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

    if (nit > 0) fetch(kbeg);
    for (int it = 0; it < nit; ++it) {      // (a k-tile past the end loads zeros)
        const int k0 = kbeg + it * kstep;
        stash(k0);
        __syncthreads();
        if (it + 1 < nit) fetch(k0 + kstep);
#pragma unroll
        for (int u = 0; u < KT / 16; ++u) {
            bf16x8 xa[NI], yv[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) xa[i] = tr_frag(Xs, xb[i], RBX, u);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (YKF) yv[j] = *(const bf16x8*)(Ys + yb[j] + 32 * u);
                else yv[j] = tr_frag(Ys, yb[j], RBY, u);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i], yv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
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
        float* C = (float*)a.C;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const long long c = c0 + wc + 32 * j + (lane & 31);
                if (c >= a.Cn) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long rr = r0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (rr < a.R) atomicAdd(C + rr * a.ldc + c, acc[i][j][r]);
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
    // ---- transposed store: the wave writes one 32-column (c) slab of its tile at a time into its LDS buffer as
    //      [c][r] rows, then every lane takes 16 bytes of one row: coalesced mask / C reads and C stores ---------------------
    unsigned char* tb = lds + wave * (32 * EPI_ROW);
    constexpr int EB = (EPI == EPI_T_F32) ? 4 : 2;            // bytes per stored element
    constexpr int NCH = WR * EB / 16;                         // 16-byte pieces per row
    constexpr int ROWS_PP = 64 / NCH;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned char* p = tb + (lane & 31) * EPI_ROW + (32 * i + 8 * q + 4 * h) * EB;
                if (EPI == EPI_T_F32) *(float4*)p = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                else *(uint2*)p = make_uint2(pk2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pk2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 32 / ROWS_PP; ++ps) {
            const int row = lane / NCH + ROWS_PP * ps, piece = lane % NCH;
            const long long c = c0 + wc + 32 * j + row;
            const long long rr = r0 + wr + piece * (16 / EB);
            const uint4 t = *(const uint4*)(tb + row * EPI_ROW + 16 * piece);
            if (c >= a.Cn || rr >= a.R) continue;
            if (EPI == EPI_T_F32) {
                float4 v = __builtin_bit_cast(float4, t);
                float* cp = (float*)a.C + c * a.ldc + rr;
                if (a.accumulate == 2) {
                    const float4 o = *(const float4*)cp;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *(float4*)cp = v;
            } else {
                unsigned short* cp = (unsigned short*)a.C + c * a.ldc + rr;
                uint4 v = t;
                if (a.accumulate == 2) {
                    const uint4 o = *(const uint4*)cp;
                    v = make_uint4(pk2(bflo(v.x) + bflo(o.x), bfhi(v.x) + bfhi(o.x)), pk2(bflo(v.y) + bflo(o.y), bfhi(v.y) + bfhi(o.y)),
                                   pk2(bflo(v.z) + bflo(o.z), bfhi(v.z) + bfhi(o.z)), pk2(bflo(v.w) + bflo(o.w), bfhi(v.w) + bfhi(o.w)));
                }
                if (a.bits != nullptr) {
                    const unsigned bm = a.bits[c * a.ldb + (rr >> 3)];
                    auto keepb = [](unsigned vw, unsigned two) { return vw & (((two & 1u) ? 0x0000ffffu : 0u) | ((two & 2u) ? 0xffff0000u : 0u)); };
                    v = make_uint4(keepb(v.x, bm), keepb(v.y, bm >> 2), keepb(v.z, bm >> 4), keepb(v.w, bm >> 6));
                } else if (a.mask != nullptr) {
                    const uint4 m = *(const uint4*)(a.mask + c * a.ldm + rr);
                    // bf16 > 0  <=>  its 16 bits, as a signed integer, > 0
                    auto keep = [](unsigned vw, unsigned mw) {
                        const unsigned lo = ((short)(mw & 0xffffu) > 0) ? 0x0000ffffu : 0u;
                        const unsigned hi = ((int)mw > 0xffff) ? 0xffff0000u : 0u;      // high half positive and nonzero
                        return vw & (lo | hi);
                    };
                    v = make_uint4(keep(v.x, m.x), keep(v.y, m.y), keep(v.z, m.z), keep(v.w, m.w));
                }
                *(uint4*)cp = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int TR, int TC, bool X32, bool YKF, bool Y32, int EPI, int NG = 1>
int g3_launch(const G3Args& a, unsigned splits, hipStream_t st) {
    hipLaunchKernelGGL((gemm3_kernel<TR, TC, X32, YKF, Y32, EPI, NG>), dim3(a.gr * a.gc * splits), dim3(256 * NG), 0, st, a);
    return (int)hipGetLastError();
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}   // namespace

// Route of moda_gemm_f32_ex: returns true (and the launch status in *rc) when the call is one of the two forms above with
// operands these kernels can take as they lie; false leaves it to the generic kernel.
bool moda_g3_try(const moda_gemm_desc* d, void* stream, int* rc) {
    const int fl = d->reserved;
    if (!(fl & MODA_GEMM_BF16) || !(fl & MODA_GEMM_A_BF16)) return false;
    static const bool off = [] { const char* e = getenv("MODA_GEMM3"); return e && e[0] == '0'; }();
    if (off) return false;
    if (d->A2 || d->bias || d->rowbias || d->act != 0 || d->K <= 0) return false;
    const bool b_bf = (fl & MODA_GEMM_B_BF16) != 0, c_bf = (fl & MODA_GEMM_C_BF16) != 0, m_bf = (fl & MODA_GEMM_MASK_BF16) != 0;
    if (d->sbn != 1) return false;                                     // B is [k][n] with n contiguous in both forms
    const int64_t lim = 0x7fffffff;
    if (d->M > lim || d->N > lim || d->K > lim) return false;
    hipStream_t st = (hipStream_t)stream;
    G3Args a;
    a.mask = nullptr; a.ldm = 0; a.xsum = nullptr; a.accumulate = 0; a.bits = nullptr; a.ldb = 0;
    if (d->sam == 1 && d->sak != 1) {
        // ---- dW form: A(m, k) = dZ[k * sak + m] (m-fast), B(k, n) = X[k * sbk + n]; C += A B with atomics ------------
        if (d->accumulate != 1 || d->mask_src || c_bf) return false;
        const int64_t R = d->M, Cn = d->N, K = d->K;
        if (!al16(d->A) || d->sak % 8 || (R + 7) / 8 * 8 > d->sak) return false;
        if (!al16(d->B) || d->sbk % (b_bf ? 8 : 4) || (Cn + 7) / 8 * 8 > d->sbk) return false;
        a.X = d->A; a.ldx = d->sak; a.Y = d->B; a.ldy = d->sbk; a.C = d->C; a.ldc = d->ldc;
        a.R = (int)R; a.Cn = (int)Cn; a.K = (int)K; a.xsum = d->a_sum;
        if (d->mask_bits) {
            if (!b_bf || Cn % 8 || d->ld_bits < Cn / 8) return false;
            a.bits = (unsigned char*)d->mask_bits; a.ldb = d->ld_bits;
        }
        const bool big = R > 64 || Cn > 64;                            // 128 x 128 tiles, or 64 x 64 for the 64-wide nets
        const int T = big ? 128 : 64;
        a.gr = (unsigned)((R + T - 1) / T); a.gc = (unsigned)((Cn + T - 1) / T);
        // split over k: one workgroup per CU (NG wave groups each), every engine with at least two k-tiles
        static const long long target_env = [] { const char* e = getenv("MODA_GEMM3_BLOCKS"); return e ? atoll(e) : 0LL; }();
        const long long target = target_env > 0 ? target_env : (big ? 256 : 128);    // measured best (64 .. 512 swept)
        const int ng = big ? 2 : 4;
        const long long nkt = (K + KT - 1) / KT;
        long long splits = target / ((long long)a.gr * a.gc);
        if (splits > nkt / (2 * ng)) splits = nkt / (2 * ng);
        if (splits < 1) splits = 1;
        if (splits >= 8) splits &= ~7LL;              // whole groups of 8 slices: the XCD placement in the kernel
        const unsigned zs = (unsigned)splits;
        a.splits = (int)splits;
        if (big) *rc = b_bf ? g3_launch<128, 128, false, false, false, EPI_ATOMIC, 2>(a, zs, st)
                            : g3_launch<128, 128, false, false, true, EPI_ATOMIC, 2>(a, zs, st);
        else *rc = b_bf ? g3_launch<64, 64, false, false, false, EPI_ATOMIC, 4>(a, zs, st)
                        : g3_launch<64, 64, false, false, true, EPI_ATOMIC, 4>(a, zs, st);
        return true;
    }
    if (d->sak == 1) {
        // ---- dX form: A(m, k) = dZ[m * sam + k] (k-fast), B(k, n) = W[k * sbk + n] (bf16 or fp32); C(m, n) row-major ------
        if (d->a_sum || (d->accumulate != 0 && d->accumulate != 2) || d->split_k > 1) return false;
        if (d->mask_src && !(m_bf && c_bf)) return false;
        if (d->mask_bits && (!c_bf || d->mask_src || d->N % 8 || d->ld_bits < d->N / 8)) return false;
        const int64_t N = d->N, K = d->K, M = d->M;
        if (K % 8 || N % 8) return false;
        if (!al16(d->A) || d->sam % 8) return false;
        if (!al16(d->B) || d->sbk % (b_bf ? 8 : 4) || N > d->sbk) return false;
        if (!al16(d->C) || d->ldc % (c_bf ? 8 : 4)) return false;
        if (d->mask_src && (!al16(d->mask_src) || d->ld_mask % 8)) return false;
        if (!c_bf && N > 64) return false;                             // the fp32 store is built for the narrow d_pe product
        a.X = d->B; a.ldx = d->sbk; a.Y = d->A; a.ldy = d->sam; a.C = d->C; a.ldc = d->ldc;
        a.mask = (const unsigned short*)d->mask_src; a.ldm = d->ld_mask;
        a.bits = (unsigned char*)d->mask_bits; a.ldb = d->ld_bits;
        a.R = (int)N; a.Cn = (int)M; a.K = (int)K; a.accumulate = d->accumulate;
        a.splits = 1;
        a.gc = (unsigned)((M + 127) / 128);
        if (N > 64) {
            a.gr = (unsigned)((N + 127) / 128);
            *rc = b_bf ? g3_launch<128, 128, false, true, false, EPI_T_BF16>(a, 1, st)
                       : g3_launch<128, 128, true, true, false, EPI_T_BF16>(a, 1, st);
        } else {
            a.gr = 1;
            if (b_bf) *rc = c_bf ? g3_launch<64, 128, false, true, false, EPI_T_BF16>(a, 1, st)
                                 : g3_launch<64, 128, false, true, false, EPI_T_F32>(a, 1, st);
            else *rc = c_bf ? g3_launch<64, 128, true, true, false, EPI_T_BF16>(a, 1, st)
                            : g3_launch<64, 128, true, true, false, EPI_T_F32>(a, 1, st);
        }
        return true;
    }
    return false;
}
