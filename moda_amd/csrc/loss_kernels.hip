// Loss heads that sit behind compositing inside inference_deform (reference nnutils/rendering.py:410-437,
// 475-477, 573-578): CSE feature matching on a 20^3 canonical grid (nnutils/loss_utils.py:273-405, softmax and
// Sinkhorn optimal-transport forms), row normalisation (F.normalize), and the visibility loss' log-sigmoid sums
// (loss_utils.py:125-149).  Forward and hand-derived backward; exact fp32.
//
// The (N rays x G grid points) matching matrix  Kmat[n,g] = exp((<f_n, v_g> - 1) * kappa)  is materialised once
// (64 MB at the training size, Infinity-Cache resident) and so is its transpose KmatT (G x N), so that every later
// step -- a sum over the grid points of a pixel or over the pixels of a grid point -- is a row sweep: one wavefront
// per row, 16-byte coalesced loads, shuffle reduction, no atomics (deterministic sums).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

constexpr int kF = 16;              // CSE feature width (nerf_feat out_channels, moda.py:447)
constexpr int kFdRows = 16;         // rows of the matching matrix per workgroup of featdot_exp_kernel

// The matching matrix Kmat / KmatT may be held as bf16 (`bf`: the throughput mode of the training route -- the 78 Sinkhorn
// sweeps of a step read it 78 times); element i of either form:
DEVINL float kld(const float* K, long long i, int bf) {
    return bf ? __builtin_bit_cast(float, (unsigned)((const unsigned short*)K)[i] << 16) : K[i];
}
DEVINL unsigned short f2bf(float v) {      // round to nearest even
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {v, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_)) & 0xffffu);
}
constexpr float kSinkEps = 1e-8f;   // loss_utils.py:366,369

DEVINL float wave_sum(float v) { return comp_wave_sum(v); }      // (DPP / lane-swap form, bitwise the shuffle butterfly: moda_dev.h)

// ---- F.normalize(x, 2, -1): y = x / max(|x|, 1e-12) ------------------------------------------------
__global__ void normalize_rows_kernel(const float* __restrict__ x, long long M, int F, float* __restrict__ y,
                                      const float* __restrict__ g, float* __restrict__ dx) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* xr = x + m * F;
    float n2 = 0.f;
    for (int i = 0; i < F; ++i) n2 += xr[i] * xr[i];
    const float nrm = sqrtf(n2);
    const float den = fmaxf(nrm, 1e-12f);
    if (g == nullptr) {
        for (int i = 0; i < F; ++i) y[m * F + i] = xr[i] / den;
        return;
    }
    const float* gr = g + m * F;
    if (nrm > 1e-12f) {
        float dot = 0.f;
        for (int i = 0; i < F; ++i) dot += gr[i] * (xr[i] / den);
        for (int i = 0; i < F; ++i) dx[m * F + i] = (gr[i] - (xr[i] / den) * dot) / den;
    } else {
        for (int i = 0; i < F; ++i) dx[m * F + i] = gr[i] / den;
    }
}

// ---- Kmat[n,g] = exp((<f_n, v_g> - 1) * kappa) ------------------------------------------------------
__global__ __launch_bounds__(256) void featdot_exp_kernel(const float* __restrict__ fn, const float* __restrict__ vn,
                                                          int N, int G, const float* __restrict__ kappa_p,
                                                          float* __restrict__ Kmat, int bf) {
    // a thread keeps its column's feature vector in registers for kFdRows rows (blockIdx.y = row group): one row per
    // workgroup re-read the 64-byte vector from L2 for every row (1 GB of L2 reads per call for a 65 MB matrix)
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float kappa = kappa_p[0];
    const float4* v4 = (const float4*)(vn + (long long)g * kF);
    float4 b[kF / 4];
#pragma unroll
    for (int i = 0; i < kF / 4; ++i) b[i] = v4[i];
    const int n0 = blockIdx.y * kFdRows;
    const int n1 = min(N, n0 + kFdRows);
    for (int n = n0; n < n1; ++n) {
        const float4* f4 = (const float4*)(fn + (long long)n * kF);      // uniform: scalar loads
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < kF / 4; ++i) {
            const float4 a = f4[i];
            d += a.x * b[i].x + a.y * b[i].y + a.z * b[i].z + a.w * b[i].w;
        }
        const float kv = expf((d - 1.f) * kappa);
        if (bf) ((unsigned short*)Kmat)[(long long)n * G + g] = f2bf(kv);
        else Kmat[(long long)n * G + g] = kv;
    }
}

// The same matrix for feat_match(init_pts=...): every pixel n has a lattice of its own, so its row of the matrix is taken against
// ITS G feature vectors vol[n, g, :] (loss_utils.py:322-335).
__global__ __launch_bounds__(256) void featdot_rows_exp_kernel(const float* __restrict__ fn, const float* __restrict__ vol, int N, int G,
                                                               const float* __restrict__ kappa_p, float* __restrict__ Kmat) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (g >= G) return;
    const float4* f4 = (const float4*)(fn + (long long)n * kF);                         // uniform: scalar loads
    const float4* v4 = (const float4*)(vol + ((long long)n * G + g) * kF);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < kF / 4; ++i) {
        const float4 a = f4[i], b = v4[i];
        d += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    Kmat[(long long)n * G + g] = expf((d - 1.f) * kappa_p[0]);
}

// epilogue of a matrix-vector sweep: 0 plain sum, 1 p / (sum + eps) (Sinkhorn update, loss_utils.py:363-369),
// 2 -sum * c^2 / p (the reverse-mode step through that update)
DEVINL float sweep_epilogue(float s, int mode, float p, float c) {
    if (mode == 1) return p / (s + kSinkEps);
    if (mode == 2) return -s * c * c / p;
    return s;
}

// out[r] = epi(sum_c Mat[r,c] x[c]) for a row-major (R,C) matrix: one wavefront per row, 16-byte loads, four in
// flight per lane.  Column sums of Kmat are row sums of its transposed copy KmatT, so every sweep has this form.
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ Mat, const float* __restrict__ x, int R,
                                                        int C, int mode, float p, const float* __restrict__ c,
                                                        float* __restrict__ out, int bf) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    if (bf) {       // bf16 rows: 16-byte loads of eight elements, two in flight per lane
        const unsigned short* rowh = (const unsigned short*)Mat + (long long)r * C;
        const int C8 = ((((uintptr_t)rowh) & 15) == 0 && (((uintptr_t)x) & 15) == 0) ? (C >> 3) : 0;
        const uint4* row8 = (const uint4*)rowh;
        const float4* x4 = (const float4*)x;
        float t0 = 0.f, t1 = 0.f;
        auto dot8 = [](const uint4& a, const float4& b0, const float4& b1) {
            return __builtin_bit_cast(float, a.x << 16) * b0.x + __builtin_bit_cast(float, a.x & 0xffff0000u) * b0.y +
                   __builtin_bit_cast(float, a.y << 16) * b0.z + __builtin_bit_cast(float, a.y & 0xffff0000u) * b0.w +
                   __builtin_bit_cast(float, a.z << 16) * b1.x + __builtin_bit_cast(float, a.z & 0xffff0000u) * b1.y +
                   __builtin_bit_cast(float, a.w << 16) * b1.z + __builtin_bit_cast(float, a.w & 0xffff0000u) * b1.w;
        };
        int q = lane;
        for (; q + 64 < C8; q += 128) {
            const uint4 a0 = row8[q], a1 = row8[q + 64];
            t0 += dot8(a0, x4[2 * q], x4[2 * q + 1]);
            t1 += dot8(a1, x4[2 * q + 128], x4[2 * q + 129]);
        }
        for (; q < C8; q += 64) t0 += dot8(row8[q], x4[2 * q], x4[2 * q + 1]);
        for (int g = 8 * C8 + lane; g < C; g += 64) t1 += __builtin_bit_cast(float, (unsigned)rowh[g] << 16) * x[g];
        const float sb = wave_sum(t0 + t1);
        if (lane == 0) out[r] = sweep_epilogue(sb, mode, p, c ? c[r] : 0.f);
        return;
    }
    const float* row = Mat + (long long)r * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int C4 = ((((uintptr_t)row) & 15) == 0 && (((uintptr_t)x) & 15) == 0) ? (C >> 2) : 0;
    const float4* row4 = (const float4*)row;
    const float4* x4 = (const float4*)x;
    int q = lane;
    for (; q + 192 < C4; q += 256) {
        const float4 a0 = row4[q], a1 = row4[q + 64], a2 = row4[q + 128], a3 = row4[q + 192];
        const float4 b0 = x4[q], b1 = x4[q + 64], b2 = x4[q + 128], b3 = x4[q + 192];
        s0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
        s1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
        s2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
        s3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
    }
    for (; q < C4; q += 64) {
        const float4 a0 = row4[q], b0 = x4[q];
        s0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
    }
    for (int g = 4 * C4 + lane; g < C; g += 64) s1 += row[g] * x[g];
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) out[r] = sweep_epilogue(s, mode, p, c ? c[r] : 0.f);
}

// prob[n,g] = Kmat[n,g] b[g] / s[n],  s[n] = sum_g Kmat[n,g] b[g];  pred[n] = sum_g prob[n,g] q[g]
// (loss_utils.py:371-374 / :376 and :389).  b == NULL means b = 1 (the softmax form).
__global__ __launch_bounds__(256) void match_expect_kernel(const float* __restrict__ Kmat, const float* __restrict__ b,
                                                           const float* __restrict__ q, int N, int G,
                                                           float* __restrict__ pred, float* __restrict__ s_out, int bf) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int g = lane; g < G; g += 64) {
        const float w = kld(Kmat, (long long)n * G + g, bf) * (b ? b[g] : 1.f);
        s += w;
        ax += w * q[3 * g + 0];
        ay += w * q[3 * g + 1];
        az += w * q[3 * g + 2];
    }
    s = wave_sum(s); ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
    if (lane == 0) {
        pred[3 * n + 0] = ax / s;
        pred[3 * n + 1] = ay / s;
        pred[3 * n + 2] = az / s;
        s_out[n] = s;
    }
}

// prob[n,g] written out (only the back-correspondence term needs the matrix itself)
__global__ __launch_bounds__(256) void match_prob_kernel(const float* __restrict__ Kmat, const float* __restrict__ b,
                                                         const float* __restrict__ s, int N, int G, float* __restrict__ P, int bf) {
    const int n = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < G) P[(long long)n * G + g] = kld(Kmat, (long long)n * G + g, bf) * (b ? b[g] : 1.f) / s[n];
}

// e[n,g] = prob[n,g] (<gbar_n, q_g> - <gbar_n, pred_n>): the gradient w.r.t. the logit log(Kmat[n,g] b[g])
DEVINL float match_e(float kval, float bg, float sn, const float* gb, const float* pr, const float* qg) {
    const float gq = gb[0] * qg[0] + gb[1] * qg[1] + gb[2] * qg[2];
    const float gp = gb[0] * pr[0] + gb[1] * pr[1] + gb[2] * pr[2];
    return kval * bg / sn * (gq - gp);
}

// out[g] = -(sum_n e[n,g]) * b[g] / p2  : ubar of the last Sinkhorn iteration (bbar_g = sum_n e / b_g, then mode-2 step).
// Row g of the transposed matrix KmatT (G,N): one wavefront per grid point, lanes over the pixels n.
// gPT (G,N) | NULL: an extra upstream gradient on the probabilities themselves (the back-correspondence term P P^T of
// use_corr, loss_utils.py:386-391), transposed; sP[n] = sum_g gP[n,g] prob[n,g].  It adds prob (gP - sP) to e.
__global__ __launch_bounds__(256) void match_ecols_kernel(const float* __restrict__ KmatT, const float* __restrict__ b,
                                                          const float* __restrict__ s, const float* __restrict__ gbar,
                                                          const float* __restrict__ pred, const float* __restrict__ q,
                                                          const float* __restrict__ gPT, const float* __restrict__ sP,
                                                          int N, int G, float p2, float* __restrict__ out, int bf) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const float qg[3] = {q[3 * g], q[3 * g + 1], q[3 * g + 2]};
    const float bg = b[g];
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float kv = kld(KmatT, (long long)g * N + n, bf);
        acc += match_e(kv, bg, s[n], gbar + 3 * n, pred + 3 * n, qg);
        if (gPT) acc += kv * bg / s[n] * (gPT[(long long)g * N + n] - sP[n]);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[g] = -acc * bg / p2;
}

// Dbar[n,g] = kappa * (e[n,g] + Kmat[n,g] * (sum_t A[t,n] Ubar[t,g] + sum_t Wbar[t,n] Bm[t,g]))
// (gradient w.r.t. the dot products <f_n, v_g>); kbar += sum e[n,g] log(Kmat[n,g]) / kappa (gradient w.r.t. kappa,
// used by the softmax form where kappa = |beta| + 1e-9).
__global__ __launch_bounds__(256) void match_dbar_kernel(const float* __restrict__ Kmat, const float* __restrict__ b,
                                                         const float* __restrict__ s, const float* __restrict__ gbar,
                                                         const float* __restrict__ pred, const float* __restrict__ q,
                                                         const float* __restrict__ A, const float* __restrict__ Ubar, int T1,
                                                         const float* __restrict__ Wbar, const float* __restrict__ Bm, int T2,
                                                         const float* __restrict__ gP, const float* __restrict__ sP,
                                                         int N, int G, const float* __restrict__ kappa_p,
                                                         float* __restrict__ Dbar, float* __restrict__ kbar, int bf) {
    const int n = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    float kb = 0.f;
    if (g < G) {
        const float kappa = kappa_p[0];
        const float kval = kld(Kmat, (long long)n * G + g, bf);
        const float qg[3] = {q[3 * g], q[3 * g + 1], q[3 * g + 2]};
        float e = match_e(kval, b ? b[g] : 1.f, s[n], gbar + 3 * n, pred + 3 * n, qg);
        if (gP) e += kval * (b ? b[g] : 1.f) / s[n] * (gP[(long long)n * G + g] - sP[n]);
        float lin = 0.f;
        for (int t = 0; t < T1; ++t) lin += A[(long long)t * N + n] * Ubar[(long long)t * G + g];
        for (int t = 0; t < T2; ++t) lin += Wbar[(long long)t * N + n] * Bm[(long long)t * G + g];
        Dbar[(long long)n * G + g] = kappa * (e + kval * lin);
        if (kbar) kb = e * logf(fmaxf(kval, 1e-37f)) / kappa;
    }
    if (kbar) {
        kb = wave_sum(kb);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kb;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kbar, red[0] + red[1] + red[2] + red[3]);
    }
}

// The same for T1 + T2 <= TMAX (MoDA's 20 Sinkhorn iterations: 39), a workgroup per (256 grid points, 32 pixels): a thread keeps
// the Ubar / Bm values of its grid point in registers for all 32 rows and the rows' A / Wbar values sit in LDS (broadcast
// reads), where the one-row form fetched its 39 vector elements from L2 again for every row -- 2.5 GB of L2 reads per
// call against 0.13 GB of matrix traffic.  Same operations in the same order per element.
template <int TMAX>
__global__ __launch_bounds__(256) void match_dbar_tiled_kernel(const float* __restrict__ Kmat, const float* __restrict__ b,
                                                               const float* __restrict__ s, const float* __restrict__ gbar,
                                                               const float* __restrict__ pred, const float* __restrict__ q,
                                                               const float* __restrict__ A, const float* __restrict__ Ubar, int T1,
                                                               const float* __restrict__ Wbar, const float* __restrict__ Bm, int T2,
                                                               const float* __restrict__ gP, const float* __restrict__ sP,
                                                               int N, int G, const float* __restrict__ kappa_p,
                                                               float* __restrict__ Dbar, float* __restrict__ kbar, int bf) {
    constexpr int ROWS = 32;
    __shared__ float av[TMAX][ROWS];
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * ROWS;
    const int nr = min(ROWS, N - n0);
    const int TT = T1 + T2;
    for (int idx = threadIdx.x; idx < TMAX * ROWS; idx += 256) {
        const int t = idx / ROWS, r = idx - t * ROWS;
        const int n = n0 + r;
        float v = 0.f;
        if (t < TT && n < N) v = t < T1 ? A[(long long)t * N + n] : Wbar[(long long)(t - T1) * N + n];
        av[t][r] = v;
    }
    __syncthreads();
    const bool gok = g < G;
    const int gc = gok ? g : G - 1;
    float u[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        u[t] = 0.f;
        if (t < TT) u[t] = t < T1 ? Ubar[(long long)t * G + gc] : Bm[(long long)(t - T1) * G + gc];
    }
    const float kappa = kappa_p[0];
    const float qg[3] = {q[3 * gc], q[3 * gc + 1], q[3 * gc + 2]};
    const float bg = b ? b[gc] : 1.f;
    float kb = 0.f;
    for (int r = 0; r < nr; ++r) {
        const int n = n0 + r;
        if (gok) {
            const float kval = kld(Kmat, (long long)n * G + g, bf);
            float e = match_e(kval, bg, s[n], gbar + 3 * n, pred + 3 * n, qg);
            if (gP) e += kval * bg / s[n] * (gP[(long long)n * G + g] - sP[n]);
            float lin = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) lin += av[t][r] * u[t];       // (the padding terms are exact zeros)
            Dbar[(long long)n * G + g] = kappa * (e + kval * lin);
            if (kbar) kb += e * logf(fmaxf(kval, 1e-37f)) / kappa;
        }
    }
    if (kbar) {
        kb = wave_sum(kb);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kb;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kbar, red[0] + red[1] + red[2] + red[3]);
    }
}

// ---- visibility loss pieces (loss_utils.py:125-149): out += scale * sum_i -logsigmoid(sign * x_i) * (w_i | 1);
//      with g given: dx_i = g * scale * (-sign * sigmoid(-sign x_i)) * (w_i | 1) ---------------------------------
__global__ __launch_bounds__(256) void logsig_loss_kernel(const float* __restrict__ x, const float* __restrict__ w, long long n,
                                                          float sign, float scale, float* __restrict__ out,
                                                          const float* __restrict__ g, float* __restrict__ dx) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (long long i = i0; i < n; i += stride) {
        const float z = sign * x[i];
        const float wi = w ? w[i] : 1.f;
        if (g == nullptr) {
            const float ls = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));   // torch's logsigmoid formula
            acc += -ls * wi;
        } else {
            const float sg = 1.f / (1.f + expf(z));                      // sigmoid(-z)
            dx[i] = g[0] * scale * (-sign * sg) * wi;
        }
    }
    if (g == nullptr) {
        acc = wave_sum(acc);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(out, scale * (red[0] + red[1] + red[2] + red[3]));
    }
}

// ------------------------------------------------------------------------------------------------
// S3IM (opts.s3im_loss; reference loss_utils.py:575-702 S3IM / SSIM / _ssim, called at rendering.py:528-532):
// 1 - mean SSIM over the 4x4 / stride 4 / padding 1 Gaussian windows of the (3, H, Wt) virtual patch whose pixel (h, w) is
// row index[h * Wt + w] % N of the masked colours.  ONE workgroup walks all windows (3 * 8 * 80 at the reference's sizes):
// a deterministic sum, and a launch that costs less than the reference's index_select alone.  The backward pass re-forms each
// window and scatters d(1 - mean ssim) / d rgb with atomics (a row sits in ~10 windows).
struct S3imWin { float w[4]; };

template <bool BWD>
__global__ __launch_bounds__(256) void s3im_kernel(const float* __restrict__ rgb, const float* __restrict__ tar,
                                                   const float* __restrict__ mask, int N, const int32_t* __restrict__ index,
                                                   int H, int Wt, S3imWin g, float* __restrict__ loss,
                                                   const float* __restrict__ g_loss, float* __restrict__ d_rgb) {
    const int oh = (H + 2 - 4) / 4 + 1, ow = (Wt + 2 - 4) / 4 + 1;
    const int nwin = 3 * oh * ow;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float acc = 0.f;
    const float gscale = BWD ? -g_loss[0] / (float)nwin : 0.f;
    for (int win = threadIdx.x; win < nwin; win += 256) {
        const int c = win / (oh * ow), oy = (win / ow) % oh, ox = win % ow;
        float x[16], y[16], m[16];
        int row[16];
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int hh = 4 * oy - 1 + (k >> 2), ww = 4 * ox - 1 + (k & 3);
            const bool in = hh >= 0 && hh < H && ww >= 0 && ww < Wt;
            const int r = in ? index[hh * Wt + ww] % N : 0;
            const float mk = in ? mask[r] : 0.f;          // zero padding of conv2d
            row[k] = r; m[k] = mk;
            x[k] = rgb[r * 3 + c] * mk;
            y[k] = tar[r * 3 + c] * mk;
            const float wk = g.w[k >> 2] * g.w[k & 3];
            mu1 = fmaf(wk, x[k], mu1); mu2 = fmaf(wk, y[k], mu2);
            e11 = fmaf(wk, x[k] * x[k], e11); e22 = fmaf(wk, y[k] * y[k], e22); e12 = fmaf(wk, x[k] * y[k], e12);
        }
        const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + C1, Bn = 2.f * s12 + C2, Cc = mu1 * mu1 + mu2 * mu2 + C1, D = s1 + s2 + C2;
        const float ssim = (A * Bn) / (Cc * D);
        if (!BWD) {
            acc += ssim;
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (m[k] == 0.f) continue;
                const float wk = g.w[k >> 2] * g.w[k & 3];
                const float dA = 2.f * mu2 * wk, dB = 2.f * wk * (y[k] - mu2), dC = 2.f * mu1 * wk, dD = 2.f * wk * (x[k] - mu1);
                const float ds = (dA * Bn + A * dB) / (Cc * D) - ssim * (dC / Cc + dD / D);
                atomicAdd(&d_rgb[row[k] * 3 + c], gscale * ds * m[k]);
            }
        }
    }
    if (!BWD) {
        __shared__ float part[4];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) loss[0] = 1.f - ((part[0] + part[1]) + (part[2] + part[3])) / (float)nwin;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-ray loss terms of inference_deform (rendering.py:518-571): img / sil / flo terms with the batch-level statistics they
// depend on (silhouette class balance :535-539, confidence normalisation :549-556) in ONE launch: a first pass of one
// workgroup reduces the seven sums, a second writes the terms.  The reference issues ~25 eager ops (and two host syncs).
// stats[8]: vsum, pos, neg, sil_sum, nsil_sum, n_flo, cfd_sum, (unused)
DEVINL float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

DEVINL bool ray_sil_flo(float sil_at, float valid, float cfd) { return sil_at > 0.f && valid == 1.f && !(cfd == 0.f); }

__global__ __launch_bounds__(1024) void ray_loss_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ sil,
                                                            const float* __restrict__ flo, const float* __restrict__ valid,
                                                            const float* __restrict__ img_at, const float* __restrict__ sil_at,
                                                            const float* __restrict__ vis_at, const float* __restrict__ flo_at,
                                                            const float* __restrict__ cfd_at, int N, int training,
                                                            float* __restrict__ img_loss, float* __restrict__ sil_loss,
                                                            float* __restrict__ flo_loss, unsigned char* __restrict__ sil_flo,
                                                            float* __restrict__ stats) {
    __shared__ float red[16];
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < N; i += 1024) {
        const float s = sil_at[i], v = vis_at[i], c = cfd_at[i];
        const float vp = v > 0.f ? 1.f : 0.f;
        a[0] += v; a[1] += s * vp; a[2] += (1.f - s) * vp; a[3] += s; a[4] += 1.f - s;
        const bool f = ray_sil_flo(s, valid[i], c);
        a[5] += f ? 1.f : 0.f; a[6] += f ? c : 0.f;
    }
    float t[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) t[k] = block_sum_1024(a[k], red);
    if (threadIdx.x < 7) stats[threadIdx.x] = t[threadIdx.x];
    const bool balance = training && t[3] > 0.f && t[4] > 0.f;
    const float pos_wt = t[0] / t[1], neg_wt = t[0] / t[2];
    const float cfd_mean = t[6] / fmaxf(t[5], 1.f);
    for (int i = threadIdx.x; i < N; i += 1024) {
        const float s = sil_at[i];
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = rgb[i * 3 + c] - img_at[i * 3 + c]; e += d * d; }
        img_loss[i] = e / 3.f * s;
        const float bal = balance ? 0.5f * pos_wt * s + 0.5f * neg_wt * (1.f - s) : 1.f;
        const float ds = sil[i] - s;
        sil_loss[i] = ds * ds * bal * vis_at[i];
        const float c = cfd_at[i];
        const float cn = t[5] > 0.f ? c / cfd_mean : c;
        const float d0 = flo[i * 2] - flo_at[i * 2], d1 = flo[i * 2 + 1] - flo_at[i * 2 + 1];
        flo_loss[i] = (d0 * d0 + d1 * d1) * cn * s;
        sil_flo[i] = ray_sil_flo(s, valid[i], c) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void ray_loss_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ sil,
                                                           const float* __restrict__ flo, const float* __restrict__ img_at,
                                                           const float* __restrict__ sil_at, const float* __restrict__ vis_at,
                                                           const float* __restrict__ flo_at, const float* __restrict__ cfd_at, int N,
                                                           int training, const float* __restrict__ stats,
                                                           const float* __restrict__ g_img, const float* __restrict__ g_sil,
                                                           const float* __restrict__ g_flo, float* __restrict__ d_rgb,
                                                           float* __restrict__ d_sil, float* __restrict__ d_flo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float s = sil_at[i];
    const bool balance = training && stats[3] > 0.f && stats[4] > 0.f;
    const float gi = g_img ? g_img[i] : 0.f, gs = g_sil ? g_sil[i] : 0.f, gf = g_flo ? g_flo[i] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_rgb[i * 3 + c] = gi * s * (2.f / 3.f) * (rgb[i * 3 + c] - img_at[i * 3 + c]);
    const float bal = balance ? 0.5f * (stats[0] / stats[1]) * s + 0.5f * (stats[0] / stats[2]) * (1.f - s) : 1.f;
    d_sil[i] = gs * 2.f * (sil[i] - s) * bal * vis_at[i];
    const float c = cfd_at[i];
    const float cn = stats[5] > 0.f ? c / (stats[6] / fmaxf(stats[5], 1.f)) : c;
#pragma unroll
    for (int k = 0; k < 2; ++k) d_flo[i * 2 + k] = gf * 2.f * (flo[i * 2 + k] - flo_at[i * 2 + k]) * cn * s;
}

// x[m].mean() of a trainer's loss assembly (moda.py:540-640) without the boolean gather: out = sum(x * m) / (k * sum(m)), x (N, k),
// m (N) any non-zero = selected; one workgroup.  Backward: dx = g * m / (k * sum(m)).
__global__ __launch_bounds__(1024) void masked_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m, int N, int k,
                                                               float* __restrict__ out) {
    __shared__ float red[16];
    float sx = 0.f, sm = 0.f;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const float mk = m[i] != 0.f ? 1.f : 0.f;
        sm += mk;
        float r = 0.f;
        for (int c = 0; c < k; ++c) r += x[(long long)i * k + c];
        sx += r * mk;
    }
    const float tx = block_sum_1024(sx, red);
    const float tm = block_sum_1024(sm, red);
    if (threadIdx.x == 0) { out[0] = tx / (tm * (float)k); out[1] = tm * (float)k; }
}

__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ m, int N, int k, const float* __restrict__ cnt,
                                                              const float* __restrict__ g, float* __restrict__ dx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * k) return;
    dx[i] = m[i / k] != 0.f ? g[0] / cnt[1] : 0.f;
}

// Per-row distance of two (N, F) arrays, the small reductions of the loss heads (loss_utils.py:200 feat_err, :216-221 the
// reprojection error: ||a - b||_2;  rendering.py:573-577 the rendered-feature error: mean_c (a - b)^2) -- one thread per row.
// Backward: da = g (a - b) / ||a - b|| (0 where the norm is 0, torch's norm backward), or g 2 (a - b) / F;  db = -da when asked.
template <bool BWD>
__global__ __launch_bounds__(256) void row_dist_kernel(const float* __restrict__ a, const float* __restrict__ b, long long N, int F,
                                                       int mean_sq, float* __restrict__ out, const float* __restrict__ g,
                                                       float* __restrict__ da, float* __restrict__ db) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float s = 0.f;
    for (int c = 0; c < F; ++c) {
        const float d = a[i * F + c] - b[i * F + c];
        s += d * d;
    }
    if (!BWD) {
        out[i] = mean_sq ? s / (float)F : sqrtf(s);
        return;
    }
    float k;
    if (mean_sq) k = 2.f * g[i] / (float)F;
    else { const float nrm = sqrtf(s); k = nrm > 0.f ? g[i] / nrm : 0.f; }
    for (int c = 0; c < F; ++c) {
        const float d = (a[i * F + c] - b[i * F + c]) * k;
        if (da) da[i * F + c] = d;
        if (db) db[i * F + c] = -d;
    }
}

// The weighted sum of a trainer's loss terms (moda.py:540-705) as ONE launch each way: term t = weight_t * mean over the selected
// rows of x_t (n_t, k_t); rows are selected by a mask of one of three kinds (or all of them).  A single workgroup walks the terms
// (a few thousand rays each): out[0] = the sum, out[1 + t] = term t, out[1 + T + t] = its denominator k_t * #selected.
constexpr int kMaxLossTerms = 16;
struct LossTerms { moda_loss_term t[kMaxLossTerms]; int n; };

DEVINL bool loss_row_selected(const moda_loss_term& q, long long i) {
    if (q.mask_kind == 1) return ((const float*)q.mask)[i] > 0.f;
    if (q.mask_kind == 2) return ((const unsigned char*)q.mask)[i] != 0;
    if (q.mask_kind == 3) return ((const float*)q.mask)[i] != 0.f;
    return true;
}

// sums of one term over the rows lane, lane + 64, ...: eight rows per lane in flight
template <int MK, bool K1>
DEVINL void loss_term_sums(const moda_loss_term& q, int lane, float& sx, float& sm) {
    for (long long base = 0; base < q.n; base += 64 * 8) {
        float xs[8], ms[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = base + u * 64 + lane;
            const bool ok = i < q.n;
            const long long ii = ok ? i : 0;
            bool sel = ok;                      // (the row index is clamped: every load is unconditional, nothing short-circuits)
            if (MK == 1) { const float mv = ((const float*)q.mask)[ii]; sel = ok & (mv > 0.f); }
            if (MK == 2) { const unsigned char mv = ((const unsigned char*)q.mask)[ii]; sel = ok & (mv != 0); }
            if (MK == 3) { const float mv = ((const float*)q.mask)[ii]; sel = ok & (mv != 0.f); }
            ms[u] = sel ? 1.f : 0.f;
            if (K1) {
                xs[u] = q.x[ii];
            } else {
                float r = 0.f;
                for (int c = 0; c < q.k; ++c) r += q.x[ii * q.k + c];
                xs[u] = r;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sm += ms[u];
            sx += ms[u] != 0.f ? xs[u] : 0.f;
        }
    }
}

__global__ __launch_bounds__(1024) void loss_terms_fwd_kernel(LossTerms a, float* __restrict__ out) {
    __shared__ float term_s[kMaxLossTerms];
    // (readfirstlane: the wave index is uniform, so the term is fetched with scalar loads from the kernel arguments; indexed by
    //  a per-lane value the compiler would copy all sixteen terms to scratch first -- 20-30 us of dispatch + spill for this kernel)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (wave < a.n) {                       // one wavefront per term (16 waves, <= 16 terms): the terms are summed concurrently
        const moda_loss_term q = a.t[wave];
        float sx = 0.f, sm = 0.f;
        // the mask kind and k == 1 are chosen ONCE, outside the row loop: with the tests inside it every load sat behind a branch
        // and the wave waited for each of them in turn (27 us for 8 x 2048 rows)
        switch (q.mask_kind * 2 + (q.k == 1 ? 1 : 0)) {
            case 0: loss_term_sums<0, false>(q, lane, sx, sm); break;
            case 1: loss_term_sums<0, true>(q, lane, sx, sm); break;
            case 2: loss_term_sums<1, false>(q, lane, sx, sm); break;
            case 3: loss_term_sums<1, true>(q, lane, sx, sm); break;
            case 4: loss_term_sums<2, false>(q, lane, sx, sm); break;
            case 5: loss_term_sums<2, true>(q, lane, sx, sm); break;
            case 6: loss_term_sums<3, false>(q, lane, sx, sm); break;
            default: loss_term_sums<3, true>(q, lane, sx, sm); break;
        }
        const float tx = wave_sum(sx), tm = wave_sum(sm);
        const float den = tm * (float)q.k;
        const float term = q.weight * (tx / den);        // nothing selected: 0 / 0 = NaN, as the mean of an empty selection
        if (lane == 0) { term_s[wave] = term; out[1 + wave] = term; out[1 + a.n + wave] = den; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int t = 0; t < a.n; ++t) total += term_s[t];    // in term order, as the reference adds them
        out[0] = total;
    }
}

__global__ __launch_bounds__(256) void loss_terms_bwd_kernel(LossTerms a, const float* __restrict__ out, const float* __restrict__ g) {
    const moda_loss_term q = a.t[blockIdx.y];
    if (!q.dx) return;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= q.n * q.k) return;
    q.dx[e] = loss_row_selected(q, e / q.k) ? g[0] * q.weight / out[1 + a.n + blockIdx.y] : 0.f;
}


// ---- the Sinkhorn iterations as ONE persistent launch (round 5) ----------------------------------------------------------
// feat_match's 20 iterations are 40 dependent matrix-vector sweeps forward (loss_utils.py:361-370) and 38 backward, each over the
// (N, G) matching matrix or its transpose: 32.8 MB as bf16 at cfg4's size, ~7 us per sweep as a launch of its own at the Infinity
// Cache's 4.6 TB/s -- 0.55 ms of a 6.5 ms training step.  The matrix never changes between the sweeps, and 1 / 256 of BOTH copies
// fits one CU: this kernel runs one workgroup per CU, which keeps its rows of Kmat in LDS (N / 256 = 8 rows x G bf16 = 128 KB) and
// its rows of KmatT in registers (G / 256 = 32 rows x N bf16 = 64 VGPRs of 512 threads), loads them ONCE, and then walks all the
// sweeps with a grid barrier between them; a sweep reads only the 8-32 KB vector of the previous one (from L2) and costs the
// barrier, the staging of that vector and ~1 us of arithmetic.
// Grid barrier: one flag word per workgroup, written (agent-scope release, after the workgroup's own barrier has drained its
// stores) with the sweep's number and polled by 256 lanes at once (agent-scope acquire) -- no read-modify-write on a shared
// address (256 workgroups' atomics on one word serialise at ~37 ns each).  It needs every workgroup resident at the same time:
// one per CU by its LDS, the grid = the CU count, launched on a stream that runs nothing else beside it (the captured training
// step is one stream; moda_match_sinkhorn refuses other shapes and the host falls back to the per-sweep launches).  Every spin is
// bounded: a workgroup that is never joined gives up after ~0.2 s and raises flags[nwg] (results are then garbage, not a hang).
constexpr int kSkThreads = 512;
constexpr int kSkMaxRowsK = 8;        // rows of Kmat per workgroup (LDS)
constexpr int kSkRowsT = 32;          // rows of KmatT per workgroup: 4 per wave (registers)
constexpr int kSkMaxN = 2048;         // 4 x 512 elements of a KmatT row per lane group

struct SinkArgs {
    const unsigned short* Kmat;       // (N, G) bf16
    const unsigned short* KmatT;      // (G, N) bf16
    int N, G, iters, dir;             // dir 0: forward (A[0] given -> Bm[t], A[t+1]); 1: backward (Ubar[T-1] given -> Wbar, Ubar)
    float* A;                         // (iters + 1, N)
    float* Bm;                        // (iters, G)
    float* Ubar;                      // (iters, G)
    float* Wbar;                      // (iters - 1, N)
    int* flags;                       // (nwg + 1) zeros on entry: per-workgroup sweep counters, then the time-out word
    int rowsK;                        // rows of Kmat per workgroup = ceil(N / nwg)
    int dbg;                          // timing experiments (MODA_SINK_DBG): 1 no barrier wait, 2 no vector staging, 4 no arithmetic -- wrong results
};

DEVINL float sk_dot8(const uint4& a, const float4& b0, const float4& b1) {
    return __builtin_bit_cast(float, a.x << 16) * b0.x + __builtin_bit_cast(float, a.x & 0xffff0000u) * b0.y +
           __builtin_bit_cast(float, a.y << 16) * b0.z + __builtin_bit_cast(float, a.y & 0xffff0000u) * b0.w +
           __builtin_bit_cast(float, a.z << 16) * b1.x + __builtin_bit_cast(float, a.z & 0xffff0000u) * b1.y +
           __builtin_bit_cast(float, a.w << 16) * b1.z + __builtin_bit_cast(float, a.w & 0xffff0000u) * b1.w;
}

__global__ __launch_bounds__(kSkThreads) void sinkhorn_resident_kernel(SinkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    const int N = a.N, G = a.G, nwg = gridDim.x, w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G8 = G >> 3;                                   // 16-byte chunks of a Kmat row
    uint4* krows = (uint4*)sk_smem;                          // [rowsK][G8]
    float* xbuf = (float*)(sk_smem + (size_t)a.rowsK * G8 * 16);           // max(N, G) floats
    float* part = xbuf + (N > G ? N : G);                                   // [8 waves][kSkMaxRowsK]
    // ---- this workgroup's rows: Kmat -> LDS, KmatT -> registers
    const int rk0 = w * a.rowsK;
    for (int i = tid; i < a.rowsK * G8; i += kSkThreads) {
        const int r = rk0 + i / G8;
        krows[i] = r < N ? ((const uint4*)(a.Kmat + (long long)r * G))[i % G8] : make_uint4(0u, 0u, 0u, 0u);
    }
    const int NK = N >> 9;                                   // 512-element groups of a KmatT row (N % 512 == 0, <= 4)
    uint4 treg[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = w * kSkRowsT + wave * 4 + i;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            treg[i][k] = (r < G && k < NK) ? ((const uint4*)(a.KmatT + (long long)r * N))[k * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
    }
    const int T = a.iters;
    const int nsweeps = a.dir == 0 ? 2 * T : 2 * (T - 1);
    const float p1 = 1.f / (float)N, p2 = 1.f / (float)G;
    bool dead = false;
    for (int s = 0; s < nsweeps; ++s) {
        // which matrix, which vectors (loss_utils.py:361-370 forward; the reverse sweep of autograd.FeatMatchFn backward)
        bool onT;
        const float* x; const float* cvec; float* out; int mode; float p;
        if (a.dir == 0) {
            const int t = s >> 1;
            onT = (s & 1) == 0;
            x = onT ? a.A + (long long)t * N : a.Bm + (long long)t * G;
            out = onT ? a.Bm + (long long)t * G : a.A + (long long)(t + 1) * N;
            cvec = nullptr; mode = 1; p = onT ? p2 : p1;
        } else {
            const int t = T - (s >> 1);                      // T, T-1, ..., 2
            onT = (s & 1) == 1;
            x = onT ? a.Wbar + (long long)(t - 2) * N : a.Ubar + (long long)(t - 1) * G;
            out = onT ? a.Ubar + (long long)(t - 2) * G : a.Wbar + (long long)(t - 2) * N;
            cvec = onT ? a.Bm + (long long)(t - 2) * G : a.A + (long long)(t - 1) * N;
            mode = 2; p = onT ? p2 : p1;
        }
        const int L = onT ? N : G;                           // length of x
        __syncthreads();                                     // (xbuf / part of the previous sweep are free)
        // (every load of bytes another workgroup wrote in THIS launch is an agent-scope relaxed atomic load -- global_load_dword
        //  sc1, past the non-coherent per-XCD L2 -- and every such store below the matching sc1 store: the hand-off then needs
        //  no release / acquire, i.e. no L2 write-back and no invalidate per sweep; MI355X_MICROARCH.md 'Valid forms', row 1)
        if (!(a.dbg & 2)) {   // 16-byte sc1 loads, up to four in flight per thread and ONE wait (a relaxed atomic load per dword was waited for
            // one at a time: 16 dependent round trips per sweep)
            const int L4 = L >> 2;
            typedef float sk_f4 __attribute__((ext_vector_type(4)));
            sk_f4 v0, v1, v2, v3;
            const int i0 = tid, i1 = tid + kSkThreads, i2 = tid + 2 * kSkThreads, i3 = tid + 3 * kSkThreads;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v0) : "v"(x + 4 * (i0 < L4 ? i0 : 0)) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v1) : "v"(x + 4 * (i1 < L4 ? i1 : 0)) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v2) : "v"(x + 4 * (i2 < L4 ? i2 : 0)) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v3) : "v"(x + 4 * (i3 < L4 ? i3 : 0)) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "memory");
            if (i0 < L4) ((sk_f4*)xbuf)[i0] = v0;
            if (i1 < L4) ((sk_f4*)xbuf)[i1] = v1;
            if (i2 < L4) ((sk_f4*)xbuf)[i2] = v2;
            if (i3 < L4) ((sk_f4*)xbuf)[i3] = v3;
            for (int i = tid + 4 * kSkThreads; i < L4; i += kSkThreads) {        // (vectors longer than 8192 floats: not served today)
                sk_f4 t;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(x + 4 * i) : "memory");
                ((sk_f4*)xbuf)[i] = t;
            }
        }
        __syncthreads();
        if (a.dbg & 4) {
            if (tid < 8) __hip_atomic_store(out + (onT ? w * kSkRowsT : rk0) + tid, xbuf[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (onT) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < NK) {
                    const float4 b0 = ((const float4*)xbuf)[(k * 512 + lane * 8) >> 2], b1 = ((const float4*)xbuf)[((k * 512 + lane * 8) >> 2) + 1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += sk_dot8(treg[i][k], b0, b1);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sum = wave_sum(acc[i]);
                const int r = w * kSkRowsT + wave * 4 + i;
                if (lane == 0 && r < G)
                    __hip_atomic_store(out + r, sweep_epilogue(sum, mode, p, cvec ? cvec[r] : 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            float acc[kSkMaxRowsK];
#pragma unroll
            for (int r = 0; r < kSkMaxRowsK; ++r) acc[r] = 0.f;
            for (int c = tid; c < G8; c += kSkThreads) {
                const float4 b0 = ((const float4*)xbuf)[2 * c], b1 = ((const float4*)xbuf)[2 * c + 1];
#pragma unroll
                for (int r = 0; r < kSkMaxRowsK; ++r)
                    if (r < a.rowsK) acc[r] += sk_dot8(krows[r * G8 + c], b0, b1);
            }
#pragma unroll
            for (int r = 0; r < kSkMaxRowsK; ++r) {
                const float sum = wave_sum(acc[r]);
                if (lane == 0) part[wave * kSkMaxRowsK + r] = sum;
            }
            __syncthreads();
            if (tid < a.rowsK && rk0 + tid < N) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < kSkThreads / 64; ++j) sum += part[j * kSkMaxRowsK + tid];      // fixed order: deterministic
                __hip_atomic_store(out + rk0 + tid, sweep_epilogue(sum, mode, p, cvec ? cvec[rk0 + tid] : 0.f), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (s + 1 == nsweeps) break;
        // ---- grid barrier: every workgroup's slice of `out` is visible to all before the next sweep stages it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its sc1 stores have reached memory ...
        __syncthreads();                                     // ... before the one lane that signals for all of them
        if (tid == 0) __hip_atomic_store(a.flags + w, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!dead && tid < nwg && !(a.dbg & 1)) {            // relaxed (sc1) polls: acquire polls would invalidate the L2 per iteration
            int spins = 0;
            while (__hip_atomic_load(a.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= s) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { dead = true; __hip_atomic_store(a.flags + nwg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        // (the polling waves join the barrier at the top of the next sweep before anyone loads the vector)
    }
    // A time-out must be VISIBLE without a host read of flags[nwg] (the launch may sit inside a captured graph): a workgroup that
    // gave up on a barrier overwrites its own slices of the chain's last vectors with NaN.  Every consumer sums over the whole
    // vector (match_expect over Bm[T-1], match_dbar over A / Ubar / Wbar), so the predictions, the loss and every gradient
    // behind them come out NaN instead of plausible garbage.
    // (the vote through this kernel's own LDS: __syncthreads_or brings a static __shared__ word, and dynamic + static LDS must
    //  stay within the 160 KB this launch already asks for)
    __syncthreads();
    int* dflag = (int*)part;
    if (tid == 0) *dflag = 0;
    __syncthreads();
    if (dead) *dflag = 1;
    __syncthreads();
    if (*dflag) {
        const float qnan = __builtin_nanf("");
        float* vg = a.dir == 0 ? a.Bm + (long long)(T - 1) * G : a.Ubar;      // (G,) vectors: rows w * kSkRowsT ... of KmatT
        float* vn = a.dir == 0 ? a.A + (long long)T * N : a.Wbar;              // (N,) vectors: rows rk0 ... of Kmat
        if (tid < kSkRowsT && w * kSkRowsT + tid < G) vg[w * kSkRowsT + tid] = qnan;
        if (tid < a.rowsK && rk0 + tid < N) vn[rk0 + tid] = qnan;
    }
}
}   // namespace

extern "C" int moda_normalize_rows(const float* x, int64_t M, int32_t F, float* y, const float* g, float* dx, void* stream) {
    if (M <= 0) return 0;
    if (!x || F < 1 || (!g && !y) || (g && !dx)) return MODA_EINVAL;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)M, (int)F, y, g, dx);
    return (int)hipGetLastError();
}

extern "C" int moda_match_matrix(const float* feats_n, const float* vol_n, int64_t N, int64_t G, int32_t F,
                                 const float* kappa, float* Kmat, int32_t kmat_bf16, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (F != kF) return MODA_ESHAPE;
    if (!feats_n || !vol_n || !kappa || !Kmat || N > 65535 || G > 0x7fffffff) return MODA_EINVAL;
    hipLaunchKernelGGL(featdot_exp_kernel, dim3((unsigned)((G + 255) / 256), (unsigned)((N + kFdRows - 1) / kFdRows)), dim3(256), 0, (hipStream_t)stream,
                       feats_n, vol_n, (int)N, (int)G, kappa, Kmat, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_match_matrix_rows(const float* feats_n, const float* vol_n, int64_t N, int64_t G, int32_t F, const float* kappa,
                                      float* Kmat, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (!feats_n || !vol_n || !kappa || !Kmat || F != kF || N > 65535) return MODA_EINVAL;
    hipLaunchKernelGGL(featdot_rows_exp_kernel, dim3((unsigned)((G + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, feats_n,
                       vol_n, (int)N, (int)G, kappa, Kmat);
    return (int)hipGetLastError();
}

extern "C" int moda_match_sweep(const float* Mat, int64_t R, int64_t C, const float* vec, int32_t mode, float p,
                                const float* c, float* out, int32_t kmat_bf16, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    if (!Mat || !vec || !out || mode < 0 || mode > 2 || (mode == 2 && !c)) return MODA_EINVAL;
    // (round 6, measured and dropped -- tools/gemv_ab.py: four 16-byte loads in flight per lane instead of two: 5.8 / 4.85 us per
    //  sweep of K / K^T against 5.7 / 4.9; two or four waves per row for the 2048 x 8000 orientation: 7.1 / 7.6 us.  Back to back
    //  on one matrix a sweep already runs at 5.8-6.7 TB/s from the Infinity Cache; inside the step, where the two orientations
    //  alternate and each sweep waits for the previous one's vector, it takes 7.0-7.3 us)
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Mat, vec, (int)R,
                       (int)C, mode, p, c, out, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_match_sinkhorn(const void* Kmat, const void* KmatT, int64_t N, int64_t G, int32_t iters, int32_t backward,
                                   float* A, float* Bm, float* Ubar, float* Wbar, int32_t* flags, int32_t flags_len, void* stream) {
    if (!Kmat || !KmatT || !A || !Bm || !flags || iters < 2 || (backward && (!Ubar || !Wbar))) return MODA_EINVAL;
    // the resident form: bf16 matrices, N a multiple of 512 up to 2048, G a multiple of 8; one workgroup per CU with its
    // N / CUs rows of Kmat in LDS; anything else is MODA_ESHAPE and the caller keeps the per-sweep launches
    if (N < 512 || N > kSkMaxN || (N & 511) || G < 8 || (G & 7)) return MODA_ESHAPE;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MODA_ESHAPE;
    static const int cap = [] { const char* e = getenv("MODA_SINKHORN_WGS"); return e ? atoi(e) : 0; }();
    int nwg = cap > 0 && cap < cus ? cap : cus;
    if (nwg < 1 || flags_len < nwg + 1) return MODA_ESHAPE;
    if ((int64_t)nwg * kSkRowsT < G || nwg > kSkThreads) return MODA_ESHAPE;   // KmatT rows must fit the workgroups' registers; one polling lane per workgroup
    const int rowsK = (int)((N + nwg - 1) / nwg);
    if (rowsK > kSkMaxRowsK) return MODA_ESHAPE;
    const size_t lds = (size_t)rowsK * (size_t)(G >> 3) * 16 + (size_t)(N > G ? N : G) * 4 + (kSkThreads / 64) * kSkMaxRowsK * 4;
    if (lds > 160 * 1024) return MODA_ESHAPE;
    static std::atomic<unsigned long long> attr_set{0ull};
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev > 63 || !(attr_set.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)sinkhorn_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MODA_ESHAPE;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sinkhorn_resident_kernel, kSkThreads, lds) != hipSuccess || per_cu < 1) return MODA_ESHAPE;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    SinkArgs a;
    a.Kmat = (const unsigned short*)Kmat; a.KmatT = (const unsigned short*)KmatT; a.N = (int)N; a.G = (int)G; a.iters = iters;
    a.dir = backward ? 1 : 0; a.A = A; a.Bm = Bm; a.Ubar = Ubar; a.Wbar = Wbar; a.flags = flags; a.rowsK = rowsK;
    { const char* e = getenv("MODA_SINK_DBG"); a.dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(sinkhorn_resident_kernel, dim3((unsigned)nwg), dim3(kSkThreads), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int moda_match_expect(const float* Kmat, const float* b, const float* query, int64_t N, int64_t G, float* pred,
                                 float* rowsum, int32_t kmat_bf16, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (!Kmat || !query || !pred || !rowsum) return MODA_EINVAL;
    hipLaunchKernelGGL(match_expect_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Kmat, b, query,
                       (int)N, (int)G, pred, rowsum, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_match_prob(const float* Kmat, const float* b, const float* rowsum, int64_t N, int64_t G, float* prob,
                               int32_t kmat_bf16, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (!Kmat || !rowsum || !prob || N > 65535) return MODA_EINVAL;
    hipLaunchKernelGGL(match_prob_kernel, dim3((unsigned)((G + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                       Kmat, b, rowsum, (int)N, (int)G, prob, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_match_ecols(const float* KmatT, const float* b, const float* rowsum, const float* g_pred, const float* pred,
                                const float* query, const float* g_probT, const float* s_prob, int64_t N, int64_t G, float p2,
                                float* ubar, int32_t kmat_bf16, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (!KmatT || !b || !rowsum || !g_pred || !pred || !query || !ubar) return MODA_EINVAL;
    if (g_probT && !s_prob) return MODA_EINVAL;
    hipLaunchKernelGGL(match_ecols_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, (hipStream_t)stream, KmatT, b, rowsum,
                       g_pred, pred, query, g_probT, s_prob, (int)N, (int)G, p2, ubar, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_match_dbar(const float* Kmat, const float* b, const float* rowsum, const float* g_pred, const float* pred,
                               const float* query, const float* A, const float* Ubar, int32_t T1, const float* Wbar,
                               const float* Bm, int32_t T2, const float* g_prob, const float* s_prob, int64_t N, int64_t G,
                               const float* kappa, float* Dbar, float* kappa_bar, int32_t kmat_bf16, void* stream) {
    if (N <= 0 || G <= 0) return 0;
    if (!Kmat || !rowsum || !g_pred || !pred || !query || !kappa || !Dbar || N > 65535) return MODA_EINVAL;
    if (g_prob && !s_prob) return MODA_EINVAL;
    if ((T1 > 0 && (!A || !Ubar)) || (T2 > 0 && (!Wbar || !Bm))) return MODA_EINVAL;
    if (T1 + T2 <= 40)
        hipLaunchKernelGGL(match_dbar_tiled_kernel<40>, dim3((unsigned)((G + 255) / 256), (unsigned)((N + 31) / 32)), dim3(256), 0,
                           (hipStream_t)stream, Kmat, b, rowsum, g_pred, pred, query, A, Ubar, (int)T1, Wbar, Bm, (int)T2, g_prob, s_prob,
                           (int)N, (int)G, kappa, Dbar, kappa_bar, (int)kmat_bf16);
    else
        hipLaunchKernelGGL(match_dbar_kernel, dim3((unsigned)((G + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                           Kmat, b, rowsum, g_pred, pred, query, A, Ubar, (int)T1, Wbar, Bm, (int)T2, g_prob, s_prob, (int)N, (int)G,
                           kappa, Dbar, kappa_bar, (int)kmat_bf16);
    return (int)hipGetLastError();
}

extern "C" int moda_logsig_loss(const float* x, const float* w, int64_t n, float sign, float scale, float* out,
                                const float* g_out, float* dx, void* stream) {
    if (n <= 0) return 0;
    if (!x || (!g_out && !out) || (g_out && !dx)) return MODA_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(logsig_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, (long long)n, sign,
                       scale, out, g_out, dx);
    return (int)hipGetLastError();
}

extern "C" int moda_s3im(const float* rgb, const float* tar, const float* mask, int64_t N, const int32_t* index, int32_t patch_h,
                         int32_t patch_w_total, float* loss, const float* g_loss, float* d_rgb, void* stream) {
    if (N <= 0 || N > (1 << 30) / 3 || patch_h < 3 || patch_w_total < 3 || !rgb || !tar || !mask || !index) return MODA_EINVAL;
    if ((!g_loss && !loss) || (g_loss && !d_rgb)) return MODA_EINVAL;
    S3imWin g;                                   // gaussian(4, 1.5) of loss_utils.py:575-577, normalised
    float s = 0.f, e[4];                         // fp32 values, fp32 sum and division, as torch.Tensor([...]) / .sum() there
    for (int x = 0; x < 4; ++x) { e[x] = (float)exp(-(double)((x - 2) * (x - 2)) / (2.0 * 1.5 * 1.5)); s += e[x]; }
    for (int x = 0; x < 4; ++x) g.w[x] = e[x] / s;
    if (g_loss)
        hipLaunchKernelGGL(s3im_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, rgb, tar, mask, (int)N, index, (int)patch_h,
                           (int)patch_w_total, g, loss, g_loss, d_rgb);
    else
        hipLaunchKernelGGL(s3im_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, rgb, tar, mask, (int)N, index, (int)patch_h,
                           (int)patch_w_total, g, loss, g_loss, d_rgb);
    return (int)hipGetLastError();
}

extern "C" int moda_ray_loss(const float* rgb, const float* sil, const float* flo, const float* valid, const float* img_at,
                             const float* sil_at, const float* vis_at, const float* flo_at, const float* cfd_at, int64_t N,
                             int32_t training, float* img_loss, float* sil_loss, float* flo_loss, uint8_t* sil_flo, float* stats,
                             const float* g_img, const float* g_sil, const float* g_flo, float* d_rgb, float* d_sil, float* d_flo,
                             void* stream) {
    if (N <= 0) return 0;
    if (N > (1 << 24) || !rgb || !sil || !flo || !img_at || !sil_at || !vis_at || !flo_at || !cfd_at || !stats) return MODA_EINVAL;
    if (d_rgb || d_sil || d_flo) {
        if (!d_rgb || !d_sil || !d_flo) return MODA_EINVAL;
        hipLaunchKernelGGL(ray_loss_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, sil, flo,
                           img_at, sil_at, vis_at, flo_at, cfd_at, (int)N, (int)training, stats, g_img, g_sil, g_flo, d_rgb, d_sil, d_flo);
    } else {
        if (!valid || !img_loss || !sil_loss || !flo_loss || !sil_flo) return MODA_EINVAL;
        hipLaunchKernelGGL(ray_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb, sil, flo, valid, img_at, sil_at, vis_at,
                           flo_at, cfd_at, (int)N, (int)training, img_loss, sil_loss, flo_loss, sil_flo, stats);
    }
    return (int)hipGetLastError();
}

extern "C" int moda_masked_mean(const float* x, const float* mask, int64_t N, int32_t k, float* out2, const float* g, float* dx,
                                void* stream) {
    if (N <= 0 || k < 1 || N > (1 << 24) || !mask || !out2) return MODA_EINVAL;
    if (dx) {
        if (!g) return MODA_EINVAL;
        hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((unsigned)((N * k + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, (int)N,
                           (int)k, out2, g, dx);
    } else {
        if (!x) return MODA_EINVAL;
        hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, mask, (int)N, (int)k, out2);
    }
    return (int)hipGetLastError();
}

extern "C" int moda_loss_terms(const moda_loss_term* terms, int32_t n_terms, float* out, const float* g, void* stream) {
    if (n_terms <= 0) return 0;
    if (!terms || n_terms > kMaxLossTerms || !out) return MODA_EINVAL;
    LossTerms a;
    a.n = n_terms;
    long long most = 0;
    for (int t = 0; t < n_terms; ++t) {
        const moda_loss_term& q = terms[t];
        if (!q.x && !g) return MODA_EINVAL;
        if (q.n < 1 || q.k < 1 || q.n > (1 << 24) || q.mask_kind < 0 || q.mask_kind > 3 || (q.mask_kind && !q.mask)) return MODA_EINVAL;
        a.t[t] = q;
        if (q.n * q.k > most) most = q.n * q.k;
    }
    if (g) hipLaunchKernelGGL(loss_terms_bwd_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n_terms), dim3(256), 0, (hipStream_t)stream, a, out, g);
    else hipLaunchKernelGGL(loss_terms_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, out);
    return (int)hipGetLastError();
}

extern "C" int moda_row_dist(const float* a, const float* b, int64_t N, int32_t F, int32_t mean_sq, float* out, const float* g,
                             float* da, float* db, void* stream) {
    if (N <= 0) return 0;
    if (!a || !b || F < 1 || (!g && !out) || (g && !da && !db)) return MODA_EINVAL;
    const dim3 grid((unsigned)((N + 255) / 256));
    if (g) hipLaunchKernelGGL(row_dist_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (long long)N, (int)F, (int)mean_sq, out, g, da, db);
    else hipLaunchKernelGGL(row_dist_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (long long)N, (int)F, (int)mean_sq, out, g, da, db);
    return (int)hipGetLastError();
}
