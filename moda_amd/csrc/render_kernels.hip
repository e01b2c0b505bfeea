// Per-sample geometry, sampling and compositing kernels of MoDA's rendering path for gfx950.
// These are HBM-/VALU-bound elementwise or per-ray-scan kernels; the MFMA work lives in mlp_fused.hip.
// Every kernel cites the reference lines (under nnutils/) whose arithmetic it reproduces.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "moda_hip.h"
#include <type_traits>

#include "moda_dev.h"

namespace {

constexpr int kBlock = 256;

static inline int nblocks(long long n, int per = kBlock) { return (int)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------------
// moda_linear_fwd: Y[r,o] = act(b[o] + sum_k W[o, col0+k] X[r,k]) -- 64x64 tile, 4x4 per thread, K step 16
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ X, long long R, long long K, long long ldx,
                                                     const float* __restrict__ Wt, long long O, long long ldw,
                                                     long long col0, const float* __restrict__ b, int act,
                                                     float* __restrict__ Y, long long ldy) {
    __shared__ float xs[16][64 + 1];
    __shared__ float ws[16][64 + 1];
    const int tx = threadIdx.x & 15;   // output column group
    const int ty = threadIdx.x >> 4;   // row group
    const long long r0 = (long long)blockIdx.x * 64;
    const long long o0 = (long long)blockIdx.y * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long long k0 = 0; k0 < K; k0 += 16) {
        // each thread stages 4 elements of X and of W: element e -> (row e/16 .. , k e%16)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = threadIdx.x + e * 256;   // 0..1023
            const int rr = idx >> 4;                // 0..63
            const int kk = idx & 15;
            const long long k = k0 + kk;
            xs[kk][rr] = (r0 + rr < R && k < K) ? X[(r0 + rr) * ldx + k] : 0.f;
            ws[kk][rr] = (o0 + rr < O && k < K) ? Wt[(o0 + rr) * ldw + col0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float xv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = xs[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long r = r0 + ty * 4 + i;
        if (r >= R) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long o = o0 + tx * 4 + j;
            if (o >= O) continue;
            float v = acc[i][j] + (b ? b[o] : 0.f);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = 1.f / (1.f + expf(-v));
            Y[r * ldy + o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// moda_embed_fwd (nerf.py:35-75)
// ------------------------------------------------------------------------------------------------
struct Window { float w[16]; };

__global__ void embed_kernel(const float* __restrict__ x, long long M, int C, int F, Window win, int normalize,
                             float* __restrict__ out, long long ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    const long long m = i / C;
    const int c = (int)(i - m * C);
    float v = x[i];
    if (normalize) {   // rays_d / rays_d.norm(2,-1) (rendering.py:64)
        float n2 = 0.f;
        for (int k = 0; k < C; ++k) n2 += x[m * C + k] * x[m * C + k];
        v = v / sqrtf(n2);
    }
    float* o = out + m * ldo;
    o[c] = v;
    for (int k = 0; k < F; ++k) {
        float sn, cs;
        sincosf(ldexpf(v, k), &sn, &cs);
        o[C + (2 * k) * C + c] = win.w[k] * sn;
        o[C + (2 * k + 1) * C + c] = win.w[k] * cs;
    }
}

// ------------------------------------------------------------------------------------------------
// moda_bone_transform_fwd (geom_utils.py:59-111, neudbs branch)
// ------------------------------------------------------------------------------------------------
__global__ void bone_transform_kernel(const float* __restrict__ bones, const float* __restrict__ rts, long long N, int B,
                                      float* __restrict__ out, const int* __restrict__ run_start) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * B) return;
    const int b = (int)(i % B);
    // rows that repeat their predecessor (the reference's per-ray copies of a frame's pose, moda.py:1302-1310) are left unwritten:
    // their consumers read the run's first row (moda_row_runs)
    if (run_start != nullptr && run_start[i / B] != (int)(i / B)) return;
    const float* bn = bones + b * 10;
    const float* dq = rts + i * 8;
    const Quat r = {dq[0], dq[1], dq[2], dq[3]};
    const Quat d = {dq[4], dq[5], dq[6], dq[7]};
    float R[9];
    quat_to_mat(r, R);                                   // :80
    const Quat rinv = {r.w, -r.x, -r.y, -r.z};           // quaternion_invert
    const Quat t = qmul(d, rinv);                        // :81  Tmat = 2 * (dq_d (x) dq_r^-1)[1:]
    const float cx = bn[0], cy = bn[1], cz = bn[2];
    float* o = out + i * 10;
    o[0] = R[0] * cx + R[1] * cy + R[2] * cz + 2.f * t.x;   // :85
    o[1] = R[3] * cx + R[4] * cy + R[5] * cz + 2.f * t.y;
    o[2] = R[6] * cx + R[7] * cy + R[8] * cz + 2.f * t.z;
    const Quat q0 = {bn[3], bn[4], bn[5], bn[6]};
    Quat q = qmul(r, q0);                                 // :86 quaternion_multiply = raw product, real part >= 0
    if (q.w < 0.f) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
    o[7] = bn[7]; o[8] = bn[8]; o[9] = bn[9];            // :109
}

// ------------------------------------------------------------------------------------------------
// Skinning logits (geom_utils.py:237-277) and the DQS warp (geom_utils.py:457-493), shared device code
// ------------------------------------------------------------------------------------------------
// logit_b = -10 * sum_k exp(ls_k) * (R^T (c - p))_k^2 * 100 * exp(aux0)   with R = matrix(normalize(q))
DEVINL float gauss_logit(const float* __restrict__ bn, float px, float py, float pz, float e_aux) {
    Quat q = {bn[3], bn[4], bn[5], bn[6]};
    const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);   // F.normalize (:196)
    q.w /= nrm; q.x /= nrm; q.y /= nrm; q.z /= nrm;
    float R[9];
    quat_to_mat(q, R);
    const float dx = bn[0] - px, dy = bn[1] - py, dz = bn[2] - pz;   // :256
    // R^T d  (:252,233): component i = sum_j R[j][i] d_j
    const float m0 = R[0] * dx + R[3] * dy + R[6] * dz;
    const float m1 = R[1] * dx + R[4] * dy + R[7] * dz;
    const float m2 = R[2] * dx + R[5] * dy + R[8] * dz;
    // :261 scale * mdis^2, :265 * 100 * exp(log_scale), :266 -10 * sum -- same operation order as the reference
    const float t0 = expf(bn[7]) * (m0 * m0) * 100.f * e_aux;
    const float t1 = expf(bn[8]) * (m1 * m1) * 100.f * e_aux;
    const float t2 = expf(bn[9]) * (m2 * m2) * 100.f * e_aux;
    return -10.f * (t0 + t1 + t2);
}

// ---- per-(set, bone) preparation: everything that does not depend on the sample ----------------------
// bones (nsets,B,10) -> prep (nsets,B,16) = [c(3) | R = matrix(normalize(q)) row-major (9) | exp(log scale)(3) | 0]
__global__ void bone_prep_kernel(const float* __restrict__ bones, long long n, float* __restrict__ prep) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* bn = bones + i * 10;
    Quat q = {bn[3], bn[4], bn[5], bn[6]};
    const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);   // F.normalize (:196)
    q.w /= nrm; q.x /= nrm; q.y /= nrm; q.z /= nrm;
    float R[9];
    quat_to_mat(q, R);
    float* o = prep + i * 16;
    o[0] = bn[0]; o[1] = bn[1]; o[2] = bn[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[3 + k] = R[k];
    o[12] = expf(bn[7]); o[13] = expf(bn[8]); o[14] = expf(bn[9]);   // :198
    o[15] = 0.f;
}

// dq (n,8) -> dq or dq_inverse(dq) (dual_quat.py:87-94)
__global__ void dq_prep_kernel(const float* __restrict__ dq, int invert, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = dq + i * 8;
    float* o = out + i * 8;
    if (invert) {
        const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        o[0] = q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = -q[3] / n2;
        o[4] = q[4] / n2; o[5] = -q[5] / n2; o[6] = -q[6] / n2; o[7] = -q[7] / n2;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = q[k];
    }
}

// ---- MFMA tables of the fused skin-MLP + warp kernel (layout: moda_dev.h) ---------------------------------------------
// One wave per (set, bone tile): lane l serves bone 32 tile + (l & 31) of the quadratic-form table, k half l >> 5.
__global__ __launch_bounds__(64) void warp_qtab_kernel(const float* __restrict__ bones, long long nsets, int B, int tiles,
                                                       const float* __restrict__ skin_aux, float* __restrict__ qtab,
                                                       const int* __restrict__ run_start) {
    const long long st = blockIdx.x;                 // set * tiles + tile
    const long long set = st / tiles;
    if (run_start != nullptr && run_start[set] != (int)set) return;     // a repeat of an earlier set: its slot is never read
    const int tile = (int)(st - set * tiles);
    const int lane = threadIdx.x, h = lane >> 5;
    const int b = tile * 32 + (lane & 31);
    float co[10];
    if (b < B) {
        const float* bn = bones + (set * B + b) * 10;
        Quat q = {bn[3], bn[4], bn[5], bn[6]};
        const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);   // F.normalize (:196)
        q.w /= nrm; q.x /= nrm; q.y /= nrm; q.z /= nrm;
        float R[9];
        quat_to_mat(q, R);
        // logit = -10 * 100 e^{aux} sum_k e^{ls_k} (sum_j R[j][k] d_j)^2 = -d^T A d,  d = c - p  (:251-266)
        const float g = 1000.f * expf(skin_aux[0]);
        const float s0 = g * expf(bn[7]), s1 = g * expf(bn[8]), s2 = g * expf(bn[9]);
        float A[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                A[i][j] = s0 * R[3 * i + 0] * R[3 * j + 0] + s1 * R[3 * i + 1] * R[3 * j + 1] + s2 * R[3 * i + 2] * R[3 * j + 2];
        const float cx = bn[0], cy = bn[1], cz = bn[2];
        const float ax = A[0][0] * cx + A[0][1] * cy + A[0][2] * cz;
        const float ay = A[1][0] * cx + A[1][1] * cy + A[1][2] * cz;
        const float az = A[2][0] * cx + A[2][1] * cy + A[2][2] * cz;
        // monomial order: (x^2, y^2 | z^2, xy | xz, yz | x, y | z, 1)
        co[0] = -A[0][0]; co[1] = -A[1][1]; co[2] = -A[2][2]; co[3] = -2.f * A[0][1]; co[4] = -2.f * A[0][2];
        co[5] = -2.f * A[1][2]; co[6] = 2.f * ax; co[7] = 2.f * ay; co[8] = 2.f * az; co[9] = -(cx * ax + cy * ay + cz * az);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) co[i] = 0.f;
        co[9] = -1e30f;                              // a padding bone: its softmax weight underflows to exactly 0
    }
    float* o = qtab + st * kWarpQFloats + lane;
#pragma unroll
    for (int f = 0; f < kWarpQFrags; ++f) o[f * 64] = h ? co[2 * f + 1] : co[2 * f];
}

__global__ __launch_bounds__(64) void warp_dqtab_kernel(const float* __restrict__ dq, int invert, long long nsets, int B, int tiles,
                                                        uint4* __restrict__ dqtab, const int* __restrict__ run_start) {
    const long long st = blockIdx.x;
    const long long set = st / tiles;
    if (run_start != nullptr && run_start[set] != (int)set) return;
    const int tile = (int)(st - set * tiles);
    const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
    // row r: component (r & 3) of the real / dual part, hi / lo half; rows 16..31 repeat 0..15 with real and dual swapped
    const int comp = (r & 3) + ((((r >> 2) ^ (r >> 4)) & 1) ? 4 : 0);
    const bool lo = (r & 8) != 0;
#pragma unroll
    for (int u = 0; u < kWarpDqFrags; ++u) {
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = tile * 32 + 16 * u + 8 * (j >> 2) + 4 * h + (j & 3);
            float v = 0.f;
            if (b < B) {
                const float* q = dq + (set * B + b) * 8;
                v = q[comp];
                if (invert) {                        // dq_inverse (dual_quat.py:87-94): conjugate / |real|^2
                    const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
                    v = ((comp & 3) ? -v : v) / n2;
                }
            }
            const __bf16 hi = (__bf16)v;
            const __bf16 res = lo ? (__bf16)(v - (float)hi) : hi;
            e[j] = __builtin_bit_cast(unsigned short, res);
        }
        uint4 w;
        w.x = e[0] | ((unsigned)e[1] << 16); w.y = e[2] | ((unsigned)e[3] << 16);
        w.z = e[4] | ((unsigned)e[5] << 16); w.w = e[6] | ((unsigned)e[7] << 16);
        dqtab[(st * kWarpDqFrags + u) * 64 + lane] = w;
    }
}

// logit from prepared bone data, in the reference's operation order (:256-266)
DEVINL float prep_logit(const float* __restrict__ P, float px, float py, float pz, float e_aux) {
    const float dx = P[0] - px, dy = P[1] - py, dz = P[2] - pz;
    const float m0 = P[3] * dx + P[6] * dy + P[9] * dz;     // R^T d
    const float m1 = P[4] * dx + P[7] * dy + P[10] * dz;
    const float m2 = P[5] * dx + P[8] * dy + P[11] * dz;
    const float t0 = P[12] * (m0 * m0) * 100.f * e_aux;
    const float t1 = P[13] * (m1 * m1) * 100.f * e_aux;
    const float t2 = P[14] * (m2 * m2) * 100.f * e_aux;
    return -10.f * (t0 + t1 + t2);
}

// One thread per sample.  Bones are processed in groups of kG with one running-max rescale per group
// (online softmax), so no per-bone array is needed and B stays a runtime value; the blended dual
// quaternion is accumulated under the same rescaling.  When every lane of a wave belongs to one ray
// (S a multiple of 64, or simply a long ray) the per-ray bone data is addressed through a wave-uniform
// pointer, which turns those loads into scalar loads.
#ifndef MODA_WARP_G
#define MODA_WARP_G 5
#endif
#ifndef MODA_WARP_SPT
#define MODA_WARP_SPT 4            // samples per thread of the hot warp configuration (4, 2, or 1 = one-sample kernel only)
#endif
constexpr int kG = MODA_WARP_G;   // bones per online-softmax group

// STAGE: the sample's row of B logits / skinning weights lives in LDS (`trow`, staged and written back by the kernel with
// coalesced accesses); without it the row is read and written in place in global memory, B floats per lane at a B-float
// stride -- every wave-instruction touches up to 50 cache lines, and the row is written twice and read once.
typedef float __attribute__((address_space(3))) lds_float;
template <bool WRITE_SKIN, bool DO_WARP, bool UNIFORM, bool HAS_DSKIN, bool STAGE = false>
DEVINL void warp_body(const float* __restrict__ prep, int bones_per_ray, int rps, const float* __restrict__ dqp,
                      const float* __restrict__ pts, const float* __restrict__ pts_tf, const float* __restrict__ dskin,
                      int dskin_bns, float e_aux,
                      long long i, long long n, long long S, int B, float* __restrict__ xyz_out,
                      float* __restrict__ skin_out, const float* __restrict__ cyc_ref, float* __restrict__ cyc_out,
                      lds_float* trow = nullptr) {
    const float px = pts[i * 3 + 0], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    // bone / transform sets: one per `rps` consecutive rays (rps = 1: per ray; rps = rays of a frame: per-frame tables)
    const int ray = UNIFORM ? __builtin_amdgcn_readfirstlane((int)n) : (int)n;
    const long long set = rps > 1 ? ray / rps : ray;
    const float* P0 = prep + (bones_per_ray ? set * B * 16 : 0);
    const float* Q0 = DO_WARP ? dqp + set * B * 8 : nullptr;
    const long long s_in_ray = i - n * S;
    const long long ds_base = dskin_bns ? n * B * S + s_in_ray : i * B;
    const long long ds_step = dskin_bns ? S : 1;

    float mx = -INFINITY, sum = 0.f;
    float bl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The loop body carries no branch: bone indices past B are clamped for the loads and their logits set to -inf by a
    // select (exp(-inf) = 0 drops them), and the MLP logits of group g+1 are requested before group g is evaluated.  The
    // earlier form guarded every bone and the dskin pointer with branches, so each logit load sat in its own block with
    // its own s_waitcnt vmcnt(0): 25 serialised memory round trips per sample (PMC: waves parked 3/4 of the time).
    float dnext[kG];
#pragma unroll
    for (int j = 0; j < kG; ++j)
        dnext[j] = HAS_DSKIN ? (STAGE ? trow[min(j, B - 1)] : dskin[ds_base + (long long)min(j, B - 1) * ds_step]) : 0.f;
    for (int g0 = 0; g0 < B; g0 += kG) {
        float l[kG];
#pragma unroll
        for (int j = 0; j < kG; ++j) l[j] = dnext[j];
#ifdef MODA_ABL_DSKIN_RT
        if (dskin) {
#else
        if (HAS_DSKIN) {   // compile-time: a run-time test here would put the prefetch in its own block, waited for at the join
#endif
#pragma unroll
            for (int j = 0; j < kG; ++j)
                dnext[j] = STAGE ? trow[min(g0 + kG + j, B - 1)] : dskin[ds_base + (long long)min(g0 + kG + j, B - 1) * ds_step];
        }
        float gm = -INFINITY;
#pragma unroll
        for (int j = 0; j < kG; ++j) {
            const int b = g0 + j;
            const float lg = prep_logit(P0 + min(b, B - 1) * 16, px, py, pz, e_aux) + l[j];   // :269
            l[j] = b < B ? lg : -INFINITY;
            // the logits are O(1e3): recomputing them for the normalised weights below could differ by an ulp (another
            // fma contraction) and un-normalise the softmax by 1e-5, so they are parked in the output row instead
            if (WRITE_SKIN && b < B) {
                if (STAGE) trow[b] = lg;      // (bones past b are still unread MLP logits: the row is rewritten front to back)
                else skin_out[i * B + b] = lg;
            }
            gm = fmaxf(gm, l[j]);
        }
        const float nm = fmaxf(mx, gm);
        const float sc = __expf(mx - nm);   // 0 for the first group (mx = -inf)
        sum *= sc;
        if (DO_WARP) {
#pragma unroll
            for (int k = 0; k < 8; ++k) bl[k] *= sc;
        }
#pragma unroll
        for (int j = 0; j < kG; ++j) {
            const float e = __expf(l[j] - nm);   // 0 for a clamped (masked) bone
            sum += e;
            if (DO_WARP) {
                const float* q = Q0 + min(g0 + j, B - 1) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) bl[k] = fmaf(e, q[k], bl[k]);   // :470 (un-normalised softmax weights)
            }
        }
        mx = nm;
    }
    if (WRITE_SKIN) {
        const float inv = 1.f / sum;
        if (STAGE) {
            for (int b = 0; b < B; ++b) trow[b] = __expf(trow[b] - mx) * inv;
        } else {
            float* so = skin_out + i * B;
            for (int b = 0; b < B; ++b) so[b] = __expf(so[b] - mx) * inv;   // :276 (the logits parked above, by this thread)
        }
    }
    if (DO_WARP) {
        // the common 1/sum factor cancels in dq_normalize up to rounding; apply it to follow the reference
        const float inv = 1.f / sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) bl[k] *= inv;
        float ox, oy, oz;
        // the weights come from `pts`; the transform applies to pts_tf when given (neu_dbs forward with a residual
        // field: skin at x, DQS of x + nerf_dis(x), geom_utils.py:420-425)
        const float* tp = pts_tf ? pts_tf : pts;
        dqs_apply(bl, tp[i * 3 + 0], tp[i * 3 + 1], tp[i * 3 + 2], &ox, &oy, &oz);
        xyz_out[i * 3 + 0] = ox;
        xyz_out[i * 3 + 1] = oy;
        xyz_out[i * 3 + 2] = oz;
        if (cyc_ref) {
            const float dx = cyc_ref[i * 3 + 0] - ox, dy = cyc_ref[i * 3 + 1] - oy, dz = cyc_ref[i * 3 + 2] - oz;
            cyc_out[i] = sqrtf(dx * dx + dy * dy + dz * dz);   // rendering.py:341
        }
    }
}

template <bool WRITE_SKIN, bool DO_WARP>
__global__ __launch_bounds__(kBlock) void warp_kernel(const float* __restrict__ prep, int bones_per_ray, int rps,
                                                     const float* __restrict__ dqp, const float* __restrict__ pts,
                                                     const float* __restrict__ pts_tf, const float* __restrict__ dskin,
                                                     int dskin_bns,
                                                     const float* __restrict__ skin_aux, long long N, long long S, int B,
                                                     float* __restrict__ xyz_out, float* __restrict__ skin_out,
                                                     const float* __restrict__ cyc_ref, float* __restrict__ cyc_out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N * S;
    if (!live) i = N * S - 1;   // keep the wave whole for the uniformity vote; the result is discarded
    const long long n = i / S;
    const float e_aux = expf(skin_aux[0]);   // log_scale.exp() (:265)
    const int n0 = __builtin_amdgcn_readfirstlane((int)n);
    const bool uniform = __all((int)n == n0) && live;
    if (__all(live) && uniform) {
        if (dskin)
            warp_body<WRITE_SKIN, DO_WARP, true, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, dskin_bns, e_aux, i, n, S, B, xyz_out,
                                                       skin_out, cyc_ref, cyc_out);
        else
            warp_body<WRITE_SKIN, DO_WARP, true, false>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, dskin_bns, e_aux, i, n, S, B, xyz_out,
                                                        skin_out, cyc_ref, cyc_out);
    } else if (live) {
        if (dskin)
            warp_body<WRITE_SKIN, DO_WARP, false, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, dskin_bns, e_aux, i, n, S, B, xyz_out,
                                                        skin_out, cyc_ref, cyc_out);
        else
            warp_body<WRITE_SKIN, DO_WARP, false, false>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, dskin_bns, e_aux, i, n, S, B, xyz_out,
                                                         skin_out, cyc_ref, cyc_out);
    }
}

// The training route's form (skinning weights saved for the backward, MLP logits sample-major (N,S,B)): the workgroup's
// 256 rows of B floats are one contiguous piece of memory.  It is copied into LDS with 16-byte coalesced loads, every
// thread rewrites its own row there (logits -> parked logits -> weights; rows are B words apart, B odd: no bank
// conflicts), and the finished tile goes out the same way.  98 -> see DESIGN.md (per-lane rows in global memory: 50 cache
// lines per wave-instruction).
template <bool DO_WARP>
__global__ __launch_bounds__(kBlock) void warp_staged_kernel(const float* __restrict__ prep, int bones_per_ray, int rps,
                                                            const float* __restrict__ dqp, const float* __restrict__ pts,
                                                            const float* __restrict__ pts_tf, const float* __restrict__ dskin,
                                                            const float* __restrict__ skin_aux, long long N, long long S, int B,
                                                            float* __restrict__ xyz_out, float* __restrict__ skin_out,
                                                            const float* __restrict__ cyc_ref, float* __restrict__ cyc_out) {
    extern __shared__ __attribute__((aligned(16))) float wtile[];
    const long long i_base = (long long)blockIdx.x * kBlock;
    const long long total = N * S;
    const int rows = (int)(total - i_base < kBlock ? total - i_base : kBlock);
    const int cnt = rows * B;
    if (dskin) {
        const float* src = dskin + i_base * B;
        if ((((uintptr_t)src) & 15) == 0) {
            for (int j = threadIdx.x; j < cnt / 4; j += kBlock) ((float4*)wtile)[j] = ((const float4*)src)[j];
            for (int j = (cnt / 4) * 4 + threadIdx.x; j < cnt; j += kBlock) wtile[j] = src[j];
        } else {
            for (int j = threadIdx.x; j < cnt; j += kBlock) wtile[j] = src[j];
        }
    }
    __syncthreads();
    long long i = i_base + threadIdx.x;
    const bool live = i < total;
    if (!live) i = total - 1;
    const long long n = i / S;
    const float e_aux = expf(skin_aux[0]);
    const int n0 = __builtin_amdgcn_readfirstlane((int)n);
    const bool uniform = __all((int)n == n0) && live;
    lds_float* trow = (lds_float*)wtile + (int)(i - i_base) * B;
    if (__all(live) && uniform) {
        if (dskin) warp_body<true, DO_WARP, true, true, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, 0, e_aux, i, n, S, B, xyz_out, skin_out, cyc_ref, cyc_out, trow);
        else warp_body<true, DO_WARP, true, false, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, 0, e_aux, i, n, S, B, xyz_out, skin_out, cyc_ref, cyc_out, trow);
    } else if (live) {
        if (dskin) warp_body<true, DO_WARP, false, true, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, 0, e_aux, i, n, S, B, xyz_out, skin_out, cyc_ref, cyc_out, trow);
        else warp_body<true, DO_WARP, false, false, true>(prep, bones_per_ray, rps, dqp, pts, pts_tf, dskin, 0, e_aux, i, n, S, B, xyz_out, skin_out, cyc_ref, cyc_out, trow);
    }
    __syncthreads();
    float* dst = skin_out + i_base * B;
    if ((((uintptr_t)dst) & 15) == 0) {
        for (int j = threadIdx.x; j < cnt / 4; j += kBlock) ((float4*)dst)[j] = ((const float4*)wtile)[j];
        for (int j = (cnt / 4) * 4 + threadIdx.x; j < cnt; j += kBlock) dst[j] = wtile[j];
    } else {
        for (int j = threadIdx.x; j < cnt; j += kBlock) dst[j] = wtile[j];
    }
}
// rows of B floats for kBlock samples must fit the default dynamic LDS limit
static inline bool warp_can_stage(int B, int dskin_bns) { return !dskin_bns && (size_t)kBlock * B * sizeof(float) <= 48 * 1024; }

// Hot configuration of the inference path (channel-major MLP logits (N,B,S), warp only, S a multiple of 64*SPT): every
// thread serves SPT consecutive samples of one ray.  The per-bone data is wave-uniform (scalar loads, amortised over
// SPT x 64 samples), each logit load is SPT x 4 bytes per lane (16 B at SPT = 4: 1 KiB per wave-instruction, SPT x
// the bytes in flight of the one-sample form) and the samples' independent arithmetic gives the VALU its ILP.  Same
// operation order per sample as warp_body.
template <int SPT>
__global__ __launch_bounds__(kBlock) void warp_multi_kernel(const float* __restrict__ prep, int bones_per_ray, int rps,
                                                           const float* __restrict__ dqp, const float* __restrict__ pts,
                                                           const float* __restrict__ pts_tf, const float* __restrict__ dskin,
                                                           const float* __restrict__ skin_aux,
                                                           long long N, long long S, int B, float* __restrict__ xyz_out,
                                                           const float* __restrict__ cyc_ref, float* __restrict__ cyc_out) {
    typedef float vecT __attribute__((ext_vector_type(SPT)));
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // thread -> SPT samples
    const long long i0 = t * SPT;
    if (i0 >= N * S) return;                                                   // whole waves: N*S is a multiple of 64*SPT
    const long long n = i0 / S;
    const int ray = __builtin_amdgcn_readfirstlane((int)n);
    const long long set = rps > 1 ? ray / rps : ray;
    const float e_aux = expf(skin_aux[0]);
    const float* P0 = prep + (bones_per_ray ? set * B * 16 : 0);
    const float* Q0 = dqp + set * B * 8;
    const long long ds_base = (long long)ray * B * S + (i0 - n * S);
    float px[SPT], py[SPT], pz[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        px[s] = pts[(i0 + s) * 3 + 0];
        py[s] = pts[(i0 + s) * 3 + 1];
        pz[s] = pts[(i0 + s) * 3 + 2];
    }
    float mx[SPT], sum[SPT], bl[SPT][8];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        mx[s] = -INFINITY;
        sum[s] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) bl[s][k] = 0.f;
    }
    vecT dnext[kG];
#pragma unroll
    for (int j = 0; j < kG; ++j) dnext[j] = *(const vecT*)(dskin + ds_base + (long long)min(j, B - 1) * S);
    for (int g0 = 0; g0 < B; g0 += kG) {
        vecT dcur[kG];
#pragma unroll
        for (int j = 0; j < kG; ++j) dcur[j] = dnext[j];
#pragma unroll
        for (int j = 0; j < kG; ++j) dnext[j] = *(const vecT*)(dskin + ds_base + (long long)min(g0 + kG + j, B - 1) * S);
        float l[kG][SPT], gm[SPT];
#pragma unroll
        for (int s = 0; s < SPT; ++s) gm[s] = -INFINITY;
#pragma unroll
        for (int j = 0; j < kG; ++j) {
            const int b = g0 + j;
            const float* P = P0 + min(b, B - 1) * 16;
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                const float lg = prep_logit(P, px[s], py[s], pz[s], e_aux) + dcur[j][s];   // :269
                l[j][s] = b < B ? lg : -INFINITY;
                gm[s] = fmaxf(gm[s], l[j][s]);
            }
        }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const float nm = fmaxf(mx[s], gm[s]);
            const float sc = __expf(mx[s] - nm);
            sum[s] *= sc;
#pragma unroll
            for (int k = 0; k < 8; ++k) bl[s][k] *= sc;
            mx[s] = nm;
        }
#pragma unroll
        for (int j = 0; j < kG; ++j) {
            const float* q = Q0 + min(g0 + j, B - 1) * 8;
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                const float e = __expf(l[j][s] - mx[s]);
                sum[s] += e;
#pragma unroll
                for (int k = 0; k < 8; ++k) bl[s][k] = fmaf(e, q[k], bl[s][k]);           // :470
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const float inv = 1.f / sum[s];
#pragma unroll
        for (int k = 0; k < 8; ++k) bl[s][k] *= inv;
        float ox, oy, oz;
        const float* tp = pts_tf ? pts_tf : pts;
        dqs_apply(bl[s], tp[(i0 + s) * 3 + 0], tp[(i0 + s) * 3 + 1], tp[(i0 + s) * 3 + 2], &ox, &oy, &oz);
        xyz_out[(i0 + s) * 3 + 0] = ox;
        xyz_out[(i0 + s) * 3 + 1] = oy;
        xyz_out[(i0 + s) * 3 + 2] = oz;
        if (cyc_ref) {
            const float dx = cyc_ref[(i0 + s) * 3 + 0] - ox, dy = cyc_ref[(i0 + s) * 3 + 1] - oy, dz = cyc_ref[(i0 + s) * 3 + 2] - oz;
            cyc_out[i0 + s] = sqrtf(dx * dx + dy * dy + dz * dz);   // rendering.py:341
        }
    }
}

// dqs_blend_skinning with given weights (geom_utils.py:457-517)
__global__ __launch_bounds__(kBlock) void dqs_kernel(const float* __restrict__ dq, int invert, const float* __restrict__ skin,
                                                    const float* __restrict__ pts, long long N, long long S, int B,
                                                    float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const long long n = i / S;
    const float* dqn = dq + n * B * 8;
    const float* sk = skin + i * B;
    float bl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const float* q = dqn + b * 8;
        const float w = sk[b];
        if (invert) {
            const float inv = 1.f / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            bl[0] += w * (q[0] * inv); bl[1] += w * (-q[1] * inv); bl[2] += w * (-q[2] * inv); bl[3] += w * (-q[3] * inv);
            bl[4] += w * (q[4] * inv); bl[5] += w * (-q[5] * inv); bl[6] += w * (-q[6] * inv); bl[7] += w * (-q[7] * inv);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) bl[k] += w * q[k];
        }
    }
    float ox, oy, oz;
    dqs_apply(bl, pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], &ox, &oy, &oz);
    out[i * 3 + 0] = ox;
    out[i * 3 + 1] = oy;
    out[i * 3 + 2] = oz;
}

// ------------------------------------------------------------------------------------------------
// Ray sampling (rendering.py:64-89, 112-113)
// ------------------------------------------------------------------------------------------------
DEVINL float z_at(float nr, float fr, long long s, long long S, int use_disp) {
    // torch.linspace(0,1,S): step = 1/(S-1); values mirrored from the end for the upper half
    const float step = S > 1 ? 1.f / (float)(S - 1) : 0.f;
    const float t = (s < S / 2) ? step * (float)s : 1.f - step * (float)(S - 1 - s);
    return use_disp ? 1.f / (1.f / nr * (1.f - t) + 1.f / fr * t) : nr * (1.f - t) + fr * t;
}

__global__ void sample_rays_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ nr,
                                   const float* __restrict__ fr, const float* __restrict__ u, float perturb, int use_disp,
                                   long long N, long long S, float* __restrict__ zv, float* __restrict__ xyz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const long long n = i / S;
    const long long s = i - n * S;
    float z = z_at(nr[n], fr[n], s, S, use_disp);
    if (perturb > 0.f && u != nullptr) {                  // :76-83
        const float zp = s > 0 ? z_at(nr[n], fr[n], s - 1, S, use_disp) : z;
        const float zn = s + 1 < S ? z_at(nr[n], fr[n], s + 1, S, use_disp) : z;
        const float lower = s > 0 ? 0.5f * (zp + z) : z;
        const float upper = s + 1 < S ? 0.5f * (z + zn) : z;
        z = lower + (upper - lower) * (perturb * u[i]);
    }
    zv[i] = z;
    xyz[i * 3 + 0] = ro[n * 3 + 0] + rd[n * 3 + 0] * z;   // :88-89
    xyz[i * 3 + 1] = ro[n * 3 + 1] + rd[n * 3 + 1] * z;
    xyz[i * 3 + 2] = ro[n * 3 + 2] + rd[n * 3 + 2] * z;
}

// The same arithmetic, four consecutive samples of a ray per thread (S % 4 == 0, N * S < 2^32): one 16-byte store of depths and
// three of positions per thread instead of four scalar stores per sample, 32-bit index arithmetic instead of a 64-bit division per
// sample.  Z = false: the depths are given (points_kernel's job), only the positions are written.
template <bool Z>
__global__ void sample_rays4_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ nr,
                                    const float* __restrict__ fr, const float* __restrict__ u, float perturb, int use_disp,
                                    unsigned N, unsigned S, float* __restrict__ zv, float* __restrict__ xyz) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned S4 = S >> 2;
    const unsigned n = t / S4;
    if (n >= N) return;
    const unsigned s0 = (t - n * S4) << 2;
    const long long i0 = (long long)n * S + s0;
    float z[4];
    if (Z) {
        const float nn = nr[n], ff = fr[n];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long s = s0 + k;
            z[k] = z_at(nn, ff, s, S, use_disp);
            if (perturb > 0.f && u != nullptr) {                  // :76-83
                const float zp = s > 0 ? z_at(nn, ff, s - 1, S, use_disp) : z[k];
                const float zn = s + 1 < S ? z_at(nn, ff, s + 1, S, use_disp) : z[k];
                const float lower = s > 0 ? 0.5f * (zp + z[k]) : z[k];
                const float upper = s + 1 < S ? 0.5f * (z[k] + zn) : z[k];
                z[k] = lower + (upper - lower) * (perturb * u[i0 + k]);
            }
        }
        *(float4*)(zv + i0) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
        const float4 q = *(const float4*)(zv + i0);
        z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
    }
    const float ox = ro[n * 3 + 0], oy = ro[n * 3 + 1], oz = ro[n * 3 + 2];
    const float dx = rd[n * 3 + 0], dy = rd[n * 3 + 1], dz = rd[n * 3 + 2];
    float p[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[3 * k + 0] = ox + dx * z[k];                      // :88-89
        p[3 * k + 1] = oy + dy * z[k];
        p[3 * k + 2] = oz + dz * z[k];
    }
    float4* o = (float4*)(xyz + i0 * 3);
    o[0] = make_float4(p[0], p[1], p[2], p[3]);
    o[1] = make_float4(p[4], p[5], p[6], p[7]);
    o[2] = make_float4(p[8], p[9], p[10], p[11]);
}

__global__ void points_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ zv,
                              long long N, long long S, float* __restrict__ xyz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const long long n = i / S;
    const float z = zv[i];
    xyz[i * 3 + 0] = ro[n * 3 + 0] + rd[n * 3 + 0] * z;
    xyz[i * 3 + 1] = ro[n * 3 + 1] + rd[n * 3 + 1] * z;
    xyz[i * 3 + 2] = ro[n * 3 + 2] + rd[n * 3 + 2] * z;
}

// ------------------------------------------------------------------------------------------------
// Compositing (rendering.py:183-237): one 64-lane wave per ray; the walk itself is composite_ray (moda_dev.h), shared with the
// fused 8 x 256 kernel's epilogue.
// ------------------------------------------------------------------------------------------------
DEVINL float wave_sum(float v) { return comp_wave_sum(v); }

struct CompGlobalLoader {
    const float* rgbsigma; const float* zv; const float* noise; const float* xyz; const float* vis_pred;
    long long n, S;
    float dnorm, ibeta, cbx, cby, cbz;
    bool clip;
    DEVINL void load(long long s, float& r, float& g, float& b, float& sraw, float& z, float& alpha) const {
        const long long i = n * S + s;
        const float4 rs = *(const float4*)(rgbsigma + i * 4);
        r = rs.x; g = rs.y; b = rs.z; sraw = rs.w;
        z = zv[i];
        const float delta = (s + 1 < S ? zv[i + 1] - z : 1e10f) * dnorm;       // :183-191
        alpha = comp_alpha(rs.w, noise != nullptr, noise ? noise[i] : 0.f, delta, ibeta);
        if (clip) {                                                             // :210-213
            const float* p = xyz + i * 3;
            if (fabsf(p[0]) > cbx || fabsf(p[1]) > cby || fabsf(p[2]) > cbz) alpha = 0.f;
        }
        if (vis_pred && vis_pred[i] < 0.5f) alpha = 0.f;                        // :214-215
    }
};

template <bool FEAT>
__global__ __launch_bounds__(kBlock) void composite_kernel(
    const float* __restrict__ rgbsigma, const float* __restrict__ feat, int F, const float* __restrict__ zv,
    const float* __restrict__ rd, const float* __restrict__ beta, const float* __restrict__ noise,
    const float* __restrict__ xyz, const float* __restrict__ clip, const float* __restrict__ vis_pred,
    const float* __restrict__ cyc, float rgb_filter_scale, long long N, long long S, float* __restrict__ rgb,
    float* __restrict__ feat_out, float* __restrict__ depth, float* __restrict__ sil, float* __restrict__ weights,
    float* __restrict__ visibility, float* __restrict__ vis_out, float* __restrict__ cyc_out,
    const int* __restrict__ n_live, float term_tau, int* __restrict__ n_used) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (n >= N) return;   // whole wave exits together
    // Early ray termination (opt-in; the reference composes all S samples, rendering.py:217-221): samples at or beyond
    // n_live[n] (a caller-supplied bound: their inputs may never have been computed) and, with term_tau > 0, samples whose
    // incoming transmittance T has fallen below term_tau get weight 0; what is dropped is at most term_tau of the ray's
    // weight.  n_used[n] = the number of samples that kept their weight.
    const long long s_end = n_live ? min((long long)n_live[n], S) : S;
    CompGlobalLoader ld;
    ld.rgbsigma = rgbsigma; ld.zv = zv; ld.noise = noise; ld.xyz = xyz; ld.vis_pred = vis_pred; ld.n = n; ld.S = S;
    ld.dnorm = comp_dnorm(rd, n);
    ld.ibeta = comp_ibeta(beta);
    ld.clip = clip != nullptr;
    ld.cbx = ld.cby = ld.cbz = 0.f;
    if (clip) { ld.cbx = clip[0]; ld.cby = clip[1]; ld.cbz = clip[2]; }
    const CompOut o{rgb, feat_out, depth, sil, weights, visibility, vis_out, cyc_out, n_used};
    composite_ray<FEAT>(ld, lane, n, S, s_end, term_tau, rgb_filter_scale, feat, F, vis_pred, cyc, o);
}

// ------------------------------------------------------------------------------------------------
// Hierarchical resampling (rendering.py:582-623): one workgroup per ray.
//   bins (N,nb), weights (N,nb-1) -> n_imp samples per ray; u (N,n_imp) or NULL for linspace(0,1,n_imp)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxBins = 1024;

__global__ __launch_bounds__(kBlock) void sample_pdf_kernel(const float* __restrict__ bins_g, const float* __restrict__ wts,
                                                           const float* __restrict__ u, long long N, int nb, int n_imp,
                                                           float* __restrict__ out) {
    __shared__ float cdf[kMaxBins];
    __shared__ float bins[kMaxBins];
    __shared__ float w[kMaxBins];
    const long long n = blockIdx.x;
    const int nw = nb - 1;   // pdf entries (N_samples_ in the reference)
    const float eps = 1e-5f;
    for (int i = threadIdx.x; i < nb; i += kBlock) bins[i] = bins_g[n * nb + i];
    // the weights are staged by the whole workgroup (one coalesced load): the running sum below keeps the reference's
    // sequential order (torch.cumsum), and read from global memory by one lane it was ~250 dependent round trips per ray
    for (int i = threadIdx.x; i < nw; i += kBlock) w[i] = wts[n * nw + i];
    __syncthreads();
    if (threadIdx.x < 64) {
        // pdf = (w + eps) / sum, cdf = [0, cumsum(pdf)] (:597-600) by one wavefront: every lane owns a run of consecutive
        // entries (sequential inside the run, as torch.cumsum is), the runs are joined by a shuffle scan.  The summation
        // order differs from a strictly sequential one by association only (~1e-7 relative on the cdf).
        const int lane = threadIdx.x;
        const int per = (nw + 63) / 64;
        const int b0 = min(lane * per, nw), b1 = min(b0 + per, nw);
        float part = 0.f;
        for (int i = b0; i < b1; ++i) part += w[i] + eps;
        const float tot = wave_sum(part);
        float run = 0.f;
        for (int i = b0; i < b1; ++i) run += (w[i] + eps) / tot;
        float incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float q = __shfl_up(incl, o, 64);
            if (lane >= o) incl += q;
        }
        float acc = incl - run;                      // sum of the pdf entries before this lane's run
        if (lane == 0) cdf[0] = 0.f;
        for (int i = b0; i < b1; ++i) {
            acc += (w[i] + eps) / tot;
            cdf[i + 1] = acc;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_imp; k += kBlock) {
        float uu;
        if (u) uu = u[n * n_imp + k];
        else {                                                                            // :604 linspace(0,1,n_imp)
            const float step = n_imp > 1 ? 1.f / (float)(n_imp - 1) : 0.f;
            uu = (k < n_imp / 2) ? step * (float)k : 1.f - step * (float)(n_imp - 1 - k);
        }
        // searchsorted(cdf, u, right=True): number of cdf entries <= u   (:610)
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;                                        // :611
        const int above = lo < nw ? lo : nw;                                              // :612
        float denom = cdf[above] - cdf[below];                                            // :618
        if (denom < eps) denom = 1.f;                                                     // :619
        out[n * n_imp + k] = bins[below] + (uu - cdf[below]) / denom * (bins[above] - bins[below]);   // :622
    }
}

// z_out (N, La+Lb) = sort(cat(a, b)) ascending (rendering.py:110): bitonic sort in LDS, one workgroup per ray
constexpr int kMaxSort = 2048;

// The same sorted depths WITH their origin (round 6): z_out[n][p] = the p-th smallest of cat(a[n], b[n]) and src[n][p] = its index in
// that concatenation (< La: coarse depth src, else importance depth src - La).  By rank, not by a sorting network: the position of
// an element is the number of elements in front of it (equal keys: a before b, then by index), so the keys come out exactly as
// torch.sort / merge_sort_kernel leave them and every position knows where it came from.  What it is for: the hierarchical final
// pass evaluates the networks at the merged depths, half of which are the coarse depths the pre-pass has just evaluated at
// (rendering.py:96-114 evaluates them twice); with the origin known the final pass runs on the importance depths alone and the
// coarse results are merged in (merge_rows_kernel).  One workgroup per ray, both rows in LDS, La + Lb compares per element.
__global__ __launch_bounds__(kBlock) void merge_index_kernel(const float* __restrict__ a, int La, const float* __restrict__ b, int Lb,
                                                            float* __restrict__ z_out, int* __restrict__ src) {
    __shared__ float keys[kMaxSort];
    const long long n = blockIdx.x;
    const int tot = La + Lb;
    for (int j = threadIdx.x; j < tot; j += kBlock) keys[j] = j < La ? a[n * La + j] : b[n * Lb + (j - La)];
    __syncthreads();
    // a half that is already ascending (the coarse depths always are; the importance depths whenever sample_pdf ran on sorted
    // uniforms: every deterministic call) is ranked by binary search instead of by counting: the rank of element t inside its own
    // sorted half is t, and its rank inside the other sorted half a lower / upper bound
    bool asc_a = true, asc_b = true;
    for (int j = threadIdx.x; j + 1 < tot; j += kBlock) {
        if (j + 1 < La) asc_a = asc_a && keys[j] <= keys[j + 1];
        else if (j >= La) asc_b = asc_b && keys[j] <= keys[j + 1];
    }
    const bool sa = __syncthreads_and(asc_a ? 1 : 0) != 0;
    const bool sb = __syncthreads_and(asc_b ? 1 : 0) != 0;
    auto lower = [&](int lo, int hi, float x) {          // first index in [lo, hi) with keys[i] >= x  (count of keys < x, + lo)
        while (lo < hi) { const int m = (lo + hi) >> 1; if (keys[m] < x) lo = m + 1; else hi = m; }
        return lo;
    };
    auto upper = [&](int lo, int hi, float x) {          // first index in [lo, hi) with keys[i] > x   (count of keys <= x, + lo)
        while (lo < hi) { const int m = (lo + hi) >> 1; if (keys[m] <= x) lo = m + 1; else hi = m; }
        return lo;
    };
    for (int t = threadIdx.x; t < tot; t += kBlock) {
        const float x = keys[t];
        int pos = 0;
        if (t < La) {        // among a: keys < x, or equal with a smaller index; among b: keys < x
            if (sa) pos += t;
            else for (int i = 0; i < La; ++i) pos += (keys[i] < x || (keys[i] == x && i < t)) ? 1 : 0;
            if (sb) pos += lower(La, tot, x) - La;
            else for (int j = La; j < tot; ++j) pos += keys[j] < x ? 1 : 0;
        } else {             // among a: keys <= x (a goes first on equal keys); among b: keys < x, or equal with a smaller index
            if (sa) pos += upper(0, La, x);
            else for (int i = 0; i < La; ++i) pos += keys[i] <= x ? 1 : 0;
            if (sb) pos += t - La;
            else for (int j = La; j < tot; ++j) pos += (keys[j] < x || (keys[j] == x && j < t)) ? 1 : 0;
        }
        z_out[n * tot + pos] = x;
        src[n * tot + pos] = t;
    }
}

// out (N, L, C) rows picked by origin: out[n][p] = src[n][p] < La ? a[n][src] : b[n][src - La]  (a (N, La, C), b (N, L - La, C))
__global__ __launch_bounds__(kBlock) void merge_rows_kernel(const int* __restrict__ src, long long NL, int L, int La, int C,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;      // one thread per (ray, position)
    if (i >= NL) return;
    const long long n = i / L;
    const int s = src[i];
    const float* from = s < La ? a + (n * La + s) * C : b + (n * (L - La) + (s - La)) * C;
    float* to = out + i * C;
    if (C == 4) *(float4*)to = *(const float4*)from;
    else for (int c = 0; c < C; ++c) to[c] = from[c];
}

__global__ __launch_bounds__(kBlock) void merge_sort_kernel(const float* __restrict__ a, int La, const float* __restrict__ b,
                                                           int Lb, long long N, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float keys[kMaxSort];
    const long long n = blockIdx.x;
    const int tot = La + Lb;
    int L = 1;
    while (L < tot) L <<= 1;
    for (int j = threadIdx.x; j < L; j += kBlock)
        keys[j] = j < La ? a[n * La + j] : (j < tot ? b[n * Lb + (j - La)] : INFINITY);
    __syncthreads();
    if (L <= kBlock) {
        // Short rows (the path's depths, L <= 256): one key per thread in a register; the compare-exchange stages whose
        // partner lies in the same wavefront (distance < 64: 33 of the 36 stages at L = 256) exchange through DPP operands /
        // lane swaps (comp_xor_lane, moda_dev.h: no LDS round trip), only the others go through LDS with a barrier.  The network
        // is unrolled for the compile-time sizes so that every stage has a constant distance.  Same network, same result.
        const int i = threadIdx.x;
        float v = keys[i < L ? i : 0];
        if (i >= L) v = INFINITY;
        auto stage = [&](int k, int j, float p) __attribute__((always_inline)) {
            const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
            v = keep_min ? fminf(v, p) : fmaxf(v, p);
        };
        auto network = [&](auto lc) __attribute__((always_inline)) {
            constexpr int LL = decltype(lc)::value;
#pragma unroll
            for (int k = 2; k <= LL; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    float p;
                    if (j >= 64) {
                        __syncthreads();
                        if (i < LL) keys[i] = v;
                        __syncthreads();
                        p = keys[(i ^ j) < LL ? (i ^ j) : 0];
                    } else {
                        p = j == 1 ? comp_xor_lane<1>(v) : j == 2 ? comp_xor_lane<2>(v) : j == 4 ? comp_xor_lane<4>(v)
                            : j == 8 ? comp_xor_lane<8>(v) : j == 16 ? comp_xor_lane<16>(v) : comp_xor_lane<32>(v);
                    }
                    stage(k, j, p);
                }
            }
        };
        // Both inputs already ascending (the coarse depths always are, the importance depths whenever sample_pdf ran on sorted
        // uniforms -- every deterministic call): [a ascending | +inf padding | b DESCENDING] is a bitonic sequence, and its
        // final merge phase alone (log2 L stages instead of log2 L (log2 L + 1) / 2) sorts it -- to the same keys in the same
        // places as the full network, the order of equal keys aside.
        bool asc = true;
        if (i + 1 < La) asc = keys[i] <= keys[i + 1];
        else if (i >= La && i + 1 < tot) asc = keys[i] <= keys[i + 1];
        const bool presorted = __syncthreads_and(asc ? 1 : 0) != 0;
        auto merge_phase = [&](auto lc) __attribute__((always_inline)) {
            constexpr int LL = decltype(lc)::value;
            v = i < La ? keys[i] : (i >= LL - Lb && i < LL ? keys[La + (LL - 1 - i)] : INFINITY);
#pragma unroll
            for (int j = LL >> 1; j > 0; j >>= 1) {
                float p;
                if (j >= 64) {
                    __syncthreads();
                    if (i < LL) keys[i] = v;
                    __syncthreads();
                    p = keys[(i ^ j) < LL ? (i ^ j) : 0];
                } else {
                    p = j == 1 ? comp_xor_lane<1>(v) : j == 2 ? comp_xor_lane<2>(v) : j == 4 ? comp_xor_lane<4>(v)
                        : j == 8 ? comp_xor_lane<8>(v) : j == 16 ? comp_xor_lane<16>(v) : comp_xor_lane<32>(v);
                }
                stage(LL, j, p);
            }
        };
        if (presorted && L == 256) merge_phase(std::integral_constant<int, 256>{});
        else if (presorted && L == 128) merge_phase(std::integral_constant<int, 128>{});
        else if (presorted && L == 64) merge_phase(std::integral_constant<int, 64>{});
        else if (L == 256) network(std::integral_constant<int, 256>{});
        else if (L == 128) network(std::integral_constant<int, 128>{});
        else if (L == 64) network(std::integral_constant<int, 64>{});
        else {
            for (int k = 2; k <= L; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    float p;
                    if (j >= 64) {
                        __syncthreads();
                        if (i < L) keys[i] = v;
                        __syncthreads();
                        p = keys[(i ^ j) < L ? (i ^ j) : 0];
                    } else {
                        p = __shfl_xor(v, j, 64);
                    }
                    stage(k, j, p);
                }
            }
        }
        if (i < tot) out[n * tot + i] = v;
        return;
    }
    for (int k = 2; k <= L; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < L; i += kBlock) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int j = threadIdx.x; j < tot; j += kBlock) out[n * tot + j] = keys[j];
}

// vec_to_sim3 (geom_utils.py:187-199): (n,10) -> center (n,3), orient (n,3,3), scale (n,3)
__global__ void vec_to_sim3_kernel(const float* __restrict__ vec, long long n, float* __restrict__ center,
                                   float* __restrict__ orient, float* __restrict__ scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = vec + i * 10;
    Quat q = {v[3], v[4], v[5], v[6]};
    const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);
    q.w /= nrm; q.x /= nrm; q.y /= nrm; q.z /= nrm;
    float R[9];
    quat_to_mat(q, R);
    for (int k = 0; k < 3; ++k) { center[i * 3 + k] = v[k]; scale[i * 3 + k] = expf(v[7 + k]); }
    for (int k = 0; k < 9; ++k) orient[i * 9 + k] = R[k];
}

// ------------------------------------------------------------------------------------------------
// dual_quat.py elementwise ops
// ------------------------------------------------------------------------------------------------
__global__ void dq_op_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, long long n,
                             float* __restrict__ out, int* __restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == MODA_DQ_QMUL) {
        const Quat x = {a[i * 4], a[i * 4 + 1], a[i * 4 + 2], a[i * 4 + 3]};
        const Quat y = {b[i * 4], b[i * 4 + 1], b[i * 4 + 2], b[i * 4 + 3]};
        const Quat o = qmul(x, y);
        out[i * 4] = o.w; out[i * 4 + 1] = o.x; out[i * 4 + 2] = o.y; out[i * 4 + 3] = o.z;
        return;
    }
    if (op == MODA_DQ_QNORMALIZE) {
        const float* q = a + i * 4;
        const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (flag && fabsf(nrm) <= 1e-8f) *flag = 1;   // torch.isclose(norm, 0): |norm| <= atol
        for (int k = 0; k < 4; ++k) out[i * 4 + k] = q[k] / nrm;
        return;
    }
    const float* p = a + i * 8;
    float* o = out + i * 8;
    if (op == MODA_DQ_DQMUL) {
        const float* q = b + i * 8;
        const Quat r1 = {p[0], p[1], p[2], p[3]}, d1 = {p[4], p[5], p[6], p[7]};
        const Quat r2 = {q[0], q[1], q[2], q[3]}, d2 = {q[4], q[5], q[6], q[7]};
        const Quat rr = qmul(r1, r2), x = qmul(r1, d2), y = qmul(d1, r2);
        o[0] = rr.w; o[1] = rr.x; o[2] = rr.y; o[3] = rr.z;
        o[4] = x.w + y.w; o[5] = x.x + y.x; o[6] = x.y + y.y; o[7] = x.z + y.z;
    } else if (op == MODA_DQ_NORMALIZE) {
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
        if (flag && fabsf(nrm) <= 1e-8f) *flag = 1;
        for (int k = 0; k < 8; ++k) o[k] = p[k] / nrm;
    } else if (op == MODA_DQ_QCONJ) {
        o[0] = p[0]; o[1] = -p[1]; o[2] = -p[2]; o[3] = -p[3]; o[4] = p[4]; o[5] = -p[5]; o[6] = -p[6]; o[7] = -p[7];
    } else if (op == MODA_DQ_CCONJ) {
        o[0] = p[0]; o[1] = -p[1]; o[2] = -p[2]; o[3] = -p[3]; o[4] = -p[4]; o[5] = p[5]; o[6] = p[6]; o[7] = p[7];
    } else if (op == MODA_DQ_INVERSE) {
        const float n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
        o[0] = p[0] / n2; o[1] = -p[1] / n2; o[2] = -p[2] / n2; o[3] = -p[3] / n2;
        o[4] = p[4] / n2; o[5] = -p[5] / n2; o[6] = -p[6] / n2; o[7] = -p[7] / n2;
    }
}

// ------------------------------------------------------------------------------------------------
// moda_mlp_pack: a NeRF's parameters -> the fused kernels' weight stream (moda_amd/mlp_pack.py layout) and bias block in
// ONE launch, straight from the parameter tensors.  Cheap enough to run at every call, so no host-side cache of packed
// weights exists that an optimiser could leave stale (torch's fused AdamW and graph-replayed steps update parameters
// without moving their version counters).  code = (source id << 24) | element offset, < 0 -> 0.
struct PackSrc {
    const float* w[16];
    const float* b[16];
};

__global__ __launch_bounds__(kBlock) void mlp_pack_kernel(PackSrc src, const int32_t* __restrict__ wcode, long long n_w8,
                                                          int bf16, void* __restrict__ wstream,
                                                          const int32_t* __restrict__ bcode, long long n_b,
                                                          float* __restrict__ bias, int* __restrict__ ovf) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_w8) {                       // 8 consecutive stream elements: one 16-byte (bf16) or two 16-byte (fp32) stores
        const int4 c0 = ((const int4*)wcode)[2 * i], c1 = ((const int4*)wcode)[2 * i + 1];
        const int32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c[e] < 0 ? 0.f : src.w[(c[e] >> 24) & 15][c[e] & 0xffffff];
        if (bf16 == 3) {
            // fp16 mode: round-to-nearest-even; a weight beyond fp16's range (it would become an infinity) is reported.  An element
            // whose code carries bit 30 belongs to a residual fragment (MODA_MLP_F16_HEADS): f16(v - f16(v))
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            union { f16x2 h[4]; uint4 u; unsigned w[4]; } o, r;
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o.h[e] = __builtin_convertvector((f32x2){v[2 * e], v[2 * e + 1]}, f16x2);
                bad = bad || (o.w[e] & 0x7fffu) >= 0x7c00u || ((o.w[e] >> 16) & 0x7fffu) >= 0x7c00u;
                const f32x2 back = __builtin_convertvector(o.h[e], f32x2);
                r.h[e] = __builtin_convertvector((f32x2){v[2 * e] - back[0], v[2 * e + 1] - back[1]}, f16x2);
                const bool lo0 = c[2 * e] >= 0 && (c[2 * e] & (1 << 30)), lo1 = c[2 * e + 1] >= 0 && (c[2 * e + 1] & (1 << 30));
                o.w[e] = (lo0 ? (r.w[e] & 0xffffu) : (o.w[e] & 0xffffu)) | (lo1 ? (r.w[e] & 0xffff0000u) : (o.w[e] & 0xffff0000u));
            }
            ((uint4*)wstream)[i] = o.u;
            if (bad && ovf != nullptr) __hip_atomic_store(ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (bf16) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            union { bf16x2 h[4]; uint4 u; unsigned w[4]; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.h[e] = __builtin_convertvector((f32x2){v[2 * e], v[2 * e + 1]}, bf16x2);   // RNE
            if (bf16 == 2) {
                // split mode: an element whose code carries bit 30 belongs to a residual fragment, bf16(v - bf16(v)); the
                // fragment before it holds bf16(v).  (Zero elements, code < 0, are zero in both.)
                union { bf16x2 h[4]; uint4 u; unsigned w[4]; } r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r0 = v[2 * e] - __builtin_bit_cast(float, o.w[e] << 16);
                    const float r1 = v[2 * e + 1] - __builtin_bit_cast(float, o.w[e] & 0xffff0000u);
                    r.h[e] = __builtin_convertvector((f32x2){r0, r1}, bf16x2);
                    const bool lo0 = c[2 * e] >= 0 && (c[2 * e] & (1 << 30)), lo1 = c[2 * e + 1] >= 0 && (c[2 * e + 1] & (1 << 30));
                    o.w[e] = (lo0 ? (r.w[e] & 0xffffu) : (o.w[e] & 0xffffu)) | (lo1 ? (r.w[e] & 0xffff0000u) : (o.w[e] & 0xffff0000u));
                }
            }
            ((uint4*)wstream)[i] = o.u;
        } else {
            ((float4*)wstream)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
            ((float4*)wstream)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    } else if (i - n_w8 < n_b) {
        const int32_t c = bcode[i - n_w8];
        bias[i - n_w8] = c < 0 ? 0.f : src.b[(c >> 24) & 15][c & 0xffffff];
    }
}

// ------------------------------------------------------------------------------------------------
// moda_fold_rows: the per-row code folds of a fused call -- Y_f[r, o] = b_f[o] + sum_k W_f[o, col0_f + k] X_f[r, k] for up to
// four (X, W, b) in ONE launch (blockIdx.y = fold), 16 rows per workgroup, one output column per thread, fp32 fmaf chains in k
// order.  The 64 x 64-tiled linear_kernel needs ~20 us for the single rest-pose row (eight barrier-separated k-steps in one
// workgroup) and the generic MFMA GEMM ~20-34 us for 8192 per-ray rows (64 workgroups): two or three such launches per network
// call were a third of a small call's fixed cost.
struct FoldDesc {
    const float* X; const float* W; const float* b; float* Y;
    const int* runs;          // null, or per row the first row of its run of identical rows: only those rows are computed
    int R, K, ldx, O, ldw, col0, ldy;
};
struct FoldArgs { FoldDesc f[4]; };

__global__ __launch_bounds__(256) void fold_rows_kernel(FoldArgs a) {
    const FoldDesc d = a.f[blockIdx.y];
    constexpr int RT = 16;
    extern __shared__ float xs[];                          // [RT][K]
    const int r0 = blockIdx.x * RT;
    if (r0 >= d.R) return;
    const int nr = min(RT, d.R - r0);
    unsigned live = 0xffffu;                               // rows of this tile that start a run (bit rr)
    if (d.runs != nullptr) {
        live = 0u;
        for (int rr = 0; rr < nr; ++rr) live |= (d.runs[r0 + rr] == r0 + rr ? 1u : 0u) << rr;      // uniform: scalar loads
        if (live == 0u) return;                            // every row repeats an earlier one: nothing to do (before any barrier)
    }
    for (int i = threadIdx.x; i < RT * d.K; i += 256) {
        const int rr = i / d.K, k = i - rr * d.K;
        xs[i] = rr < nr ? d.X[(long long)(r0 + rr) * d.ldx + k] : 0.f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d.O; o += 256) {
        const float* w = d.W + (long long)o * d.ldw + d.col0;
        float acc[RT];
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) acc[rr] = 0.f;
#pragma unroll 4
        for (int k = 0; k < d.K; ++k) {
            const float wv = w[k];
#pragma unroll
            for (int rr = 0; rr < RT; ++rr) acc[rr] = fmaf(wv, xs[rr * d.K + k], acc[rr]);
        }
        const float bv = d.b ? d.b[o] : 0.f;
#pragma unroll
        for (int rr = 0; rr < RT; ++rr)
            if (rr < nr && ((live >> rr) & 1u)) d.Y[(long long)(r0 + rr) * d.ldy + o] = bv + acc[rr];
    }
}

// fold_rows_mfma_kernel: the same products for the large folds, on the matrix pipe.  v_mfma_f32_32x32x2_f32 accumulates its two k
// steps as an fmaf chain in k order (the property the exact-fp32 MLP kernels rest on), so the results are BIT-IDENTICAL to the
// VALU kernel above (tests/test_gpu_parity.py::test_fold_rows_matrix_form_is_bit_identical_to_the_tile_form).  The VALU form walks
// its weight rows from global memory -- a thread per output column, 64 different cache lines per load instruction -- and reads
// every row element from LDS once per multiply-add: the per-ray direction fold of config 2 (65 536 rows, K = 91, O = 128:
// 0.76 G multiply-adds) took 92-150 us; a first rewrite with LDS-resident weights and 128-bit broadcast reads of the rows was
// LDS-issue-bound at 85 us.  Here a PERSISTENT workgroup stages the weight slice once, transposed ([k][o]), walks 32-row tiles
// (staged transposed too, [k][33]: both operand reads are conflict-free ds_read_b32), and each wave owns 32-column output blocks:
// A = W^T (lane: column o = l & 31, k-half l >> 5), B = the rows (lane: row r = l & 31), two operand registers per 2 048
// multiply-adds; a lane ends up with four consecutive output columns per accumulator quad and stores them as one float4.
typedef float fold_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void fold_rows_mfma_kernel(FoldArgs a) {
    const FoldDesc d = a.f[blockIdx.y];
    constexpr int RT = 32, XS = 33;
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int K2 = (d.K + 1) & ~1;                       // pad k to whole MFMAs (zeros: fmaf(0, 0, acc) = acc)
    const int NB = (d.O + 31) >> 5, OP = NB * 32;
    float* ws = fl;                                      // [K2][OP]
    float* xs = fl + (size_t)K2 * OP;                    // [K2][XS]
    const int ntiles = (d.R + RT - 1) / RT;
    if ((int)blockIdx.x >= ntiles) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
    // Staging walks a (rows x K) index space linearly (consecutive threads: consecutive k of one row -- coalesced) in steps of
    // 256 elements WITHOUT divisions (row / column advance by 256 / K and 256 % K with a carry) and four loads at a time: written
    // as `for (i ...) dst[f(i / K, i % K)] = src[...]` every element paid a division and a full memory round trip of its own
    // (the first version of this kernel spent 100 us there against 10 us of MFMAs).
    const int K = d.K, dr = 256 / K, dk = 256 - dr * K;
    const int row0 = (int)threadIdx.x / K, col0k = (int)threadIdx.x - row0 * K;
    auto stage = [&](const float* __restrict__ src, long long ld, int nrows, int nvalid, float* dst, int dst_ld) __attribute__((always_inline)) {
        // dst[k * dst_ld + row] = src[row * ld + k] for row < nvalid, 0 for the other rows of the nrows x K space; the pad
        // column K (K odd) is zeroed separately
        int row = row0, k = col0k;
        const int total = nrows * K;
        for (int i = threadIdx.x; i < total; i += 1024) {
            float v[4];
            int rw[4], kc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rw[u] = row; kc[u] = k;
                v[u] = (i + 256 * u < total && row < nvalid) ? src[(long long)row * ld + k] : 0.f;
                k += dk; row += dr;
                if (k >= K) { k -= K; row += 1; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + 256 * u < total) dst[kc[u] * dst_ld + rw[u]] = v[u];
        }
        if (K2 != K)
            for (int r = threadIdx.x; r < nrows; r += 256) dst[K * dst_ld + r] = 0.f;
    };
    stage(d.W + d.col0, d.ldw, OP, d.O, ws, OP);
    const bool vec_ok = (d.ldy & 3) == 0 && (((uintptr_t)d.Y) & 15) == 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = t * RT;
        const int nr = min(RT, d.R - r0);
        unsigned live = 0xffffffffu;
        if (d.runs != nullptr) {                         // one load per lane, the same answer in every wave
            const bool st = l32 < nr && half == 0 && d.runs[r0 + l32] == r0 + l32;
            live = (unsigned)__ballot(st);
            if (live == 0u) continue;                    // uniform over the workgroup
        }
        __syncthreads();                                 // the previous tile's readers are done (first time: ws is complete)
        stage(d.X + (long long)r0 * d.ldx, d.ldx, RT, nr, xs, XS);
        __syncthreads();
        for (int ob = wave; ob < NB; ob += 4) {
            fold_f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const float* wp = ws + ob * 32 + l32 + half * OP;
            const float* xp = xs + l32 + half * XS;
#pragma unroll 4
            for (int k = 0; k < K2; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[k * OP], xp[k * XS], acc, 0, 0, 0);
            // lane (row r = l32 of the tile, half): accumulator register 4q + i = output column 32 ob + 8q + 4 half + i
            if (l32 < nr && ((live >> l32) & 1u)) {
                float* y = d.Y + (long long)(r0 + l32) * d.ldy;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = ob * 32 + 8 * q + 4 * half;
                    if (vec_ok && o + 3 < d.O) {
                        float4 v;
                        v.x = (d.b ? d.b[o] : 0.f) + acc[4 * q]; v.y = (d.b ? d.b[o + 1] : 0.f) + acc[4 * q + 1];
                        v.z = (d.b ? d.b[o + 2] : 0.f) + acc[4 * q + 2]; v.w = (d.b ? d.b[o + 3] : 0.f) + acc[4 * q + 3];
                        *(float4*)(y + o) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (o + i < d.O) y[o + i] = (d.b ? d.b[o + i] : 0.f) + acc[4 * q + i];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// moda_row_runs: run_start[n] = index of the first row of the run of identical consecutive rows that row n belongs to.
// The reference's ray layout repeats every per-frame row (bone_rts, the bones transformed by it) for each of a frame's rays
// (moda.py:1281-1311): per-set data derived from such rows (the MFMA operand tables of the fused warp) need to be built once
// per RUN, at the slot of the run's first row, instead of once per ray (134 MB of tables per warp at config 2).
// Three small launches: flags (one wavefront per row), per-block max-scan, the carry of the earlier blocks.
// one wavefront per row: is it bit-identical to the row before?
struct RunSrc { const float* p[4]; int f[4]; };
__global__ __launch_bounds__(256) void row_runs_flag_kernel(RunSrc src, int N, int* __restrict__ run_start) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    bool diff = n == 0;
    if (n > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (src.p[s] == nullptr) continue;
            const int f = src.f[s];
            const unsigned* p = (const unsigned*)(src.p[s] + (long long)n * f);
            for (int i = lane; i < f; i += 64) diff = diff || p[i] != p[i - f];
        }
    }
    const bool starts = __ballot(diff) != 0ull;
    if (lane == 0) run_start[n] = starts ? n : -1;
}

__global__ __launch_bounds__(256) void row_runs_local_kernel(int N, int* __restrict__ run_start, int* __restrict__ block_last) {
    __shared__ int sc[256];
    const int n = blockIdx.x * 256 + threadIdx.x;
    sc[threadIdx.x] = n < N ? run_start[n] : -1;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                    // inclusive max-scan: the nearest run start at or before this row
        const int v = threadIdx.x >= o ? sc[threadIdx.x - o] : -1;
        __syncthreads();
        sc[threadIdx.x] = max(sc[threadIdx.x], v);
        __syncthreads();
    }
    if (n < N) run_start[n] = sc[threadIdx.x];             // -1: the run began in an earlier block
    if (threadIdx.x == 255) block_last[blockIdx.x] = sc[255];
}

__global__ __launch_bounds__(256) void row_runs_carry_kernel(int N, int* __restrict__ run_start, const int* __restrict__ block_last) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N || run_start[n] >= 0) return;
    int b = blockIdx.x - 1;                                // (row 0 always starts a run, so the walk ends)
    while (b > 0 && block_last[b] < 0) --b;
    run_start[n] = block_last[b];
}

// Diagnostic (tools/poison_check.py): fill the LDS of every CU with a pattern, so that a kernel that reads LDS it has not written
// -- whose result then depends on what ran before it -- shows up as a difference between two patterns.
__global__ __launch_bounds__(1024) void dbg_poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned plds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) plds[i] = pattern;
    __syncthreads();
    if (sink != nullptr && plds[(threadIdx.x * 37) % (160 * 1024 / 4)] == 0x12345u) sink[0] = 1u;      // keeps the stores alive
}

}   // namespace

#define ST(s) ((hipStream_t)(s))
#define LAUNCH_RC() ((int)hipGetLastError())

extern "C" int moda_abi_version(void) { return 9; }

extern "C" uint64_t moda_stream_capture_id(void* stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(ST(stream), &status, &id) != hipSuccess || status != hipStreamCaptureStatusActive) return 0;
    return id ? (uint64_t)id : 1;
}

extern "C" int moda_linear_fwd(const float* X, int64_t R, int64_t K, int64_t ldx, const float* Wt, int64_t O, int64_t ldw,
                               int64_t col0, const float* b, int32_t act, float* Y, int64_t ldy, void* stream) {
    if (R <= 0 || O <= 0) return 0;
    if (!X || !Wt || !Y || K < 0) return MODA_EINVAL;
    dim3 grid((unsigned)((R + 63) / 64), (unsigned)((O + 63) / 64));
    hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, ST(stream), X, (long long)R, (long long)K, (long long)ldx, Wt,
                       (long long)O, (long long)ldw, (long long)col0, b, act, Y, (long long)ldy);
    return LAUNCH_RC();
}

extern "C" int moda_mlp_pack(const void* const* wsrc, int32_t n_wsrc, const int32_t* wcode, int64_t n_w, int32_t bf16,
                             void* wstream, const void* const* bsrc, int32_t n_bsrc, const int32_t* bcode, int64_t n_b,
                             float* bias, int32_t* overflow, void* stream) {
    if (n_w <= 0 && n_b <= 0) return 0;
    if (n_wsrc < 0 || n_wsrc > 16 || n_bsrc < 0 || n_bsrc > 16 || (n_w & 7) || bf16 < 0 || bf16 > 3 || (n_w > 0 && (!wsrc || !wcode || !wstream)) ||
        (n_b > 0 && (!bsrc || !bcode || !bias)))
        return MODA_EINVAL;
    PackSrc src;
    for (int i = 0; i < 16; ++i) {
        src.w[i] = i < n_wsrc ? (const float*)wsrc[i] : nullptr;
        src.b[i] = i < n_bsrc ? (const float*)bsrc[i] : nullptr;
    }
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(nblocks(n_w / 8 + n_b)), dim3(kBlock), 0, ST(stream), src, wcode, (long long)(n_w / 8),
                       bf16, wstream, bcode, (long long)n_b, bias, (int*)overflow);
    return LAUNCH_RC();
}

extern "C" int moda_fold_rows(int32_t n, const float* const* X, const int64_t* R, const int64_t* K, const int64_t* ldx,
                              const float* const* W, const int64_t* O, const int64_t* ldw, const int64_t* col0,
                              const float* const* b, float* const* Y, const int64_t* ldy, const int32_t* const* run_start,
                              void* stream) {
    if (n <= 0) return 0;
    if (n > 4 || !X || !R || !K || !ldx || !W || !O || !ldw || !col0 || !b || !Y || !ldy) return MODA_EINVAL;
    FoldArgs a;
    long long rmax = 0, kmax = 0;
    for (int i = 0; i < 4; ++i) {
        const int j = i < n ? i : 0;
        if (!X[j] || !W[j] || !Y[j] || R[j] < 1 || K[j] < 1 || K[j] > 512 || O[j] < 1 || R[j] > 0x7fffffffLL) return MODA_EINVAL;
        a.f[i] = FoldDesc{X[j], W[j], b[j], Y[j], run_start ? (const int*)run_start[j] : nullptr, (int)(i < n ? R[j] : 0), (int)K[j], (int)ldx[j], (int)O[j], (int)ldw[j], (int)col0[j],
                          (int)ldy[j]};
        if (i < n) { rmax = R[j] > rmax ? R[j] : rmax; kmax = K[j] > kmax ? K[j] : kmax; }
    }
    // many rows: the persistent matrix-pipe form with LDS-resident weights (bit-identical arithmetic); few rows: a workgroup per
    // 16-row tile
    long long omax = 0;
    for (int i = 0; i < n; ++i) omax = O[i] > omax ? O[i] : omax;
    const long long K2 = (kmax + 1) & ~1LL, OPm = (omax + 31) / 32 * 32;
    const size_t lds2 = (size_t)(K2 * OPm + K2 * 33) * sizeof(float);
    static const bool v2_off = [] { const char* e = getenv("MODA_FOLD_MFMA"); return e && e[0] == '0'; }();
    if (rmax >= 1024 && lds2 <= 64 * 1024 && !v2_off) {
        const long long ntiles = (rmax + 31) / 32;
        hipLaunchKernelGGL(fold_rows_mfma_kernel, dim3((unsigned)(ntiles < 512 ? ntiles : 512), (unsigned)n), dim3(256), lds2, ST(stream), a);
        return LAUNCH_RC();
    }
    hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)((rmax + 15) / 16), (unsigned)n), dim3(kBlock), (size_t)(16 * kmax * sizeof(float)),
                       ST(stream), a);
    return LAUNCH_RC();
}

extern "C" int moda_dbg_poison_lds(uint32_t pattern, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)dbg_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MODA_EINVAL;
        attr = true;
    }
    hipLaunchKernelGGL(dbg_poison_lds_kernel, dim3(1024), dim3(1024), 160 * 1024, ST(stream), pattern, (unsigned*)nullptr);
    return LAUNCH_RC();
}

extern "C" int moda_embed_fwd(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window,
                              int32_t normalize, float* out, int64_t ldo, void* stream) {
    if (M <= 0) return 0;
    if (!x || !out || C < 1 || n_freq < 0 || n_freq > 16 || (n_freq > 0 && !window)) return MODA_EINVAL;
    if (ldo < (int64_t)C * (1 + 2 * n_freq)) return MODA_EINVAL;
    Window w;
    for (int i = 0; i < 16; ++i) w.w[i] = i < n_freq ? window[i] : 0.f;
    hipLaunchKernelGGL(embed_kernel, dim3(nblocks(M * C)), dim3(kBlock), 0, ST(stream), x, (long long)M, C, n_freq, w, normalize, out,
                       (long long)ldo);
    return LAUNCH_RC();
}

extern "C" int moda_bone_transform_fwd(const float* bones, const float* rts, int64_t N, int32_t B, float* out,
                                       const int32_t* run_start, void* stream) {
    if (N <= 0 || B <= 0) return 0;
    if (!bones || !rts || !out) return MODA_EINVAL;
    hipLaunchKernelGGL(bone_transform_kernel, dim3(nblocks(N * B)), dim3(kBlock), 0, ST(stream), bones, rts, (long long)N, B, out,
                       (const int*)run_start);
    return LAUNCH_RC();
}

static long long warp_ws_floats(long long N, int B, int per_ray) { return (per_ray ? N : 1) * (long long)B * 16 + N * (long long)B * 8; }

extern "C" int64_t moda_warp_workspace_floats(int64_t N, int32_t B, int32_t bones_per_ray) {
    return warp_ws_floats(N, B, bones_per_ray);
}

extern "C" int moda_skinning_fwd(const float* bones, int32_t bones_per_ray, const float* pts, const float* dskin,
                                 const float* skin_aux, int64_t N, int64_t S, int32_t B, float* skin, float* workspace,
                                 void* stream) {
    if (N <= 0 || S <= 0 || B <= 0) return 0;
    if (!bones || !pts || !skin_aux || !skin || !workspace) return MODA_EINVAL;
    const long long nsets = bones_per_ray ? N : 1;
    hipLaunchKernelGGL(bone_prep_kernel, dim3(nblocks(nsets * B)), dim3(kBlock), 0, ST(stream), bones, nsets * B, workspace);
    hipLaunchKernelGGL((warp_kernel<true, false>), dim3(nblocks(N * S)), dim3(kBlock), 0, ST(stream), workspace, bones_per_ray, 1,
                       (const float*)nullptr, pts, (const float*)nullptr, dskin, 0, skin_aux, (long long)N, (long long)S, B, (float*)nullptr, skin,
                       (const float*)nullptr, (float*)nullptr);
    return LAUNCH_RC();
}

extern "C" int moda_dqs_fwd(const float* dq, int32_t invert, const float* skin, const float* pts, int64_t N, int64_t S,
                            int32_t B, float* out, void* stream) {
    if (N <= 0 || S <= 0 || B <= 0) return 0;
    if (!dq || !skin || !pts || !out) return MODA_EINVAL;
    hipLaunchKernelGGL(dqs_kernel, dim3(nblocks(N * S)), dim3(kBlock), 0, ST(stream), dq, invert, skin, pts, (long long)N,
                       (long long)S, B, out);
    return LAUNCH_RC();
}

extern "C" int moda_warp_frames_fwd(const float* bones, int32_t bones_per_set, const float* dq, int64_t rays_per_set,
                                    int32_t invert, const float* pts, const float* pts_tf, const float* dskin, int32_t dskin_bns,
                                    const float* skin_aux, int64_t N, int64_t S, int32_t B, float* xyz_out, float* skin_out,
                                    const float* cyc_ref, float* cyc_out, float* workspace, void* stream) {
    if (N <= 0 || S <= 0 || B <= 0) return 0;
    if (!bones || !dq || !pts || !skin_aux || !xyz_out || !workspace) return MODA_EINVAL;
    if (cyc_ref && !cyc_out) return MODA_EINVAL;
    if (rays_per_set < 1 || N % rays_per_set != 0 || N > 0x7fffffffLL) return MODA_EINVAL;
    const long long nsets = N / rays_per_set;           // transform sets; bone sets: the same, or one for all rays
    const long long nbsets = bones_per_set ? nsets : 1;
    const int rps = (int)rays_per_set;
    float* prep = workspace;
    float* dqp = workspace + nbsets * B * 16;
    hipLaunchKernelGGL(bone_prep_kernel, dim3(nblocks(nbsets * B)), dim3(kBlock), 0, ST(stream), bones, nbsets * B, prep);
    hipLaunchKernelGGL(dq_prep_kernel, dim3(nblocks(nsets * B)), dim3(kBlock), 0, ST(stream), dq, invert, nsets * B, dqp);
    dim3 grid(nblocks(N * S)), block(kBlock);
    const bool al16 = ((((uintptr_t)dskin) | ((uintptr_t)pts)) & 15) == 0;
    if (!skin_out && dskin && dskin_bns && S % 256 == 0 && al16 && MODA_WARP_SPT == 4)
        hipLaunchKernelGGL((warp_multi_kernel<4>), dim3(nblocks(N * S / 4)), block, 0, ST(stream), prep, bones_per_set, rps, dqp,
                           pts, pts_tf, dskin, skin_aux, (long long)N, (long long)S, B, xyz_out, cyc_ref, cyc_out);
    else if (!skin_out && dskin && dskin_bns && S % 128 == 0 && al16 && MODA_WARP_SPT >= 2)
        hipLaunchKernelGGL((warp_multi_kernel<2>), dim3(nblocks(N * S / 2)), block, 0, ST(stream), prep, bones_per_set, rps, dqp,
                           pts, pts_tf, dskin, skin_aux, (long long)N, (long long)S, B, xyz_out, cyc_ref, cyc_out);
    else if (skin_out && warp_can_stage(B, dskin_bns))
        hipLaunchKernelGGL((warp_staged_kernel<true>), grid, block, (size_t)kBlock * B * sizeof(float), ST(stream), prep, bones_per_set,
                           rps, dqp, pts, pts_tf, dskin, skin_aux, (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    else if (skin_out)
        hipLaunchKernelGGL((warp_kernel<true, true>), grid, block, 0, ST(stream), prep, bones_per_set, rps, dqp, pts, pts_tf, dskin,
                           dskin_bns, skin_aux, (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    else
        hipLaunchKernelGGL((warp_kernel<false, true>), grid, block, 0, ST(stream), prep, bones_per_set, rps, dqp, pts, pts_tf, dskin,
                           dskin_bns, skin_aux, (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    return LAUNCH_RC();
}

extern "C" int32_t moda_warp_tiles(int32_t B) { return (B + 31) / 32; }

static int row_runs_launch(const RunSrc& src, int64_t N, int32_t* run_start, int32_t* workspace, void* stream) {
    const unsigned nb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(row_runs_flag_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ST(stream), src, (int)N, (int*)run_start);
    hipLaunchKernelGGL(row_runs_local_kernel, dim3(nb), dim3(256), 0, ST(stream), (int)N, (int*)run_start, (int*)workspace);
    if (nb > 1) hipLaunchKernelGGL(row_runs_carry_kernel, dim3(nb), dim3(256), 0, ST(stream), (int)N, (int*)run_start, (const int*)workspace);
    return LAUNCH_RC();
}

extern "C" int moda_row_runs(const float* rows_a, int64_t floats_a, const float* rows_b, int64_t floats_b, int64_t N,
                             int32_t* run_start, int32_t* workspace, void* stream) {
    if (N <= 0) return 0;
    if (!rows_a || floats_a < 1 || (rows_b && floats_b < 1) || !run_start || !workspace || N > 0x7fffffffLL) return MODA_EINVAL;
    RunSrc src{{rows_a, rows_b, nullptr, nullptr}, {(int)floats_a, (int)floats_b, 0, 0}};
    return row_runs_launch(src, N, run_start, workspace, stream);
}

extern "C" int moda_row_runs_multi(int32_t n_src, const float* const* rows, const int64_t* floats, int64_t N, int32_t* run_start,
                                   int32_t* workspace, void* stream) {
    if (N <= 0) return 0;
    if (n_src < 1 || n_src > 4 || !rows || !floats || !run_start || !workspace || N > 0x7fffffffLL) return MODA_EINVAL;
    RunSrc src{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};
    for (int i = 0; i < n_src; ++i) {
        if (!rows[i] || floats[i] < 1 || floats[i] > 0x7fffffffLL) return MODA_EINVAL;
        src.p[i] = rows[i];
        src.f[i] = (int)floats[i];
    }
    return row_runs_launch(src, N, run_start, workspace, stream);
}

extern "C" int moda_warp_tables_fwd(const float* bones, int64_t n_bone_sets, const float* dq, int64_t n_dq_sets, int32_t invert,
                                    const float* skin_aux, int32_t B, float* qtab, void* dqtab, const int32_t* run_start,
                                    void* stream) {
    if (B <= 0) return 0;
    if (B > 64 || n_bone_sets < 0 || n_dq_sets < 0) return MODA_ESHAPE;
    const int tiles = (B + 31) / 32;
    if (n_bone_sets * tiles > 0x7fffffffLL || n_dq_sets * tiles > 0x7fffffffLL) return MODA_ESHAPE;
    if (n_bone_sets > 0) {
        if (!bones || !skin_aux || !qtab) return MODA_EINVAL;
        hipLaunchKernelGGL(warp_qtab_kernel, dim3((unsigned)(n_bone_sets * tiles)), dim3(64), 0, ST(stream), bones,
                           (long long)n_bone_sets, B, tiles, skin_aux, qtab, n_bone_sets > 1 ? (const int*)run_start : nullptr);
    }
    if (n_dq_sets > 0) {
        if (!dq || !dqtab) return MODA_EINVAL;
        hipLaunchKernelGGL(warp_dqtab_kernel, dim3((unsigned)(n_dq_sets * tiles)), dim3(64), 0, ST(stream), dq, invert,
                           (long long)n_dq_sets, B, tiles, (uint4*)dqtab, (const int*)run_start);
    }
    return LAUNCH_RC();
}

extern "C" int moda_warp_fwd(const float* bones, int32_t bones_per_ray, const float* dq, int32_t invert, const float* pts,
                             const float* dskin, int32_t dskin_bns, const float* skin_aux, int64_t N, int64_t S, int32_t B,
                             float* xyz_out, float* skin_out, const float* cyc_ref, float* cyc_out, float* workspace,
                             void* stream) {
    return moda_warp_frames_fwd(bones, bones_per_ray, dq, 1, invert, pts, nullptr, dskin, dskin_bns, skin_aux, N, S, B, xyz_out, skin_out,
                                cyc_ref, cyc_out, workspace, stream);
}

extern "C" int moda_warp_prepped_fwd(const float* prep, int32_t per_ray, const float* q, const float* pts, const float* pts_tf,
                                     const float* dskin,
                                     int32_t dskin_bns, const float* skin_aux, int64_t N, int64_t S, int32_t B, float* xyz_out,
                                     float* skin_out, const float* cyc_ref, float* cyc_out, void* stream) {
    if (N <= 0 || S <= 0 || B <= 0) return 0;
    if (!prep || !q || !pts || !skin_aux || !xyz_out) return MODA_EINVAL;
    if (cyc_ref && !cyc_out) return MODA_EINVAL;
    dim3 grid(nblocks(N * S)), block(kBlock);
    if (skin_out && warp_can_stage(B, dskin_bns))
        hipLaunchKernelGGL((warp_staged_kernel<true>), grid, block, (size_t)kBlock * B * sizeof(float), ST(stream), prep, per_ray, 1, q,
                           pts, pts_tf, dskin, skin_aux, (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    else if (skin_out)
        hipLaunchKernelGGL((warp_kernel<true, true>), grid, block, 0, ST(stream), prep, per_ray, 1, q, pts, pts_tf, dskin, dskin_bns, skin_aux,
                           (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    else
        hipLaunchKernelGGL((warp_kernel<false, true>), grid, block, 0, ST(stream), prep, per_ray, 1, q, pts, pts_tf, dskin, dskin_bns, skin_aux,
                           (long long)N, (long long)S, B, xyz_out, skin_out, cyc_ref, cyc_out);
    return LAUNCH_RC();
}

extern "C" int moda_sample_rays_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                                    const float* u, float perturb, int32_t use_disp, int64_t N, int64_t S, float* z_vals,
                                    float* xyz, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!rays_o || !rays_d || !near || !far || !z_vals || !xyz) return MODA_EINVAL;
    if (perturb > 0.f && !u) return MODA_EINVAL;
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if (S % 4 == 0 && N * S < (1LL << 32) && al16(z_vals) && al16(xyz))      // four samples per thread, 16-byte stores
        hipLaunchKernelGGL(sample_rays4_kernel<true>, dim3(nblocks(N * (S / 4))), dim3(kBlock), 0, ST(stream), rays_o, rays_d, near, far,
                           u, perturb, use_disp, (unsigned)N, (unsigned)S, z_vals, xyz);
    else
        hipLaunchKernelGGL(sample_rays_kernel, dim3(nblocks(N * S)), dim3(kBlock), 0, ST(stream), rays_o, rays_d, near, far, u,
                           perturb, use_disp, (long long)N, (long long)S, z_vals, xyz);
    return LAUNCH_RC();
}

extern "C" int moda_points_fwd(const float* rays_o, const float* rays_d, const float* z_vals, int64_t N, int64_t S,
                               float* xyz, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!rays_o || !rays_d || !z_vals || !xyz) return MODA_EINVAL;
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if (S % 4 == 0 && N * S < (1LL << 32) && al16(z_vals) && al16(xyz))
        hipLaunchKernelGGL(sample_rays4_kernel<false>, dim3(nblocks(N * (S / 4))), dim3(kBlock), 0, ST(stream), rays_o, rays_d, nullptr,
                           nullptr, nullptr, 0.f, 0, (unsigned)N, (unsigned)S, (float*)z_vals, xyz);
    else
        hipLaunchKernelGGL(points_kernel, dim3(nblocks(N * S)), dim3(kBlock), 0, ST(stream), rays_o, rays_d, z_vals, (long long)N,
                           (long long)S, xyz);
    return LAUNCH_RC();
}

extern "C" int moda_composite_fwd(const float* rgbsigma, const float* feat, int32_t F, const float* z_vals,
                                  const float* rays_d, const float* beta, const float* noise, const float* xyz,
                                  const float* clip_bound, const float* vis_pred, const float* cyc, float rgb_filter_scale,
                                  int64_t N, int64_t S, float* rgb, float* feat_out, float* depth, float* sil, float* weights,
                                  float* visibility, float* vis_out, float* cyc_out, const int32_t* n_live, float term_tau,
                                  int32_t* n_used, void* stream) {
    if (N <= 0 || S <= 0) return 0;
    if (!rgbsigma || !z_vals || !rays_d || !beta || !rgb || !depth || !sil) return MODA_EINVAL;      // weights / visibility: optional
    if (feat && (F < 1 || F > kMaxFeat)) return MODA_ESHAPE;
    if (clip_bound && !xyz) return MODA_EINVAL;
    if (!(term_tau >= 0.f) || term_tau >= 1.f || S > 0x7fffffffLL) return MODA_EINVAL;
    if (feat)
        hipLaunchKernelGGL(composite_kernel<true>, dim3(nblocks(N, kBlock / 64)), dim3(kBlock), 0, ST(stream), rgbsigma, feat, F, z_vals,
                           rays_d, beta, noise, xyz, clip_bound, vis_pred, cyc, rgb_filter_scale, (long long)N, (long long)S, rgb,
                           feat_out, depth, sil, weights, visibility, vis_out, cyc_out, (const int*)n_live, term_tau, (int*)n_used);
    else      // no feature channels to composite: the instantiation without their accumulators (8 instead of 6 waves per SIMD)
        hipLaunchKernelGGL(composite_kernel<false>, dim3(nblocks(N, kBlock / 64)), dim3(kBlock), 0, ST(stream), rgbsigma, feat, F, z_vals,
                           rays_d, beta, noise, xyz, clip_bound, vis_pred, cyc, rgb_filter_scale, (long long)N, (long long)S, rgb,
                           feat_out, depth, sil, weights, visibility, vis_out, cyc_out, (const int*)n_live, term_tau, (int*)n_used);
    return LAUNCH_RC();
}

extern "C" int moda_sample_pdf_fwd(const float* bins, const float* weights, const float* u, int64_t N, int32_t n_bins,
                                   int32_t n_importance, float* samples, void* stream) {
    if (N <= 0 || n_importance <= 0) return 0;
    if (!bins || !weights || !samples) return MODA_EINVAL;
    if (n_bins < 2 || n_bins > kMaxBins) return MODA_ESHAPE;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)N), dim3(kBlock), 0, ST(stream), bins, weights, u, (long long)N, n_bins,
                       n_importance, samples);
    return LAUNCH_RC();
}

extern "C" int moda_merge_sort_fwd(const float* a, int32_t La, const float* b, int32_t Lb, int64_t N, float* out, void* stream) {
    if (N <= 0) return 0;
    if (!a || !out || La < 0 || Lb < 0 || (Lb > 0 && !b)) return MODA_EINVAL;
    if (La + Lb > kMaxSort || La + Lb < 1) return MODA_ESHAPE;
    hipLaunchKernelGGL(merge_sort_kernel, dim3((unsigned)N), dim3(kBlock), 0, ST(stream), a, La, b, Lb, (long long)N, out);
    return LAUNCH_RC();
}

extern "C" int moda_merge_index_fwd(const float* a, int32_t La, const float* b, int32_t Lb, int64_t N, float* z_out, int32_t* src,
                                    void* stream) {
    if (N <= 0) return 0;
    if (!a || !b || !z_out || !src || La < 1 || Lb < 1) return MODA_EINVAL;
    if (La + Lb > kMaxSort) return MODA_ESHAPE;
    hipLaunchKernelGGL(merge_index_kernel, dim3((unsigned)N), dim3(kBlock), 0, ST(stream), a, La, b, Lb, z_out, src);
    return LAUNCH_RC();
}

extern "C" int moda_merge_rows_fwd(const int32_t* src, int64_t N, int32_t L, int32_t La, int32_t C, const float* a, const float* b,
                                   float* out, void* stream) {
    if (N <= 0 || L <= 0) return 0;
    if (!src || !a || !b || !out || La < 0 || La > L || C < 1 || C > 64) return MODA_EINVAL;
    if (C == 4 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15)) return MODA_EINVAL;
    hipLaunchKernelGGL(merge_rows_kernel, dim3(nblocks(N * L)), dim3(kBlock), 0, ST(stream), src, (long long)N * L, L, La, C, a, b, out);
    return LAUNCH_RC();
}

extern "C" int moda_vec_to_sim3_fwd(const float* vec, int64_t n, float* center, float* orient, float* scale, void* stream) {
    if (n <= 0) return 0;
    if (!vec || !center || !orient || !scale) return MODA_EINVAL;
    hipLaunchKernelGGL(vec_to_sim3_kernel, dim3(nblocks(n)), dim3(kBlock), 0, ST(stream), vec, (long long)n, center, orient, scale);
    return LAUNCH_RC();
}

extern "C" int moda_dq_op(int32_t op, const float* a, const float* b, int64_t n, float* out, int32_t* flag, void* stream) {
    if (n <= 0) return 0;
    if (op < 0 || op > MODA_DQ_QNORMALIZE || !a || !out) return MODA_EINVAL;
    if ((op == MODA_DQ_QMUL || op == MODA_DQ_DQMUL) && !b) return MODA_EINVAL;
    hipLaunchKernelGGL(dq_op_kernel, dim3(nblocks(n)), dim3(kBlock), 0, ST(stream), op, a, b, (long long)n, out, flag);
    return LAUNCH_RC();
}
