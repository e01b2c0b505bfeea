// Per-(ray, bone) preparation of the skinning data for the training route, forward and hand-derived backward:
//   bone_transform (neudbs branch, reference nnutils/geom_utils.py:59-111), vec_to_sim3 in the warp kernels' layout
//   (geom_utils.py:187-199 -> [centre | R | exp(scale) | 0]), dq_inverse (nnutils/dual_quat.py:87-94).
// The reference differentiates these through a few hundred eager ops on (N,B,.) tensors; here each is one kernel
// forward and one backward (a few dozen flops per (ray, bone)), so a training step no longer spends its launch
// budget on them.  Quaternions are real-first; matrix(q) is pytorch3d's quaternion_to_matrix (scaled by 2/|q|^2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "moda_hip.h"

namespace {

#define DEVINL __device__ __forceinline__
constexpr int kBlock = 256;

struct Q4 { float w, x, y, z; };

DEVINL Q4 qmul(const Q4& a, const Q4& b) {   // Hamilton product a (x) b
    Q4 o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return o;
}
DEVINL Q4 qconj(const Q4& a) { return {a.w, -a.x, -a.y, -a.z}; }
DEVINL void qacc(Q4& a, const Q4& b) { a.w += b.w; a.x += b.x; a.y += b.y; a.z += b.z; }

DEVINL void quat_to_mat(const Q4& q, float R[9]) {
    const float ts = 2.f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    R[0] = 1.f - ts * (q.y * q.y + q.z * q.z);
    R[1] = ts * (q.x * q.y - q.z * q.w);
    R[2] = ts * (q.x * q.z + q.y * q.w);
    R[3] = ts * (q.x * q.y + q.z * q.w);
    R[4] = 1.f - ts * (q.x * q.x + q.z * q.z);
    R[5] = ts * (q.y * q.z - q.x * q.w);
    R[6] = ts * (q.x * q.z - q.y * q.w);
    R[7] = ts * (q.y * q.z + q.x * q.w);
    R[8] = 1.f - ts * (q.x * q.x + q.y * q.y);
}

// dL/dq of R = matrix(q) given g = dL/dR: every entry is const + ts * f(q) with ts = 2/|q|^2, d ts/dq = -ts^2 q
DEVINL Q4 quat_to_mat_bwd(const Q4& q, const float g[9]) {
    const float r = q.w, i = q.x, j = q.y, k = q.z;
    const float ts = 2.f / (r * r + i * i + j * j + k * k);
    const float G = -g[0] * (j * j + k * k) + g[1] * (i * j - k * r) + g[2] * (i * k + j * r) + g[3] * (i * j + k * r)
                    - g[4] * (i * i + k * k) + g[5] * (j * k - i * r) + g[6] * (i * k - j * r) + g[7] * (j * k + i * r)
                    - g[8] * (i * i + j * j);
    Q4 a;
    a.w = -k * g[1] + j * g[2] + k * g[3] - i * g[5] - j * g[6] + i * g[7];
    a.x = j * g[1] + k * g[2] + j * g[3] - 2.f * i * g[4] - r * g[5] + k * g[6] + r * g[7] - 2.f * i * g[8];
    a.y = -2.f * j * g[0] + i * g[1] + r * g[2] + i * g[3] + k * g[5] - r * g[6] + k * g[7] - 2.f * j * g[8];
    a.z = -2.f * k * g[0] - r * g[1] + i * g[2] + r * g[3] - 2.f * k * g[4] + j * g[5] + i * g[6] + j * g[7];
    const float c = ts * ts * G;
    return {ts * a.w - c * r, ts * a.x - c * i, ts * a.y - c * j, ts * a.z - c * k};
}

// bones (n,10) -> prep (n,16) = [c | matrix(q / |q|) | exp(log scale) | 0]; with g (n,16): d_bones (n,10) instead
__global__ void bone_prep_kernel(const float* __restrict__ bones, long long n, float* __restrict__ prep,
                                 const float* __restrict__ g, float* __restrict__ d_bones) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* bn = bones + i * 10;
    const Q4 q = {bn[3], bn[4], bn[5], bn[6]};
    const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);   // F.normalize (:196)
    const Q4 u = {q.w / nrm, q.x / nrm, q.y / nrm, q.z / nrm};
    if (g == nullptr) {
        float R[9];
        quat_to_mat(u, R);
        float* o = prep + i * 16;
        o[0] = bn[0]; o[1] = bn[1]; o[2] = bn[2];
#pragma unroll
        for (int k = 0; k < 9; ++k) o[3 + k] = R[k];
        o[12] = expf(bn[7]); o[13] = expf(bn[8]); o[14] = expf(bn[9]);                        // :198
        o[15] = 0.f;
        return;
    }
    const float* gi = g + i * 16;
    float* d = d_bones + i * 10;
    d[0] = gi[0]; d[1] = gi[1]; d[2] = gi[2];
    const Q4 du = quat_to_mat_bwd(u, gi + 3);
    const float dot = du.w * u.w + du.x * u.x + du.y * u.y + du.z * u.z;
    d[3] = (du.w - u.w * dot) / nrm; d[4] = (du.x - u.x * dot) / nrm;
    d[5] = (du.y - u.y * dot) / nrm; d[6] = (du.z - u.z * dot) / nrm;
    d[7] = gi[12] * expf(bn[7]); d[8] = gi[13] * expf(bn[8]); d[9] = gi[14] * expf(bn[9]);
}

// backward of bone_transform: g (N,B,10) -> d_rts (N,B,8), per-ray partial d_bones (N,B,10) (summed over rays by the caller)
__global__ void bone_transform_bwd_kernel(const float* __restrict__ bones, const float* __restrict__ rts, long long N, int B,
                                          const float* __restrict__ g, float* __restrict__ d_bones_ray,
                                          float* __restrict__ d_rts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * B) return;
    const int b = (int)(i % B);
    const float* bn = bones + b * 10;
    const float* dq = rts + i * 8;
    const float* gi = g + i * 10;
    const Q4 r = {dq[0], dq[1], dq[2], dq[3]};
    const Q4 d = {dq[4], dq[5], dq[6], dq[7]};
    float R[9];
    quat_to_mat(r, R);
    const float c[3] = {bn[0], bn[1], bn[2]};
    const float gc[3] = {gi[0], gi[1], gi[2]};
    float* db = d_bones_ray + i * 10;
    // centre' = R c + t
    db[0] = R[0] * gc[0] + R[3] * gc[1] + R[6] * gc[2];
    db[1] = R[1] * gc[0] + R[4] * gc[1] + R[7] * gc[2];
    db[2] = R[2] * gc[0] + R[5] * gc[1] + R[8] * gc[2];
    float dR[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) dR[3 * a + k] = gc[a] * c[k];
    Q4 dr = quat_to_mat_bwd(r, dR);
    // t = 2 (d (x) conj(r))[1:]:  p = a (x) b  =>  da = dp (x) conj(b), db = conj(a) (x) dp
    const Q4 du = {0.f, 2.f * gc[0], 2.f * gc[1], 2.f * gc[2]};
    const Q4 dd = qmul(du, r);
    const Q4 drc = qmul(qconj(d), du);
    qacc(dr, qconj(drc));
    // orient' = +-(r (x) q0), sign so that the real part is >= 0
    const Q4 q0 = {bn[3], bn[4], bn[5], bn[6]};
    const Q4 p = qmul(r, q0);
    const float sg = p.w < 0.f ? -1.f : 1.f;
    const Q4 gp = {sg * gi[3], sg * gi[4], sg * gi[5], sg * gi[6]};
    qacc(dr, qmul(gp, qconj(q0)));
    const Q4 dq0 = qmul(qconj(r), gp);
    db[3] = dq0.w; db[4] = dq0.x; db[5] = dq0.y; db[6] = dq0.z;
    db[7] = gi[7]; db[8] = gi[8]; db[9] = gi[9];
    float* o = d_rts + i * 8;
    o[0] = dr.w; o[1] = dr.x; o[2] = dr.y; o[3] = dr.z;
    o[4] = dd.w; o[5] = dd.x; o[6] = dd.y; o[7] = dd.z;
}

// out = conj(dq) / |r|^2 (dual_quat.py:87-94); backward with g
__global__ void dq_inverse_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ g, long long n,
                                      float* __restrict__ d_dq) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = dq + i * 8;
    const float* gi = g + i * 8;
    const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const float sg[8] = {1.f, -1.f, -1.f, -1.f, 1.f, -1.f, -1.f, -1.f};
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot += gi[k] * sg[k] * q[k];
    const float dn2 = -dot / (n2 * n2);
    float* o = d_dq + i * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = sg[k] * gi[k] / n2 + (k < 4 ? 2.f * q[k] * dn2 : 0.f);
}

inline unsigned nblocks(long long n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}   // namespace

extern "C" int moda_bone_prep(const float* bones, int64_t n, float* prep, const float* g_prep, float* d_bones, void* stream) {
    if (n <= 0) return 0;
    if (!bones || (!g_prep && !prep) || (g_prep && !d_bones)) return MODA_EINVAL;
    hipLaunchKernelGGL(bone_prep_kernel, dim3(nblocks(n)), dim3(kBlock), 0, (hipStream_t)stream, bones, (long long)n, prep, g_prep,
                       d_bones);
    return (int)hipGetLastError();
}

extern "C" int moda_bone_transform_bwd(const float* bones, const float* rts, int64_t N, int32_t B, const float* g_out,
                                       float* d_bones_ray, float* d_rts, void* stream) {
    if (N <= 0 || B <= 0) return 0;
    if (!bones || !rts || !g_out || !d_bones_ray || !d_rts) return MODA_EINVAL;
    hipLaunchKernelGGL(bone_transform_bwd_kernel, dim3(nblocks(N * B)), dim3(kBlock), 0, (hipStream_t)stream, bones, rts,
                       (long long)N, (int)B, g_out, d_bones_ray, d_rts);
    return (int)hipGetLastError();
}

extern "C" int moda_dq_inverse_bwd(const float* dq, const float* g_out, int64_t n, float* d_dq, void* stream) {
    if (n <= 0) return 0;
    if (!dq || !g_out || !d_dq) return MODA_EINVAL;
    hipLaunchKernelGGL(dq_inverse_bwd_kernel, dim3(nblocks(n)), dim3(kBlock), 0, (hipStream_t)stream, dq, g_out, (long long)n,
                       d_dq);
    return (int)hipGetLastError();
}

// ================================================================================================
// Per-frame feeders of the path: ray construction and the body-pose head's tail
// ================================================================================================
namespace {

DEVINL float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// raycast (geom_utils.py:746-794): per frame b, pixel s:  v = Kinv_b [x, y, 1],  rays_d = R_b^T v,  rays_o = -R_b^T T_b.
// One wavefront per frame.  With g_d / g_o given it runs the backward: dR += v g_d^T - T g_o^T, dT = -R sum g_o,
// dKinv += (R g_d) [x, y, 1]^T, summed over the frame's pixels by wavefront shuffles (no atomics).
__global__ __launch_bounds__(256) void raycast_kernel(const float* __restrict__ xys, const float* __restrict__ Rm,
                                                      const float* __restrict__ Tm, const float* __restrict__ Kinv, int bs,
                                                      int ns, float* __restrict__ rays_d, float* __restrict__ rays_o,
                                                      const float* __restrict__ g_d, const float* __restrict__ g_o,
                                                      float* __restrict__ dR, float* __restrict__ dT,
                                                      float* __restrict__ dK) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= bs) return;
    float R[9], K[9], T[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) { R[k] = Rm[b * 9 + k]; K[k] = Kinv[b * 9 + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) T[k] = Tm[b * 3 + k];
    const bool bwd = g_d != nullptr;
    float aR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, aK[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, aT[3] = {0, 0, 0};
    for (int s = lane; s < ns; s += 64) {
        const long long i = (long long)b * ns + s;
        const float x = xys[i * 2 + 0], y = xys[i * 2 + 1];
        const float v[3] = {K[0] * x + K[1] * y + K[2], K[3] * x + K[4] * y + K[5], K[6] * x + K[7] * y + K[8]};
        if (!bwd) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rays_d[i * 3 + c] = v[0] * R[c] + v[1] * R[3 + c] + v[2] * R[6 + c];          // (v^T R)_c
                rays_o[i * 3 + c] = -(T[0] * R[c] + T[1] * R[3 + c] + T[2] * R[6 + c]);
            }
        } else {
            const float gd[3] = {g_d[i * 3], g_d[i * 3 + 1], g_d[i * 3 + 2]};
            const float go[3] = {g_o ? g_o[i * 3] : 0.f, g_o ? g_o[i * 3 + 1] : 0.f, g_o ? g_o[i * 3 + 2] : 0.f};
            const float xy1[3] = {x, y, 1.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dv = R[3 * j] * gd[0] + R[3 * j + 1] * gd[1] + R[3 * j + 2] * gd[2];   // (R g_d)_j
                aT[j] -= R[3 * j] * go[0] + R[3 * j + 1] * go[1] + R[3 * j + 2] * go[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    aR[3 * j + c] += v[j] * gd[c] - T[j] * go[c];
                    aK[3 * j + c] += dv * xy1[c];
                }
            }
        }
    }
    if (bwd) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { aR[k] = wave_sum64(aR[k]); aK[k] = wave_sum64(aK[k]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) aT[k] = wave_sum64(aT[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) { dR[b * 9 + k] = aR[k]; dK[b * 9 + k] = aK[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) dT[b * 3 + k] = aT[k];
        }
    }
}

// DQ_RTHead tail (nerf.py:260-279): rts (n,7) = [t(3) | q(4)] -> dq (n,8) = [q/|q|, 1/2 (0, 0.1 t) (x) q/|q|]
__global__ void rt_to_dq_kernel(const float* __restrict__ rts, long long n, float* __restrict__ dq,
                                const float* __restrict__ g, float* __restrict__ d_rts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rts + i * 7;
    const Q4 tq = {0.f, 0.1f * r[0], 0.1f * r[1], 0.1f * r[2]};
    const Q4 q = {r[3], r[4], r[5], r[6]};
    const float nrm = fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);
    const Q4 u = {q.w / nrm, q.x / nrm, q.y / nrm, q.z / nrm};
    if (g == nullptr) {
        const Q4 d = qmul(tq, u);
        float* o = dq + i * 8;
        o[0] = u.w; o[1] = u.x; o[2] = u.y; o[3] = u.z;
        o[4] = 0.5f * d.w; o[5] = 0.5f * d.x; o[6] = 0.5f * d.y; o[7] = 0.5f * d.z;
        return;
    }
    const float* gi = g + i * 8;
    const Q4 gd = {0.5f * gi[4], 0.5f * gi[5], 0.5f * gi[6], 0.5f * gi[7]};
    Q4 du = {gi[0], gi[1], gi[2], gi[3]};
    qacc(du, qmul(qconj(tq), gd));                 // d = tq (x) u  =>  du += conj(tq) (x) gd
    const Q4 dtq = qmul(gd, qconj(u));             //               dtq = gd (x) conj(u)
    const float dot = du.w * u.w + du.x * u.x + du.y * u.y + du.z * u.z;
    float* o = d_rts + i * 7;
    o[0] = 0.1f * dtq.x; o[1] = 0.1f * dtq.y; o[2] = 0.1f * dtq.z;
    o[3] = (du.w - u.w * dot) / nrm; o[4] = (du.x - u.x * dot) / nrm;
    o[5] = (du.y - u.y * dot) / nrm; o[6] = (du.z - u.z * dot) / nrm;
}

}   // namespace

extern "C" int moda_raycast(const float* xys, const float* Rmat, const float* Tmat, const float* Kinv, int64_t bs, int64_t ns,
                            float* rays_d, float* rays_o, const float* g_rays_d, const float* g_rays_o, float* d_Rmat,
                            float* d_Tmat, float* d_Kinv, void* stream) {
    if (bs <= 0 || ns <= 0) return 0;
    if (!xys || !Rmat || !Tmat || !Kinv || bs > 0x7fffffff || ns > 0x7fffffff) return MODA_EINVAL;
    if (g_rays_d ? (!d_Rmat || !d_Tmat || !d_Kinv) : (!rays_d || !rays_o)) return MODA_EINVAL;
    hipLaunchKernelGGL(raycast_kernel, dim3((unsigned)((bs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xys, Rmat, Tmat, Kinv,
                       (int)bs, (int)ns, rays_d, rays_o, g_rays_d, g_rays_o, d_Rmat, d_Tmat, d_Kinv);
    return (int)hipGetLastError();
}

extern "C" int moda_rt_to_dq(const float* rts, int64_t n, float* dq, const float* g_dq, float* d_rts, void* stream) {
    if (n <= 0) return 0;
    if (!rts || (!g_dq && !dq) || (g_dq && !d_rts)) return MODA_EINVAL;
    hipLaunchKernelGGL(rt_to_dq_kernel, dim3(nblocks(n)), dim3(kBlock), 0, (hipStream_t)stream, rts, (long long)n, dq, g_dq, d_rts);
    return (int)hipGetLastError();
}
