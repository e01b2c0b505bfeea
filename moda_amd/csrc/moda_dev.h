// Device helpers shared by the kernels of libmoda_hip.so (render_kernels.hip, mlp_fused.hip): quaternion algebra, the
// DQS point transform, and the layout of the per-set MFMA tables of the fused skin-MLP + warp kernel.
#pragma once
#include <hip/hip_runtime.h>

#define DEVINL __device__ __forceinline__

// gemm_bf16.hip: the bf16-native forms of moda_gemm_f32_ex; true when the call was taken (launch status in *rc)
struct moda_gemm_desc;
bool moda_g3_try(const moda_gemm_desc* d, void* stream, int* rc);

namespace {

// ------------------------------------------------------------------------------------------------
// quaternion helpers (real first)
// ------------------------------------------------------------------------------------------------
struct Quat { float w, x, y, z; };

DEVINL Quat qmul(const Quat& a, const Quat& b) {   // Hamilton product a (x) b
    Quat o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return o;
}

// rotation matrix of a (not necessarily unit) quaternion, scaled by 2/|q|^2 (pytorch3d quaternion_to_matrix)
DEVINL void quat_to_mat(const Quat& q, float R[9]) {
    const float two_s = 2.f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    R[0] = 1.f - two_s * (q.y * q.y + q.z * q.z);
    R[1] = two_s * (q.x * q.y - q.z * q.w);
    R[2] = two_s * (q.x * q.z + q.y * q.w);
    R[3] = two_s * (q.x * q.y + q.z * q.w);
    R[4] = 1.f - two_s * (q.x * q.x + q.z * q.z);
    R[5] = two_s * (q.y * q.z - q.x * q.w);
    R[6] = two_s * (q.x * q.z - q.y * q.w);
    R[7] = two_s * (q.y * q.z + q.x * q.w);
    R[8] = 1.f - two_s * (q.x * q.x + q.y * q.y);
}

// out = v + 2 d0 x (d0 x v + a0 v) + 2 (a0 de - ae d0 + d0 x de)   (:489-491), c = blend / |blend_r|
DEVINL void dqs_apply(const float bl[8], float px, float py, float pz, float* ox, float* oy, float* oz) {
    const float nrm = sqrtf(bl[0] * bl[0] + bl[1] * bl[1] + bl[2] * bl[2] + bl[3] * bl[3]);   // dq_normalize (:471)
    const float a0 = bl[0] / nrm, d0x = bl[1] / nrm, d0y = bl[2] / nrm, d0z = bl[3] / nrm;
    const float ae = bl[4] / nrm, dex = bl[5] / nrm, dey = bl[6] / nrm, dez = bl[7] / nrm;
    // inner = d0 x v + a0 v
    const float ix = d0y * pz - d0z * py + a0 * px;
    const float iy = d0z * px - d0x * pz + a0 * py;
    const float iz = d0x * py - d0y * px + a0 * pz;
    // rotated = v + 2 d0 x inner
    const float rx = px + 2.f * (d0y * iz - d0z * iy);
    const float ry = py + 2.f * (d0z * ix - d0x * iz);
    const float rz = pz + 2.f * (d0x * iy - d0y * ix);
    // trans = 2 (a0 de - ae d0 + d0 x de)
    const float tx = 2.f * (a0 * dex - ae * d0x + (d0y * dez - d0z * dey));
    const float ty = 2.f * (a0 * dey - ae * d0y + (d0z * dex - d0x * dez));
    const float tz = 2.f * (a0 * dez - ae * d0z + (d0x * dey - d0y * dex));
    *ox = rx + tx;
    *oy = ry + ty;
    *oz = rz + tz;
}

// The same transform for the throughput-mode kernel: the eight components are scaled by ONE hardware reciprocal square root
// of the real part's squared norm (v_rsq_f32, ~1 ulp) instead of eight IEEE divisions by its square root (~90 instructions);
// the exact-fp32 kernels keep dqs_apply's division, which is what the reference computes (geom_utils.py:471).
DEVINL void dqs_apply_fast(const float bl[8], float px, float py, float pz, float* ox, float* oy, float* oz) {
    const float inv = __builtin_amdgcn_rsqf(bl[0] * bl[0] + bl[1] * bl[1] + bl[2] * bl[2] + bl[3] * bl[3]);
    const float a0 = bl[0] * inv, d0x = bl[1] * inv, d0y = bl[2] * inv, d0z = bl[3] * inv;
    const float ae = bl[4] * inv, dex = bl[5] * inv, dey = bl[6] * inv, dez = bl[7] * inv;
    const float ix = d0y * pz - d0z * py + a0 * px;
    const float iy = d0z * px - d0x * pz + a0 * py;
    const float iz = d0x * py - d0y * px + a0 * pz;
    *ox = px + 2.f * (d0y * iz - d0z * iy) + 2.f * (a0 * dex - ae * d0x + (d0y * dez - d0z * dey));
    *oy = py + 2.f * (d0z * ix - d0x * iz) + 2.f * (a0 * dey - ae * d0y + (d0z * dex - d0x * dez));
    *oz = pz + 2.f * (d0x * iy - d0y * ix) + 2.f * (a0 * dez - ae * d0z + (d0x * dey - d0y * dex));
}

// ---- per-set MFMA tables of the fused skin-MLP + warp kernel (moda_warp_tables_fwd -> moda_mlp_warp_fwd) -----------------
// A "set" is the bone data of one ray (or of one frame of rays).  Bones are tiled by 32 (the MFMA M dimension).
//   qtab : per set, per bone tile, 5 fragments of 64 floats -- the A operand of v_mfma_f32_32x32x2_f32 (lane l: row l & 31 =
//          bone, k = l >> 5) holding the Gaussian skinning logit (geom_utils.py:251-266) as a quadratic form in the sample
//          position:  logit_b(p) = - (p - c)^T A (p - c),  A = 1000 e^{aux} R diag(s) R^T;  the matching B operand rows are
//          the monomials (x^2, y^2 | z^2, xy | xz, yz | x, y | z, 1).  Padding bones have constant -1e30 (weight 0).
//   dqtab: per set, per bone tile, 2 fragments of 64 x 16 B -- the A operand of v_mfma_f32_32x32x16_bf16 for the blend
//          sum_b w_b dq_b (geom_utils.py:470): row r of the 32 carries component (r & 3) of the real (r & 4 == 0) or dual
//          part of dq_b, as bf16 "hi" (r & 8 == 0) or "lo" = bf16(v - hi) (the two add up to 16 mantissa bits); rows 16-31
//          repeat rows 0-15 with real and dual swapped so that BOTH lane halves of the 32x32 result hold all 8 sums.
//          k element j of lane half h of fragment u is bone 32 tile + 16 u + 8 (j >> 2) + 4 h + (j & 3): the order in which
//          a 32x32 accumulator tile, packed pairwise to bf16, is a B operand.
constexpr int kWarpQFrags = 5;                 // fp32 fragments (64 floats) per bone tile
constexpr int kWarpDqFrags = 2;                // bf16 fragments (64 x 16 B) per bone tile
constexpr int kWarpQFloats = kWarpQFrags * 64;          // per (set, tile)
constexpr int kWarpDqBytes = kWarpDqFrags * 64 * 16;    // per (set, tile)

}   // namespace
