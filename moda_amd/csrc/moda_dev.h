// Device helpers shared by the kernels of libmoda_hip.so (render_kernels.hip, mlp_fused.hip): quaternion algebra, the
// DQS point transform, and the layout of the per-set MFMA tables of the fused skin-MLP + warp kernel.
#pragma once
#include <hip/hip_runtime.h>

#define DEVINL __device__ __forceinline__

// gemm_bf16.hip: the bf16-native forms of moda_gemm_f32_ex; true when the call was taken (launch status in *rc)
struct moda_gemm_desc;
bool moda_g3_try(const moda_gemm_desc* d, void* stream, int* rc);
// bwd64_chain.hip: the hidden-layer chain of a 64-wide network's bf16-storage backward (dW, db, masked dX) as one launch
long long moda_chain64_part_floats(long long M);
int moda_chain64_bwd(const void* dh_in, long long ld_in, const void* const* h, long long ld_h, const void* const* wb, void* dh_out,
                     long long ld_out, float* const* gW, const long long* ldw, float* const* gb, int n, long long M, float* part,
                     void* stream);
int moda_heads64_bwd(const void* dzb, long long ld_dzb, const void* dd, long long ld_dd, const void* hD, long long ld_hD, const void* wrgb,
                     const void* wext, void* dh, long long ld_dh, float* g_rgb, long long ld_grgb, float* g_brgb, int n_out, float* Tm,
                     float* svec, long long M, float* part, void* stream);
int moda_pe_ends64_bwd(const void* dha, const void* dhb, long long ld_dh, const void* pe, long long ld_pe, const void* wa, const void* wb,
                       const float* xyz, int n_freq, const float* window, float* gWa, long long lda, float* gWb, long long ldb,
                       float* gb_b, float* d_xyz, long long M, float* part, void* stream);
// bwd256_fused.hip: one W -> W hidden layer (W = 256 or 128) of the bf16-storage backward (dW, db, masked dX) as one launch
int moda_bwd256_layer(int W, const void* dz, long long ldz, const void* x, long long ldx, const void* wb, long long ldw, void* dx,
                      long long ldo, float* gW, long long ldg, float* gb, long long M, void* stream);
bool moda_x3_try(const moda_gemm_desc* d, int ns, void* stream, int* rc);   // gemm_x3.hip: the MODA_GEMM_BF16X3 / X6 forms (ns = 2 / 3)

namespace {

// sin and cos of an fp32 argument of any size the positional encodings meet (|t| up to a few thousand), absolute error <= 1.3e-7
// (2 ulp of 1.0: the accuracy class of sincosf, i.e. of torch.sin / cos(freq * x) in the reference): t / 2pi as an exact product
// in two floats, the whole revolutions and the nearest quarter taken off exactly, the remainder (|.| <= 1/8 revolution) through
// degree-7 / degree-8 polynomials.  ~35 VALU operations where sincosf's general argument reduction needs well over a hundred.
// (mlp_fused.hip's split-bf16 encoding carries the same arithmetic as sin_quarter_shifted.)
DEVINL void sincos_rr(float t, float& sn_out, float& cs_out) {
    const float INV_HI = 0.15915494f, INV_LO = 6.4206382e-09f, TP_HI = 6.2831855f, TP_LO = -1.7484555e-07f;
    const float ph = t * INV_HI;
    float pl = __builtin_fmaf(t, INV_HI, -ph);
    pl = __builtin_fmaf(t, INV_LO, pl);
    const float fh = ph - __builtin_rintf(ph);
    const float q = __builtin_rintf(fh * 4.f);
    const float g = __builtin_fmaf(q, -0.25f, fh) + pl;
    const float a = __builtin_fmaf(g, TP_LO, g * TP_HI);
    const float zz = a * a;
    float sn = __builtin_fmaf(zz, -1.9515295891e-4f, 8.3321608736e-3f);
    sn = __builtin_fmaf(sn, zz, -1.6666654611e-1f);
    sn = __builtin_fmaf(sn * zz, a, a);
    float cs = __builtin_fmaf(zz, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cs = __builtin_fmaf(cs, zz, 4.166664568298827e-2f);
    cs = __builtin_fmaf(cs * zz, zz, __builtin_fmaf(zz, -0.5f, 1.f));
    const int qi = (int)q & 3;                    // angle = a + qi * pi/2
    const float s1 = (qi & 1) ? cs : sn, c1 = (qi & 1) ? sn : cs;
    sn_out = (qi & 2) ? -s1 : s1;
    cs_out = ((qi + 1) & 2) ? -c1 : c1;
}

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

// ------------------------------------------------------------------------------------------------
// Compositing core (rendering.py:183-237): ONE 64-lane wavefront walks a ray in blocks of 64 samples -- exclusive transmittance
// product by a wavefront shuffle scan with a carried prefix, per-lane partial sums reduced by a butterfly at the end.  Shared by
// composite_kernel (samples read from HBM) and the epilogue of the fused 8 x 256 kernel (samples read from LDS, where the
// workgroup's waves have just put them): the two routes run the same instruction sequence on the same values, floating-point
// contraction off, so their results are bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxFeat = 16;

struct CompOut {
    float *rgb, *feat_out, *depth, *sil, *weights, *visibility, *vis_out, *cyc_out;
    int* n_used;
};

DEVINL float comp_dnorm(const float* __restrict__ rd, long long n) {            // |rays_d| of ray n (:186)
#pragma clang fp contract(off)
    return sqrtf(rd[n * 3] * rd[n * 3] + rd[n * 3 + 1] * rd[n * 3 + 1] + rd[n * 3 + 2] * rd[n * 3 + 2]);
}

DEVINL float comp_ibeta(const float* __restrict__ beta) { return 1.f / (fabsf(beta[0]) + 1e-9f); }   // :199

// SDF-to-density (VolSDF Laplace CDF with the learnt beta, :199-205) and alpha of one sample (:207); delta already times |d|
DEVINL float comp_alpha(float sigma_raw, bool has_noise, float noise, float delta, float ibeta) {
#pragma clang fp contract(off)
    float sg = sigma_raw;
    if (has_noise) sg += noise;                                                 // :196
    const float sdf = -sg;                                                      // :201
    const float sgn = sdf > 0.f ? 1.f : (sdf < 0.f ? -1.f : 0.f);
    const float dens = (0.5f + 0.5f * sgn * expm1f(-fabsf(sdf) * ibeta)) * ibeta;   // :202-205
    return 1.f - expf(-delta * dens);                                           // :207
}

// The association every route uses (so that they agree bit for bit).  Samples are taken in GROUPS of 32:
//   transmittance  T_i = carry_g * excl_i, excl_i = the exclusive product of t = 1 - alpha + 1e-10 inside group g by a 32-lane
//                  Hillis-Steele scan, carry_g = ((1 * tot_0) * tot_1) ... * tot_{g-1} over the earlier groups' totals in order;
//   sums           a 32-lane butterfly per group, the groups' sums added in order: acc = ((0 + s_0) + s_1) + ...
// composite_ray walks a ray with one 64-lane wave, two groups per step; the fused 8 x 256 kernel gives every wave one group.
struct CompTerms { float r, g, b, d, s, v, c; };       // per-sample products summed over a ray (rgb, depth, sil, vis, cyc)

// Cross-lane steps as DPP operands / lane swaps: one VALU instruction each.  (The `__shfl*` forms of rounds 1-3 compile to
// ds_bpermute_b32 -- an LDS-pipe round trip and an s_waitcnt per step; composite_kernel had 170 of them and spent its time
// waiting on their dependent chains, not on memory.)  The forms below reproduce the shuffle forms' association STEP FOR STEP, so
// every result is bitwise what rounds 1-3 computed; probed on the hardware against the shuffle forms, 200 random waves,
// bitwise equal: tools/probes/dpp_reduce_probe.hip.  Callers must have all 64 lanes ACTIVE (every call site is in wave-uniform
// control flow): DPP and lane swaps read an inactive lane's stale register, where ds_bpermute returned 0 for it.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
DEVINL float comp_dpp(float old, float src) {      // lanes that are masked off or have no source lane keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                   ROW_MASK, BANK_MASK, false));
}
DEVINL float comp_lane(float v, int l) {           // lane l's value in every lane (l wave-uniform)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// v_permlane16_swap exchanges the odd rows (of 16 lanes) of its first register with the even rows of its second: with one value
// in both, every lane of rows 0 / 1 ends up with (row 0's value, row 1's value) of its column, rows 2 / 3 likewise.  Two pitfalls
// of hipcc 7.2, both probed: given ONE value twice the operands may share a register and the swap degenerates (hence the opaque
// copy), and bit-casting element 1 of the returned pair directly reads element 0 (hence the two unsigned temporaries).
DEVINL void comp_rows_pair(float v, float& even_row, float& odd_row) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    even_row = __uint_as_float(r0);
    odd_row = __uint_as_float(r1);
}

// butterfly over the 32 lanes of a half wave: v[l] += v[l ^ o] for o = 16, 8, 4, 2, 1; the sum ends up in every lane of the half
DEVINL float comp_group_sum(float v) {
    float e, o;
    comp_rows_pair(v, e, o);
    v = e + o;                                          // xor 16
    v += comp_dpp<0x128>(v, v);                         // xor 8 = row_ror:8
    float q = comp_dpp<0x124, 0xf, 0xa>(v, v);          // xor 4: banks 1, 3 take lane - 4 (row_ror:4) ...
    q = comp_dpp<0x12C, 0xf, 0x5>(q, v);                //        banks 0, 2 take lane + 4 (row_ror:12)
    v += q;
    v += comp_dpp<0x4E>(v, v);                          // xor 2 = quad_perm [2,3,0,1]
    v += comp_dpp<0xB1>(v, v);                          // xor 1 = quad_perm [1,0,3,2]
    return v;
}

// butterfly over all 64 lanes: v[l] += v[l ^ o] for o = 32, 16, ..., 1 (xor 32 through v_permlane32_swap: the upper half of its
// first register <-> the lower half of its second; same pitfalls as comp_rows_pair); bitwise the `__shfl_xor` loop it replaces
DEVINL float comp_wave_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return comp_group_sum(__uint_as_float(r0) + __uint_as_float(r1));
}

// v of lane (l ^ J), J = 1, 2, 4, 8, 16, 32, in every lane (all 64 active) -- `__shfl_xor` without the LDS pipe
template <int J>
DEVINL float comp_xor_lane(float v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "one address bit");
    if constexpr (J == 1) return comp_dpp<0xB1>(v, v);
    else if constexpr (J == 2) return comp_dpp<0x4E>(v, v);
    else if constexpr (J == 4) {
        const float q = comp_dpp<0x124, 0xf, 0xa>(v, v);
        return comp_dpp<0x12C, 0xf, 0x5>(q, v);
    } else if constexpr (J == 8) return comp_dpp<0x128>(v, v);
    else if constexpr (J == 16) {
        float e, o;
        comp_rows_pair(v, e, o);
        return (__lane_id() & 16) ? e : o;
    } else {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        asm volatile("" : "+v"(b));
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);     // r0 = lower halves, r1 = upper halves
        const unsigned r0 = r[0], r1 = r[1];
        return (__lane_id() & 32) ? __uint_as_float(r0) : __uint_as_float(r1);
    }
}

// inclusive product scan of t inside each 32-lane half (Hillis-Steele: p[l] *= p[l - of] for of = 1, 2, 4, 8, 16); returns the
// exclusive value, *total = the half's product.  Lanes whose source lies in the row before take it from the swapped copy
// (comp_rows_pair), rotated into place.
DEVINL float comp_group_scan(float t, int lane32, float* total) {
#pragma clang fp contract(off)
    float p = t, q, e, o;
    q = comp_dpp<0x138>(1.f, p);                        // of = 1: wave_shr:1
    p *= (lane32 >= 1) ? q : 1.f;
    q = comp_dpp<0x138>(1.f, p);                        // of = 2: two single shifts
    q = comp_dpp<0x138>(1.f, q);
    p *= (lane32 >= 2) ? q : 1.f;
    comp_rows_pair(p, e, o);                            // of = 4: row_shr:4 inside a row, the first four lanes of rows 1 / 3 from the row before
    q = comp_dpp<0x114>(1.f, p);
    q = comp_dpp<0x124, 0xa, 0x1>(q, e);                //   row_ror:4 of the even row's values: rows 1 / 3, bank 0
    p *= q;
    comp_rows_pair(p, e, o);                            // of = 8
    q = comp_dpp<0x118>(1.f, p);
    q = comp_dpp<0x128, 0xa, 0x3>(q, e);                //   row_ror:8: rows 1 / 3, banks 0 - 1
    p *= q;
    comp_rows_pair(p, e, o);                            // of = 16: rows 1 / 3 times the row before, same column
    p *= (lane32 >= 16) ? e : 1.f;
    float excl = comp_dpp<0x138>(1.f, p);
    if (lane32 == 0) excl = 1.f;
    const float t0 = comp_lane(p, 31), t1 = comp_lane(p, 63);
    *total = (__lane_id() & 32) ? t1 : t0;
    return excl;
}

// weight and per-sample terms of one sample (rendering.py:220-235, 408, 473); `last` = the ray's last sample
DEVINL float comp_sample_terms(float alpha, float T, bool live, bool last, float cr, float cg, float cb, float z, float sraw,
                               float rgb_filter_scale, float visp, float cycv, CompTerms* o) {
#pragma clang fp contract(off)
    const float w = live ? alpha * T : 0.f;                                     // :220
    // rgb_filter (:171, 225, 229-230): colour weighted by w * scale_rgb * sigmoid(-10 sigma_raw), last sample excluded
    const float wr = rgb_filter_scale > 0.f ? (!last ? w * rgb_filter_scale * (1.f / (1.f + expf(10.f * sraw))) : 0.f) : w;
    o->r = live ? wr * cr : 0.f; o->g = live ? wr * cg : 0.f; o->b = live ? wr * cb : 0.f;   // :232
    o->d = live ? w * z : 0.f;                                                  // :234
    o->s = (live && !last) ? w : 0.f;                                           // :235
    o->v = live ? w * visp : 0.f;                                               // :408
    o->c = live ? w * cycv : 0.f;                                               // :473
    return w;
}

// Loader: void load(long long s, float& r, float& g, float& b, float& sigma_raw, float& z, float& alpha) for an existing sample s
// FEAT = false: the instantiation without composited features (F = 0) -- no 16 feature accumulators: 75 -> under 64 registers,
// 6 -> 8 waves per SIMD for a walk that lives on occupancy
template <bool FEAT = true, class Loader>
DEVINL void composite_ray(const Loader& ld, int lane, long long n, long long S, long long s_end, float term_tau,
                          float rgb_filter_scale, const float* __restrict__ feat_in, int F, const float* __restrict__ vis_pred,
                          const float* __restrict__ cyc, const CompOut& o) {
#pragma clang fp contract(off)
    const float* __restrict__ feat = FEAT ? feat_in : nullptr;
    const int lane32 = lane & 31, half = lane >> 5;
    long long used = s_end;
    float carry = 1.f;   // product of (1 - alpha + 1e-10) over all earlier groups
    float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f, a_s = 0.f, a_v = 0.f, a_c = 0.f;
    float a_f[FEAT ? kMaxFeat : 1];
#pragma unroll
    for (int f = 0; f < (FEAT ? kMaxFeat : 1); ++f) a_f[f] = 0.f;
    long long s0 = 0;
    for (; s0 < s_end; s0 += 64) {
        const long long s = s0 + lane;
        const bool in = s < S;
        const bool comp = s < s_end;            // this sample's inputs exist
        const long long i = n * S + (in ? s : S - 1);
        float t = 1.f, alpha = 0.f, z = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, sraw = 0.f;
        if (comp) {
            ld.load(s, cr, cg, cb, sraw, z, alpha);
            t = 1.f - alpha + 1e-10f;                                           // :218
        }
        float tot;
        const float excl = comp_group_scan(t, lane32, &tot);
        const float tot0 = comp_lane(tot, 0), tot1 = comp_lane(tot, 32);
        const float c1 = carry * tot0;
        const float T = (half ? c1 : carry) * excl;                             // :219
        carry = c1 * tot1;
        const bool dead = term_tau > 0.f && T < term_tau;                       // T is non-increasing: dead lanes form a suffix
        const unsigned long long dmask = __ballot(dead && comp);
        const bool live = comp && !dead;                                        // (terminated samples' inputs are never read)
        CompTerms q;
        const float w = comp_sample_terms(alpha, T, live, s + 1 >= S, cr, cg, cb, z, sraw, rgb_filter_scale,
                                          (live && vis_pred) ? vis_pred[i] : 0.f, (live && cyc) ? cyc[i] : 0.f, &q);
        if (in) {
            if (o.weights) o.weights[i] = w;
            if (o.visibility) o.visibility[i] = comp ? T : 0.f;                 // :224
        }
        // the two groups' sums, added in order
        const float gr = comp_group_sum(q.r), gg = comp_group_sum(q.g), gb = comp_group_sum(q.b), gd = comp_group_sum(q.d),
                    gs = comp_group_sum(q.s);
        a_r = (a_r + comp_lane(gr, 0)) + comp_lane(gr, 32);
        a_g = (a_g + comp_lane(gg, 0)) + comp_lane(gg, 32);
        a_b = (a_b + comp_lane(gb, 0)) + comp_lane(gb, 32);
        a_d = (a_d + comp_lane(gd, 0)) + comp_lane(gd, 32);
        a_s = (a_s + comp_lane(gs, 0)) + comp_lane(gs, 32);
        if (vis_pred) { const float gv = comp_group_sum(q.v); a_v = (a_v + comp_lane(gv, 0)) + comp_lane(gv, 32); }
        if (cyc) { const float gc = comp_group_sum(q.c); a_c = (a_c + comp_lane(gc, 0)) + comp_lane(gc, 32); }
        if constexpr (FEAT) {
            if (feat) {
                // the sample's F features first (16-byte loads when they can be, all in flight together), then the sums
                const float* fp = feat + i * F;
                float fv[kMaxFeat];
                if ((F & 3) == 0 && ((((uintptr_t)feat) & 15) == 0)) {
#pragma unroll
                    for (int f4 = 0; f4 < kMaxFeat / 4; ++f4)
                        if (4 * f4 < F) {
                            const float4 q = live ? ((const float4*)fp)[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
                            fv[4 * f4] = q.x; fv[4 * f4 + 1] = q.y; fv[4 * f4 + 2] = q.z; fv[4 * f4 + 3] = q.w;
                        }
                } else {
#pragma unroll
                    for (int f = 0; f < kMaxFeat; ++f)
                        if (f < F) fv[f] = live ? fp[f] : 0.f;
                }
#pragma unroll
                for (int f = 0; f < kMaxFeat; ++f)
                    if (f < F) {
                        const float gf = comp_group_sum(live ? w * fv[f] : 0.f);    // :233
                        a_f[f] = (a_f[f] + comp_lane(gf, 0)) + comp_lane(gf, 32);
                    }
            }
        }
        if (dmask != 0ull) {                                                    // the ray ends in this block
            used = s0 + __builtin_ctzll(dmask);
            s0 += 64;
            break;
        }
    }
    for (; s0 < S; s0 += 64) {                                                  // terminated tail: weights 0, nothing read
        const long long s = s0 + lane;
        if (s < S) {
            if (o.weights) o.weights[n * S + s] = 0.f;
            if (o.visibility) o.visibility[n * S + s] = 0.f;
        }
    }
    if (lane == 0) {
        if (o.n_used) o.n_used[n] = (int)used;
        o.rgb[n * 3 + 0] = a_r; o.rgb[n * 3 + 1] = a_g; o.rgb[n * 3 + 2] = a_b;
        o.depth[n] = a_d;
        o.sil[n] = a_s;
        if (o.vis_out && vis_pred) o.vis_out[n] = a_v;
        if (o.cyc_out && cyc) o.cyc_out[n] = a_c;
        if constexpr (FEAT) {
            if (feat && o.feat_out) {
#pragma unroll
                for (int f = 0; f < kMaxFeat; ++f)
                    if (f < F) o.feat_out[n * F + f] = a_f[f];
            }
        }
    }
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
