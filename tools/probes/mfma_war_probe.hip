// Does gfx950 need protection between an MFMA and a VALU write to one of its A / B source registers (write-after-read)?
// hipcc's hazard recogniser inserts nothing there.  Every wave runs a loop of
//     MFMA acc += A * B ;  [D wait states] ;  VALU: overwrite one A / B register with garbage ;  restore it ;  (safe gap)
// in hand-placed physical registers (one asm block: nothing is rescheduled), 2-4 waves per SIMD so that an MFMA regularly
// queues behind another wave's; the accumulators are compared with a run whose "garbage" equals the original value.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_war_probe tools/probes/mfma_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// KIND 0: v_mfma_f32_32x32x16_bf16 (A, B: 4 VGPRs each)  1: ..._f16  2: v_mfma_f32_32x32x2_f32 (A, B: 1 VGPR each)
// WHICH: register overwritten: 40..43 = A[0..3], 44..47 = B[0..3]
#define PROBE(NAME, MFMA, WHICH, GAP) PROBE2(NAME, MFMA, WHICH, GAP, "v_mov_b32 v" STR(WHICH) ", %[poison]\n", "")
#define PROBE2(NAME, MFMA, WHICH, GAP, OVW, PRE)                                                                                     \
    __global__ __launch_bounds__(512) void NAME(const unsigned* ab, float* out, int iters, unsigned poison_in, int clean) {              \
        const int lane = threadIdx.x & 63;                                                                                 \
        unsigned a0 = ab[lane * 8 + 0], a1 = ab[lane * 8 + 1], a2 = ab[lane * 8 + 2], a3 = ab[lane * 8 + 3];               \
        unsigned b0 = ab[lane * 8 + 4], b1 = ab[lane * 8 + 5], b2 = ab[lane * 8 + 6], b3 = ab[lane * 8 + 7];               \
        unsigned orig = ab[lane * 8 + (WHICH - 40)];                                                                       \
        const unsigned poison = clean ? orig : poison_in;   /* clean run: the overwrite rewrites the same value */          \
        float o0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10, o11, o12, o13, o14, o15;                                        \
        asm volatile(                                                                                                      \
            "v_mov_b32 v40, %[a0]\n v_mov_b32 v41, %[a1]\n v_mov_b32 v42, %[a2]\n v_mov_b32 v43, %[a3]\n"                  \
            "v_mov_b32 v44, %[b0]\n v_mov_b32 v45, %[b1]\n v_mov_b32 v46, %[b2]\n v_mov_b32 v47, %[b3]\n"                  \
            "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n" \
            "v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n" \
            "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"                                   \
            "s_mov_b32 s20, %[iters]\n s_nop 15\n"                                                                         \
            "1:\n" PRE MFMA "\n" GAP OVW                                                                                   \
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                                  \
            "v_mov_b32 v" STR(WHICH) ", %[orig]\n"                                                                         \
            "s_nop 15\n"                                                                                                   \
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                            \
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                       \
            "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v49\n v_mov_b32 %[o2], v50\n v_mov_b32 %[o3], v51\n"                  \
            "v_mov_b32 %[o4], v52\n v_mov_b32 %[o5], v53\n v_mov_b32 %[o6], v54\n v_mov_b32 %[o7], v55\n"                  \
            "v_mov_b32 %[o8], v56\n v_mov_b32 %[o9], v57\n v_mov_b32 %[o10], v58\n v_mov_b32 %[o11], v59\n"                \
            "v_mov_b32 %[o12], v60\n v_mov_b32 %[o13], v61\n v_mov_b32 %[o14], v62\n v_mov_b32 %[o15], v63\n"              \
            : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3), [o4] "=v"(o4), [o5] "=v"(o5), [o6] "=v"(o6),      \
              [o7] "=v"(o7), [o8] "=v"(o8), [o9] "=v"(o9), [o10] "=v"(o10), [o11] "=v"(o11), [o12] "=v"(o12),               \
              [o13] "=v"(o13), [o14] "=v"(o14), [o15] "=v"(o15)                                                            \
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2),             \
              [b3] "v"(b3), [iters] "s"(iters), [poison] "v"(poison), [orig] "v"(orig)                                     \
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",      \
              "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s20", "scc", "memory");                       \
        float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;                                             \
        o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4; o[5] = o5; o[6] = o6; o[7] = o7;                            \
        o[8] = o8; o[9] = o9; o[10] = o10; o[11] = o11; o[12] = o12; o[13] = o13; o[14] = o14; o[15] = o15;                \
    }

#define M_BF16 "v_mfma_f32_32x32x16_bf16 v[48:63], v[40:43], v[44:47], v[48:63]"
#define M_F16 "v_mfma_f32_32x32x16_f16 v[48:63], v[40:43], v[44:47], v[48:63]"
#define M_F32 "v_mfma_f32_32x32x2_f32 v[48:63], v40, v44, v[48:63]"
#define G0 ""
#define G1 "s_nop 0\n"
#define G2 "s_nop 1\n"
#define G4 "s_nop 3\n"
#define G8 "s_nop 7\n"
#define G16 "s_nop 15\n"
#define G32 "s_nop 15\n s_nop 15\n"

#define ALLGAPS(P, M, W)                                                                                       \
    PROBE(P##_g0, M, W, G0) PROBE(P##_g1, M, W, G1) PROBE(P##_g2, M, W, G2) PROBE(P##_g4, M, W, G4)              \
    PROBE(P##_g8, M, W, G8) PROBE(P##_g16, M, W, G16) PROBE(P##_g32, M, W, G32)

ALLGAPS(bf16_a0, M_BF16, 40) ALLGAPS(bf16_a3, M_BF16, 43) ALLGAPS(bf16_b0, M_BF16, 44) ALLGAPS(bf16_b3, M_BF16, 47)
ALLGAPS(f16_a0, M_F16, 40) ALLGAPS(f16_b3, M_F16, 47)
ALLGAPS(f32_a, M_F32, 40) ALLGAPS(f32_b, M_F32, 44)


// overwrite by a transcendental (v_exp_f32: exp2(poison') with poison' chosen by the host so that the result is the garbage), by
// the bf16 pack the epilogues use, and the NEGATIVE CONTROL: garbage written BEFORE the MFMA (must be seen)
#define OVW_EXP(W) "v_exp_f32 v" STR(W) ", %[poison]\n"
#define OVW_CVT(W) "v_cvt_pk_bf16_f32 v" STR(W) ", %[poison], %[poison]\n"
PROBE2(bf16_a0_exp_g0, M_BF16, 40, G0, OVW_EXP(40), "") PROBE2(bf16_b3_exp_g0, M_BF16, 47, G0, OVW_EXP(47), "")
PROBE2(f32_a_exp_g0, M_F32, 40, G0, OVW_EXP(40), "") PROBE2(f32_b_exp_g0, M_F32, 44, G0, OVW_EXP(44), "")
PROBE2(bf16_a0_cvt_g0, M_BF16, 40, G0, OVW_CVT(40), "") PROBE2(f16_b3_cvt_g0, M_F16, 47, G0, OVW_CVT(47), "")
PROBE2(bf16_a0_neg, M_BF16, 40, G0, "", "v_mov_b32 v40, %[poison]\n s_nop 7\n")
PROBE2(f32_b_neg, M_F32, 44, G0, "", "v_mov_b32 v44, %[poison]\n s_nop 7\n")

typedef void (*kern_t)(const unsigned*, float*, int, unsigned, int);
struct Case { const char* name; int kind; int which; kern_t k[7]; };
#define CASE(P, KIND, W) {#P, KIND, W, {P##_g0, P##_g1, P##_g2, P##_g4, P##_g8, P##_g16, P##_g32}}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static unsigned short f16_of_small_int(int v) {   // exact for 0..3
    static const unsigned short t[4] = {0x0000, 0x3c00, 0x4000, 0x4200};
    return t[v & 3];
}

int main() {
    Case cases[] = {CASE(bf16_a0, 0, 40), CASE(bf16_a3, 0, 43), CASE(bf16_b0, 0, 44), CASE(bf16_b3, 0, 47),
                    CASE(f16_a0, 1, 40), CASE(f16_b3, 1, 47), CASE(f32_a, 2, 40), CASE(f32_b, 2, 44)};
    const int gaps[7] = {0, 1, 2, 4, 8, 16, 32};
    const int iters = 2000;
    for (int wpb : {8, 4}) {   // (workgroups of <= 512 threads: several co-reside on a CU, up to 8 waves per SIMD)          // waves per workgroup: 2 / 1 / 4 per SIMD
        const int threads = wpb * 64, blocks = 512;
        const size_t n = (size_t)blocks * threads * 16;
        unsigned* d_ab; float* d_out;
        hipMalloc(&d_ab, 64 * 8 * 4);
        hipMalloc(&d_out, n * 4);
        std::vector<float> ref(n), got(n);
        printf("== %d waves per workgroup (%d per SIMD), %d workgroups, %d MFMAs per wave\n", wpb, wpb / 4, blocks, iters);
        for (auto& c : cases) {
            // operands: small integers, so that every fp32 sum is exact
            std::vector<unsigned> ab(64 * 8);
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 8; ++r) {
                    const int v0 = (l * 7 + r * 3) % 3, v1 = (l * 5 + r * 11 + 1) % 3;
                    if (c.kind == 0) ab[l * 8 + r] = bf16_of((float)v0) | ((unsigned)bf16_of((float)v1) << 16);
                    else if (c.kind == 1) ab[l * 8 + r] = f16_of_small_int(v0) | ((unsigned)f16_of_small_int(v1) << 16);
                    else { float f = (float)(v0 + 1); memcpy(&ab[l * 8 + r], &f, 4); }
                }
            hipMemcpy(d_ab, ab.data(), ab.size() * 4, hipMemcpyHostToDevice);
            printf("%-8s:", c.name);
            for (int g = 0; g < 7; ++g) {
                // poison: a pattern that changes every product it enters (bf16 / f16: both halves 3.0; f32: 5.0)
                const unsigned poison = c.kind == 0 ? 0x40404040u : (c.kind == 1 ? 0x42004200u : 0x40a00000u);
                hipMemset(d_out, 0, n * 4);
                hipLaunchKernelGGL(c.k[g], dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters, poison, 1);   // clean reference
                hipDeviceSynchronize();
                hipMemcpy(ref.data(), d_out, n * 4, hipMemcpyDeviceToHost);
                hipMemset(d_out, 0, n * 4);
                hipLaunchKernelGGL(c.k[g], dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters, poison, 0);
                hipDeviceSynchronize();
                hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost);
                size_t bad = 0, waves_bad = 0;
                for (size_t w = 0; w < n / 1024; ++w) {
                    size_t b = 0;
                    for (size_t i = 0; i < 1024; ++i) b += got[w * 1024 + i] != ref[w * 1024 + i] ? 1 : 0;
                    bad += b;
                    waves_bad += b ? 1 : 0;
                }
                printf("  gap%-2d %zu waves (%zu values)", gaps[g], waves_bad, bad);
            }
            printf("\n");
        }
        struct X { const char* name; int kind; kern_t k; unsigned poison; } xs[] = {
            {"bf16_a0 <- v_exp_f32", 0, bf16_a0_exp_g0, 0x40000000u}, {"bf16_b3 <- v_exp_f32", 0, bf16_b3_exp_g0, 0x40000000u},
            {"f32_a <- v_exp_f32", 2, f32_a_exp_g0, 0x40400000u}, {"f32_b <- v_exp_f32", 2, f32_b_exp_g0, 0x40400000u},
            {"bf16_a0 <- v_cvt_pk_bf16_f32", 0, bf16_a0_cvt_g0, 0x40400000u}, {"f16_b3 <- v_cvt_pk_bf16_f32", 1, f16_b3_cvt_g0, 0x40400000u},
            {"NEGATIVE CONTROL bf16_a0 (garbage BEFORE the MFMA)", 0, bf16_a0_neg, 0x40404040u},
            {"NEGATIVE CONTROL f32_b (garbage BEFORE the MFMA)", 2, f32_b_neg, 0x40a00000u}};
        for (auto& x : xs) {
            std::vector<unsigned> ab(64 * 8);
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 8; ++r) {
                    const int v0 = (l * 7 + r * 3) % 3, v1 = (l * 5 + r * 11 + 1) % 3;
                    if (x.kind == 0) ab[l * 8 + r] = bf16_of((float)v0) | ((unsigned)bf16_of((float)v1) << 16);
                    else if (x.kind == 1) ab[l * 8 + r] = f16_of_small_int(v0) | ((unsigned)f16_of_small_int(v1) << 16);
                    else { float f = (float)(v0 + 1); memcpy(&ab[l * 8 + r], &f, 4); }
                }
            hipMemcpy(d_ab, ab.data(), ab.size() * 4, hipMemcpyHostToDevice);
            // clean reference: the plain kernel of the same MFMA kind with the rewrite of the original value
            kern_t clean = x.kind == 0 ? bf16_a0_g32 : (x.kind == 1 ? f16_a0_g32 : f32_a_g32);
            hipMemset(d_out, 0, n * 4);
            hipLaunchKernelGGL(clean, dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters, 0u, 1);
            hipDeviceSynchronize();
            hipMemcpy(ref.data(), d_out, n * 4, hipMemcpyDeviceToHost);
            hipMemset(d_out, 0, n * 4);
            hipLaunchKernelGGL(x.k, dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters, x.poison, 0);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i] ? 1 : 0;
            printf("%-52s gap 0: %zu of %zu values differ from the clean run\n", x.name, bad, n);
        }
        hipFree(d_ab); hipFree(d_out);
    }
    return 0;
}
