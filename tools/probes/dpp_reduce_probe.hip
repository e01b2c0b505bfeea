// Probe (gfx950): the DPP forms of the compositing helpers against their shuffle forms on random data, one wave.
//   hipcc --offload-arch=gfx950 -O3 -o dpp_reduce_probe tools/probes/dpp_reduce_probe.hip && ./dpp_reduce_probe
// The DPP forms reproduce the shuffle forms' association step for step: every output must be BITWISE equal to the shuffle form's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define DEVINL __device__ __forceinline__
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
DEVINL float dpp_f(float old, float src) {      // lanes that are masked off or have no source lane keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                   ROW_MASK, BANK_MASK, false));
}
DEVINL float sum_shfl(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// v_permlane16_swap exchanges the odd rows of its first register with the even rows of its second: with one value in both, every
// lane of rows 0 / 1 ends up with (row 0's value, row 1's value) of its column, rows 2 / 3 likewise.  Two pitfalls of hipcc 7.2,
// both seen here: given ONE value twice the operands may share a register and the swap degenerates (hence the opaque copy), and
// bit-casting element 1 of the returned pair directly reads element 0 (hence the two unsigned temporaries).
DEVINL void rows_pair(float v, float& even_row, float& odd_row) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    even_row = __uint_as_float(r0);
    odd_row = __uint_as_float(r1);
}
// the xor butterfly 16, 8, 4, 2, 1 of sum_shfl, step for step (bitwise the same result)
DEVINL float sum_dpp(float v) {
    float e, o;
    rows_pair(v, e, o);
    v = e + o;                                         // xor 16
    v += dpp_f<0x128>(v, v);                           // xor 8 = row_ror:8
    float q = dpp_f<0x124, 0xf, 0xa>(v, v);            // xor 4: banks 1, 3 take lane - 4 (row_ror:4) ...
    q = dpp_f<0x12C, 0xf, 0x5>(q, v);                  //        banks 0, 2 take lane + 4 (row_ror:12)
    v += q;
    v += dpp_f<0x4E>(v, v);                            // xor 2
    v += dpp_f<0xB1>(v, v);                            // xor 1
    return v;
}
DEVINL float wsum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// xor 32 through v_permlane32_swap (upper half of the first register <-> lower half of the second), then the 32-lane butterfly
DEVINL float wsum_dpp(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return sum_dpp(__uint_as_float(r0) + __uint_as_float(r1));
}
DEVINL float scan_shfl(float t, int lane32, float* total) {
#pragma clang fp contract(off)
    float p = t;
#pragma unroll
    for (int of = 1; of < 32; of <<= 1) {
        const float q = __shfl_up(p, of, 32);
        if (lane32 >= of) p *= q;
    }
    float excl = __shfl_up(p, 1, 32);
    if (lane32 == 0) excl = 1.f;
    *total = __shfl(p, 31, 32);
    return excl;
}
// the Hillis-Steele scan of scan_shfl, step for step (bitwise the same result): p[l] *= p[l - of] for of = 1, 2, 4, 8, 16 inside
// each 32-lane group.  Lanes whose source lies in the row before take it from the swapped copy (rows_pair), rotated into place.
DEVINL float scan_dpp(float t, int lane32, float* total) {
#pragma clang fp contract(off)
    float p = t, q, e, o;
    q = dpp_f<0x138>(1.f, p);                          // of = 1: wave_shr:1
    p *= (lane32 >= 1) ? q : 1.f;
    q = dpp_f<0x138>(1.f, p);                          // of = 2: two single shifts
    q = dpp_f<0x138>(1.f, q);
    p *= (lane32 >= 2) ? q : 1.f;
    rows_pair(p, e, o);                                // of = 4: in-row lanes from row_shr:4, the first four of rows 1 / 3 from the row before
    q = dpp_f<0x114>(1.f, p);
    q = dpp_f<0x124, 0xa, 0x1>(q, e);                  //   row_ror:4 of the even row's values, rows 1 / 3, bank 0
    p *= q;
    rows_pair(p, e, o);                                // of = 8
    q = dpp_f<0x118>(1.f, p);
    q = dpp_f<0x128, 0xa, 0x3>(q, e);                  //   row_ror:8, rows 1 / 3, banks 0 - 1
    p *= q;
    rows_pair(p, e, o);                                // of = 16: rows 1 / 3 times the row before, same column
    p *= (lane32 >= 16) ? e : 1.f;
    float excl = dpp_f<0x138>(1.f, p);
    if (lane32 == 0) excl = 1.f;
    const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 31));
    const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 63));
    *total = (threadIdx.x & 32) ? t1 : t0;
    return excl;
}
__global__ void probe(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l], t = in[64 + l];
    float ta, tb;
    out[l] = sum_shfl(v);
    out[64 + l] = sum_dpp(v);
    out[128 + l] = scan_shfl(t, l & 31, &ta);
    out[192 + l] = scan_dpp(t, l & 31, &tb);
    out[256 + l] = ta;
    out[320 + l] = tb;
    out[384 + l] = wsum_shfl(v);
    out[448 + l] = wsum_dpp(v);
}
int main() {
    float h[128], o[512], *di, *dout;
    srand(7);
    int bad = 0, nb[6] = {0, 0, 0, 0, 0, 0};
    hipMalloc(&di, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 64; ++i) h[i] = (float)rand() / RAND_MAX - 0.3f;
        for (int i = 0; i < 64; ++i) h[64 + i] = 0.5f + 0.5f * (float)rand() / RAND_MAX;
        hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, di, dout);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {
            double ref = 0, pr = 1;
            for (int i = 0; i < 32; ++i) ref += h[32 * g + i];
            for (int i = 0; i < 32; ++i) {
                const int l = 32 * g + i;
                if (o[64 + l] != o[64 + 32 * g]) { ++bad; ++nb[0]; }                                  // uniform over the group, bitwise
                if (fabs(o[64 + l] - ref) > 1e-5) { ++bad; ++nb[1]; } if (fabs(o[l] - ref) > 1e-5) { ++bad; ++nb[2]; }      // both forms are the sum
                if (fabs(o[192 + l] - pr) > 1e-5 * pr) { ++bad; ++nb[3]; } if (fabs(o[128 + l] - pr) > 1e-5 * pr) { ++bad; ++nb[4]; }  // exclusive products
                pr *= h[64 + l];
                if (o[448 + l] != o[384 + l]) { ++bad; ++nb[5]; }
                if (o[320 + l] != o[256 + l] || o[192 + l] != o[128 + l] || o[64 + l] != o[l]) { ++bad; ++nb[5]; }     // totals
            }
            if (fabs(o[320 + 32 * g] - pr) > 1e-5 * pr) ++bad;
        }
    }
    printf("dpp_reduce_probe: %d mismatches over 200 random waves (sum uniform %d, dpp sum %d, shfl sum %d, dpp scan %d, shfl scan %d, dpp != shfl bitwise %d)\n", bad, nb[0], nb[1], nb[2], nb[3], nb[4], nb[5]);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: sum %.6f / %.6f   excl %.6g / %.6g   tot %.6g / %.6g\n", l, o[l], o[64 + l], o[128 + l], o[192 + l], o[256 + l], o[320 + l]);
    return bad != 0;
}
