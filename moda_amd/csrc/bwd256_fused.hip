// One hidden layer of the 256-wide network's bf16-storage backward as ONE launch (gfx950): dW, db and the masked dX together.
//
// Reference: the autograd backward of one nn.Linear(256, 256) + ReLU of NeRF.forward (nnutils/nerf.py:147-198) --
//   dW += dZ^T @ X,   db += 1^T dZ,   dX = (dZ @ W) (.) [X > 0]
// with dZ (M x 256) the gradient at the layer's pre-activation, X (M x 256) its input = the ReLU output of the layer below
// (so the ReLU mask of dX is the sign of X itself), M = rays x samples.  gemm_bf16.hip runs this as two launches, each bound
// by its HBM stream: the dW form reads dZ and X (and leaves X's sign map behind), the dX form reads dZ and the map and writes
// dX -- 0.55 GB per layer at cfg4's M = 262 144, dZ crossing the fabric twice.  Here a workgroup reads a 64-sample tile of dZ
// and of X once and does both products from the same LDS image: 0.40 GB per layer.
//
// Work split: workgroup = (stream p, half); it owns 128 of the 256 INPUT columns i (the half) and walks the 64-sample tiles
// p, p + P, p + 2P, ... (persistent: one workgroup per CU).  Both halves of a stream read the same dZ tiles; their dispatch
// ids are 8 apart, i.e. on one XCD, and they run in step, so the second read is an L2 hit.  Eight waves, wave = (ib, mh):
//   dX^T[i][m] = sum_o W[o][i] dZ[m][o]   one 32 x 32 tile per wave: i-block ib of the half, sample rows 32 mh .. +31;
//                                         the A operand (W^T rows i, all 256 o) lives in 64 VGPRs for the whole kernel,
//                                         the B operand is a ds_read_b128 of the dZ image (k = o is the fast index);
//   dW[o][i]  += sum_m dZ[m][o] X[m][i]   four 32 x 32 tiles per wave: o-blocks 4 mh .. 4 mh + 3, i-block ib; accumulated in
//                                         64 VGPRs over ALL tiles of the stream, one set of atomics per workgroup at the end;
//                                         both operands have k = m as the slow index: ds_read_b64_tr_b16 of the same images.
// One image per operand serves both read patterns: 16-byte chunk c of row r sits at chunk c ^ f(r), f(r) = (r & 3) << 2 |
// (r >> 2) & 3 -- sixteen consecutive rows of one chunk column land in sixteen different bank groups (the b128 reads), and the
// four rows of a transposing read in four different 64-byte groups.
// Per tile and workgroup: 48 KB in (dZ 32, X 16), 16 KB out, 256 MFMAs (32 per wave), one barrier.  The stream is LDS-DMA into
// three stages (tiles k+1 and k+2 in flight while tile k is multiplied: one tile ahead, the first version of this kernel, left
// it latency-bound at 4.2 TB/s without any arithmetic), the swizzle applied to the global address of each lane; db is summed
// from the dZ image by the workgroups of half 0; dX^T leaves through a 2 KB staging buffer per wave (64-byte row pieces).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef B256_PIN
#define B256_PIN 1
#endif
constexpr int MT = 64;                 // samples per tile
constexpr int HI = 128;                // input columns per workgroup
constexpr int XROW = HI * 2, XS = MT * XROW;
constexpr int OWAVE = 64 * 64;         // a dX wave's dX^T staging: [64 m][32 i] bf16
// WD = the layer's width (both sides): 256 (two workgroups per stream, one per half of the input columns) or 128 (one)
template <int WD> struct Geo {
    static constexpr int ZROW = WD * 2, ZS = MT * ZROW, STAGE = ZS + XS;       // 32 + 16 KB | 16 + 16 KB
    static constexpr int NSTAGE = WD == 256 ? 3 : 4;                             // tiles in the ring: NSTAGE - 1 in flight beside the one read
    static constexpr int LDS_BYTES = NSTAGE * STAGE + 4 * OWAVE;                 // 163 840 (all of a CU's LDS) | 147 456
    static constexpr int ZPW = ZS / 1024 / 8;                                    // dZ DMA instructions per wave and tile (1 KB each): 4 | 2
    static constexpr int ZRPI = 1024 / ZROW;                                     // rows per instruction: 2 | 4
    static constexpr int DPT = ZPW + 2;                                          // DMA instructions per wave and tile
    static constexpr int KW = WD / 16, NOB = WD / 32;                            // k-steps of the dX product, o-blocks of dW
    static constexpr int ZCH = ZROW / 16;                                        // 16-byte chunks per dZ row
};

struct B256Args {
    const unsigned short* dz; long long ldz;
    const unsigned short* x; long long ldx;
    const unsigned short* wb; long long ldw;      // [o][i] bf16
    unsigned short* dx; long long ldo;
    float* gW; long long ldg;
    float* gb;
    long long M;
    int streams;
};

DEVINL int fsw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
DEVINL unsigned pk2(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}
DEVINL float bflo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
DEVINL float bfhi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// bf16 > 0  <=>  its 16 bits, as a signed integer, > 0 (gemm_bf16.hip's mask rule)
DEVINL unsigned keep(unsigned vw, unsigned mw) {
    const unsigned lo = ((short)(mw & 0xffffu) > 0) ? 0x0000ffffu : 0u;
    const unsigned hi = ((int)mw > 0xffff) ? 0xffff0000u : 0u;
    return vw & (lo | hi);
}

template <int WD>
__global__ __launch_bounds__(512, 1) void bwd256_kernel(B256Args a) {
    typedef Geo<WD> G_;
    constexpr int ZROW = G_::ZROW, ZS = G_::ZS, STAGE = G_::STAGE, NSTAGE = G_::NSTAGE, ZPW = G_::ZPW, ZRPI = G_::ZRPI, DPT = G_::DPT,
                  KW = G_::KW, NOB = G_::NOB, ZCH = G_::ZCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef void __attribute__((address_space(3))) * lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, l31 = lane & 31;
    const int ib = wave & 3, mh = wave >> 2;
    const unsigned b = blockIdx.x;
    const int half = WD == 256 ? (int)((b >> 3) & 1u) : 0;
    const int p = WD == 256 ? (int)((b >> 4) * 8 + (b & 7u)) : (int)b;
    // (tiles are dealt round-robin, p, p + P, ...: at any moment the workgroups read CONSECUTIVE tiles.  Contiguous shares per
    // stream -- tried in order to stagger the workgroups' ends and run the dW atomics of the early ones under the stream of the
    // late ones -- put every workgroup's address a multiple of 256 KB from its neighbour's: 104 -> 106-120 us at W = 256,
    // 51 -> 175 us at W = 128, whatever the stagger)
    const int P = a.streams;
    const long long ntiles = (a.M + MT - 1) / MT;
    if (p >= ntiles) return;
    const int nit = (int)((ntiles - p + P - 1) / P);

    // ---- the stream: LDS-DMA, 16 bytes per lane, 1 KB per instruction (two rows of the dZ image / four of the X image); the
    //      destination is linear in the lane index, so the chunk swizzle is applied on the GLOBAL side: lane -> (row, physical
    //      chunk pc) fetches logical chunk pc ^ f(row).  Six instructions per wave and tile.  A row past M is out of the buffer's
    //      range and arrives as zeros (the tile's base is part of the VECTOR offset: the range check does not see the scalar one).
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, (int)(unsigned)(a.M * a.ldo * 2), 0x00020000);
    const unsigned ldz2 = (unsigned)a.ldz * 2u, ldx2 = (unsigned)a.ldx * 2u, ldo2 = (unsigned)a.ldo * 2u;
    unsigned zdma[ZPW], xdma[2];
#pragma unroll
    for (int i = 0; i < ZPW; ++i) {
        const int row = ZRPI * (ZPW * wave + i) + lane / ZCH;
        zdma[i] = (unsigned)row * ldz2 + 16u * (unsigned)((lane % ZCH) ^ fsw(row));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (2 * wave + i) + (lane >> 4);
        xdma[i] = (unsigned)row * ldx2 + (unsigned)half * (HI * 2) + 16u * (unsigned)((lane & 15) ^ fsw(row));
    }
    // The DMA instructions are asm statements: hipcc then neither counts them nor orders LDS reads behind them (through the
    // builtin every ds_read that follows waits with vmcnt(0) -- it cannot tell the stages apart -- and the stream is drained
    // once per tile); their completion is counted by hand below.  M0 = the wave-uniform LDS address, saved and restored.
    const u32x4_ dz_rs = {(unsigned)(uintptr_t)a.dz, (unsigned)((uintptr_t)a.dz >> 32) & 0xffffu, (unsigned)(a.M * a.ldz * 2), 0x00020000u};
    const u32x4_ x_rs = {(unsigned)(uintptr_t)a.x, (unsigned)((uintptr_t)a.x >> 32) & 0xffffu, (unsigned)(a.M * a.ldx * 2), 0x00020000u};
    auto dma16 = [&](const u32x4_& rs, unsigned lds_dst, unsigned voff) __attribute__((always_inline)) {
        unsigned keep_m0;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "s"(lds_dst), "v"(voff), "s"(rs) : "memory");
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int k) __attribute__((always_inline)) {     // tile k of this stream -> stage k % 3 (any k: past the end it is zeros)
#ifdef B256_ABL_NODMA
        const unsigned m0 = 0xffffff00u;     // every row out of range: the instruction stream stays, nothing is fetched
#else
        const unsigned m0 = ((unsigned)p + (unsigned)k * (unsigned)P) * MT;
#endif
        const unsigned st = lds0 + (unsigned)(k % NSTAGE) * STAGE;
#pragma unroll
        for (int i = 0; i < ZPW; ++i) dma16(dz_rs, st + ((unsigned)ZPW * wave_u + i) * 1024u, zdma[i] + m0 * ldz2);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(x_rs, st + ZS + (2u * wave_u + i) * 1024u, xdma[i] + m0 * ldx2);
    };

#pragma unroll
    for (int k = 0; k < NSTAGE - 1; ++k) issue(k);
    const int zc = tid % ZCH, zr = (tid & 255) / ZCH;   // db (the dX waves' 256 threads): thread -> chunk column zc of rows zr + (256 / ZCH) e
    const bool do_db = a.gb != nullptr && half == 0;
    // (the lgkmcnt(0) in front of every barrier retires the wave's own LDS reads of the tile: the stage is restaged by the DMA
    // the other waves issue right behind that barrier -- the guide's WAR rule; hipcc moves the last MFMAs below the barrier)
    // vmcnt counts LDS-DMA, loads and stores together in issue order: behind a tile's six DMA instructions a wave issues, per
    // iteration, six more (the tile after) and its stores, and that many operations may stay in flight when the tile is needed.
    // The two roles below run the same loop: [DMA of tile k+2 into the stage tile k-1 was read from] [tile k] [wait for tile k+1,
    // barrier: everybody is done with tile k].

    if (wave < 4) {
        // ================= dX waves: dX^T[i-block ib][all 64 m] = W^T dZ^T, mask, store ======================================
        float xsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // db: this thread's chunk column of dZ, summed over its rows of every tile
        auto db_tile = [&](const unsigned char* St) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < MT * ZCH / 256; ++e) {
                const int row = zr + (256 / ZCH) * e;
                const u32x4_ v = *(const u32x4_*)(St + row * ZROW + 16 * (zc ^ fsw(row)));
                xsum8[0] += bflo(v.x); xsum8[1] += bfhi(v.x); xsum8[2] += bflo(v.y); xsum8[3] += bfhi(v.y);
                xsum8[4] += bflo(v.z); xsum8[5] += bfhi(v.z); xsum8[6] += bflo(v.w); xsum8[7] += bfhi(v.w);
            }
        };
        bf16x8 wreg[KW];        // W^T rows of the i-block, all WD o: the A operand, resident
        {
            const unsigned short* wp = a.wb + half * HI + ib * 32 + l31;
            const int ldw = (int)a.ldw;
#pragma unroll
            for (int u = 0; u < KW; ++u) {
                union { unsigned short s[8]; bf16x8 v; } t;
#pragma unroll
#ifdef B256_ABL_NOWREG
                for (int e = 0; e < 8; ++e) t.s[e] = (unsigned short)(0x3c00 + lane + u + e + ldw);
#else
                for (int e = 0; e < 8; ++e) t.s[e] = wp[(16 * u + 8 * h + e) * ldw];
#endif
                wreg[u] = t.v;
            }
        }
        const int fz = fsw(l31);                          // f of rows l31 and 32 + l31 alike
        const int zkf = l31 * ZROW;                       // + 32 mb ZROW + 16 ((2u + h) ^ fz)
        const int xmk = ZS + l31 * XROW + 8 * h;          // + 32 mb XROW + 16 ((4 ib + j) ^ fz): mask words of accumulator rows 4j .. 4j+3
        // the wave's staging buffer: [64 m][32 i] bf16, row m = four 16-byte chunks, chunk j at j ^ (m >> 2) & 3
        unsigned char* const Ow = lds + NSTAGE * STAGE + wave * OWAVE;
        const int ost = l31 * 64 + 8 * h, osw = (l31 >> 2) & 3;
        const int frow = lane >> 2, fch = lane & 3;       // flush: lane -> row frow + 16 e, logical chunk fch
        const unsigned ovo = (unsigned)frow * ldo2 + (unsigned)half * (HI * 2) + (unsigned)ib * 64u + 16u * fch;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * DPT) : "memory");       // tile 0 (the later tiles' DMA behind it)
        for (int k = 0; k < nit; ++k) {
            issue(k + NSTAGE - 1);
            asm volatile("" ::: "memory");
            const unsigned char* St = lds + (k % NSTAGE) * STAGE;
            const unsigned m0 = ((unsigned)p + (unsigned)k * (unsigned)P) * MT;
            if (do_db) db_tile(St);
            f32x16 accx[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[mb][r] = 0.f;
            constexpr int AH = 4;               // k-steps read ahead
            bf16x8 zb[AH][2];
#pragma unroll
            for (int u = 0; u < AH; ++u)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) zb[u][mb] = *(const bf16x8*)(St + zkf + 32 * mb * ZROW + 16 * ((2 * u + h) ^ fz));
#pragma unroll
            for (int u = 0; u < KW; ++u) {
#if B256_PIN
                __builtin_amdgcn_sched_barrier(0);      // keep the read-ahead: hipcc otherwise sinks every read to its MFMA (lgkmcnt(0) per MFMA)
#endif
#ifndef B256_ABL_NODX
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) accx[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[u], zb[u % AH][mb], accx[mb], 0, 0, 0);
#else
                accx[0][u] += (float)zb[u % AH][0][0] + (float)zb[u % AH][1][1];
#endif
                if (u + AH < KW)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) zb[u % AH][mb] = *(const bf16x8*)(St + zkf + 32 * mb * ZROW + 16 * ((2 * (u + AH) + h) ^ fz));
            }
            // C/D map: lane l register r -> tile row (r & 3) + 8 (r >> 2) + 4 (l >> 5) = i, tile column l & 31 = m
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint2 mk = *(const uint2*)(St + xmk + 32 * mb * XROW + 16 * ((4 * ib + j) ^ fz));
                    *(uint2*)(Ow + 32 * mb * 64 + ost + 16 * (j ^ osw)) =
                        make_uint2(keep(pk2(accx[mb][4 * j], accx[mb][4 * j + 1]), mk.x), keep(pk2(accx[mb][4 * j + 2], accx[mb][4 * j + 3]), mk.y));
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = frow + 16 * e;
                const u32x4_ v = *(const u32x4_*)(Ow + row * 64 + 16 * (fch ^ ((row >> 2) & 3)));
#ifdef B256_ABL_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, 0xffffff00u, 0, 0);
#else
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ovo + (m0 + 16u * e) * ldo2, 0, 0);
#endif
            }
            // tile k+1 is needed: the DMA of the NSTAGE - 2 tiles behind it and this tile's 4 stores may stay in flight
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * DPT + 4) : "memory");
        }
        if (do_db) {        // (every wave is past its last tile: the stages are free once EVERY wave's DMA of the tiles past the end has
            // landed -- a wave's own vmcnt(0) says nothing about the rows other waves DMA into the bytes `red` occupies: all waves
            // drain, then meet (the dW waves at the matching barrier behind their loop), then `red` is written)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            float* red = (float*)lds;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(zr * ZCH + zc) * 8 + e] = xsum8[e];
        }
    } else {
        // ================= dW waves: dW[all 256 o][i-block ib] += dZ^T X over the tile's 64 samples ===========================
        // transposing read (gemm_bf16.hip): lane (h, g & 1, q4, p4) supplies row 16u + 8h + q4 (+ 4 for the high half), elements
        // 16 (g & 1) + 4 p4 .. + 3 of the 32-wide tile; f of that row = q4 << 2 | (2h + e) & 3, whatever u.  o-blocks ob and
        // ob + 4 are 256 bytes apart in a row (f touches the low four chunk bits only).
        int ztr[4][2], xtr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = 8 * h + q4 + 4 * e;
            const int sub = 2 * (g & 1) + (p4 >> 1);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) ztr[ob][e] = row * ZROW + 16 * ((4 * ob + sub) ^ fsw(row)) + 8 * (p4 & 1);
            xtr[e] = ZS + row * XROW + 16 * ((4 * ib + sub) ^ fsw(row)) + 8 * (p4 & 1);
        }
        typedef s16x4 __attribute__((address_space(3))) lds_s16x4;
        auto tr_frag = [&](const unsigned char* base, int lo_off, int hi_off) __attribute__((always_inline)) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + lo_off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + hi_off));
            union { struct { s16x4 a, b; } s; bf16x8 v; } o;
            o.s.a = lo;
            o.s.b = hi;
            return o.v;
        };
        f32x16 accw[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[ob][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * DPT) : "memory");
        for (int k = 0; k < nit; ++k) {
            issue(k + NSTAGE - 1);
            asm volatile("" ::: "memory");
            // groups of four o-blocks: group g = (k-step u = g / NH, o-blocks 4 (g % NH) .. + 3); the fragments of group g + 1 are
            // read while group g multiplies (pinned: hipcc otherwise sinks every read to its MFMA, one lgkmcnt(0) per MFMA)
            constexpr int NH = NOB / 4, NGRP = (MT / 16) * NH;
            bf16x8 xb[2], za[2][4];
            // this tile's ten read bases, opaque to hipcc: every fragment address is then base + an immediate (left to itself it
            // keeps one address register per fragment, ~40, and spills the accumulators' neighbours)
            const int st_off = (k % NSTAGE) * STAGE;
            int zt[4][2], xt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xt[e] = xtr[e] + st_off;
                asm volatile("" : "+v"(xt[e]));
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    zt[ob][e] = ztr[ob][e] + st_off;
                    asm volatile("" : "+v"(zt[ob][e]));
                }
            }
            auto frags = [&](int g_, int s_) __attribute__((always_inline)) {
                const int u = g_ / NH, hh = g_ % NH;
                if (hh == 0) xb[u & 1] = tr_frag(lds, xt[0] + 16 * u * XROW, xt[1] + 16 * u * XROW);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) za[s_][ob] = tr_frag(lds, zt[ob][0] + 256 * hh + 16 * u * ZROW, zt[ob][1] + 256 * hh + 16 * u * ZROW);
            };
            frags(0, 0);
#pragma unroll
            for (int g_ = 0; g_ < NGRP; ++g_) {
#if B256_PIN
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (g_ + 1 < NGRP) frags(g_ + 1, (g_ + 1) & 1);
#if B256_PIN
                __builtin_amdgcn_sched_barrier(0);
#endif
                const int u = g_ / NH, hh = g_ % NH;
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#ifndef B256_ABL_NODW
                    accw[4 * hh + ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za[g_ & 1][ob], xb[u & 1], accw[4 * hh + ob], 0, 0, 0);
#else
                    accw[4 * hh + ob][u] += (float)za[g_ & 1][ob][0] + (float)xb[u & 1][1];
#endif
            }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * DPT) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (do_db) asm volatile("s_barrier" ::: "memory");      // (the dX waves' barrier in front of their `red` writes: see there)
        // one set of atomics per workgroup
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            float* gp = a.gW + (long long)(32 * ob + 4 * h) * a.ldg + half * HI + 32 * ib + l31;
#pragma unroll
#ifdef B256_ABL_NOATOM
            for (int r = 0; r < 16; ++r) if (accw[ob][r] == 1.2345f) gp[r] = 0.f;
#else
            for (int r = 0; r < 16; ++r) atomicAdd(gp + (long long)((r & 3) + 8 * (r >> 2)) * a.ldg, accw[ob][r]);
#endif
        }
    }
    if (do_db) {        // the dX waves' threads of a chunk column hold partial sums of the same eight o
        const float* red = (const float*)lds;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < WD) {
            float s = 0.f;
#pragma unroll
            for (int rw = 0; rw < 256 / ZCH; ++rw) s += red[(rw * ZCH + (tid >> 3)) * 8 + (tid & 7)];
            atomicAdd(a.gb + tid, s);
        }
    }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int WD>
int b256_launch(const B256Args& a0, long long M, hipStream_t st) {
    static std::atomic<unsigned long long> attr_set{0ull};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) devid = 0;
    const unsigned long long bit = 1ull << (devid & 63);
    if (devid > 63 || !(attr_set.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)bwd256_kernel<WD>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<WD>::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    static const long long streams_env = [] { const char* e = getenv("MODA_BWD256_STREAMS"); return e ? atoll(e) : 0LL; }();
    const long long ntiles = (M + MT - 1) / MT;
    long long P = streams_env > 0 ? streams_env : (WD == 256 ? 128 : 256);      // one workgroup per CU
    if (P > ntiles) P = ntiles;
    if (WD == 256) P = (P + 7) / 8 * 8;                                          // whole groups of 8: the (half, stream) <-> dispatch id map
    B256Args a = a0;
    a.streams = (int)P;
    hipLaunchKernelGGL(bwd256_kernel<WD>, dim3((unsigned)(WD == 256 ? 2 * P : P)), dim3(512), Geo<WD>::LDS_BYTES, st, a);
    return (int)hipGetLastError();
}

}   // namespace

// One W -> W hidden layer, W = 256 or 128:  dX (M x W, bf16) = (dZ @ Wt) . [X > 0];  gW (W x W at ldg, fp32) += dZ^T X;
// gb (W) += column sums of dZ (or null).  Returns MODA_ESHAPE when the operands do not fit the kernel (the caller keeps its
// two-launch route then).
int moda_bwd256_layer(int W, const void* dz, long long ldz, const void* x, long long ldx, const void* wb, long long ldw, void* dx,
                      long long ldo, float* gW, long long ldg, float* gb, long long M, void* stream) {
    if (!dz || !x || !wb || !dx || !gW) return MODA_EINVAL;
    if (W != 256 && W != 128) return MODA_ESHAPE;
    if (M <= 0) return 0;
    if (!al16(dz) || !al16(x) || !al16(dx) || ldz % 8 || ldx % 8 || ldo % 8 || ldz < W || ldx < W || ldo < W || ldw < W || ldg < W)
        return MODA_ESHAPE;
    const long long lim = 0x7fffffffLL;               // buffer offsets are 32-bit
    if (M * ldz * 2 > lim || M * ldx * 2 > lim || M * ldo * 2 > lim || ldw > 65536) return MODA_ESHAPE;
    B256Args a;
    a.dz = (const unsigned short*)dz; a.ldz = ldz; a.x = (const unsigned short*)x; a.ldx = ldx;
    a.wb = (const unsigned short*)wb; a.ldw = ldw; a.dx = (unsigned short*)dx; a.ldo = ldo;
    a.gW = gW; a.ldg = ldg; a.gb = gb; a.M = M; a.streams = 0;
    return W == 256 ? b256_launch<256>(a, M, (hipStream_t)stream) : b256_launch<128>(a, M, (hipStream_t)stream);
}
