// Fused positional-encoding + NeRF MLP forward for gfx950 (MI355X).
//
// Replaces, for every sample of every ray, the reference chain
//   Embedding.forward (nnutils/nerf.py:35-75) -> evaluate_mlp's concat (nnutils/geom_utils.py:33-50)
//   -> NeRF.forward (nnutils/nerf.py:147-198)
// without ever writing the 63-wide embedding or any hidden activation to HBM.
//
// Design (see DESIGN.md "mlp_fused"):
//  * Activations are kept TRANSPOSED, H^T[feature][sample]: a wave owns 32*CB consecutive samples
//    (the MFMA column = lane & 31) and all W features of every layer live in its accumulator
//    registers.  A 32x32 accumulator tile is, unchanged (fp32) or after a pairwise bf16 pack, the B
//    operand of the next layer's MFMA (rows of the tile are the next layer's k index), so hidden
//    activations never touch LDS or HBM.
//  * Weights are the A operand.  The host packs them once per weight update into the exact
//    lane-linear order the MFMAs consume ("fragments" of 64 lanes x 16 B, see mlp_pack.py); the four
//    waves of a workgroup stream them through a kRing-deep LDS ring with LDS-DMA
//    (global_load_lds_dwordx4), counted vmcnt waits and one raw s_barrier per 16 KB chunk.
//  * Per-ray inputs (pose / environment codes, direction embedding) are folded into a per-row bias
//    by moda_linear_fwd beforehand, so the per-sample K of those layers is only the 63 PE features.
//  * Two instantiations per width: exact fp32 (v_mfma_f32_32x32x2_f32, bit-equivalent to an fmaf chain,
//    the parity mode) and bf16 operands / fp32 accumulate (v_mfma_f32_32x32x16_bf16, the throughput mode).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "moda_hip.h"
#include "moda_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef MODA_BF16_WAVES
#define MODA_BF16_WAVES 8          // waves per workgroup of the bf16 instantiations (4: one per SIMD, 8: two)
#endif
#ifndef MODA_BF16_CB128
#define MODA_BF16_CB128 2          // ... of the 128-wide bf16 / fp16 inference kernels on long uniform batches (dispatch); 1 = off
#endif
#ifndef MODA_BF16_CB
#define MODA_BF16_CB 1             // 32-sample column blocks per wave of the bf16 instantiations
#endif
#ifndef MODA_BF16_CB64
#define MODA_BF16_CB64 1           // column blocks per wave of the 64-wide bf16 instantiation
#endif
#ifndef MODA_BF16_WAVES64
#define MODA_BF16_WAVES64 16       // waves per workgroup of the 64-wide bf16 instantiation (resident weights, <= 128 VGPRs)
#endif
#ifndef MODA_RING
#define MODA_RING 6
#endif
#ifndef MODA_STAGGER
#define MODA_STAGGER 0             // 8-wave kernels: SIMD partners run one ring chunk apart (measured: no gain)
#endif
#ifndef MODA_RESIDENT
#define MODA_RESIDENT 1            // 64-wide bf16 nets keep their whole weight stream in LDS
#endif
#ifndef MODA_APIPE
#define MODA_APIPE 2               // A fragments read ahead of their MFMA
#endif
#ifndef MODA_HEAD_PREFETCH
#define MODA_HEAD_PREFETCH 0       // 1: UNI kernels load a tile's positions and row-bias rows one tile ahead.  It won 8 % on the
#endif                             // 5 x 64 kernel in round 1; now its 15 registers cost more (spills) than the wait: off is
                                   // 3.7 % faster on the 8 x 256 kernel and 2.5 % on the skin + warp kernel (A/B, r02)
#ifndef MODA_DMA_LEADERS
#define MODA_DMA_LEADERS 0         // 1: one wave per SIMD issues all LDS-DMA pieces of the weight stream (measured: 12 % slower)
#endif
#ifndef MODA_DMA_SPLIT
#define MODA_DMA_SPLIT 0           // 1: a wave issues its LDS-DMA pieces of one chunk half a chunk apart (measured: no gain)
#endif
#ifndef MODA_XLAYER
#define MODA_XLAYER 0              // 1: hidden layers hand their last output tile's epilogue to the next layer's first tile.
#endif                             // Measured: +1.6 % slower alone (the accumulator set kept across the layer boundary spills),
                                   // +2 % slower than off with the head prefetch off as well -- a negative result, kept as an option
#ifndef MODA_X3_EPI_PIPE
#define MODA_X3_EPI_PIPE 0         // split-bf16 kernels: software-pipelined tile epilogue (needs the second accumulator set)
#endif
#ifndef MODA_X3_WAVES256
#define MODA_X3_WAVES256 4         // waves per workgroup of the 256-wide split-bf16 kernel (2 x 128 activation registers)
#endif
#ifndef MODA_X3_WAVES128
#define MODA_X3_WAVES128 4         // ... of the 128-wide one (8 waves' PE stash + ring do not fit the 160 KB of LDS)
#endif
#ifndef MODA_X3_WAVES
#define MODA_X3_WAVES 8            // ... of the 64-wide one
#endif
#ifndef MODA_RING_SAFE
#define MODA_RING_SAFE 1           // refill the slot of the chunk before last (see Ring::kInFlight); 0: the racy schedule of rounds 1-3
#endif
#ifndef MODA_F16_TRACK
#define MODA_F16_TRACK 1           // fp16 kernels: keep the running maximum of the packed activations for the overflow report (0: timing A/B)
#endif
#ifndef MODA_AGPR_APIPE
#define MODA_AGPR_APIPE 4          // the AGPR kernel's fragment read-ahead (it has the registers for more than MODA_APIPE; a power of
                                   // two: with 3 or 6 the unrolled layer loses its constant register numbers).  2 -> 4: -1.0 % (A/B, one box)
#endif
#ifndef MODA_AGPR_PREFETCH
#define MODA_AGPR_PREFETCH 0       // the AGPR kernel loads a tile's positions and row-bias rows one tile ahead (MODA_HEAD_PREFETCH)
#endif
#ifndef MODA_AGPR_PREQ
#define MODA_AGPR_PREQ 0           // 1: read the next layer's first fragments ahead of the current layer's last epilogue.  Measured +1 % SLOWER (the chunk wait then blocks in front of the epilogue instead of behind it; layer 1 doubles): off
#endif
#ifndef MODA_AGPR_XLAYER
#define MODA_AGPR_XLAYER 0         // the AGPR kernel: hidden layers hand their last tile's epilogue to the next layer's first tile (see MODA_XLAYER)
#endif
#ifndef MODA_MLP_AGPR_DEFAULT
#define MODA_MLP_AGPR_DEFAULT 1    // 1: the 8 x 256 bf16 inference kernel takes the AGPR form by default (MODA_MLP_AGPR overrides)
#endif
#ifndef MODA_EPI_PIPE
#define MODA_EPI_PIPE 1            // the epilogue of an output tile is issued between the MFMAs of the next one
#endif
constexpr int kAPipe = MODA_APIPE;
constexpr int kFragBytes = 1024;   // one fragment: 64 lanes x 16 B

// Diagnostic build (-DMODA_STAMPS): wave 0 of every workgroup adds the s_memtime deltas of its phases into
// 16 slots of `stamps` (a buffer nothing else reads); never compiled into the shipped library.
#ifdef MODA_STAMPS
#define STAMP(slot)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        stamp_acc[slot] += t_ - stamp_prev;                                                      \
        stamp_prev = t_;                                                                         \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

struct MlpArgs {
    unsigned long long* stamps;
    const uint8_t* wstream;
    const float* bias;
    const float* xyz;
    const uint8_t* flip;
    const float* rb1;
    const float* rb5;
    const float* rbd;
    float* out;
    int M;                // samples (< 2^31, checked by the launcher)
    int R1, div1, Rd, divd;
    int out_stride;
    int out_tr_S;         // > 0: out is (M/S, out_stride, S)
    int nchunks;     // chunks in one pass over the network
    int nbias;       // floats in the bias block
    int n_pre, n_post;
    int n_out;
    int flags;
    int n_freq;
    float window[16];
    // WARP epilogue (moda_mlp_warp_fwd): the network's outputs are per-bone skinning logits, consumed in registers
    const float* qtab;        // per bone set: Gaussian-logit quadratic forms as fp32 MFMA A fragments (moda_dev.h)
    const f32x4* dqtab;       // per transform set: dual quaternions as bf16 hi/lo MFMA A fragments
    const float* pts_tf;      // points the blended transform is applied to (null: xyz)
    const float* cyc_ref;     // null, or (M,3): cyc_out[m] = |cyc_ref[m] - out[m]|
    float* cyc_out;
    int warp_S;               // samples per ray (a multiple of 32)
    int q_rps, dq_rps;        // rays per bone set (0: one set for all rays) / per transform set (>= 1)
    const int* run_start;     // null, or per set: the set whose table slot holds this set's data (moda_row_runs)
    int rows_at_runs;         // rb1 / rb5 rows are valid at run starts only: read row run_start[set] (MODA_MLP_ROWS_AT_RUNS)
    // training forward (moda_mlp_dump_fwd): every hidden layer's post-ReLU activations, fp32 row-major
    float* dump_h;            // (D, M, W): layer l at dump_h + l * M * W; null: nothing is dumped
    float* dump_dd;           // (M, W/2): the dir_encoding activations
    int dump_bf16;            // the dumps are stored as bf16 (same element offsets; MODA_MLP_DUMP_BF16 in moda_mlp_desc.reserved)
    // early ray termination (moda_mlp_live_fwd): 32-sample groups at or beyond n_live[ray] are not evaluated
    const int* n_live;        // null: every sample is evaluated
    int live_S;               // samples per ray (a multiple of 32)
    // fused compositing epilogue (moda_mlp_composite_fwd): the tile's [rgb, sigma] never leave the chip
    const float* comp_zv;     // (M) depths
    const float* comp_rd;     // (M / comp_S, 3) ray directions
    const float* comp_beta;   // (1)
    const float* comp_noise;  // (M) | null
    const float* comp_cyc;    // (M) | null
    int comp_S;               // samples per ray: 32, 64, 128 or 256 (divides the workgroup tile)
    CompOut comp_out;
    // fp16 mode: set to 1 (system-scope store) by any workgroup one of whose hidden activations left fp16's range
    int* ovf;                 // moda_mlp_desc.overflow; null: not reported
};

// ---------------------------------------------------------------------------------------------
// Weight ring: every wave of the workgroup consumes the same fragment sequence.
// ---------------------------------------------------------------------------------------------
// RESIDENT: the whole stream fits in LDS (the 64-wide bf16 nets: <= 80 KB).  It is loaded once per workgroup;
// afterwards there is no LDS-DMA and no barrier in the tile loop, so the waves drift apart and one wave's
// VALU / global-load phases overlap its neighbours' MFMAs.
// (The kernels that also STORE inside the tile loop -- the activation dumps of the training forward -- pay for it here:
// vmcnt counts loads, stores and LDS-DMA together in issue order, so the wait below also waits for every store issued
// since, and with them for younger chunks: less prefetch.  Widening the wait by the number of stores issued after the
// awaited chunk needs a run-time count; done with a compare chain the 8 x 256 kernel spilled (6x slower), done by polling
// the wave's own counter in IB_STS (s_getreg_b32: vm_cnt[3:0] bits 3:0, vm_cnt[5:4] bits 23:22 -- the decode is right,
// 20 after 20 loads) it was 3-5x slower.  Measured negative results; the dump kernels keep the strict wait.)
#ifndef MODA_DMA_SPREAD
#define MODA_DMA_SPREAD 1          // one-wave-per-SIMD kernels: a chunk's LDS-DMA pieces are issued one at a time, evenly over the chunk
#endif
#ifndef MODA_AGPR_REGSTAGE
#define MODA_AGPR_REGSTAGE 0       // 1: the AGPR kernel stages its weight chunks L2 -> registers -> LDS (buffer_load + ds_write_b128) instead of
                                   // LDS-DMA.  Measured (round 5, profiles/r05/coarse_kernel_cb2_agpr_ab.md): bit-identical and SLOWER, 12.63-12.74 ms
                                   // against 12.23 ms -- the LDS-DMA pieces are not what the one-wave-per-SIMD kernel loses its cycles to.
                                   // (building it needs -mllvm -pragma-unroll-threshold=131072: the staging code pushes the fully unrolled
                                   // layer past hipcc's default limit and the literal-AGPR operands stop being constants)
#endif
#ifndef MODA_STAGE_LOAD0
#define MODA_STAGE_LOAD0 1         // fragment (of the chunk being fetched) in front of which the first piece of the NEXT chunk is loaded
#endif
#ifndef MODA_STAGE_LOADSTEP
#define MODA_STAGE_LOADSTEP 2
#endif
#ifndef MODA_STAGE_STORE0
#define MODA_STAGE_STORE0 12       // ... and written to LDS
#endif
#ifndef MODA_STAGE_STORESTEP
#define MODA_STAGE_STORESTEP 1
#endif
#ifndef MODA_DMA_SPREAD_FIRST
#define MODA_DMA_SPREAD_FIRST 9    // fragments 9, 11, 13, 15 of a chunk: 0.978 -> 0.966 of the eight-wave form's time against 0, 4, 8, 12 (A/B, one box)
#endif
#ifndef MODA_DMA_SPREAD_STEP
#define MODA_DMA_SPREAD_STEP 2
#endif
template <int CHF, int NWAVES, bool RESIDENT, int kRing = MODA_RING, bool SPREAD = false>
struct Ring {
    __amdgpu_buffer_rsrc_t rsrc;   // packed stream (global), as a buffer resource
    uint8_t* lds;          // ring base (LDS)
    int nchunks;           // chunks per network pass (the stream is cyclic)
    int slot;              // ring slot of the chunk this wave is consuming
    int issue_slot;        // ring slot the next LDS-DMA chunk goes to
    int pos;               // stream position of the next chunk to issue, modulo nchunks
    int fcount;            // fragments already consumed from the current chunk
    int late_slot, late_pos;   // chunk whose second half of LDS-DMA pieces is still to be issued (kSplit)
    int lane, wave;
    bool leader;           // waves [0, NWAVES/2) run one chunk ahead of their SIMD partners [NWAVES/2, NWAVES)
    f32x4 st[4];           // kStage: this wave's pieces of the NEXT chunk on their way L2 -> registers -> LDS
    int nxt_slot, nxt_pos; // kStage: where that chunk goes / comes from

    static constexpr int kChunkBytes = CHF * kFragBytes;
    // LDS-DMA instructions per wave per chunk; a resident stream may be loaded by the first CHF of more than CHF waves
    // MODA_DMA_LEADERS: only the first half of the waves (one per SIMD) issue the LDS-DMA pieces of a streamed chunk, twice
    // as many each; their SIMD partners go straight to their MFMAs.  An LDS-DMA piece costs its wave 60-185 issue cycles
    // (4 waves x 2 column blocks pay the same 13 % for the stream as 8 x 1: it is issue time, not bytes); with both
    // partners issuing at the same point of the chunk the matrix pipe of their SIMD idles meanwhile.
    static constexpr int kWantLoaders = (!RESIDENT && MODA_DMA_LEADERS != 0 && NWAVES >= 8) ? NWAVES / 2 : NWAVES;
    static constexpr int kLoaders = (CHF >= kWantLoaders) ? kWantLoaders : CHF;
    static constexpr int kPerWave = CHF / kLoaders;
    static constexpr bool kStagger = (NWAVES == 8) && (MODA_STAGGER != 0);
    static constexpr bool kSplit = !RESIDENT && !kStagger && (kPerWave >= 2) && (MODA_DMA_SPLIT != 0) && !SPREAD;
    // SPREAD (the four-wave AGPR kernel): with one wave per SIMD nobody issues MFMAs while this wave issues an LDS-DMA piece (60-185
    // cycles each by the guide's table, against 24 cycles of shadow behind an MFMA), and a chunk's four pieces back to back behind
    // the barrier idle the matrix pipe for a fifth of the chunk.  Piece i goes out in front of fragment i * CHF / kPerWave instead.
    static constexpr bool kSpread = SPREAD && !RESIDENT && !kStagger && (kPerWave >= 2) && (MODA_DMA_SPREAD != 0);
    // ... and where: in the hidden layers a chunk IS one output tile (16 fragments), whose MFMAs 1..8 carry the previous tile's
    // epilogue pieces; the LDS-DMA pieces go behind the later, filler-free MFMAs (MODA_DMA_SPREAD_FIRST + i * MODA_DMA_SPREAD_STEP)
    static constexpr int kSpreadStep = (CHF == 16 && kPerWave == 4) ? MODA_DMA_SPREAD_STEP : CHF / kPerWave;
    static constexpr int kSpreadFirst = (CHF == 16 && kPerWave == 4) ? MODA_DMA_SPREAD_FIRST : 0;
    static_assert(!kSpread || (kSpreadFirst + (kPerWave - 1) * kSpreadStep < CHF), "pieces inside the chunk");
    // chunks that may still be in flight when the chunk a leader needs must have landed
    // The slot refilled at step s (after its barrier) must be one no wave can still be READING.  A wave issues its fragment
    // reads kAPipe ahead of the MFMAs, so when it arrives at the barrier of step s the ds_reads of chunk s-1's last fragments may
    // still sit in the LDS queue; refilling chunk s-1's slot right after that barrier (kInFlight = kRing - 2, rounds 1-3) races
    // the LDS-DMA against those reads.  The DMA's memory round trip usually wins the race for the reads by a wide margin -- but
    // not always when the LDS pipe is congested: the 128-wide training-forward kernel (4 waves, activation dumps transposed
    // through LDS) computed 32 ... 128 rows of a launch with a stale 1 KiB weight fragment in ~1 % of its launches (round 3's
    // "unexplained" gradient outliers; found by the soak tests of round 4, tools/dump_fwd_repro.py).  With MODA_RING_SAFE the slot
    // refilled is chunk s-2's, whose reads every wave has CONSUMED (a whole chunk of MFMAs ago): kRing - 3 chunks in flight,
    // which measured the same speed as kRing - 2 (a 5-deep ring was within 0.2 % of the 6-deep one).
    // (With the stagger the followers read chunk s-1 at step s, so the slot that is safe to refill lies one chunk further back
    //  still: kRing - 4 chunks in flight when both switches are on -- ADVICE r04; MODA_STAGGER is off by default.)
    // kStage (MODA_AGPR_REGSTAGE, off: an experiment of round 5 kept for its negative result): no LDS-DMA at all.  The idea: an
    // LDS-DMA piece costs the issuing wave ~60 cycles (guide: 60-185) of which an MFMA's shadow hides 24, and with one wave per
    // SIMD the rest is matrix-pipe idle time, four times per chunk.  So the four pieces of the NEXT chunk travel as plain 16-byte
    // buffer loads issued early in the current chunk and are written to the ring with ds_write_b128 late in it, each in a gap of
    // its own; the chunk barrier stays (with an lgkmcnt wait for the writes instead of a vmcnt wait for the DMA).  Two slots of
    // the ring are in use at a time (read / being written).  Result: same bits, 3-4 % slower than the DMA ring.
    static constexpr bool kStage = SPREAD && !RESIDENT && !kStagger && (kPerWave == 4) && (CHF == 16) && (MODA_AGPR_REGSTAGE != 0);
    static constexpr int kInFlight = (kStagger && MODA_RING_SAFE) ? kRing - 4 : ((kStagger || MODA_RING_SAFE) ? kRing - 3 : kRing - 2);
    static_assert(kInFlight >= 1, "the ring is too shallow for this refill schedule");
    static_assert(CHF % kLoaders == 0, "chunk fragments must divide over the loader waves");

    DEVINL void stage_load(int i) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, nxt_pos * kChunkBytes + (i * kLoaders + wave) * kFragBytes, 0);
        st[i] = __builtin_bit_cast(f32x4, v);
    }
    DEVINL void stage_store(int i) {
        *(f32x4*)(lds + nxt_slot * kChunkBytes + (i * kLoaders + wave) * kFragBytes + lane * 16) = st[i];
    }
    DEVINL void stage_at(int f) {       // the staging work that belongs in front of fragment f of the chunk being fetched
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (f == MODA_STAGE_LOAD0 + i * MODA_STAGE_LOADSTEP) stage_load(i);
            if (f == MODA_STAGE_STORE0 + i * MODA_STAGE_STORESTEP) stage_store(i);
        }
    }
    DEVINL void issue(int to_slot, int stream_pos, int i0 = 0, int i1 = kPerWave) {
        // buffer form: descriptor + scalar chunk/fragment offset in SGPRs, the per-lane 16 B offset in one VGPR that
        // never changes -- no vector address arithmetic per issue
        if (kLoaders < NWAVES && wave >= kLoaders) return;
        uint8_t* l = lds + to_slot * kChunkBytes + wave * kFragBytes;
        const int soff = stream_pos * kChunkBytes + wave * kFragBytes;
#pragma unroll
        for (int i = 0; i < kPerWave; ++i)
            if (i >= i0 && i < i1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)(l + i * kLoaders * kFragBytes),
                                                         16, lane * 16, soff + i * kLoaders * kFragBytes, 0, 0);
    }
    // One step of the workgroup-wide schedule: wait for the oldest outstanding chunk, rendezvous, refill the slot
    // that no wave reads any more.  With the stagger, at step s the leaders read chunk s and their SIMD partners
    // chunk s-1, so a wave in its VALU epilogue (end of a layer) sits beside a partner that is still issuing
    // MFMAs; the slot of chunk s-2 is the one refilled (with chunk s + kRing - 2).
    DEVINL void acquire() {
        if (RESIDENT) return;
        if (kStage) {       // every wave's pieces of this chunk are written (lgkmcnt) before anyone reads it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            nxt_slot = (slot + 1 == kRing) ? 0 : slot + 1;
            nxt_pos = pos;
            pos = (pos + 1 == nchunks) ? 0 : pos + 1;
            return;
        }
#ifdef MODA_ABL_NOBAR
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight * kPerWave) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kInFlight * kPerWave) : "memory");
#endif
#ifndef MODA_ABL_NODMA
        // MODA_DMA_SPLIT: only the first LDS-DMA piece goes out here; the rest follows half a chunk later (next()), so
        // that two pieces do not queue behind each other at the address unit while the wave should be issuing MFMAs
        issue(issue_slot, pos, 0, kSpread ? (kSpreadFirst == 0 ? 1 : 0) : (kSplit ? kPerWave / 2 : kPerWave));
#endif
        late_slot = issue_slot;
        late_pos = pos;
        issue_slot = (issue_slot + 1 == kRing) ? 0 : issue_slot + 1;
        pos = (pos + 1 == nchunks) ? 0 : pos + 1;
    }
    DEVINL void prime() {
        if (RESIDENT) {
            for (int c = 0; c < nchunks; ++c) issue(c, c);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            slot = 0;
            fcount = 0;
            return;
        }
        if (kStage) {       // chunk 0 through the registers into slot 0; chunk 1 follows during chunk 0's fragments
            nxt_slot = 0;
            nxt_pos = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_load(i);
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_store(i);
            pos = 1 % nchunks;
            slot = 0;
            fcount = 0;
            issue_slot = late_slot = late_pos = 0;
            return;
        }
        constexpr int kPrimed = kInFlight + 1;   // chunks issued before the first step
#pragma unroll
        for (int c = 0; c < kPrimed; ++c) issue(c, c % nchunks);
        pos = kPrimed % nchunks;
        issue_slot = kPrimed % kRing;
        slot = 0;
        fcount = 0;
        if (kStagger && !leader) acquire();   // followers sit out step 0
    }
    DEVINL void finish() {
        if (RESIDENT || kStage) return;
        if (kStagger && leader) acquire();    // leaders sit out the last step
        // every LDS-DMA this wave issued must land before the workgroup's LDS is released
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (fcount is a compile-time fact at every call site of the unrolled layers, so these tests fold away)
    DEVINL void issue_rest() {    // SPREAD: a wave that consumes no fragment of this chunk still owes its remaining pieces
        if (kStage) {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_load(i);
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_store(i);
            return;
        }
#ifndef MODA_ABL_NODMA
        if (kSpread) issue(late_slot, late_pos, kSpreadFirst == 0 ? 1 : 0, kPerWave);
#endif
    }
    DEVINL void issue_late() {
#ifndef MODA_ABL_NODMA
        issue(late_slot, late_pos, kPerWave / 2, kPerWave);
#endif
    }
    DEVINL void advance() {
        fcount = 0;
        slot = (slot + 1 == (RESIDENT ? nchunks : kRing)) ? 0 : slot + 1;
    }
    DEVINL f32x4 next() {
        if (fcount == 0) acquire();
#ifdef MODA_ABL_NOLDS
        f32x4 v = {1.f, 2.f, 3.f, 4.f};
        asm volatile("" : "+v"(v));
#else
        const f32x4 v = *(const f32x4*)(lds + slot * kChunkBytes + fcount * kFragBytes + lane * 16);
#endif
        if (kStage) stage_at(fcount);
        if (kSplit && fcount == CHF / 2) issue_late();     // before fragment CHF/2 is consumed
        if (!kStage && kSpread && fcount > 0 && fcount >= kSpreadFirst && (fcount - kSpreadFirst) % kSpreadStep == 0 &&
            (fcount - kSpreadFirst) / kSpreadStep < kPerWave) {
#ifndef MODA_ABL_NODMA
            issue(late_slot, late_pos, (fcount - kSpreadFirst) / kSpreadStep, (fcount - kSpreadFirst) / kSpreadStep + 1);
#endif
        }
        if (++fcount == CHF) advance();
        return v;
    }
    DEVINL void end_layer() {   // layers are padded to whole chunks
        if (fcount != 0) {
            if (kSplit && fcount <= CHF / 2) issue_late();   // ended before the half-way point: the second half is still owed
            if (kStage) {                                    // the next chunk's loads / writes that this short layer did not reach
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (MODA_STAGE_LOAD0 + i * MODA_STAGE_LOADSTEP >= fcount) stage_load(i);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (MODA_STAGE_STORE0 + i * MODA_STAGE_STORESTEP >= fcount) stage_store(i);
            }
            if (!kStage && kSpread) {                        // pieces of this chunk's refill that are still owed
#ifndef MODA_ABL_NODMA
#pragma unroll
                for (int i = (kSpreadFirst == 0 ? 1 : 0); i < kPerWave; ++i)
                    if (kSpreadFirst + i * kSpreadStep >= fcount) issue(late_slot, late_pos, i, i + 1);
#endif
            }
            advance();
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Precision policies.  A "fragment" always carries 16 B per lane of A operand.
//   F32 : 4 consecutive k-steps of v_mfma_f32_32x32x2_f32 (8 input features)
//   BF16: 1 v_mfma_f32_32x32x16_bf16 (16 input features)
//
// Positional-encoding slots.  The 64 (63 + one zero pad) embedding features are assigned to MFMA k
// positions so that the two lane halves differ only by a quarter-turn phase:
//   slot p < 30 : (k, c) = (p / 3, p % 3);  half 0 holds sin(2^k x_c), half 1 holds cos(2^k x_c)
//   slot 30     : half 0 holds x, half 1 holds y
//   slot 31     : half 0 holds z, half 1 holds the zero pad
// (reference feature order, nerf.py:58-72: f = 3 + 6k + 3*fn + c).  The packer puts the matching weight
// column under each slot, so any assignment is legal; this one costs ~5 VALU per slot.
// Slot p is element j = p % PE_ELEMS of fragment group g = p / PE_ELEMS.
// ---------------------------------------------------------------------------------------------
struct PrecF32 {
    static constexpr int SUBS = 4;       // fragments per 32-feature activation tile
    static constexpr int PEG = 8;        // fragments covering the 64 PE slots
    static constexpr int PE_ELEMS = 4;   // slots per fragment
    typedef f32x4 Frag;                  // what one step of the weight stream hands the MFMAs
    template <class R> static DEVINL Frag fetch(R& ring) { return ring.next(); }
    struct Act { f32x16 v; };
    struct Pe { float v[32]; };
    // activation tile as B operand: k-step s uses accumulator register s; lane half h holds row
    // (s&3) + 8(s>>2) + 4h of the tile (the MFMA C/D map), which is the k index the packer assumes.
    static DEVINL void mma_act(f32x16& acc, const f32x4& a, const Act& x, int sub) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x.v[4 * sub + j], acc, 0, 0, 0);
    }
    static DEVINL void mma_pe(f32x16& acc, const f32x4& a, const Pe& p, int g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], p.v[4 * g + j], acc, 0, 0, 0);
    }
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x.v[i] = relu ? fmaxf(acc[i], 0.f) : acc[i];
    }
    // one eighth of store_act (accumulator registers 2p, 2p+1): the epilogue is issued in pieces between MFMAs
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned& trk) {
        x.v[2 * p] = relu ? fmaxf(acc[2 * p], 0.f) : acc[2 * p];
        x.v[2 * p + 1] = relu ? fmaxf(acc[2 * p + 1], 0.f) : acc[2 * p + 1];
    }
    // a fresh (unspecified) value without an instruction: see fresh() in the kernel
    static DEVINL void fresh_act(Act& x) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "=v"(x.v[i]));
    }
    static DEVINL void note(unsigned&, float) {}
    // exact path: sincosf (<= 2 ulp) of the exactly scaled argument, as torch.sin/cos(freq * x) in the reference
    static DEVINL void encode(Pe& p, float x, float y, float z, int h, const float* win_lds) {
#pragma unroll
        for (int i = 0; i < 32; ++i) p.v[i] = 0.f;
#pragma nounroll
        for (int q = 0; q < 30; ++q) {
            const int k = q / 3;
            const int c = q - 3 * k;
            const float v = c == 0 ? x : (c == 1 ? y : z);
            float sn, cs;
            sincosf(ldexpf(v, k), &sn, &cs);
            const float val = win_lds[k] * (h ? cs : sn);
#pragma unroll
            for (int i = 0; i < 30; ++i) p.v[i] = (i == q) ? val : p.v[i];
        }
        p.v[30] = h ? y : x;
        p.v[31] = h ? 0.f : z;
    }
};

struct PrecBF16 {
    static constexpr int SUBS = 2;
    static constexpr int PEG = 4;
    static constexpr int PE_ELEMS = 8;
    typedef f32x4 Frag;
    template <class R> static DEVINL Frag fetch(R& ring) { return ring.next(); }
    struct Act { bf16x8 b[2]; };
    struct Pe { bf16x8 b[PEG]; };
    static DEVINL bf16x8 as_bf16(const f32x4& a) {
        union { f32x4 f; bf16x8 b; } u;
        u.f = a;
        return u.b;
    }
    static DEVINL void mma_act(f32x16& acc, const f32x4& a, const Act& x, int sub) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(a), x.b[sub], acc, 0, 0, 0);
    }
    static DEVINL void mma_pe(f32x16& acc, const f32x4& a, const Pe& p, int g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(a), p.b[g], acc, 0, 0, 0);
    }
    // two floats -> one dword of two bf16 (round-to-nearest-even): the 2-vector conversion selects v_cvt_pk_bf16_f32 and,
    // unlike an inline-asm statement, is visible to hipcc's hazard recogniser (MFMA result -> VALU read wait states)
    // and scheduler, so the activation epilogue can be placed in the shadow of later MFMAs.
    static DEVINL unsigned cvt_pk(float lo, float hi) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        union { bf16x2 b; unsigned u; } o;
        const f32x2 v = {lo, hi};
        o.b = __builtin_convertvector(v, bf16x2);
        return o.u;
    }
    // registers 8u..8u+7 of the accumulator, packed pairwise, are the B fragment of sub-step u:
    // element j of lane half h is row 16u + 8(j>>2) + 4h + (j&3) of the tile.  ReLU is applied on the
    // packed bf16 pairs as a signed 16-bit max with 0 (v_pk_max_i16): negative floats are negative ints.
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            union { u32x4 w; bf16x8 b; } o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                union { unsigned w; s16x2 s; } c;
                c.w = cvt_pk(acc[8 * u + 2 * q], acc[8 * u + 2 * q + 1]);
                if (relu) {
                    const s16x2 zero = {0, 0};
                    c.s = __builtin_elementwise_max(c.s, zero);
                }
                o.w[q] = c.w;
            }
            x.b[u] = o.b;
        }
    }
    // one eighth of store_act: accumulator registers 2p, 2p+1 -> dword (p & 3) of sub-step p >> 2
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned& trk) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        union { unsigned w; s16x2 s; } c;
        c.w = cvt_pk(acc[2 * p], acc[2 * p + 1]);
        if (relu) {
            const s16x2 zero = {0, 0};
            c.s = __builtin_elementwise_max(c.s, zero);
        }
        union { u32x4 w; bf16x8 b; } o;
        o.b = x.b[p >> 2];
        o.w[p & 3] = c.w;
        x.b[p >> 2] = o.b;
    }
    static DEVINL void fresh_act(Act& x) {
        asm volatile("" : "=v"(x.b[0]));
        asm volatile("" : "=v"(x.b[1]));
    }
    static DEVINL void note(unsigned&, float) {}
    // throughput path: hardware sine of the argument in revolutions, t = x / 2pi scaled exactly by 2^k
    static DEVINL void encode(Pe& p, float x, float y, float z, int h, const float* win_lds) {
        const float inv2pi = 0.15915494309189535f;
        const float t[3] = {x * inv2pi, y * inv2pi, z * inv2pi};
        const float phase = h ? 0.25f : 0.f;
        float v[32];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
            const int k = q / 3;
            const int c = q - 3 * k;
#ifdef MODA_ABL_NOPE   // timing-only ablation build: no sine
            v[q] = win_lds[k] * __builtin_fmaf(t[c], (float)(1 << k), phase);
#else
            const float rev = __builtin_amdgcn_fractf(__builtin_fmaf(t[c], (float)(1 << k), phase));
            v[q] = win_lds[k] * __builtin_amdgcn_sinf(rev);
#endif
        }
        v[30] = h ? y : x;
        v[31] = h ? 0.f : z;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int g = 0; g < PEG; ++g) {
            union { u32x4 w; bf16x8 b; } o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o.w[q] = cvt_pk(v[8 * g + 2 * q], v[8 * g + 2 * q + 1]);
            p.b[g] = o.b;
        }
    }
};

// bf16 with the ACTIVATIONS in the accumulator file (round 5).  A 4-wave workgroup runs one wave per SIMD, so a wave owns all 512
// registers of its lane: 256 architectural VGPRs + 256 AGPRs.  Two 32-sample column blocks per wave need 2 x 2 x 64 = 256
// activation registers (X and Y of both blocks) -- with them in VGPRs nothing else fits (round 3: -7 %, round 4: 156 B of spills
// and every AGPR filled with COPIES), because hipcc never uses an AGPR as an MFMA A / B operand on its own, although the hardware
// takes them there (gfx90a and later).  Values of register class "a" did not help either (round 5, first attempt: each dword born
// in an asm statement with an "=a" output and consumed through an "a" tuple operand): the allocator does not coalesce the dwords
// into their tuples -- one v_accvgpr_mov per dword -- and at 256 live AGPRs it spills them to VGPRs and back (1 724
// v_accvgpr_write, 720 B of scratch).  So the activation buffers are asm-OWNED, literally named registers:
//     X: block 0 a[0:63], block 1 a[64:127];   Y: block 0 a[128:191], block 1 a[192:255];   tile t at + 8 t, sub-step s at + 4 s;
//     dir_encoding's output goes where the buffer that does not hold the last hidden layer lies.
// An `Act` is just the tile's first register number; every use is an "i" operand spliced into the register text (the layer code
// is fully unrolled, so the numbers are constants by instruction selection).  A fragment is born in an asm statement
// (v_cvt_pk_bf16_f32 + v_pk_max_i16 in a VGPR temporary, v_accvgpr_write_b32) and read by an asm MFMA as its B operand: it never
// visits a VGPR again.  One 1 KiB weight fragment read from LDS then feeds TWO MFMAs and a ring chunk (one barrier) 32 instead of
// 16: half the LDS traffic and barriers per MFMA of the one-block kernel.
// hipcc knows nothing of these registers: it must not touch the AGPR file itself (no spills: audit the ISA for v_accvgpr_* outside
// ASMSTART / ASMEND, tools/agpr_audit.py), and the one "a255" clobber in settle() makes the kernel descriptor allocate all 256.
// It neither schedules nor pads what is inside an asm statement (cdna_hip_programming.md section 5.7); the wait states are this
// code's business:  (1) MFMA result -> the epilogue's v_cvt_pk: the software pipeline of `layer` issues the pieces of tile rt-1
// only after two MFMA rounds of tile rt (>= 3 x 32 cycles behind the producing MFMA; MFMAs and pieces are all `asm volatile`,
// which keeps their source order); at a layer's end, where the last tile is converted right behind its MFMAs, settle() pads 16
// states;  (2) v_accvgpr_write -> MFMA reading it: a written tile is first read a whole output tile of MFMAs later, and the
// layer-end pieces close with s_nop 1;  (3) accumulators read by compiler-generated code (the heads' outputs): settle();
// (4) an accumulator whose registers the compiler may REUSE right behind the chain's last MFMA statement (the sigma head: only
// row 0 is read) -- the hardware writes all 16 registers ~32 cycles after issue, into whatever lives there by then: settle()
// right behind the chain (found as non-repeatable outputs; with the builtin the hazard recogniser pads this WAW itself).
// debugging knobs (numbers, -DMODA_AGPR_PRE_NOP=7 ...): wait states ahead of an epilogue piece / behind it / ahead of an MFMA
#define MODA_STR2(x) #x
#define MODA_STR(x) MODA_STR2(x)
#ifdef MODA_AGPR_PRE_NOP
#define MODA_AGPR_PRE "s_nop " MODA_STR(MODA_AGPR_PRE_NOP) "\n\t"
#else
#define MODA_AGPR_PRE ""
#endif
#ifdef MODA_AGPR_POST_NOP
#define MODA_AGPR_POST "\n\ts_nop " MODA_STR(MODA_AGPR_POST_NOP)
#else
#define MODA_AGPR_POST ""
#endif
#ifdef MODA_AGPR_MFMA_NOP
#define MODA_AGPR_MFMA_PRE "s_nop " MODA_STR(MODA_AGPR_MFMA_NOP) "\n\t"
#else
#define MODA_AGPR_MFMA_PRE ""
#endif
struct PrecBF16A : PrecBF16 {
    struct Act { int base; };            // first of the tile's eight AGPRs
    static constexpr int kRegX = 0, kRegY = 128, kRegBlock = 64, kRegTile = 8;
    static DEVINL void mma_act(f32x16& acc, const f32x4& a, const Act& x, int sub) {
        asm volatile(MODA_AGPR_MFMA_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(x.base + 4 * sub), "i"(x.base + 4 * sub + 3));
    }
    static DEVINL void mma_pe(f32x16& acc, const f32x4& a, const Pe& p, int g) {
        asm volatile(MODA_AGPR_MFMA_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(p.b[g]));
    }
    // accumulator registers 2p, 2p+1 -> dword (p & 3) of sub-step p >> 2 (PrecBF16::store_piece), i.e. AGPR base + p
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned&) {
        float tmp;
        if (relu)
            asm volatile(MODA_AGPR_PRE "v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0\n\tv_accvgpr_write_b32 a[%c3], %0" MODA_AGPR_POST
                         : "=&v"(tmp) : "v"(acc[2 * p]), "v"(acc[2 * p + 1]), "i"(x.base + p));
        else
            asm volatile(MODA_AGPR_PRE "v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_accvgpr_write_b32 a[%c3], %0" MODA_AGPR_POST
                         : "=&v"(tmp) : "v"(acc[2 * p]), "v"(acc[2 * p + 1]), "i"(x.base + p));
    }
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
#pragma unroll
        for (int p = 0; p < 8; ++p) store_piece(x, acc, relu, p, trk);
        asm volatile("s_nop 1");
    }
    static DEVINL void fresh_act(Act&) {}
    // an accumulator written by an asm MFMA is about to be read by something other than the next MFMA of its chain
    // (the "a255" clobber: the kernel descriptor must allocate the whole AGPR file, see above)
    static DEVINL void settle(f32x16& acc) { asm volatile("s_nop 15" : "+v"(acc) : : "a255"); }
    // ahead of a chain's first MFMA: whatever compiler-generated VALU instruction wrote the accumulator last (both column blocks
    // start from the SAME bias rows, so hipcc loads them once and v_mov's them into the second tile -- right in front of the MFMA)
    // must be 2 wait states behind (VALU write -> MFMA operand read; the recogniser pads builtins, not asm).  The operands pin the
    // order: copies -> this statement -> the MFMA that consumes its output.
    static DEVINL void guard(f32x16& acc) { asm volatile("s_nop 1" : "+v"(acc)); }
};

// fp16 operands / fp32 accumulate (v_mfma_f32_32x32x16_f16: the bf16 MFMA's rate, 11 significand bits instead of 8) -- the
// parity-grade mode at throughput-mode speed, round 4.  Same fragment geometry, stream layout and encoding as PrecBF16; every
// operand of this path is O(1) (PE in [-1, 1], weights U(+-1/sqrt(fan_in)), post-ReLU activations), far inside fp16's range,
// but nothing here saturates silently: an operand that left fp16's range shows as a non-finite accumulator in the layer that
// consumes it (see note()), and the kernel raises MlpArgs::ovf.
struct PrecF16 {
    static constexpr int SUBS = 2;
    static constexpr int PEG = 4;
    static constexpr int PE_ELEMS = 8;
    typedef f32x4 Frag;
    template <class R> static DEVINL Frag fetch(R& ring) { return ring.next(); }
    struct Act { f16x8 b[2]; };
    struct Pe { f16x8 b[PEG]; };
    static DEVINL f16x8 as_f16(const f32x4& a) {
        union { f32x4 f; f16x8 b; } u;
        u.f = a;
        return u.b;
    }
    static DEVINL void mma_act(f32x16& acc, const f32x4& a, const Act& x, int sub) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16(a), x.b[sub], acc, 0, 0, 0);
    }
    static DEVINL void mma_pe(f32x16& acc, const f32x4& a, const Pe& p, int g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16(a), p.b[g], acc, 0, 0, 0);
    }
    // two floats -> one dword of two fp16, round-to-nearest-even (v_cvt_pk_f16_f32; a 2-vector conversion, visible to the
    // hazard recogniser like PrecBF16::cvt_pk); values beyond 65504 + half an ulp become inf, which `trk` reports
    static DEVINL unsigned cvt_pk(float lo, float hi) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        union { f16x2 b; unsigned u; } o;
        const f32x2 v = {lo, hi};
        o.b = __builtin_convertvector(v, f16x2);
        return o.u;
    }
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // ReLU on the packed pair as a signed 16-bit max with 0 (negative halves, -0 and sign-carrying NaNs included, are negative
    // integers)
    static DEVINL unsigned finish_pair(unsigned w, bool relu) {
        union { unsigned w; s16x2 s; } c;
        c.w = w;
        if (relu) {
            const s16x2 zero = {0, 0};
            c.s = __builtin_elementwise_max(c.s, zero);
        }
        return c.w;
    }
    // Overflow report.  An fp16 operand that is not finite (an activation that rounded to inf, an infinite weight) makes EVERY
    // accumulator row of its sample in the layer that consumes it inf or NaN (inf * w = +-inf, inf * 0 = NaN, inf - inf = NaN),
    // and every packed activation and every PE value is consumed by a later MFMA of the same lane column.  So one accumulator
    // register per output tile, looked at before it is packed, sees every overflow of the layer before: `trk` is the running
    // maximum of |acc[0]| as an integer (inf = 0x7f800000, NaNs above).  Two VALU operations per tile (an earlier version kept
    // the maximum of every packed pair: 8 per tile, +2 % on the 8 x 256 kernel and +6 % on the skin + warp kernel).
    static DEVINL void note(unsigned& trk, float v) {
#if MODA_F16_TRACK
        const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
        trk = b > trk ? b : trk;
#endif
    }
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
        note(trk, acc[0]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            union { u32x4 w; f16x8 b; } o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o.w[q] = finish_pair(cvt_pk(acc[8 * u + 2 * q], acc[8 * u + 2 * q + 1]), relu);
            x.b[u] = o.b;
        }
    }
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned& trk) {
        if (p == 0) note(trk, acc[0]);
        union { u32x4 w; f16x8 b; } o;
        o.b = x.b[p >> 2];
        o.w[p & 3] = finish_pair(cvt_pk(acc[2 * p], acc[2 * p + 1]), relu);
        x.b[p >> 2] = o.b;
    }
    static DEVINL void fresh_act(Act& x) {
        asm volatile("" : "=v"(x.b[0]));
        asm volatile("" : "=v"(x.b[1]));
    }
    // split heads: the rounded RESIDUALS of a post-ReLU tile against the fp16 values store_piece / store_act packed (hi + lo
    // carries 22 significand bits of the activation)
    static DEVINL unsigned lo_pair(float a0, float a1) {
        const float v0 = fmaxf(a0, 0.f), v1 = fmaxf(a1, 0.f);
        union { unsigned u; _Float16 h[2]; } hh;
        hh.u = cvt_pk(v0, v1);
        return cvt_pk(v0 - (float)hh.h[0], v1 - (float)hh.h[1]);
    }
    static DEVINL void store_piece_lo(Act& x, const f32x16& acc, int p) {
        union { u32x4 w; f16x8 b; } o;
        o.b = x.b[p >> 2];
        o.w[p & 3] = lo_pair(acc[2 * p], acc[2 * p + 1]);
        x.b[p >> 2] = o.b;
    }
    static DEVINL void store_act_lo(Act& x, const f32x16& acc) {
#pragma unroll
        for (int p = 0; p < 8; ++p) store_piece_lo(x, acc, p);
    }
    static DEVINL bool overflowed(unsigned trk) { return trk >= 0x7f800000u; }
    // hardware sine of the argument in revolutions (as PrecBF16::encode; its absolute error, ~1e-6, is far below fp16's 2^-12)
    static DEVINL void encode(Pe& p, float x, float y, float z, int h, const float* win_lds) {
        const float inv2pi = 0.15915494309189535f;
        const float t[3] = {x * inv2pi, y * inv2pi, z * inv2pi};
        const float phase = h ? 0.25f : 0.f;
        float v[32];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
            const int k = q / 3;
            const int c = q - 3 * k;
            const float rev = __builtin_amdgcn_fractf(__builtin_fmaf(t[c], (float)(1 << k), phase));
            v[q] = win_lds[k] * __builtin_amdgcn_sinf(rev);
        }
        v[30] = h ? y : x;
        v[31] = h ? 0.f : z;
#pragma unroll
        for (int g = 0; g < PEG; ++g) {
            union { u32x4 w; f16x8 b; } o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o.w[q] = cvt_pk(v[8 * g + 2 * q], v[8 * g + 2 * q + 1]);
            p.b[g] = o.b;
        }
    }
};

// the two one-MFMA-per-product precisions with 16-bit operands: they share every structural choice of the kernel
// The fp16 counterpart of PrecBF16A (the parity-grade mode's 8 x 256 kernel, split rgb head included): same register map -- the
// residual tiles of dir_encoding's output (actd_lo) take tiles 4..7 of the buffer that holds its output -- same wait-state
// rules.  The overflow probe of PrecF16::note reads an accumulator register: it rides inside the first epilogue piece's asm
// statement, where the tile has landed by construction (as compiler-generated code it could be scheduled right behind the MFMA).
struct PrecF16A : PrecF16 {
    struct Act { int base; };
    static constexpr int kRegX = 0, kRegY = 128, kRegBlock = 64, kRegTile = 8;
    static DEVINL void mma_act(f32x16& acc, const f32x4& a, const Act& x, int sub) {
        asm volatile(MODA_AGPR_MFMA_PRE "v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(x.base + 4 * sub), "i"(x.base + 4 * sub + 3));
    }
    static DEVINL void mma_pe(f32x16& acc, const f32x4& a, const Pe& p, int g) {
        asm volatile(MODA_AGPR_MFMA_PRE "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(p.b[g]));
    }
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned& trk) {
        float tmp;
#if MODA_F16_TRACK
        if (p == 0) {            // the overflow probe (PrecF16::note) on accumulator register 0, inside the statement
            unsigned t2;
            asm volatile("v_and_b32 %1, 0x7fffffff, %2\n\tv_max_u32 %0, %0, %1" : "+v"(trk), "=&v"(t2) : "v"(acc[0]));
        }
#endif
        if (relu)
            asm volatile(MODA_AGPR_PRE "v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0\n\tv_accvgpr_write_b32 a[%c3], %0" MODA_AGPR_POST
                         : "=&v"(tmp) : "v"(acc[2 * p]), "v"(acc[2 * p + 1]), "i"(x.base + p));
        else
            asm volatile(MODA_AGPR_PRE "v_cvt_pk_f16_f32 %0, %1, %2\n\tv_accvgpr_write_b32 a[%c3], %0" MODA_AGPR_POST
                         : "=&v"(tmp) : "v"(acc[2 * p]), "v"(acc[2 * p + 1]), "i"(x.base + p));
    }
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
#pragma unroll
        for (int p = 0; p < 8; ++p) store_piece(x, acc, relu, p, trk);
        asm volatile("s_nop 1");
    }
    // PrecF16::lo_pair in asm: v = max(a, 0); hi = fp16(v); lo = fp16(v - float(hi)), both halves of the pair
    static DEVINL void store_piece_lo(Act& x, const f32x16& acc, int p) {
        float t0, t1, th, ta;
        asm volatile("v_max_f32 %0, 0, %4\n\tv_max_f32 %1, 0, %5\n\tv_cvt_pk_f16_f32 %2, %0, %1\n\t"
                     "v_cvt_f32_f16 %3, %2\n\tv_sub_f32 %0, %0, %3\n\tv_lshrrev_b32 %3, 16, %2\n\tv_cvt_f32_f16 %3, %3\n\t"
                     "v_sub_f32 %1, %1, %3\n\tv_cvt_pk_f16_f32 %2, %0, %1\n\tv_accvgpr_write_b32 a[%c6], %2"
                     : "=&v"(t0), "=&v"(t1), "=&v"(th), "=&v"(ta) : "v"(acc[2 * p]), "v"(acc[2 * p + 1]), "i"(x.base + p));
    }
    static DEVINL void store_act_lo(Act& x, const f32x16& acc) {
#pragma unroll
        for (int p = 0; p < 8; ++p) store_piece_lo(x, acc, p);
        asm volatile("s_nop 1");
    }
    static DEVINL void fresh_act(Act&) {}
    static DEVINL void settle(f32x16& acc) { asm volatile("s_nop 15" : "+v"(acc) : : "a255"); }
    static DEVINL void guard(f32x16& acc) { asm volatile("s_nop 1" : "+v"(acc)); }
};

template <class P> constexpr bool kIs16 = std::is_same<P, PrecBF16>::value || std::is_same<P, PrecF16>::value || std::is_same<P, PrecBF16A>::value ||
                                          std::is_same<P, PrecF16A>::value;
template <class P> constexpr bool kAsmMfma = std::is_same<P, PrecBF16A>::value || std::is_same<P, PrecF16A>::value;   // MFMAs in asm statements: see PrecBF16A
template <class P> constexpr bool kIsF16 = std::is_same<P, PrecF16>::value || std::is_same<P, PrecF16A>::value;
template <class P> DEVINL void settle_acc(f32x16& acc) {
    if constexpr (kAsmMfma<P>) P::settle(acc);
}
template <class P> DEVINL void guard_acc(f32x16& acc) {
    if constexpr (kAsmMfma<P>) P::guard(acc);
}

// Split-bf16 ("bf16x3"): every operand is carried as bf16 hi + bf16 lo (lo = bf16(v - hi): 16 mantissa bits together) and a
// product is three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in fp32 -- operand error 2^-17 instead of 2^-9 at a third of the
// bf16 rate (the exact-fp32 MFMA runs at a sixteenth).  The parity-grade throughput mode: same fragment geometry as PrecBF16,
// every stream position holds a PAIR of fragments (hi, lo), the positional encoding is the exact sincosf of the fp32 kernels.
struct PrecBF16x3 {
    static constexpr int SUBS = 2;
    static constexpr int PEG = 4;
    static constexpr int PE_ELEMS = 8;
    struct Frag { f32x4 hi, lo; };
    template <class R> static DEVINL Frag fetch(R& ring) {
        Frag f;
        f.hi = ring.next();
        f.lo = ring.next();
        return f;
    }
    struct Act { bf16x8 hi[2], lo[2]; };
    struct Pe { bf16x8 hi[PEG], lo[PEG]; };
    static DEVINL void mma3(f32x16& acc, const Frag& a, const bf16x8& bhi, const bf16x8& blo) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PrecBF16::as_bf16(a.lo), bhi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PrecBF16::as_bf16(a.hi), blo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PrecBF16::as_bf16(a.hi), bhi, acc, 0, 0, 0);
    }
    static DEVINL void mma_act(f32x16& acc, const Frag& a, const Act& x, int sub) { mma3(acc, a, x.hi[sub], x.lo[sub]); }
    static DEVINL void mma_pe(f32x16& acc, const Frag& a, const Pe& p, int g) { mma3(acc, a, p.hi[g], p.lo[g]); }
    // two floats -> the dword of their bf16 roundings and the dword of the roundings of the residuals
    static DEVINL void split_pk(float v0, float v1, unsigned& hi, unsigned& lo) {
        hi = PrecBF16::cvt_pk(v0, v1);
        lo = PrecBF16::cvt_pk(v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xffff0000u));
    }
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    static DEVINL void store_piece(Act& x, const f32x16& acc, bool relu, int p, unsigned& trk) {
        const float v0 = relu ? fmaxf(acc[2 * p], 0.f) : acc[2 * p], v1 = relu ? fmaxf(acc[2 * p + 1], 0.f) : acc[2 * p + 1];
        unsigned hi, lo;
        split_pk(v0, v1, hi, lo);
        union { u32x4_ w; bf16x8 b; } o;
        o.b = x.hi[p >> 2];
        o.w[p & 3] = hi;
        x.hi[p >> 2] = o.b;
        o.b = x.lo[p >> 2];
        o.w[p & 3] = lo;
        x.lo[p >> 2] = o.b;
    }
    static DEVINL void store_act(Act& x, const f32x16& acc, bool relu, unsigned& trk) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            union { u32x4_ w; bf16x8 b; } oh, ol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v0 = relu ? fmaxf(acc[8 * u + 2 * q], 0.f) : acc[8 * u + 2 * q];
                const float v1 = relu ? fmaxf(acc[8 * u + 2 * q + 1], 0.f) : acc[8 * u + 2 * q + 1];
                unsigned hi, lo;
                split_pk(v0, v1, hi, lo);
                oh.w[q] = hi;
                ol.w[q] = lo;
            }
            x.hi[u] = oh.b;
            x.lo[u] = ol.b;
        }
    }
    static DEVINL void fresh_act(Act& x) {
        asm volatile("" : "=v"(x.hi[0]));
        asm volatile("" : "=v"(x.hi[1]));
        asm volatile("" : "=v"(x.lo[0]));
        asm volatile("" : "=v"(x.lo[1]));
    }
    static DEVINL void note(unsigned&, float) {}
    // sin(t + shift * pi/2) of an fp32 argument of any size this path meets (|t| up to a few thousand), absolute error <= 1.3e-7
    // (2 ulp of 1.0 -- the accuracy class of sincosf, which torch.sin / cos(freq * x) of the reference are): t / 2pi as an
    // exact product in two floats (fma), the whole revolutions and the nearest quarter taken off exactly, the remainder
    // (|.| <= 1/8 revolution) through degree-7 / degree-8 polynomials.  ~30 VALU operations where sincosf's general argument
    // reduction needs well over a hundred -- the encoding was ~10 % of this kernel, which has one wave per SIMD to hide it.
    static DEVINL float sin_quarter_shifted(float t, int shift) {
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
        const int qi = ((int)q + shift) & 3;
        const float r = (qi & 1) ? cs : sn;
        return (qi & 2) ? -r : r;
    }
    // slot q is element q & 7 of fragment q >> 3; lane half h holds the sine (0) or the cosine (1) of the slot's argument
    static DEVINL void encode(Pe& p, float x, float y, float z, int h, const float* win_lds) {
        float v[32];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
            const int k = q / 3;
            const int c = q - 3 * k;
            const float t = (c == 0 ? x : (c == 1 ? y : z)) * (float)(1 << k);        // exact scaling
            v[q] = win_lds[k] * sin_quarter_shifted(t, h);
        }
        v[30] = h ? y : x;
        v[31] = h ? 0.f : z;
#pragma unroll
        for (int g = 0; g < PEG; ++g) {
            union { u32x4_ w; bf16x8 b; } oh, ol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hi, lo;
                split_pk(v[8 * g + 2 * q], v[8 * g + 2 * q + 1], hi, lo);
                oh.w[q] = hi;
                ol.w[q] = lo;
            }
            p.hi[g] = oh.b;
            p.lo[g] = ol.b;
        }
    }
};

DEVINL float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }
DEVINL void keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }   // timing-only ablation builds

// ---------------------------------------------------------------------------------------------
// Kernel.  A layer is processed one 32-row OUTPUT tile at a time: initialise one accumulator tile from the bias,
// run its MFMAs over all input tiles, convert it (ReLU + bf16 pack) into the OTHER activation buffer -- so the VALU
// epilogue of tile rt has no register in common with the MFMAs of tile rt+1 and runs in their shadow, there is no
// in-place rewrite of a live B operand (no hazard fence), and only a few accumulator tiles are live at a time.
// Activations ping-pong between two register buffers X and Y; ENDY says which one the last hidden layer writes
// ((D-1) odd: Y), a compile-time fact so that every layer body exists once.
// ---------------------------------------------------------------------------------------------
// UNI: the host has checked that the 32 samples of every column block share their row-bias rows (per-ray rows with a
// multiple of 32 samples per ray, or a single row).  The kernel then has no global-memory source for an accumulator
// at all; with both sources in one body hipcc waits at their join with vmcnt(0), which also drains the ring's LDS-DMA
// prefetch -- once per output tile of every row-bias layer.
// HX (fp16 skin + warp kernel): the network's last two layers with split operands.  With fp16 operands the error of this network's
// outputs comes almost entirely from its LAST layers -- dir_encoding (64 -> 32) and the rgb head (32 -> B): few terms per sum and
// nothing downstream that averages (measured on the float64 restatement: rounding only the rgb head 2.5e-4 of the output scale,
// only the dir layer 2.2e-4, only layer 5 0.9e-4, only any of layers 1-4 2-4e-6; all of them 2.9e-4).  So the dir layer takes
// its WEIGHTS as fp16 hi + lo (two MFMAs per product; its input, the last hidden layer, stays single fp16) and the rgb head takes
// weights AND activations split (three MFMAs): 14 more MFMAs per 32-sample tile of the ~60 the network has, on a matrix pipe this
// kernel leaves half idle.  The stream holds those two layers' fragments as (hi, lo) pairs (MODA_MLP_F16_HEADS).
// HX on the 8 x 256 kernel (HXR): the rgb head alone -- 128 terms per colour with nothing behind them but a sigmoid, 6.6e-5 of
// the colour scale with single fp16 operands against 2.3e-6 with the head's weights and activations split; dir_encoding's epilogue
// packs the residuals beside the roundings (actd_lo), the head issues 3 MFMAs per fragment pair: +16 MFMAs on ~1050 per tile.
template <int W, typename P, int CB, int NWAVES, bool ENDY, bool UNI, bool WARP = false, int DUMP = 0, int RING = MODA_RING,
          bool COMP = false, bool HX = false>
__global__ __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(NWAVES / 4, NWAVES / 4)))
void mlp_fused_kernel(MlpArgs a) {
    static_assert(!HX || (kIsF16<P> && ((WARP && W == 64) || (!WARP && W == 256 && DUMP == 0 && !COMP))),
                  "split heads: the fp16 skin + warp kernel and the fp16 8 x 256 kernel");
    constexpr bool HXR = HX && !WARP;                 // 8 x 256: the rgb head alone is split
    static_assert(!COMP || (UNI && !WARP && DUMP == 0 && CB == 1 && std::is_same<P, PrecBF16>::value),
                  "the compositing epilogue is built for the bf16 UNI inference kernels");
    static_assert(!WARP || (UNI && !std::is_same<P, PrecF32>::value), "the warp epilogue is built for the bf16-geometry UNI kernels");
    // DUMP: 0 none, 1 fp32 activation dumps, 2 bf16 dumps by lane-pair swap, 3 / 4 bf16 dumps through a per-wave LDS transpose
    // of two tiles / one tile (a template parameter: with several store forms in one body the 8 x 256 kernel spills)
    static_assert(!DUMP || (std::is_same<P, PrecBF16>::value && !WARP), "activation dumps are built for the bf16 kernels");
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int NT = W / 32;                        // 32-row tiles of a hidden layer
    constexpr int NTD = (NT / 2 > 0) ? NT / 2 : 1;    // tiles of the dir_encoding layer (W/2 rows)
    constexpr int CHF = (W == 64) ? 8 : 16;           // fragments per ring chunk
    constexpr int TILE = NWAVES * 32 * CB;            // samples per workgroup iteration
    constexpr bool RESIDENT = (W == 64) && kIs16<P> && (MODA_RESIDENT != 0);
    using RingT = Ring<CHF, NWAVES, RESIDENT, RING, kAsmMfma<P>>;
    const int ring_chunks = RESIDENT ? a.nchunks : RING;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + ring_chunks * RingT::kChunkBytes);
    float* win_lds = bias_lds + a.nbias;
    // per-wave, per-column-block staging of the three row-bias rows a tile uses (layer 1, skip layer, dir layer)
    constexpr int RBW = 2 * W + NTD * 32;             // floats per slot
    float* rb_slots = win_lds + 16;
    // PE stash: the embedding fragments are parked in LDS between layer 1 and the skip layer (lane-linear 16 B)
    f32x4* pe_lds = (f32x4*)(rb_slots + NWAVES * CB * RBW) + threadIdx.x;
    constexpr int PE_VEC = sizeof(typename P::Pe) / 16;   // 16-byte pieces per lane and column block
    // DUMP == 3: per wave and column block, a 4 KB transpose buffer of the activation dump ([32 samples][64 features] bf16)
    constexpr int TBW = (DUMP == 4) ? 2048 : 4096;       // bytes per wave and column block
    constexpr int TBRS = (DUMP == 4) ? 64 : 128;         // bytes per sample row
    unsigned char* const tbuf = (unsigned char*)(rb_slots + NWAVES * CB * RBW) + sizeof(typename P::Pe) * CB * NTHREADS +
                                (threadIdx.x >> 6) * (CB * TBW);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int h = lane >> 5;

    for (int i = threadIdx.x; i < a.nbias; i += NTHREADS) bias_lds[i] = a.bias[i];
    if (threadIdx.x < 16) win_lds[threadIdx.x] = a.window[threadIdx.x];
    __syncthreads();

    RingT ring;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wstream, 0, a.nchunks * RingT::kChunkBytes, 0x00020000);
    ring.lds = smem;
    ring.nchunks = a.nchunks;
    ring.lane = lane;
    ring.wave = wave;
    ring.leader = wave < NWAVES / 2;
    ring.prime();
    // PREQ (the AGPR kernel): the first fragments of the NEXT layer are read from LDS while the current layer's last tile is
    // still being converted -- the chunk barrier, the LDS-DMA wait and the LDS latency at every layer boundary pass under the
    // epilogue instead of in front of the layer's first MFMA (one wave per SIMD: nobody else uses the matrix pipe meanwhile).
    constexpr int PREQ = (kAsmMfma<P> && MODA_AGPR_PREQ != 0) ? MODA_AGPR_APIPE : 0;
    typename P::Frag preq[PREQ > 0 ? PREQ : 1];
    auto prefetch_next = [&]() __attribute__((always_inline)) {
        if constexpr (PREQ > 0) {
#pragma unroll
            for (int d = 0; d < PREQ; ++d) preq[d] = P::fetch(ring);
        }
    };
    prefetch_next();

    const bool with_sigma = (a.flags & (MODA_MLP_WITH_SIGMA | MODA_MLP_SIGMA_ONLY)) != 0;
    const bool sigma_only = (a.flags & MODA_MLP_SIGMA_ONLY) != 0;
    const bool do_sigmoid = (a.flags & MODA_MLP_SIGMOID) != 0;
    const int nout_t = (a.n_out + 31) >> 5;
    const int n_mid = a.n_pre + 1 + a.n_post;         // layers 2..D

#ifdef MODA_STAMPS
    unsigned long long stamp_acc[16] = {0};
    unsigned long long stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    const int ntiles = (a.M + TILE - 1) / TILE;
    unsigned trk = 0u;   // PrecF16: running maximum of the packed activations (see the policy); unused otherwise
    // What a tile reads from global memory before it can start: its samples' positions and its row-bias rows.
    struct Head {
        float x, y, z;
        f32x4 rb[3];
    };
    // sample of column block cb of tile t (clamped: out-of-range columns compute on the last sample and store nothing)
    auto sample_at = [&](int t, int cb, bool& ok) __attribute__((always_inline)) {
        const int mm = t * TILE + wave * (32 * CB) + cb * 32 + col;
        ok = mm < a.M;
        return ok ? mm : a.M - 1;
    };
    // row of the per-row bias tables a sample uses: min(m / div, R - 1)
    auto row_at = [&](int t, int cb, int div, int R) __attribute__((always_inline)) {
        bool ok;
        const unsigned r = (unsigned)sample_at(t, cb, ok) / (unsigned)div;
        return (int)(r < (unsigned)R ? r : (unsigned)R - 1u);
    };
    auto load_head = [&](int t, int cb, bool uni) __attribute__((always_inline)) {
        Head hd;
        hd.rb[0] = hd.rb[1] = hd.rb[2] = f32x4{0.f, 0.f, 0.f, 0.f};
        int r1u = __builtin_amdgcn_readfirstlane(row_at(t, cb, a.div1, a.R1));
        const int rdu = __builtin_amdgcn_readfirstlane(row_at(t, cb, a.divd, a.Rd));
        if constexpr (WARP) {
            // per-set code rows folded only at the first row of a run of identical sets (moda_fold_rows with run_start): read there
            if (a.rows_at_runs) r1u = a.run_start[r1u];
        }
#ifndef MODA_ABL_NOHEADLOAD
        if (uni) {
            if (lane < W / 4) {
                hd.rb[0] = *(const f32x4*)(a.rb1 + (long long)r1u * W + 4 * lane);
                hd.rb[1] = *(const f32x4*)(a.rb5 + (long long)r1u * W + 4 * lane);
            }
            if (lane < NTD * 8 && (a.flags & MODA_MLP_SIGMA_ONLY) == 0)
                hd.rb[2] = *(const f32x4*)(a.rbd + (long long)rdu * (NTD * 32) + 4 * lane);
        }
#endif
        bool ok;
        const long long mm = sample_at(t, cb, ok);
#ifdef MODA_ABL_NOHEADLOAD   // timing-only ablation build: no global loads at the head of a tile
        hd.x = 0.001f * (float)(mm & 1023);
        hd.y = 0.002f * (float)(lane);
        hd.z = 0.3f;
#else
        hd.x = a.xyz[mm * 3 + 0];
        hd.y = a.xyz[mm * 3 + 1];
        hd.z = a.xyz[mm * 3 + 2];
#endif
        if (a.flip != nullptr && a.flip[mm]) hd.x = -hd.x;
        return hd;
    };
    // UNI kernels request a tile's head one tile ahead (after the encoding of the current tile has consumed the
    // registers): the loads then have a whole tile to land in, instead of being waited for at the head of their own tile
    // (MODA_ABL_NOHEADLOAD: that wait is 8 % of the 5x64 kernel)
    // (not in the dump kernels: they are short of registers -- the 8 x 256 one spilled 84 dwords per lane, and a scratch
    //  reload waits on vmcnt like everything else, i.e. for every dump store in flight)
    constexpr bool PREFETCH = UNI && ((MODA_HEAD_PREFETCH != 0) || (kAsmMfma<P> && MODA_AGPR_PREFETCH != 0)) && (DUMP == 0);
    Head head[CB];
    if (PREFETCH) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) head[cb] = load_head(blockIdx.x, cb, true);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        STAMP(15);   // loop overhead / previous tile's tail
        if (UNI && a.n_live != nullptr) {
            // early ray termination: this wave's 32-sample groups all lie beyond their rays' live prefix -> nothing to
            // compute.  The wave still takes its part in the workgroup's weight ring (one wait + barrier + LDS-DMA issue
            // per chunk), which every wave must walk in step; its matrix / LDS-read slots go to its SIMD partner.
            bool any_live = false;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int m_first = tile * TILE + wave * (32 * CB) + cb * 32;
                if (m_first < a.M) {
                    const int ray = m_first / a.live_S;
                    any_live = any_live || (m_first - ray * a.live_S) < a.n_live[ray];
                }
            }
            if (!__builtin_amdgcn_readfirstlane((int)any_live)) {
                for (int c = 0; c < a.nchunks; ++c) {
                    ring.acquire();
                    ring.issue_rest();
                    ring.advance();
                }
                if (PREFETCH) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) head[cb] = load_head(tile + gridDim.x, cb, true);
                }
                continue;
            }
        }
        auto sample_of = [&](int cb, bool& ok) __attribute__((always_inline)) { return sample_at(tile, cb, ok); };
        auto row_of = [&](int cb, int div, int R) __attribute__((always_inline)) { return row_at(tile, cb, div, R); };
        // ---- stage this tile's row-bias rows in the wave's LDS slot when all 32 samples of a column block share
        //      them (a ray's samples are consecutive: always, once S is a multiple of 32) ------------------------------
        bool rb_uni[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (UNI) {
                rb_uni[cb] = true;
            } else {
                const int r1 = row_of(cb, a.div1, a.R1), rd = row_of(cb, a.divd, a.Rd);
                const int r1u = __builtin_amdgcn_readfirstlane(r1), rdu = __builtin_amdgcn_readfirstlane(rd);
                rb_uni[cb] = __builtin_amdgcn_ballot_w64(r1 != r1u || rd != rdu) == 0ull;
            }
            if (!PREFETCH) head[cb] = load_head(tile, cb, rb_uni[cb]);
        }
        typename P::Pe pe[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) P::encode(pe[cb], head[cb].x, head[cb].y, head[cb].z, h, win_lds);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float* slot = rb_slots + (wave * CB + cb) * RBW;
            if (rb_uni[cb]) {
                if (lane < W / 4) {
                    *(f32x4*)(slot + 4 * lane) = head[cb].rb[0];
                    *(f32x4*)(slot + W + 4 * lane) = head[cb].rb[1];
                }
                if (lane < NTD * 8) *(f32x4*)(slot + 2 * W + 4 * lane) = head[cb].rb[2];
            }
        }
        if (PREFETCH) {   // the next tile's head (indices past the end are clamped to valid memory; nothing is stored for them)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) head[cb] = load_head(tile + gridDim.x, cb, true);
        }
        // fused compositing: the depths this lane's sample needs are requested now and waited for at the end of the tile (loaded
        // there, their round trip -- and the transcendental chain behind it -- ran with the matrix pipe idle on all waves at once)
        float comp_z = 0.f, comp_delta = 0.f, comp_noise_v = 0.f, comp_ib = 0.f;
        if constexpr (COMP) {
            bool okc;
            const int mmc = sample_of(0, okc);
            const int nc = mmc / a.comp_S, sxc = mmc - nc * a.comp_S;
            comp_z = a.comp_zv[mmc];
            comp_delta = (sxc + 1 < a.comp_S ? a.comp_zv[mmc + 1] - comp_z : 1e10f) * comp_dnorm(a.comp_rd, nc);   // :183-191
            comp_noise_v = a.comp_noise ? a.comp_noise[mmc] : 0.f;
            comp_ib = comp_ibeta(a.comp_beta);
        }
        STAMP(0);    // xyz load + positional encoding + row-bias staging

        // Both activation buffers are (re)written by the layers below before they are read, but which layer writes
        // which one depends on run-time layer counts, so to the compiler the previous tile's contents look live: it
        // carried all 2 x 64 registers around the tile loop (and the head accumulators with them: 112 phi registers,
        // 34 copies and 7 scratch round trips per tile, 0.9 GB of scratch writes per launch).  An empty asm that
        // "defines" each register ends that liveness at no cost.
        typename P::Act actX[CB][NT], actY[CB][NT];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                P::fresh_act(actX[cb][t]);
                P::fresh_act(actY[cb][t]);
                if constexpr (kAsmMfma<P>) {        // asm-owned AGPR buffers: an Act is its first register number
                    actX[cb][t].base = P::kRegX + cb * P::kRegBlock + t * P::kRegTile;
                    actY[cb][t].base = P::kRegY + cb * P::kRegBlock + t * P::kRegTile;
                }
            }

        // ---- accumulator initialisers: lane (col, h) register i holds row (i&3) + 8(i>>2) + 4h ----
        auto init_glob = [&](f32x16& c, const float* rb, int row, int ld, int rt) __attribute__((always_inline)) {
            const float* p = rb + (long long)row * ld + 32 * rt + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(p + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) c[4 * q + i] = v[i];
            }
        };
        // the source is addressed as LDS explicitly: with a generic pointer hipcc merges this with init_glob below into
        // FLAT loads through a selected pointer, and a FLAT load is waited for with vmcnt(0) -- a full drain of the
        // ring's LDS-DMA prefetch at every row-bias layer
        typedef const f32x4 __attribute__((address_space(3))) lds_f32x4;
        auto init_lds = [&](f32x16& c, const float* row, int rt) __attribute__((always_inline)) {
            const float* p = row + 32 * rt + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(lds_f32x4*)(p + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) c[4 * q + i] = v[i];
            }
        };
        // which: 0 layer 1, 1 skip layer, 2 dir layer
        auto init_rowbias = [&](f32x16& c, int cb, int which, int rt) __attribute__((always_inline)) {
            if (UNI || rb_uni[cb]) {
                init_lds(c, rb_slots + (wave * CB + cb) * RBW + (which == 0 ? 0 : (which == 1 ? W : 2 * W)), rt);
            } else if (which == 2) {
                init_glob(c, a.rbd, row_of(cb, a.divd, a.Rd), NTD * 32, rt);
            } else {
                init_glob(c, which == 0 ? a.rb1 : a.rb5, row_of(cb, a.div1, a.R1), W, rt);
            }
        };

        // ---- one layer, output tile by output tile.  Fragment order per tile: [PE groups] then [(t, s) over the
        //      source tiles]; the A fragments are read kAPipe ahead of the MFMA that consumes them. ------------------
        // INIT: -1 plain bias at bias_lds[boff], else the row-bias kind.  dst tiles receive act(acc).
        // DUMP (the training forward): a finished 32-row output tile is also written to dptr (fp32, [sample][dld features]).
        // Registers 4q..4q+3 of the accumulator are four CONSECUTIVE rows 8q + 4h .. + 3 of the tile for this lane's sample,
        // so a lane stores float4s straight from its accumulator -- no transpose through LDS.
        auto dump_quad = [&](float* dptr, int dld, int cb, int rt, int q, const f32x16& acc, bool relu) __attribute__((always_inline)) {
            bool ok;
            const long long mm = sample_of(cb, ok);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu ? fmaxf(acc[4 * q + i], 0.f) : acc[4 * q + i];
            const unsigned voff = ((unsigned)mm * (unsigned)dld + 4u * (unsigned)h) * 4u;      // see dump_pair
            if (ok) *(f32x4*)((char*)dptr + (size_t)voff + (size_t)((32 * rt + 8 * q) * 4)) = v;
        };
        // bf16 dumps: two quads at a time.  Lane half 0 holds rows 8q .. 8q+3 of both quads, half 1 rows 8q+4 .. 8q+7; one
        // v_permlane32_swap per dword hands half 0 the whole of quad 2qp (rows 16qp .. +7) and half 1 the whole of quad
        // 2qp+1, so a lane stores 16 contiguous bytes -- half the store instructions (this epilogue is bound by their issue:
        // a 64-lane store here touches 32 different 512-byte rows whatever its width).
        auto dump_pair = [&](float* dptr, int dld, int cb, int rt, int qp, const typename P::Act& x) __attribute__((always_inline)) {
            if constexpr (std::is_same<P, PrecBF16>::value) {
                bool ok;
                const long long mm = sample_of(cb, ok);
                // the tile's packed bf16 B-operand registers ARE the dump (ReLU applied, rounded): fragment qp holds quad 2qp in
                // its dwords 0-1 and quad 2qp+1 in dwords 2-3 -- no second conversion, no copy of the accumulator kept alive
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                union { u32x4_ w; bf16x8 b; } o;
                o.b = x.b[qp];
                const auto r0 = __builtin_amdgcn_permlane32_swap(o.w[0], o.w[2], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(o.w[1], o.w[3], false, false);
                // address = uniform layer base (SGPRs) + one 32-bit per-lane byte offset that is the same for every layer of
                // this width + an immediate (64-bit per-lane pointers were hoisted out of the tile loop, one pair per dumped
                // layer, and spilled).  The entry checks M * dld * 2 < 2^32.
                const unsigned voff = ((unsigned)mm * (unsigned)dld + 8u * (unsigned)h) * 2u;
                if (ok) *(uint4*)((char*)dptr + (size_t)voff + (size_t)((32 * rt + 16 * qp) * 2)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
        };
        // DUMP == 3, bf16 dumps through LDS: the quads of one or two finished output tiles are parked in the wave's buffer as
        // rows [sample][features] (16-byte chunks XOR-swizzled by the row), then every lane stores 16 bytes of one row: a
        // 64-lane store writes whole 64- or 128-byte pieces of 16 or 8 rows instead of 16 bytes of each of 32 rows.  (The
        // dump is bound by what its stores cost the L2 / memory side, not by the kernel's arithmetic: 0.9 ms with the
        // scattered stores against 0.2 ms for the same network without the dump.)
        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        typedef u32x2v __attribute__((address_space(3))) lds_uint2;
        typedef const u32x4v __attribute__((address_space(3))) lds_uint4;
        auto tb_put = [&](int cb, int slot, int qd, const typename P::Act& x) __attribute__((always_inline)) {
            if constexpr (std::is_same<P, PrecBF16>::value) {
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                union { u32x4_ w; bf16x8 b; } o;
                o.b = x.b[qd >> 1];
                const int ch = slot * 4 + qd;
                const int sw = (DUMP == 4) ? ((col >> 1) & 3) : (col & 7);
                *(lds_uint2*)(tbuf + cb * TBW + col * TBRS + 16 * (ch ^ sw) + 8 * h) = u32x2v{o.w[2 * (qd & 1)], o.w[2 * (qd & 1) + 1]};
            }
        };
        auto tb_flush = [&](float* dptr, int dld, int cb, int rt_first, const int ntl) __attribute__((always_inline)) {
            const int m_first = tile * TILE + wave * (32 * CB) + cb * 32;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                if (ps >= 2 * ntl) continue;
                const int row = (ntl == 2 ? (lane >> 3) + 8 * ps : (lane >> 2) + 16 * ps);
                const int ch = ntl == 2 ? (lane & 7) : (lane & 3);
                const int sw = (DUMP == 4) ? ((row >> 1) & 3) : (row & 7);
                const u32x4v v = *(lds_uint4*)(tbuf + cb * TBW + row * TBRS + 16 * (ch ^ sw));
                const int mm = m_first + row;
                const unsigned voff = ((unsigned)mm * (unsigned)dld + 8u * (unsigned)ch) * 2u;    // see dump_pair
                if (mm < a.M) *(u32x4v*)((char*)dptr + (size_t)voff + (size_t)(32 * rt_first * 2)) = v;
                }
        };
        // The two accumulator sets of the pipelined layers live across layers: with XL a hidden layer leaves its LAST output
        // tile unconverted in set 1 (defer_out) and the next hidden layer converts it piecewise between the MFMAs of its
        // first output tile (pend_in) -- into the last tile of its own source buffer, which those MFMAs read last.  Without
        // it every layer ends with one tile's epilogue (MFMA result latency + 16 VALU) while the matrix pipe idles on all
        // waves at once: ~3 % of the 8 x 256 kernel by its phase stamps.
        constexpr bool XL = (((MODA_XLAYER != 0) && std::is_same<P, PrecBF16>::value) || ((MODA_AGPR_XLAYER != 0) && kAsmMfma<P>)) &&
                            (DUMP == 0) && (W >= 128) && (NT % 2 == 0);
        f32x16 cacc[2][CB];
        typename P::Act actd_lo[HXR ? CB : 1][HXR ? NTD : 1];     // split heads (8 x 256): residuals of dir_encoding's output
        auto layer = [&](auto& src, auto& dst, auto ntout_c, auto ntin_c, const bool with_pe, const bool with_act,
                         const int init_kind, const int boff, const bool relu, float* dptr = nullptr,
                         int dld = 0, const bool pend_in = false, const bool defer_out = false,
                         const bool split_out = false) __attribute__((always_inline)) {
            constexpr int NTO = decltype(ntout_c)::value;
            constexpr int NTI = decltype(ntin_c)::value;
            constexpr int PEGc = P::PEG;
            const int fpt = (with_pe ? PEGc : 0) + (with_act ? NTI * P::SUBS : 0);   // fragments per output tile
            const int NF = NTO * fpt;                                                 // fragments of this layer, in stream order
            constexpr int AP = kAsmMfma<P> ? MODA_AGPR_APIPE : kAPipe;     // fragments read ahead of their MFMA
            typename P::Frag q[AP];
#pragma unroll
            for (int d = 0; d < AP; ++d) {
                if constexpr (PREQ > 0) q[d] = preq[d];
                else if (d < NF) q[d] = P::fetch(ring);
            }
            // (bf16 kernels only: the fp32 parity kernels keep 2 x 128 activation registers and have no room for a
            // second accumulator set -- pipelined, their allocation collapsed into AGPR copies and scratch, 3x slower)
            if constexpr ((MODA_EPI_PIPE != 0) && (kIs16<P> ||
                                                   (std::is_same<P, PrecBF16x3>::value && (MODA_X3_EPI_PIPE != 0)))) {
            // Software pipeline over the output tiles, two accumulator sets in ping-pong: while tile rt accumulates, the
            // epilogue of tile rt-1 (ReLU + pack into dst, 8 pieces) is issued piecewise between its MFMAs and, once
            // that set is free again, the bias of tile rt+1 is read into it.  Written sequentially (one accumulator,
            // epilogue after the last MFMA) every tile boundary costs the MFMA result latency + 16 VALU + an LDS round
            // trip with the matrix pipe idle -- on all waves at once, since the ring barrier keeps them in step.
            auto& c = cacc;
            auto init_acc = [&](f32x16& acc, int cb, int rt) __attribute__((always_inline)) {
                if (init_kind < 0) init_lds(acc, bias_lds + boff, rt);
                else init_rowbias(acc, cb, init_kind, rt);
            };
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) init_acc(c[0][cb], cb, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) guard_acc<P>(c[0][cb]);
#pragma unroll
            for (int rt = 0; rt < NTO; ++rt) {
                const int cur = rt & 1, oth = cur ^ 1;
                // after MFMA j of this tile: pieces [p0(j), p0(j+1)) of the previous tile's epilogue, spread over MFMAs
                // 1 .. fpt-2 (the first leaves room for the previous tile's last MFMA to retire); then the next bias
                // (cbs: the column blocks whose pieces go out here.  With several blocks per wave each block's pieces follow ITS MFMA --
                //  M(cb0), pieces(cb0), M(cb1), pieces(cb1) -- so that every MFMA has a few fillers in its shadow; behind a back-to-back
                //  pair the whole step's fillers queued in the second MFMA's 32 cycles and overflowed them: the one-wave-per-SIMD AGPR
                //  kernel has no partner wave to cover that)
                auto after = [&](int j, int cb_lo = 0, int cb_hi = CB) __attribute__((always_inline)) {
                    const int span = fpt - 2 < 1 ? 1 : (fpt - 2 > 8 ? 8 : fpt - 2);   // MFMAs 1 .. span carry the pieces
                    const int lo = j < 1 ? 0 : ((j - 1) * 8 + span - 1) / span;
                    const int hi = j < 1 ? 0 : (j >= span ? 8 : (j * 8 + span - 1) / span);
                    if (rt == 0 && XL) {
                        if (pend_in) {       // the previous hidden layer's last tile (always ReLU), held in set 1
#pragma unroll
                            for (int p = 0; p < 8; ++p)
                                if (p >= lo && p < hi) {
#pragma unroll
                                    for (int cb = 0; cb < CB; ++cb)
                                        if (cb >= cb_lo && cb < cb_hi) P::store_piece(src[cb][NTI - 1], c[oth][cb], true, p, trk);
                                }
                        }
                    }
                    if (rt > 0) {
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if (p >= lo && p < hi) {
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb) {
                                    if (cb < cb_lo || cb >= cb_hi) continue;
                                    P::store_piece(dst[cb][rt - 1], c[oth][cb], relu, p, trk);
                                    if constexpr (HXR) {
                                        if (split_out) P::store_piece_lo(actd_lo[cb][(rt - 1) % NTD], c[oth][cb], p);
                                    }
                                    if (DUMP && dptr != nullptr) {
                                        if (DUMP >= 3) {
                                            constexpr int G = (DUMP == 3 && NTO % 2 == 0) ? 2 : 1;      // tiles per flush
                                            if (p & 1) tb_put(cb, (rt - 1) % G, p >> 1, dst[cb][rt - 1]);
                                            if (p == 7 && (rt - 1) % G == G - 1) tb_flush(dptr, dld, cb, rt - G, G);
                                        } else if (DUMP == 2) {
                                            if ((p & 3) == 3) dump_pair(dptr, dld, cb, rt - 1, p >> 2, dst[cb][rt - 1]);
                                        } else if (p & 1) {
                                            dump_quad(dptr, dld, cb, rt - 1, p >> 1, c[oth][cb], relu);
                                        }
                                    }
                                }
                            }
                    }
                    // the other accumulator set is free once its last piece is out: read the next tile's bias into it
                    // as early as that, so that the LDS latency passes under the remaining MFMAs of this tile
                    const int jinit = span + 1 < fpt - 1 ? span + 1 : fpt - 1;
                    if (j == jinit && rt + 1 < NTO && cb_hi == CB) {
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) init_acc(c[oth][cb], cb, rt + 1);
                    }
                };
                constexpr bool PER_CB = kAsmMfma<P> && CB > 1;      // fillers behind each block's own MFMA
                if (with_pe) {
#pragma unroll
                    for (int g = 0; g < PEGc; ++g) {
                        const int idx = rt * fpt + g;
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) {
                            P::mma_pe(c[cur][cb], q[idx % AP], pe[cb], g);
                            if (PER_CB && cb + 1 < CB) after(g, cb, cb + 1);
                        }
                        if (idx + AP < NF) q[idx % AP] = P::fetch(ring);
                        if (PER_CB) after(g, CB - 1, CB); else after(g);
                    }
                }
                if (with_act) {
#pragma unroll
                    for (int t = 0; t < NTI; ++t)
#pragma unroll
                        for (int sb = 0; sb < P::SUBS; ++sb) {
                            const int j = (with_pe ? PEGc : 0) + t * P::SUBS + sb;
                            const int idx = rt * fpt + j;
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) {
                                P::mma_act(c[cur][cb], q[idx % AP], src[cb][t], sb);
                                if (PER_CB && cb + 1 < CB) after(j, cb, cb + 1);
                            }
                            if (idx + AP < NF) q[idx % AP] = P::fetch(ring);
                            if (PER_CB) after(j, CB - 1, CB); else after(j);
                        }
                }
            }
            if constexpr (PREQ > 0) {                 // every fragment of this layer has been read: on to the next layer's chunk
                ring.end_layer();
                prefetch_next();
            }
            if (!(XL && defer_out)) {
            settle_acc<P>(c[(NTO - 1) & 1][0]);        // (asm MFMAs: the last tile is converted right behind its MFMAs)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                P::store_act(dst[cb][NTO - 1], c[(NTO - 1) & 1][cb], relu, trk);
                if constexpr (HXR) {
                    if (split_out) P::store_act_lo(actd_lo[cb][(NTO - 1) % NTD], c[(NTO - 1) & 1][cb]);
                }
                if (DUMP && dptr != nullptr) {
                    if (DUMP >= 3) {
                        constexpr int G = (DUMP == 3 && NTO % 2 == 0) ? 2 : 1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) tb_put(cb, (NTO - 1) % G, q, dst[cb][NTO - 1]);
                        tb_flush(dptr, dld, cb, NTO - G, G);
                    } else if (DUMP == 2) {
                        dump_pair(dptr, dld, cb, NTO - 1, 0, dst[cb][NTO - 1]);
                        dump_pair(dptr, dld, cb, NTO - 1, 1, dst[cb][NTO - 1]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) dump_quad(dptr, dld, cb, NTO - 1, q, c[(NTO - 1) & 1][cb], relu);
                    }
                }
            }
            }
            } else {
#pragma unroll
            for (int rt = 0; rt < NTO; ++rt) {
                f32x16 c[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    if (init_kind < 0) init_lds(c[cb], bias_lds + boff, rt);
                    else init_rowbias(c[cb], cb, init_kind, rt);
                }
                if (with_pe) {
#pragma unroll
                    for (int g = 0; g < PEGc; ++g) {
                        const int idx = rt * fpt + g;
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) P::mma_pe(c[cb], q[idx % AP], pe[cb], g);
                        if (idx + AP < NF) q[idx % AP] = P::fetch(ring);
                    }
                }
                if (with_act) {
#pragma unroll
                    for (int t = 0; t < NTI; ++t)
#pragma unroll
                        for (int sb = 0; sb < P::SUBS; ++sb) {
                            const int idx = rt * fpt + (with_pe ? PEGc : 0) + t * P::SUBS + sb;
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) P::mma_act(c[cb], q[idx % AP], src[cb][t], sb);
                            if (idx + AP < NF) q[idx % AP] = P::fetch(ring);
                        }
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) P::store_act(dst[cb][rt], c[cb], relu, trk);
            }
            }
            if constexpr (PREQ == 0) ring.end_layer();
        };
        using IC_NT = std::integral_constant<int, NT>;
        using IC_NTD = std::integral_constant<int, NTD>;

        // ---- layer 1: PE(63) -> W, ReLU (nerf.py:113,176) ---------------------------------------
        const long long dstep = (long long)a.M * W;      // one layer of the activation dump
        float* const dh = DUMP ? a.dump_h : nullptr;
        layer(actY /*unused*/, actX, IC_NT{}, IC_NT{}, true, false, 0, 0, true, dh, W, false, true);
        STAMP(2);    // layer 1
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int v = 0; v < PE_VEC; ++v) pe_lds[(cb * PE_VEC + v) * NTHREADS] = ((const f32x4*)&pe[cb])[v];

        // ---- layers 2..D alternate X -> Y, Y -> X; the skip layer (index 3 of this sequence, nerf.py:174-176:
        //      input cat[input_xyz, h]) always lands on a Y -> X step -------------------------------------------------
        int boff = 0;
        for (int i = 0; i < n_mid; i += 2) {
            layer(actX, actY, IC_NT{}, IC_NT{}, false, true, -1, boff, true, dh ? dh + (i + 1) * dstep : nullptr, W, true, i + 1 < n_mid);
            boff += W;
            if (i + 1 < n_mid) {
                if (i + 1 == a.n_pre) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int v = 0; v < PE_VEC; ++v) ((f32x4*)&pe[cb])[v] = pe_lds[(cb * PE_VEC + v) * NTHREADS];
                    layer(actY, actX, IC_NT{}, IC_NT{}, true, true, 1, 0, true, dh ? dh + (i + 2) * dstep : nullptr, W, true, i + 2 < n_mid);
                } else {
                    layer(actY, actX, IC_NT{}, IC_NT{}, false, true, -1, boff, true, dh ? dh + (i + 2) * dstep : nullptr, W, true, i + 2 < n_mid);
                    boff += W;
                }
            }
        }
        STAMP(4);    // layers 2..D
        auto& hid = ENDY ? actY : actX;     // output of the last hidden layer

        // ---- sigma head (nerf.py:178) --------------------------------------------------------------------
        f32x16 accs[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) init_lds(accs[cb], bias_lds + boff, NT);   // unconditional: defined every tile
        if (with_sigma) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) guard_acc<P>(accs[cb]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sb = 0; sb < P::SUBS; ++sb) {
                    const int k = t * P::SUBS + sb;
                    typename P::Frag w;
                    if constexpr (PREQ > 0) w = preq[k % PREQ]; else w = P::fetch(ring);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) P::mma_act(accs[cb], w, hid[cb][t], sb);
                    if constexpr (PREQ > 0) {
                        if (k + PREQ < NT * P::SUBS) preq[k % PREQ] = P::fetch(ring);
                    }
                }
            if constexpr (PREQ > 0) {
                ring.end_layer();
                prefetch_next();
            }
            // (asm MFMAs: the compiler treats the statement as finished when it is issued.  Behind this chain it (a) copies the
            //  accumulators at the join of this branch -- v_mov reads of registers the hardware has not written yet: stale sigma in
            //  87 % of the samples, different on every launch -- and (b) reuses the 15 registers of each tile whose rows nobody reads
            //  for the dir layer's bias, which the late write then overwrites.  settle() INSIDE the branch keeps the tile live and
            //  untouched across 16 wait states.)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) settle_acc<P>(accs[cb]);
        }
        if constexpr (PREQ == 0) ring.end_layer();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) P::note(trk, accs[cb][0]);     // (fp16: an overflow of the last hidden layer shows here)
        if (sigma_only) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                bool ok;
                const long long mm = sample_of(cb, ok);
                if (ok && h == 0) a.out[mm * a.out_stride] = accs[cb][0];
            }
            continue;
        }
        boff += (NT + 1) * 32;
        STAMP(7);    // sigma

        // ---- dir_encoding: cat[final, dir ++ codes] -> W/2, ReLU (nerf.py:186-187).  xyz_encoding_final (:184) has no
        //      activation, so the host folds it into this layer (mlp_pack.fold_final): the stream's dir weights are
        //      Wd[:, :W] Wf, the row bias carries Wd[:, :W] bf, and the layer reads the last hidden activations directly. ----
        typename P::Act actd[CB][NTD];
        if constexpr (kAsmMfma<P>) {                // dir_encoding's output: in the buffer that does NOT hold the last hidden layer
            static_assert(!kAsmMfma<P> || (CB == 2 && NT == 8 && NWAVES == 4 && !WARP && DUMP == 0 && !COMP), "the AGPR map of PrecBF16A / PrecF16A");
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int t = 0; t < NTD; ++t) {
                    actd[cb][t].base = (ENDY ? P::kRegX : P::kRegY) + cb * P::kRegBlock + t * P::kRegTile;
                    if constexpr (HXR) actd_lo[cb][t].base = actd[cb][t].base + NTD * P::kRegTile;     // tiles 4..7 of the same buffer
                }
        }
        f32x16 acco[CB][2];
        if constexpr (HX && WARP) {
            // dir layer, one 32-row output tile (W = 64): weights hi + lo, the input single fp16
            f32x16 accd[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) init_rowbias(accd[cb], cb, 2, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sb = 0; sb < P::SUBS; ++sb) {
                    const f32x4 whi = ring.next(), wlo = ring.next();
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        accd[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P::as_f16(wlo), hid[cb][t].b[sb], accd[cb], 0, 0, 0);
                        accd[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P::as_f16(whi), hid[cb][t].b[sb], accd[cb], 0, 0, 0);
                    }
                }
            ring.end_layer();
            // ReLU, then the activations as fp16 roundings + rounded residuals (22 significand bits together)
            typename P::Act actd_lo[CB];       // (shadows the 8 x 256 form's array)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                P::note(trk, accd[cb][0]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    union { typename P::u32x4 w; f16x8 b; } oh, ol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = fmaxf(accd[cb][8 * u + 2 * q], 0.f), v1 = fmaxf(accd[cb][8 * u + 2 * q + 1], 0.f);
                        const unsigned hi = P::cvt_pk(v0, v1);
                        union { unsigned u; _Float16 h[2]; } hh;
                        hh.u = hi;
                        oh.w[q] = hi;
                        ol.w[q] = P::cvt_pk(v0 - (float)hh.h[0], v1 - (float)hh.h[1]);
                    }
                    actd[cb][0].b[u] = oh.b;
                    actd_lo[cb].b[u] = ol.b;
                }
            }
            // rgb head: (w_hi + w_lo)(a_hi + a_lo) without the lo * lo term
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) init_lds(acco[cb][ot], bias_lds + boff, ot);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                if (ot < nout_t) {
#pragma unroll
                    for (int sb = 0; sb < P::SUBS; ++sb) {
                        const f32x4 whi = ring.next(), wlo = ring.next();
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) {
                            acco[cb][ot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P::as_f16(wlo), actd[cb][0].b[sb], acco[cb][ot], 0, 0, 0);
                            acco[cb][ot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P::as_f16(whi), actd_lo[cb].b[sb], acco[cb][ot], 0, 0, 0);
                            acco[cb][ot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P::as_f16(whi), actd[cb][0].b[sb], acco[cb][ot], 0, 0, 0);
                        }
                    }
                }
            }
            ring.end_layer();
        } else {
        layer(hid, actd, IC_NTD{}, IC_NT{}, false, true, 2, 0, true, DUMP ? a.dump_dd : nullptr, NTD * 32, false, false, HXR);
        STAMP(8);    // dir layer
        // ---- rgb head (nerf.py:188) --------------------------------------------------------------------
        // both output tiles are initialised before either is accumulated into, so that no element of the pair carries
        // the previous tile's value (see actX / actY above)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) init_lds(acco[cb][ot], bias_lds + boff, ot);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            if (ot < nout_t) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) guard_acc<P>(acco[cb][ot]);
#pragma unroll
                for (int t = 0; t < NTD; ++t)
#pragma unroll
                    for (int sb = 0; sb < P::SUBS; ++sb) {
                        if constexpr (HXR) {
                            // (w_hi + w_lo)(a_hi + a_lo) without the lo * lo term, small terms first
                            const f32x4 whi = ring.next(), wlo = ring.next();
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) {
                                P::mma_act(acco[cb][ot], wlo, actd[cb][t], sb);
                                P::mma_act(acco[cb][ot], whi, actd_lo[cb][t], sb);
                                P::mma_act(acco[cb][ot], whi, actd[cb][t], sb);
                            }
                        } else {
                            const int k = ot * (NTD * P::SUBS) + t * P::SUBS + sb;        // fragment of the head (both tiles)
                            typename P::Frag w;
                            if constexpr (PREQ > 0) w = preq[k % PREQ]; else w = P::fetch(ring);
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) P::mma_act(acco[cb][ot], w, actd[cb][t], sb);
                            if constexpr (PREQ > 0) {
                                if (k + PREQ < nout_t * (NTD * P::SUBS)) preq[k % PREQ] = P::fetch(ring);
                            }
                        }
                    }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) settle_acc<P>(acco[cb][ot]);     // (asm MFMAs: inside the branch, as for sigma above)
            }
        }
        ring.end_layer();
        prefetch_next();       // (the next tile's first layer: its chunk arrives while this tile's outputs are stored)
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) P::note(trk, acco[cb][0][0]);  // (fp16: ... of the dir layer here)

        STAMP(9);    // rgb head
        if constexpr (COMP) {
            // ---- fused compositing (rendering.py:183-237).  A wave's 32 samples are one GROUP of a ray in the association of
            //      moda_dev.h (composite_ray): the wave scans its own transmittance factors and butterflies its own sums with
            //      shuffles; what crosses waves -- the groups' products and sums, 8 floats per group -- goes through LDS, and every
            //      wave chains the earlier groups' products in order.  Same routines, same order as composite_kernel: the results
            //      are bit-identical to the two-kernel route, whose (M, 4) round trip through HBM and second launch this replaces.
            float* cx = (float*)(pe_lds - threadIdx.x);        // the (now idle) PE stash: [NWAVES][8] group records
            const int S = a.comp_S;
            const int gpr = S >> 5;                            // groups (waves) per ray
            bool ok;
            const int mm = sample_of(0, ok);
            const int n = mm / S, sx = mm - n * S;             // (clamped sample for columns past the end: nothing is stored)
            float z = 0.f, alpha = 0.f, cr = 0.f, cg = 0.f, cbv = 0.f, t = 1.f;
            // both lane halves hold the tile's columns; the lower half owns the rgb / sigma rows and does the arithmetic
            if (ok && h == 0) {
                z = comp_z;
                cr = do_sigmoid ? sigmoidf(acco[0][0][0]) : acco[0][0][0];
                cg = do_sigmoid ? sigmoidf(acco[0][0][1]) : acco[0][0][1];
                cbv = do_sigmoid ? sigmoidf(acco[0][0][2]) : acco[0][0][2];
                alpha = comp_alpha(accs[0][0], a.comp_noise != nullptr, comp_noise_v, comp_delta, comp_ib);
                t = 1.f - alpha + 1e-10f;                                                                           // :218
            }
            float tot;
            const float excl = comp_group_scan(t, col, &tot);
            const int g = wave & (gpr - 1);                    // this wave's group index inside its ray (TILE % S == 0)
            if (lane == 0) cx[wave * 8 + 0] = tot;
            // (LDS traffic only: a __syncthreads() would also wait for vmcnt(0), i.e. drain the weight ring's LDS-DMA prefetch)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float carry = 1.f;
            {
#pragma clang fp contract(off)
                for (int k = 0; k < g; ++k) carry = carry * cx[(wave - g + k) * 8 + 0];
            }
            const float T = carry * excl;                                                                           // :219
            const bool live = ok && h == 0;
            CompTerms q;
            const float w = comp_sample_terms(alpha, T, live, sx + 1 >= S, cr, cg, cbv, z, 0.f, 0.f, 0.f,
                                              (live && a.comp_cyc) ? a.comp_cyc[mm] : 0.f, &q);
            if (live) {
                if (a.comp_out.weights) a.comp_out.weights[mm] = w;
                if (a.comp_out.visibility) a.comp_out.visibility[mm] = T;                                           // :224
            }
            const float gr = comp_group_sum(q.r), gg = comp_group_sum(q.g), gb = comp_group_sum(q.b), gd = comp_group_sum(q.d),
                        gs = comp_group_sum(q.s), gc = comp_group_sum(q.c);
            if (lane == 0) {
                float* rec = cx + wave * 8;
                rec[1] = gr; rec[2] = gg; rec[3] = gb; rec[4] = gd; rec[5] = gs; rec[6] = gc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g == 0 && lane == 0 && ok) {                   // the ray's first wave adds the groups' sums in order
#pragma clang fp contract(off)
                float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, as = 0.f, ac = 0.f;
                for (int k = 0; k < gpr; ++k) {
                    const float* rec = cx + (wave + k) * 8;
                    ar = ar + rec[1]; ag = ag + rec[2]; ab = ab + rec[3]; ad = ad + rec[4]; as = as + rec[5]; ac = ac + rec[6];
                }
                a.comp_out.rgb[n * 3 + 0] = ar; a.comp_out.rgb[n * 3 + 1] = ag; a.comp_out.rgb[n * 3 + 2] = ab;
                a.comp_out.depth[n] = ad;
                a.comp_out.sil[n] = as;
                if (a.comp_out.cyc_out && a.comp_cyc) a.comp_out.cyc_out[n] = ac;
            }
            continue;
        }
        if constexpr (WARP) {
            // ---- fused skinning tail (gauss_mlp_skinning :202-217 after the MLP, skinning :237-277, dqs_blend_skinning
            //      :457-493): the rgb-head accumulators are the per-bone logit offsets dskin[bone][sample] of this wave's 32
            //      samples, which all belong to ONE ray (warp_S % 32 == 0), so the per-set tables are wave-uniform. ---------
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                bool ok;
                const int mm = sample_of(cb, ok);
                const int m_first = __builtin_amdgcn_readfirstlane(tile * TILE + wave * (32 * CB) + cb * 32);
                const int ray = min(m_first, a.M - 1) / a.warp_S;
                long long qset = a.q_rps > 0 ? ray / a.q_rps : 0;
                long long dset = ray / a.dq_rps;
                if (a.run_start != nullptr) {        // repeated per-frame rows: the tables were built once per run
                    dset = a.run_start[dset];
                    if (a.q_rps > 0) qset = a.run_start[qset];
                }
                const float* qt = a.qtab + qset * nout_t * kWarpQFloats + lane;
                const f32x4* dt = a.dqtab + dset * nout_t * (kWarpDqFrags * 64) + lane;
                float qa[2][kWarpQFrags];
                f32x4 da[2][kWarpDqFrags];
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const int o2 = ot < nout_t ? ot : 0;
#pragma unroll
                    for (int f = 0; f < kWarpQFrags; ++f) qa[ot][f] = qt[(o2 * kWarpQFrags + f) * 64];
#pragma unroll
                    for (int u = 0; u < kWarpDqFrags; ++u) da[ot][u] = dt[(o2 * kWarpDqFrags + u) * 64];
                }
                const float x = a.xyz[(long long)mm * 3 + 0], y = a.xyz[(long long)mm * 3 + 1], z = a.xyz[(long long)mm * 3 + 2];
                // Gaussian logits: acc += Q (32 bones x 10) . monomials (10 x 32 samples), exact fp32 MFMAs; lane half h
                // supplies the k = h monomial of each of the five 2-deep steps
                float mono[kWarpQFrags] = {h ? y * y : x * x, h ? x * y : z * z, h ? y * z : x * z, h ? y : x, h ? 1.f : z};
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    if (ot < nout_t) {
#pragma unroll
                        for (int f = 0; f < kWarpQFrags; ++f)
                            acco[cb][ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ot][f], mono[f], acco[cb][ot], 0, 0, 0);
                    }
                // softmax numerators over the bones of this sample: 16 (x tiles) in this lane, the rest in lane ^ 32.
                // The common 1 / sum factor is dropped: dq_normalize (:471) divides the blend by its own real-part norm.
                float mx = -INFINITY;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    if (ot < nout_t) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acco[cb][ot][i]);
                    }
                // Operand registers of the logit MFMAs are kept live -- an empty asm that "reads" them -- until a result has been
                // consumed.  Round 2 put this in against rare wrong columns 16..31 and explained them as a queued MFMA reading
                // sources that the wave's next VALU instruction had already overwritten; round 4's hardware probe
                // (tools/probes/mfma_war_probe.hip) shows gfx950 has NO such hazard, so whatever the same edit fixed was something
                // else.  The statements cost nothing and stay.
#pragma unroll
                for (int f = 0; f < kWarpQFrags; ++f) asm volatile("" ::"v"(mono[f]), "v"(qa[0][f]), "v"(qa[1][f]), "v"(mx));
                {   // the other lane half's maximum: v_permlane32_swap instead of an LDS-pipe shuffle (moda_dev.h, comp_rows_pair)
                    unsigned ma = __builtin_bit_cast(unsigned, mx), mb = ma;
                    asm volatile("" : "+v"(mb));
                    const auto mr = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
                    const unsigned m0 = mr[0], m1 = mr[1];
                    mx = fmaxf(__uint_as_float(m0), __uint_as_float(m1));
                }
                const float kLog2e = 1.4426950408889634f;
                const float nmx = -mx * kLog2e;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                union Pack { u32x4 w; bf16x8 b; };
                Pack whi[2][2], wlo[2][2];       // [bone tile][16-bone sub-step]: B operands of the blend MFMAs
                auto pack_tile = [&](int ot) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(acco[cb][ot][8 * u + 2 * q], kLog2e, nmx));
                            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(acco[cb][ot][8 * u + 2 * q + 1], kLog2e, nmx));
                            const unsigned ph = PrecBF16::cvt_pk(e0, e1);
                            whi[ot][u].w[q] = ph;
                            // residuals against the rounded values: hi + lo carries 16 mantissa bits of the weight
                            wlo[ot][u].w[q] = PrecBF16::cvt_pk(e0 - __builtin_bit_cast(float, ph << 16),
                                                               e1 - __builtin_bit_cast(float, ph & 0xffff0000u));
                        }
                };
                pack_tile(0);
                if (nout_t > 1) {
                    pack_tile(1);
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) whi[1][u].w = wlo[1][u].w = u32x4{0u, 0u, 0u, 0u};
                }
                f32x16 bl;
#pragma unroll
                for (int i = 0; i < 16; ++i) bl[i] = 0.f;
                __builtin_amdgcn_sched_barrier(0);   // every operand register is final before the first blend MFMA issues
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    if (ot < nout_t) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            bl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PrecBF16::as_bf16(da[ot][u]), whi[ot][u].b, bl, 0, 0, 0);
                            bl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PrecBF16::as_bf16(da[ot][u]), wlo[ot][u].b, bl, 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
                // rows of the blend tile (moda_dev.h): this lane half holds real / dual sums as hi + lo pairs
                float c8[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo4 = bl[k] + bl[4 + k], hi4 = bl[8 + k] + bl[12 + k];
                    c8[k] = h ? hi4 : lo4;          // real part
                    c8[4 + k] = h ? lo4 : hi4;      // dual part
                }
                // c8 consumes every register of the MFMA result: the operands stay live (see above) up to here
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        asm volatile("" ::"v"(whi[ot][u].w), "v"(wlo[ot][u].w), "v"(da[ot][u]), "v"(c8[0]), "v"(c8[1]), "v"(c8[2]),
                                     "v"(c8[3]), "v"(c8[4]), "v"(c8[5]), "v"(c8[6]), "v"(c8[7]));
                float px = x, py = y, pz = z;
                if (a.pts_tf != nullptr) {
                    px = a.pts_tf[(long long)mm * 3 + 0]; py = a.pts_tf[(long long)mm * 3 + 1]; pz = a.pts_tf[(long long)mm * 3 + 2];
                }
                float ox, oy, oz;
#ifdef MODA_WARP_EXACT_DIV
                dqs_apply(c8, px, py, pz, &ox, &oy, &oz);
#else
                dqs_apply_fast(c8, px, py, pz, &ox, &oy, &oz);
#endif
                if (ok && h == 0) {
                    if (a.out != nullptr) {            // (null: the caller wants the cycle distance only)
                        float* o = a.out + (long long)mm * 3;
                        o[0] = ox; o[1] = oy; o[2] = oz;
                    }
                    if (a.cyc_ref != nullptr) {
                        const float dx = a.cyc_ref[(long long)mm * 3 + 0] - ox, dy = a.cyc_ref[(long long)mm * 3 + 1] - oy,
                                    dz = a.cyc_ref[(long long)mm * 3 + 2] - oz;
                        a.cyc_out[mm] = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);   // rendering.py:341 (v_sqrt_f32)
                    }
                }
            }
            continue;
        }
        // ---- store: out[m, row] for the rgb rows, sigma appended (nerf.py:190-197) ---------------------
#ifdef MODA_ABL_NOSTORE   // timing-only ablation build: results kept alive, nothing written
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) { keep_alive(acco[cb][0]); keep_alive(acco[cb][1]); keep_alive(accs[cb]); }
        continue;
#endif
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            bool ok;
            const int mm = sample_of(cb, ok);
            if (!ok) continue;
            if (a.n_out == 3 && with_sigma && a.out_tr_S == 0 && a.out_stride == 4) {
                // the coarse net: [rgb, sigma] is one float4 per sample, held by the lower lane half
                if (h == 0) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 3; ++i) v[i] = do_sigmoid ? sigmoidf(acco[cb][0][i]) : acco[cb][0][i];
                    v[3] = accs[cb][0];
                    *(f32x4*)(a.out + (long long)mm * 4) = v;
                }
                continue;
            }
            float* o = a.out + (long long)mm * a.out_stride;
            int rs = 1;   // distance between consecutive output channels of one sample
            if (a.out_tr_S > 0) {
                const int ray = (unsigned)mm / (unsigned)a.out_tr_S;
                o = a.out + (long long)ray * a.out_stride * a.out_tr_S + (mm - ray * a.out_tr_S);
                rs = a.out_tr_S;
                // channel-major output, a whole number of 32-sample groups per ray, every column valid: the wave's
                // 32 x 32 tile goes through its (now idle) PE-stash slice of LDS so that a lane writes four consecutive
                // samples of one channel -- 16-byte stores, eight lanes per 128-byte row segment, instead of 16 dword stores
                const int m_first = tile * TILE + wave * (32 * CB) + cb * 32;
#ifndef MODA_ABL_NO_TRSTORE
                if ((a.out_tr_S & 31) == 0 && m_first + 31 < a.M && !with_sigma) {
                    float* tb = (float*)(pe_lds - threadIdx.x) + (long long)cb * PE_VEC * NTHREADS * 4 + wave * 256;
                    float* ob = a.out + (long long)((unsigned)m_first / (unsigned)a.out_tr_S) * a.out_stride * a.out_tr_S
                                + (m_first % a.out_tr_S);
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot) {
                        if (ot < nout_t) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                                const float v = do_sigmoid ? sigmoidf(acco[cb][ot][i]) : acco[cb][ot][i];
                                tb[(row >> 3) * (NTHREADS * 4) + (row & 7) * 32 + col] = v;
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int ps = 0; ps < 4; ++ps) {
                                const int row = 32 * ot + 8 * ps + (lane >> 3);
                                const f32x4 v = *(const f32x4*)(tb + ps * (NTHREADS * 4) + (lane >> 3) * 32 + (lane & 7) * 4);
                                if (row < a.n_out) *(f32x4*)(ob + (long long)row * rs + (lane & 7) * 4) = v;
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    continue;
                }
#endif
            }
            // The 32 row predicates and row offsets below are the same in every tile; left to itself hipcc computes them once
            // before the tile loop and carries them through it -- in spilled SGPR pairs and 64-bit scratch slots (the dump
            // kernels: 84 dwords per lane).  An opaque copy of the lane half per tile keeps them local to this epilogue.
            // (the same trick on the transposed-store branch above cost the 8 x 256 inference kernel 5 %: its allocation
            //  then spilled the prefetched tile head inside the layer loop.  Check ScratchSize and an A/B after any change here.)
            int hq = h;
            asm volatile("" : "+v"(hq));
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                if (ot < nout_t) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = 32 * ot + (i & 3) + 8 * (i >> 2) + 4 * hq;
                        if (row < a.n_out) o[(long long)row * rs] = do_sigmoid ? sigmoidf(acco[cb][ot][i]) : acco[cb][ot][i];
                    }
                }
            }
            if (with_sigma && h == 0) o[(long long)a.n_out * rs] = accs[cb][0];
        }
        STAMP(10);   // output store
    }
#ifdef MODA_STAMPS
    if (a.stamps != nullptr && threadIdx.x == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(a.stamps + i, stamp_acc[i]);
#endif
    if constexpr (kIsF16<P>) {
        if (a.ovf != nullptr && __builtin_amdgcn_ballot_w64(P::overflowed(trk)) != 0ull && lane == 0)
            __hip_atomic_store(a.ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    ring.finish();
}

#ifdef MODA_STAMPS
static unsigned long long* moda_dbg_stamps_ptr = nullptr;
#endif

// chunks per pass; must mirror moda_amd/mlp_pack.py
struct StreamShape {
    int chf, subs, peg, nt, ntd;
    long long chunks;
    long long nbias;
};

static inline long long pad_to(long long v, long long q) { return (v + q - 1) / q * q; }

static int stream_shape(const moda_mlp_desc* d, StreamShape* s) {
    if (d->W != 64 && d->W != 128 && d->W != 256) return MODA_ESHAPE;
    if (d->D < 5 || d->D > 8) return MODA_ESHAPE;
    if (d->n_out < 1 || d->n_out > 64) return MODA_ESHAPE;
    if (d->n_freq < 0 || d->n_freq > 10) return MODA_ESHAPE;
    const bool x3 = (d->flags & MODA_MLP_BF16X3) != 0;
    const bool f16 = (d->flags & MODA_MLP_F16) != 0;
    if ((int)x3 + (int)f16 + (int)((d->flags & MODA_MLP_BF16) != 0) > 1) return MODA_EINVAL;      // one precision per launch
    const bool bf16 = x3 || f16 || (d->flags & MODA_MLP_BF16) != 0;      // the split and fp16 modes have the bf16 fragment geometry
    const bool sigma_only = (d->flags & MODA_MLP_SIGMA_ONLY) != 0;
    const bool with_sigma = sigma_only || (d->flags & MODA_MLP_WITH_SIGMA) != 0;
    s->chf = d->W == 64 ? 8 : 16;
    s->subs = bf16 ? 2 : 4;
    s->peg = bf16 ? 4 : 8;
    s->nt = d->W / 32;
    s->ntd = s->nt / 2 > 0 ? s->nt / 2 : 1;
    const long long m = x3 ? 2 : 1;                        // split mode: every fragment is a (hi, lo) pair, padded per layer after pairing
    // fp16 with split heads: the 64-wide network's dir and rgb layers' fragments come as (hi, lo) pairs, the 256-wide network's
    // rgb head's alone
    const bool hx = (d->flags & MODA_MLP_F16_HEADS) != 0;
    if (hx && (!f16 || sigma_only || (d->W == 64 && with_sigma) || d->W == 128)) return MODA_EINVAL;
    const long long mh = hx ? 2 : m;
    const long long mdir = (hx && d->W == 256) ? m : mh;
    const long long act = m * s->nt * s->nt * s->subs;     // frags of a W x W layer
    const long long pef = m * s->peg * s->nt;
    long long c = 0;
    c += pad_to(pef, s->chf);                              // layer 1
    c += 3 * pad_to(act, s->chf);                          // layers 2..4
    c += pad_to(pef + act, s->chf);                        // layer 5
    c += (long long)(d->D - 5) * pad_to(act, s->chf);      // layers 6..D
    c += pad_to(with_sigma ? m * s->nt * s->subs : 0, s->chf);   // sigma (xyz_encoding_final is folded into dir)
    if (!sigma_only) {
        c += pad_to(mdir * s->ntd * s->nt * s->subs, s->chf);                    // dir
        c += pad_to(mh * ((d->n_out + 31) / 32) * s->ntd * s->subs, s->chf);     // rgb
    }
    s->chunks = c / s->chf;
    s->nbias = (long long)(d->D - 2) * d->W + (s->nt + 1) * 32 + 64;
    return 0;
}

template <int W, typename P, int CB, int NWAVES, bool ENDY, bool UNI, bool WARP = false, int DUMP = 0, int RING = MODA_RING,
          bool COMP = false, bool HX = false>
static int launch_p(const MlpArgs& a, hipStream_t stream) {
    constexpr int CHF = (W == 64) ? 8 : 16;
    constexpr int TILE = NWAVES * 32 * CB;
    const size_t pe_bytes = sizeof(typename P::Pe) * CB * NWAVES * 64;
    constexpr bool RESIDENT = (W == 64) && kIs16<P> && (MODA_RESIDENT != 0);
    const size_t ring_chunks = RESIDENT ? (size_t)a.nchunks : (size_t)RING;
    constexpr int NTD = (W / 64 > 0) ? W / 64 : 1;
    const size_t rb_bytes = (size_t)NWAVES * CB * (2 * W + NTD * 32) * sizeof(float);
    const size_t lds = ring_chunks * CHF * kFragBytes + (size_t)(a.nbias + 16) * sizeof(float) + rb_bytes + pe_bytes +
                       (DUMP == 3 ? (size_t)NWAVES * CB * 4096 : (DUMP == 4 ? (size_t)NWAVES * CB * 2048 : 0));
    if (lds > 160 * 1024) return MODA_ESHAPE;
    // the attribute is per device: one bit per device ordinal and instantiation (a benign race only repeats the call),
    // so a process that drives several GPUs sets it on each of them
    static std::atomic<unsigned long long> attr_set{0ull};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) devid = 0;
    const unsigned long long bit = 1ull << (devid & 63);
    if (devid > 63 || !(attr_set.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_fused_kernel<W, P, CB, NWAVES, ENDY, UNI, WARP, DUMP, RING, COMP, HX>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    const long long ntiles = ((long long)a.M + TILE - 1) / TILE;
    int grid = ntiles < 256 ? (int)ntiles : 256;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((mlp_fused_kernel<W, P, CB, NWAVES, ENDY, UNI, WARP, DUMP, RING, COMP, HX>), dim3(grid), dim3(NWAVES * 64), lds, stream, a);
    return (int)hipGetLastError();
}

// the last hidden layer (index D-1 of layers 2..D, alternating X->Y, Y->X) writes Y when D-1 is odd
template <int W, typename P, int CB, int NWAVES, int DUMP = 0, int RING = MODA_RING, bool HX = false>
static int launch(const MlpArgs& a, hipStream_t stream) {
    // column blocks start at multiples of 32 samples; row = min(m / div, R - 1)
    const bool uni = (a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0);
    const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
    if (uni)
        return endy ? launch_p<W, P, CB, NWAVES, true, true, false, DUMP, RING, false, HX>(a, stream)
                    : launch_p<W, P, CB, NWAVES, false, true, false, DUMP, RING, false, HX>(a, stream);
    return endy ? launch_p<W, P, CB, NWAVES, true, false, false, DUMP, RING, false, HX>(a, stream)
                : launch_p<W, P, CB, NWAVES, false, false, false, DUMP, RING, false, HX>(a, stream);
}

}   // namespace

#ifdef MODA_STAMPS
extern "C" int moda_dbg_read_stamps(unsigned long long* host16) {
    if (!moda_dbg_stamps_ptr) return -1;
    hipDeviceSynchronize();
    return (int)hipMemcpy(host16, moda_dbg_stamps_ptr, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif

extern "C" int64_t moda_mlp_stream_bytes(const moda_mlp_desc* d) {
    StreamShape s;
    if (stream_shape(d, &s) != 0) return -1;
    return s.chunks * s.chf * kFragBytes;
}

extern "C" int64_t moda_mlp_bias_floats(const moda_mlp_desc* d) {
    StreamShape s;
    if (stream_shape(d, &s) != 0) return -1;
    return s.nbias;
}

// The 128-wide network (nerf_feat: five layers, no per-ray codes).  Its tile has 180 MFMAs where the 8 x 256 network's has 1 048,
// so what a tile costs besides them -- the encoding, one LDS fragment read and one ring barrier share per MFMA, the store --
// weighs twice as much: 42 % of the matrix peak executed against 58-60 %.  TWO 32-sample column blocks per wave halve the fragment
// reads and barriers per MFMA; the registers of the second block exist here (two blocks of a 128-wide net = one of a 256-wide
// one: 247 VGPRs, no scratch, with a 4-slot ring for the doubled encoding stash), which they did not for the 8 x 256 kernel
// (-7 %, round 3).  Taken for long batches whose column blocks share their code rows and whose last hidden layer lands in X (the
// instantiations that do not spill); 2.9 -> 2.57 ms on config 5's 16.8 M samples.
template <typename P>
static int wide128(const MlpArgs& a, hipStream_t st) {
    const bool uni = (a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0);
    const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
    if (MODA_BF16_CB128 > 1 && uni && !endy && a.M >= 256 * 512 && a.n_live == nullptr)
        return launch_p<128, P, (MODA_BF16_CB128 > 1 ? MODA_BF16_CB128 : 1), MODA_BF16_WAVES, false, true, false, 0, 4>(a, st);
    return launch<128, P, MODA_BF16_CB, MODA_BF16_WAVES>(a, st);
}

static int dispatch(const moda_mlp_desc* d, const MlpArgs& a, hipStream_t st) {
    // paired head fragments: the 8 x 256 kernel here (rgb head), the 64-wide network in moda_mlp_warp_fwd only
    const bool hx = (d->flags & MODA_MLP_F16_HEADS) != 0;
    if (hx && d->W != 256) return MODA_ESHAPE;
    const bool bf16 = (d->flags & MODA_MLP_BF16) != 0;
    if (d->flags & MODA_MLP_BF16X3) {
        if (d->W == 256) return launch<256, PrecBF16x3, 1, MODA_X3_WAVES256>(a, st);
        if (d->W == 128) return launch<128, PrecBF16x3, 1, MODA_X3_WAVES128>(a, st);
        return launch<64, PrecBF16x3, 1, MODA_X3_WAVES>(a, st);
    }
    if (d->flags & MODA_MLP_F16) {
        const char* agpr_env16 = getenv("MODA_MLP_AGPR");        // (read per call, as for bf16 below)
        const int agpr16 = agpr_env16 ? atoi(agpr_env16) : MODA_MLP_AGPR_DEFAULT;
        if (d->W == 256 && hx && agpr16 && a.n_live == nullptr) {            // the four-wave AGPR form (PrecF16A), uniform column blocks only
            const bool uni = (a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0);
            const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
            if (uni) return endy ? launch_p<256, PrecF16A, 2, 4, true, true, false, 0, MODA_RING, false, true>(a, st)
                                 : launch_p<256, PrecF16A, 2, 4, false, true, false, 0, MODA_RING, false, true>(a, st);
        }
        if (d->W == 256 && hx) return launch<256, PrecF16, MODA_BF16_CB, MODA_BF16_WAVES, 0, MODA_RING, true>(a, st);
        if (d->W == 256) return launch<256, PrecF16, MODA_BF16_CB, MODA_BF16_WAVES>(a, st);
        if (d->W == 128) return wide128<PrecF16>(a, st);
        return launch<64, PrecF16, MODA_BF16_CB64, (MODA_RESIDENT ? MODA_BF16_WAVES64 : MODA_BF16_WAVES)>(a, st);
    }
    if (bf16) {
        // MODA_MLP_AGPR=1: the four-wave, two-column-block form with the activations in AGPRs (PrecBF16A), uniform column blocks
        // only.  A run-time switch so that both forms can be timed in one process (interleaved A/B on one box).
        const char* agpr_env = getenv("MODA_MLP_AGPR");            // (read per call: both forms in one process)
        const int agpr = agpr_env ? atoi(agpr_env) : MODA_MLP_AGPR_DEFAULT;
        if (d->W == 256 && agpr && a.n_live == nullptr) {
            const bool uni = (a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0);
            const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
            if (uni) return endy ? launch_p<256, PrecBF16A, 2, 4, true, true>(a, st) : launch_p<256, PrecBF16A, 2, 4, false, true>(a, st);
        }
        if (d->W == 256) return launch<256, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES>(a, st);
        if (d->W == 128) return wide128<PrecBF16>(a, st);
        return launch<64, PrecBF16, MODA_BF16_CB64, (MODA_RESIDENT ? MODA_BF16_WAVES64 : MODA_BF16_WAVES)>(a, st);
    }
    if (d->W == 256) return launch<256, PrecF32, 1, 4>(a, st);
    if (d->W == 128) return launch<128, PrecF32, 1, 4>(a, st);
    return launch<64, PrecF32, 1, 4>(a, st);
}

static int fill_args(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz, const uint8_t* flip_x,
                     const float* rb1, const float* rb5, int64_t R1, int64_t div1, const float* rbd, int64_t Rd, int64_t divd,
                     float* out, int64_t out_stride, int64_t out_tr_S, int64_t M, void* stream, MlpArgs* pa) {
    StreamShape s;
    const int rc = stream_shape(d, &s);
    if (rc != 0) return rc;
    if (!wstream || !bias || !xyz || !rb1 || !rb5 || !rbd || !out) return MODA_EINVAL;
    if (R1 < 1 || Rd < 1 || div1 < 1 || divd < 1) return MODA_EINVAL;
    const int64_t lim = 0x7fffffff;
    if (M > lim || R1 > lim || Rd > lim || out_stride > lim || out_tr_S > lim) return MODA_ESHAPE;
    if (div1 > lim) div1 = lim;   // rows = m / div clamps to row 0 anyway
    if (divd > lim) divd = lim;
    MlpArgs& a = *pa;
    a.stamps = nullptr;
#ifdef MODA_STAMPS
    {
        static unsigned long long* dbg = nullptr;
        if (!dbg) hipMalloc((void**)&dbg, 16 * sizeof(unsigned long long));
        hipMemsetAsync(dbg, 0, 16 * sizeof(unsigned long long), (hipStream_t)stream);
        a.stamps = dbg;
        moda_dbg_stamps_ptr = dbg;
    }
#endif
    a.wstream = (const uint8_t*)wstream;
    a.bias = bias;
    a.xyz = xyz;
    a.flip = flip_x;
    a.rb1 = rb1;
    a.rb5 = rb5;
    a.rbd = rbd;
    a.out = out;
    a.M = (int)M;
    a.R1 = (int)R1;
    a.div1 = (int)div1;
    a.Rd = (int)Rd;
    a.divd = (int)divd;
    a.out_stride = (int)out_stride;
    a.out_tr_S = (int)out_tr_S;
    if (out_tr_S < 0 || (out_tr_S > 0 && M % out_tr_S != 0)) return MODA_EINVAL;
    a.nchunks = (int)s.chunks;
    a.nbias = (int)s.nbias;
    a.n_pre = 3;
    a.n_post = d->D - 5;
    a.n_out = d->n_out;
    a.flags = d->flags;
    a.n_freq = d->n_freq;
    for (int i = 0; i < 16; ++i) a.window[i] = d->window[i];
    a.qtab = nullptr;
    a.dqtab = nullptr;
    a.pts_tf = nullptr;
    a.cyc_ref = nullptr;
    a.cyc_out = nullptr;
    a.warp_S = 0;
    a.q_rps = 0;
    a.dq_rps = 1;
    a.run_start = nullptr;
    a.rows_at_runs = 0;
    a.n_live = nullptr;
    a.live_S = 0;
    a.dump_h = nullptr;
    a.dump_dd = nullptr;
    a.dump_bf16 = 0;
    a.comp_zv = a.comp_rd = a.comp_beta = a.comp_noise = a.comp_cyc = nullptr;
    a.comp_S = 0;
    a.ovf = (d->flags & MODA_MLP_F16) ? (int*)d->overflow : nullptr;
    a.comp_out = CompOut{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return 0;
}

extern "C" int moda_mlp_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                            const uint8_t* flip_x, const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                            const float* rbd, int64_t Rd, int64_t divd, float* out, int64_t out_stride, int64_t out_tr_S,
                            int64_t M, void* stream) {
    if (!d) return MODA_EINVAL;
    if (M <= 0) {
        StreamShape s;
        return stream_shape(d, &s);
    }
    MlpArgs a;
    const int rc = fill_args(d, wstream, bias, xyz, flip_x, rb1, rb5, R1, div1, rbd, Rd, divd, out, out_stride, out_tr_S, M, stream, &a);
    if (rc != 0) return rc;
    return dispatch(d, a, (hipStream_t)stream);
}

extern "C" int moda_mlp_live_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                                 const uint8_t* flip_x, const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                                 const float* rbd, int64_t Rd, int64_t divd, float* out, int64_t out_stride, int64_t M,
                                 const int32_t* n_live, int64_t S, void* stream) {
    if (!d) return MODA_EINVAL;
    if (M <= 0) {
        StreamShape s;
        return stream_shape(d, &s);
    }
    if (!n_live || S < 32 || S % 32 != 0 || M % S != 0 || S > 0x7fffffffLL) return MODA_ESHAPE;
    MlpArgs a;
    const int rc = fill_args(d, wstream, bias, xyz, flip_x, rb1, rb5, R1, div1, rbd, Rd, divd, out, out_stride, 0, M, stream, &a);
    if (rc != 0) return rc;
    // the skip test is made per 32-sample group, in the kernels whose groups share their code rows
    if (!((a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0))) return MODA_ESHAPE;
    a.n_live = (const int*)n_live;
    a.live_S = (int)S;
    return dispatch(d, a, (hipStream_t)stream);
}

extern "C" int moda_mlp_composite_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                                      const uint8_t* flip_x, const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                                      const float* rbd, int64_t Rd, int64_t divd, const float* z_vals, const float* rays_d,
                                      const float* beta, const float* noise, const float* cyc, int64_t S, int64_t M, float* rgb,
                                      float* depth, float* sil, float* weights, float* visibility, float* cyc_out, void* stream) {
    if (!d) return MODA_EINVAL;
    // the 8 x 256-class bf16 colour network ([sigmoid(rgb), sigma] per sample), rays of 32 / 64 / 128 / 256 samples: whole rays
    // per 256-sample workgroup tile, code rows uniform over every 32-sample group
    if (d->W != 256 || d->n_out != 3 || (d->flags & (MODA_MLP_BF16 | MODA_MLP_SIGMOID | MODA_MLP_WITH_SIGMA | MODA_MLP_SIGMA_ONLY |
                                                   MODA_MLP_BF16X3 | MODA_MLP_F16)) != (MODA_MLP_BF16 | MODA_MLP_SIGMOID | MODA_MLP_WITH_SIGMA))
        return MODA_ESHAPE;
    if ((S != 32 && S != 64 && S != 128 && S != 256) || M % S != 0) return MODA_ESHAPE;
    if (M <= 0) return 0;
    if (!z_vals || !rays_d || !beta || !rgb || !depth || !sil || (cyc && !cyc_out)) return MODA_EINVAL;
    float dummy_out = 0.f;
    MlpArgs a;
    const int rc = fill_args(d, wstream, bias, xyz, flip_x, rb1, rb5, R1, div1, rbd, Rd, divd, &dummy_out, 4, 0, M, stream, &a);
    if (rc != 0) return rc;
    if (!((a.R1 == 1 || a.div1 % 32 == 0) && (a.Rd == 1 || a.divd % 32 == 0))) return MODA_ESHAPE;
    a.out = nullptr;
    a.comp_zv = z_vals; a.comp_rd = rays_d; a.comp_beta = beta; a.comp_noise = noise; a.comp_cyc = cyc; a.comp_S = (int)S;
    a.comp_out = CompOut{rgb, nullptr, depth, sil, weights, visibility, nullptr, cyc_out, nullptr};
    const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
    hipStream_t st = (hipStream_t)stream;
    return endy ? launch_p<256, PrecBF16, 1, 8, true, true, false, 0, MODA_RING, true>(a, st)
                : launch_p<256, PrecBF16, 1, 8, false, true, false, 0, MODA_RING, true>(a, st);
}

extern "C" int moda_mlp_warp_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                                 const float* rb1, const float* rb5, int64_t R1, int64_t div1, const float* rbd, const float* qtab,
                                 int64_t q_rps, const void* dqtab, int64_t dq_rps, const float* pts_tf, const float* cyc_ref, float* xyz_out,
                                 float* cyc_out, int64_t S, int64_t M, const int32_t* run_start, void* stream) {
    if (!d) return MODA_EINVAL;
    // the 64-wide bf16 skin net with raw outputs (one logit per bone, at most two 32-bone tiles), whole 32-sample groups per ray
    if (d->W != 64 || !(d->flags & (MODA_MLP_BF16 | MODA_MLP_F16 | MODA_MLP_BF16X3)) ||
        (d->flags & (MODA_MLP_SIGMA_ONLY | MODA_MLP_WITH_SIGMA | MODA_MLP_SIGMOID)))
        return MODA_ESHAPE;
    if (S < 32 || S % 32 != 0 || M % S != 0 || q_rps < 0 || dq_rps < 1) return MODA_ESHAPE;
    if (M <= 0) return 0;
    if (!qtab || !dqtab || (!xyz_out && !cyc_ref) || (cyc_ref && !cyc_out) || !rb1 || !rb5 || !rbd) return MODA_EINVAL;
    if (R1 != 1 && div1 % 32 != 0) return MODA_ESHAPE;          // code rows must be uniform over a 32-sample group
    MlpArgs a;
    float dummy_out = 0.f;
    const int rc = fill_args(d, wstream, bias, xyz, nullptr, rb1, rb5, R1, div1, rbd, 1, 1, xyz_out ? xyz_out : &dummy_out, 3, 0, M, stream, &a);
    if (rc != 0) return rc;
    a.out = xyz_out;                                   // may be null: only cyc_out is wanted
    a.qtab = qtab;
    a.dqtab = (const f32x4*)dqtab;
    a.pts_tf = pts_tf;
    a.cyc_ref = cyc_ref;
    a.cyc_out = cyc_out;
    a.warp_S = (int)S;
    a.q_rps = (int)(q_rps > 0x7fffffff ? 0x7fffffff : q_rps);
    a.dq_rps = (int)(dq_rps > 0x7fffffff ? 0x7fffffff : dq_rps);
    if (run_start && q_rps > 0 && q_rps != dq_rps) return MODA_EINVAL;      // one run table serves both kinds of set
    if (d->reserved & MODA_MLP_ROWS_AT_RUNS) {                              // ... and, on request, the code rows: one row per set
        if (!run_start || R1 <= 1 || div1 != S * dq_rps) return MODA_EINVAL;
        a.rows_at_runs = 1;
    }
    a.run_start = (const int*)run_start;
    constexpr int NW = MODA_RESIDENT ? MODA_BF16_WAVES64 : MODA_BF16_WAVES;
    const bool endy = ((a.n_pre + 1 + a.n_post) & 1) != 0;
    hipStream_t st = (hipStream_t)stream;
    if (d->flags & MODA_MLP_F16_HEADS)
        return endy ? launch_p<64, PrecF16, MODA_BF16_CB64, NW, true, true, true, 0, MODA_RING, false, true>(a, st)
                    : launch_p<64, PrecF16, MODA_BF16_CB64, NW, false, true, true, 0, MODA_RING, false, true>(a, st);
    if (d->flags & MODA_MLP_F16)
        return endy ? launch_p<64, PrecF16, MODA_BF16_CB64, NW, true, true, true>(a, st)
                    : launch_p<64, PrecF16, MODA_BF16_CB64, NW, false, true, true>(a, st);
    if (d->flags & MODA_MLP_BF16X3)      // the parity-grade form: split-bf16 network, same tail (round 4)
        return endy ? launch_p<64, PrecBF16x3, 1, MODA_X3_WAVES, true, true, true>(a, st)
                    : launch_p<64, PrecBF16x3, 1, MODA_X3_WAVES, false, true, true>(a, st);
    return endy ? launch_p<64, PrecBF16, MODA_BF16_CB64, NW, true, true, true>(a, st)
                : launch_p<64, PrecBF16, MODA_BF16_CB64, NW, false, true, true>(a, st);
}

extern "C" int moda_mlp_dump_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                                 const uint8_t* flip_x, const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                                 const float* rbd, int64_t Rd, int64_t divd, float* out, int64_t out_stride, float* dump_h,
                                 float* dump_dd, int64_t M, void* stream) {
    if (!d) return MODA_EINVAL;
    if (!(d->flags & MODA_MLP_BF16) || (d->flags & MODA_MLP_SIGMA_ONLY)) return MODA_ESHAPE;
    if (d->W != 64 && d->W != 128 && d->W != 256) return MODA_ESHAPE;
    if (M <= 0) {
        StreamShape s;
        return stream_shape(d, &s);
    }
    if (!dump_h || !dump_dd) return MODA_EINVAL;
    if ((long long)M * d->W * 4 >= (1LL << 32)) return MODA_ESHAPE;       // the dump stores use 32-bit byte offsets within a layer
    MlpArgs a;
    const int rc = fill_args(d, wstream, bias, xyz, flip_x, rb1, rb5, R1, div1, rbd, Rd, divd, out, out_stride, 0, M, stream, &a);
    if (rc != 0) return rc;
    a.dump_h = dump_h;
    a.dump_dd = dump_dd;
    a.dump_bf16 = (d->reserved & MODA_MLP_DUMP_BF16) != 0;
    hipStream_t st = (hipStream_t)stream;
#ifndef MODA_DUMP_MODE
#define MODA_DUMP_MODE 3           // bf16 dumps: 3 through the per-wave LDS transpose, 2 lane-pair swap + 16-byte scattered stores
#endif
    if (a.dump_bf16) {
        // 8 x 256: lane-pair swap + 16-byte stores with the usual 8 waves (0.80 ms; the LDS route needs the waves halved for
        // its buffers and loses more to the ring than the wider stores gain: 1.16 ms).  Narrower nets: through LDS.
        // (measured for 8 x 256, all 0.79-0.83 ms: the pair-swap form; one-tile LDS transposes with all 8 waves and a 5-deep
        //  ring, -DMODA_DUMP_MODE256=4; the ring's counted wait relaxed by 4 or 8 operations as an experiment.  Neither the
        //  width of the stores nor the wait is what bounds this kernel.)
#ifndef MODA_DUMP_MODE256
#define MODA_DUMP_MODE256 2
#endif
#if MODA_DUMP_MODE256 == 4
        if (d->W == 256) return launch<256, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES, 4, MODA_RING - 1>(a, st);
#endif
        if (d->W == 256) return launch<256, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES, 2>(a, st);
        if (MODA_DUMP_MODE == 3) {
            if (d->W == 128) return launch<128, PrecBF16, MODA_BF16_CB, 4, 3>(a, st);
            return launch<64, PrecBF16, MODA_BF16_CB64, 8, 3>(a, st);
        }
        if (d->W == 128) return launch<128, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES, 2>(a, st);
        return launch<64, PrecBF16, MODA_BF16_CB64, (MODA_RESIDENT ? MODA_BF16_WAVES64 : MODA_BF16_WAVES), 2>(a, st);
    }
    if (d->W == 256) return launch<256, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES, 1>(a, st);
    if (d->W == 128) return launch<128, PrecBF16, MODA_BF16_CB, MODA_BF16_WAVES, 1>(a, st);
    return launch<64, PrecBF16, MODA_BF16_CB64, (MODA_RESIDENT ? MODA_BF16_WAVES64 : MODA_BF16_WAVES), 1>(a, st);
}
