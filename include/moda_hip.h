/*
 * moda_hip.h -- C ABI of libmoda_hip.so, the MI355X (gfx950) implementation of
 * MoDA's per-ray rendering hot path.
 *
 * The reference (ChaoyueSong/MoDA) is pure Python/PyTorch: it has no FFI for
 * this path.  Each entry point below therefore names the reference *function*
 * (file:line under the reference root) whose arithmetic it replaces; the
 * Python package moda_amd/ mirrors the reference's call surface on top of
 * these through ctypes.
 *
 * Conventions (all entry points):
 *   - raw device pointers, fp32 unless stated, contiguous row-major, ray-major;
 *   - the caller allocates every output and workspace;
 *   - no allocation, no global state, no implicit synchronisation inside;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - returns 0 on success, otherwise the hipError_t of the failed launch or a
 *     negative MODA_E* code for an argument the library cannot serve;
 *   - thread-safe by statelessness.
 */
#ifndef MODA_HIP_H
#define MODA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODA_EINVAL (-1)   /* unsupported / inconsistent argument */
#define MODA_ESHAPE (-2)   /* shape outside what the kernels were instantiated for */

/* ABI version; bumped on any signature change. */
int moda_abi_version(void);

/* 0 when `stream` is not being captured into a HIP graph, else the runtime's id of that capture (hipStreamGetCaptureInfo): a
 * host-side cache of device buffers whose initialisation must be part of the graph that uses them keys on it. */
uint64_t moda_stream_capture_id(void* stream);

/* ------------------------------------------------------------------------
 * Fused positional-encoding + NeRF MLP  (nnutils/nerf.py:35-75 Embedding.forward,
 * nerf.py:147-198 NeRF.forward, driven by nnutils/geom_utils.py:19-57 evaluate_mlp)
 * ------------------------------------------------------------------------ */

#define MODA_MLP_BF16        1   /* bf16 MFMA operands, fp32 accumulate (default: exact fp32 MFMA) */
#define MODA_MLP_SIGMOID     2   /* sigmoid on the rgb head (raw_feat == False, nerf.py:193) */
#define MODA_MLP_WITH_SIGMA  4   /* also evaluate the sigma head and append it after the rgb columns */
#define MODA_MLP_SIGMA_ONLY  8   /* sigma_only=True early-out (nerf.py:179-180): out is (M,1) */
#define MODA_MLP_BF16X3     16   /* split-bf16: operands as bf16 hi + lo, three MFMAs per product (hi*hi + hi*lo + lo*hi), fp32
                                    accumulate, exact sincosf encoding -- the parity-grade throughput mode (not with MODA_MLP_BF16);
                                    the weight stream holds every fragment twice, as (roundings, residuals) pairs padded per
                                    layer after pairing (moda_mlp_stream_bytes accounts for it) */
#define MODA_MLP_F16        32   /* fp16 MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate: the bf16 mode's rate with 11
                                    significand bits instead of 8 -- the parity-grade mode at throughput speed (ABI 7; one of
                                    BF16 / BF16X3 / F16 per launch).  Stream layout and size as MODA_MLP_BF16, elements fp16.
                                    Nothing saturates silently: see moda_mlp_desc.overflow */

#define MODA_MLP_F16_HEADS  64   /* with MODA_MLP_F16: head layers with split operands, their stream fragments as (fp16 rounding, fp16
                                    residual) pairs.  W = 64 with raw outputs, moda_mlp_warp_fwd only: the dir_encoding layer uses
                                    its weights hi + lo (2 MFMAs per product), the rgb head weights and activations hi + lo (3
                                    MFMAs) -- these two layers carry ~95 % of the fp16 network's output error.  W = 256,
                                    moda_mlp_fwd / moda_mlp_live_fwd: the rgb head alone (weights and activations split), which
                                    carries the 8 x 256 network's colour error (ABI 7) */

typedef struct moda_mlp_desc {
    int32_t W;            /* hidden width: 64, 128 or 256 */
    int32_t D;            /* xyz_encoding layers, 5..8, skip connection at layer index 4 (skips=[4]) */
    int32_t n_out;        /* rows of the rgb head (out_channels), 1..64 */
    int32_t flags;        /* MODA_MLP_* */
    int32_t n_freq;       /* positional-encoding frequencies of the xyz input, <= 10 */
    int32_t reserved;
    float   window[16];   /* w_k of Embedding (nerf.py:63-68), k < n_freq */
    int32_t* overflow;    /* MODA_MLP_F16 only, NULL allowed: a device-ACCESSIBLE int32 (device memory, or pinned host memory that
                             the caller can read without synchronising) that the launch sets to 1 -- a system-scope store, never
                             cleared by the library -- when any hidden activation of any sample rounded to an fp16 infinity
                             (|v| >= 65520) or was not a number.  Ignored by the other precisions (ABI 7) */
} moda_mlp_desc;

/* Bytes of the packed weight stream / floats of the LDS-resident bias block for a descriptor.
 * The stream layout itself is produced by moda_amd/mlp_pack.py (documented there). */
int64_t moda_mlp_stream_bytes(const moda_mlp_desc* d);
int64_t moda_mlp_bias_floats(const moda_mlp_desc* d);

/* out[m, :] = NeRF(PE(xyz[m]) ++ per-row codes).
 *   xyz        (M,3)        sample positions
 *   flip_x     (M) u8|NULL  1 -> negate x before encoding (symm_shape, rendering.py:385-391)
 *   wstream                 packed weights (moda_mlp_stream_bytes)
 *   bias                    packed plain biases (moda_mlp_bias_floats)
 *   rb1,rb5    (R1,W)       layer-1 / skip-layer bias with the per-row code part of the input already
 *                           folded in:  rb[r,o] = b[o] + sum_k Wcode[o,k] code[r,k]; sample m uses row
 *                           min(m / div1, R1-1).  R1 == 1 when the net has no code input.
 *   rbd        (Rd,W/2)     same for the dir_encoding layer (dir embedding ++ env/appearance codes).
 *                           xyz_encoding_final (nerf.py:184) is a Linear WITHOUT activation in front of dir_encoding's
 *                           Linear (:186-187): the stream carries the two as ONE (W/2 x W) layer, Wd[:, :W] Wf, and rbd
 *                           must include Wd[:, :W] bf (moda_amd/mlp_pack.py fold_final; the caller forms the product).
 *   out        (M, out_stride) columns [0,n_out) = rgb head, column n_out = sigma if WITH_SIGMA;
 *              with out_tr_S = S > 0 the layout is (M/S, out_stride, S) instead: channel-major inside each
 *              group of S consecutive samples (one ray), so that consecutive samples are contiguous.
 */
int moda_mlp_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias,
                 const float* xyz, const uint8_t* flip_x,
                 const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                 const float* rbd, int64_t Rd, int64_t divd,
                 float* out, int64_t out_stride, int64_t out_tr_S, int64_t M, void* stream);

/* moda_mlp_fwd (bf16 mode) that ALSO writes what a backward pass needs: dump_h (D, M, W) fp32, layer l's post-ReLU output at
 * dump_h + l * M * W, row-major [sample][feature]; dump_dd (M, W/2) the dir_encoding activations.  Stored straight from the
 * MFMA accumulators (four consecutive features per lane and store). */
int moda_mlp_dump_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias,
                      const float* xyz, const uint8_t* flip_x,
                      const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                      const float* rbd, int64_t Rd, int64_t divd,
                      float* out, int64_t out_stride, float* dump_h, float* dump_dd, int64_t M, void* stream);

/* moda_mlp_fwd of the 8 x 256-class colour network (bf16 mode; out would be [sigmoid(rgb), sigma]) with the compositing of
 * nnutils/rendering.py:183-237 (moda_composite_fwd's plain form: no clip / vis_pred / feature / rgb_filter / termination) as the
 * kernel's epilogue: the (M, 4) network output never goes to HBM.  Rays of S = 32, 64, 128 or 256 consecutive samples (whole
 * rays per 256-sample workgroup tile), per-row codes uniform over every 32-sample group.  z_vals (M), rays_d (M/S, 3), beta (1),
 * noise (M)|NULL, cyc (M)|NULL -> rgb (M/S, 3), depth, sil (M/S), weights (M)|NULL, visibility (M)|NULL, cyc_out (M/S)|NULL.
 * Results are bit-identical to moda_mlp_fwd + moda_composite_fwd (one shared device routine walks the ray in both). */
int moda_mlp_composite_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                           const uint8_t* flip_x, const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                           const float* rbd, int64_t Rd, int64_t divd, const float* z_vals, const float* rays_d,
                           const float* beta, const float* noise, const float* cyc, int64_t S, int64_t M, float* rgb,
                           float* depth, float* sil, float* weights, float* visibility, float* cyc_out, void* stream);

/* moda_mlp_fwd with a per-ray sample bound (early ray termination, opt-in): the M samples are rays of S consecutive
 * samples (S % 32 == 0, per-row codes uniform over each 32-sample group) and the 32-sample groups that start at or
 * beyond n_live[ray] (int32, M / S entries) are NOT evaluated -- their rows of `out` are left untouched; the consumer
 * (moda_composite_fwd with the same n_live) never reads them. */
int moda_mlp_live_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias,
                      const float* xyz, const uint8_t* flip_x,
                      const float* rb1, const float* rb5, int64_t R1, int64_t div1,
                      const float* rbd, int64_t Rd, int64_t divd,
                      float* out, int64_t out_stride, int64_t M, const int32_t* n_live, int64_t S, void* stream);

/* A NeRF module's parameters (nerf.py:109-140 state-dict tensors, fp32) -> the weight stream and bias block the fused
 * kernels consume, in ONE launch.  Run at every call: there is no cache of packed weights to go stale behind an optimiser.
 *   wsrc (n_wsrc <= 16), bsrc (n_bsrc <= 16)   HOST arrays of device pointers, in the order the code tables refer to
 *   wcode (n_w, int32, device), bcode (n_b)    per output element: (source index << 24) | element offset, negative = 0
 *                                              (moda_amd/mlp_pack.py StreamIndex.codes(); n_w a multiple of 8)
 *   wstream    n_w elements.  bf16 == 0: fp32;  1: bf16 (round-to-nearest-even);  2 (MODA_MLP_BF16X3): bf16, where an
 *              element whose code has bit 30 set holds the rounded RESIDUAL bf16(v - bf16(v)) of its value (the table
 *              lists every fragment twice: values, then residuals);  3 (MODA_MLP_F16): fp16, round-to-nearest-even, an element
 *              whose code has bit 30 set holds the fp16 RESIDUAL f16(v - f16(v)) (the head layers under MODA_MLP_F16_HEADS);
 *              bias  n_b fp32
 *   overflow   mode 3 only, NULL allowed: as moda_mlp_desc.overflow -- set to 1 when a weight is not representable in fp16
 *              (|w| >= 65520 or not a number)  (ABI 7) */
int moda_mlp_pack(const void* const* wsrc, int32_t n_wsrc, const int32_t* wcode, int64_t n_w, int32_t bf16,
                  void* wstream, const void* const* bsrc, int32_t n_bsrc, const int32_t* bcode, int64_t n_b,
                  float* bias, int32_t* overflow, void* stream);

/* Up to four of the folds below in ONE launch (the per-row code folds of a fused network call: layer-1 and skip-layer pose-code
 * rows, dir_encoding rows): Y_i[r, o] = b_i[o] + sum_k W_i[o, col0_i + k] X_i[r, k], every argument a HOST array of n entries
 * (X_i (R_i, K_i; ldx_i), W_i (O_i, ldw_i), b_i (O_i)|NULL, Y_i (R_i, O_i; ldy_i)); fp32 fmaf chains in k order.
 * run_start (host array of n device pointers, or NULL; entry i: R_i int32 from moda_row_runs, or NULL): only rows that START a run
 * of identical rows are computed and written (ABI 7) -- the consumer reads row run_start[r] (moda_mlp_warp_fwd does). */
int moda_fold_rows(int32_t n, const float* const* X, const int64_t* R, const int64_t* K, const int64_t* ldx,
                   const float* const* W, const int64_t* O, const int64_t* ldw, const int64_t* col0,
                   const float* const* b, float* const* Y, const int64_t* ldy,
                   const int32_t* const* run_start, void* stream);

/* Y[r, o] = b[o] + sum_k W[o, col0 + k] * X[r, k]   (the per-row fold used by moda_mlp_fwd;
 * also the plain nn.Linear of the compatibility path).  W is (O, ldw) row-major.  act: 0 none, 1 relu, 2 sigmoid. */
int moda_linear_fwd(const float* X, int64_t R, int64_t K, int64_t ldx,
                    const float* Wt, int64_t O, int64_t ldw, int64_t col0,
                    const float* b, int32_t act, float* Y, int64_t ldy, void* stream);

/* Embedding.forward (nerf.py:35-75): out (M, C*(1+2F); row stride ldo) = [x, w_k sin(2^k x), w_k cos(2^k x)]_k.
 * normalize != 0 first divides each row by its 2-norm (rays_d / rays_d.norm, rendering.py:64). */
int moda_embed_fwd(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window,
                   int32_t normalize, float* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------
 * Skinning and dual-quaternion warp  (nnutils/geom_utils.py)
 * ------------------------------------------------------------------------ */

/* bone_transform, neudbs branch (geom_utils.py:59-111): bones (B,10), rts (N,B,8) -> out (N,B,10).
 * run_start (N int32, moda_row_runs)|NULL: only the rows that START a run of identical rows are written (ABI 7); the other rows of
 * `out` are left as they are -- for consumers that read a run's first row (moda_warp_tables_fwd with the same run_start). */
int moda_bone_transform_fwd(const float* bones, const float* rts, int64_t N, int32_t B, float* out,
                            const int32_t* run_start, void* stream);

/* Floats of caller-provided workspace for moda_skinning_fwd / moda_warp_fwd (per-bone data hoisted out of
 * the per-sample loop: centre, rotation matrix, exp(scale); the (inverted) dual quaternions). */
int64_t moda_warp_workspace_floats(int64_t N, int32_t B, int32_t bones_per_ray);

/* skinning (geom_utils.py:237-302): softmax_B(-10*100*exp(skin_aux[0]) * sum_k s_k (R^T(c-p))_k^2 + dskin).
 *   bones (N,B,10) if bones_per_ray else (B,10);  pts (N,S,3);  dskin (N,S,B)|NULL;  skin (N,S,B) */
int moda_skinning_fwd(const float* bones, int32_t bones_per_ray, const float* pts, const float* dskin,
                      const float* skin_aux, int64_t N, int64_t S, int32_t B, float* skin, float* workspace,
                      void* stream);

/* dqs_blend_skinning (geom_utils.py:457-517): dq (N,B,8), skin (N,S,B), pts (N,S,3) -> out (N,S,3).
 * invert != 0 applies dq_inverse (dual_quat.py:87-94) to dq first (neu_dbs backward=True, geom_utils.py:388). */
int moda_dqs_fwd(const float* dq, int32_t invert, const float* skin, const float* pts,
                 int64_t N, int64_t S, int32_t B, float* out, void* stream);

/* Fused gauss_mlp_skinning tail + neu_dbs (rendering.py:304-319 / :330-341): skinning weights from
 * bones and dskin, then the DQS warp, in one pass.  dskin_bns != 0: dskin is stored (N,B,S) (what
 * moda_mlp_fwd writes with out_tr_S, coalesced on both sides) instead of (N,S,B).
 * skin_out (N,S,B)|NULL, cyc_ref (N,S,3)|NULL: when given, cyc_out (N,S) = |cyc_ref - xyz_out|
 * (frame_cyc_dis, rendering.py:341).  workspace: moda_warp_workspace_floats(N,B,bones_per_ray) floats. */
int moda_warp_fwd(const float* bones, int32_t bones_per_ray, const float* dq, int32_t invert,
                  const float* pts, const float* dskin, int32_t dskin_bns, const float* skin_aux,
                  int64_t N, int64_t S, int32_t B,
                  float* xyz_out, float* skin_out, const float* cyc_ref, float* cyc_out, float* workspace, void* stream);

/* moda_warp_fwd for the frame-grouped ray layout: the rays of one frame are consecutive and share their bone
 * transforms, so the per-frame tables are passed once instead of repeated per ray (what moda.update_rays does,
 * moda.py:1281-1311: bone_rts (F,8B) -> .repeat -> (N,8B)).  dq (N/rays_per_set, B, 8); bones (N/rays_per_set, B, 10) if
 * bones_per_set else (B,10); N a multiple of rays_per_set; rays_per_set = 1 is moda_warp_fwd.
 * pts_tf (N,S,3)|NULL: the points the blended transform is applied to when they differ from the points `pts` the
 * skinning weights are evaluated at (neu_dbs forward with nerf_dis: x + nerf_dis(x), geom_utils.py:420-425).
 * workspace: moda_warp_workspace_floats(N/rays_per_set, B, bones_per_set) floats. */
int moda_warp_frames_fwd(const float* bones, int32_t bones_per_set, const float* dq, int64_t rays_per_set, int32_t invert,
                         const float* pts, const float* pts_tf, const float* dskin, int32_t dskin_bns, const float* skin_aux,
                         int64_t N, int64_t S, int32_t B,
                         float* xyz_out, float* skin_out, const float* cyc_ref, float* cyc_out, float* workspace,
                         void* stream);

/* ------------------------------------------------------------------------
 * Fused skin MLP + skinning softmax + DQS warp  (one kernel per warp direction)
 *   replaces the chain  gauss_mlp_skinning geom_utils.py:202-217 (nerf_skin through evaluate_mlp :19-57 + skinning
 *   :237-302)  ->  neu_dbs :372-456 / dqs_blend_skinning :457-517, as rendering.py:304-319 (backward warp) and
 *   :330-341 (forward warp + cycle distance) call it.  The per-bone logit offsets never leave the registers.
 * ------------------------------------------------------------------------ */

/* Bone tiles (of 32) the tables below hold per set. */
int32_t moda_warp_tiles(int32_t B);

/* Per-set MFMA operand tables (layouts: moda_amd/csrc/moda_dev.h).
 *   bones (n_bone_sets,B,10) -> qtab: n_bone_sets * tiles * 320 floats  (Gaussian logits as quadratic forms, fp32);
 *   dq    (n_dq_sets,B,8)    -> dqtab: n_dq_sets * tiles * 2048 bytes   (dual quaternions, or their inverses
 *   (dual_quat.py:87-94) with invert=1, as bf16 hi/lo pairs).  Either count may be 0 (that table is not written). B <= 64.
 *   run_start (int32 per set, moda_row_runs)|NULL: only the slots of sets that START a run of identical sets are written; the
 *   kernel below reads a set's tables at run_start[set].  (The reference's ray layout repeats each frame's rows per ray.) */
int moda_warp_tables_fwd(const float* bones, int64_t n_bone_sets, const float* dq, int64_t n_dq_sets, int32_t invert,
                         const float* skin_aux, int32_t B, float* qtab, void* dqtab, const int32_t* run_start, void* stream);

/* run_start[n] = index of the first row of the run of bit-identical consecutive rows that row n belongs to; a row is
 * rows_a[n] (floats_a floats) and, when given, rows_b[n] (floats_b floats).  workspace: ceil(N / 256) int32. */
int moda_row_runs(const float* rows_a, int64_t floats_a, const float* rows_b, int64_t floats_b, int64_t N,
                  int32_t* run_start, int32_t* workspace, void* stream);
/* The same over up to four row sources at once (a row starts a run when it differs from its predecessor in ANY source): one
 * partition that is valid for every per-frame tensor of a `rays` dict -- bone_rts, time_embedded, env_code (ABI 7). */
int moda_row_runs_multi(int32_t n_src, const float* const* rows, const int64_t* floats, int64_t N, int32_t* run_start,
                        int32_t* workspace, void* stream);

/* xyz_out[m] = DQS(softmax_b(gauss_b(xyz[m]) + nerf_skin([PE(xyz[m]), code])_b), pts_tf[m] or xyz[m]).
 *   d: the skin net (W = 64, bf16 flag, raw outputs, n_out = B <= 64); wstream / bias / rb1 / rb5 / R1 / div1 as moda_mlp_fwd;
 *   rbd (32): the dir_encoding bias row (with xyz_encoding_final's bias folded in, as above);  M = rays * S samples, S % 32 == 0 (the samples of a ray are consecutive);
 *   qtab with q_rps rays per bone set (0: one set shared by all rays);  dqtab with dq_rps >= 1 rays per transform set;
 *   pts_tf (M,3)|NULL;  cyc_ref (M,3)|NULL -> cyc_out (M) = |cyc_ref - xyz_out| (rendering.py:341);  xyz_out may be NULL when
 *   cyc_ref is given (only the cycle distance is wanted: 12 bytes per sample less to write, ABI 7);
 *   run_start (moda_row_runs over the sets)|NULL: tables are read at run_start[set] and -- with MODA_MLP_ROWS_AT_RUNS in
 *   d->reserved and one code row per set (R1 > 1, div1 = S * dq_rps) -- rb1 / rb5 at row run_start[set] (rows folded only at
 *   run starts, moda_fold_rows; the partition must then also be one of the code rows: moda_row_runs_multi).
 *   d->flags: MODA_MLP_BF16, MODA_MLP_F16 or MODA_MLP_BF16X3 (ABI 7).
 * One-MFMA / split modes only: returns MODA_ESHAPE for anything else (the caller then runs moda_mlp_fwd + moda_warp_frames_fwd). */
int moda_mlp_warp_fwd(const moda_mlp_desc* d, const void* wstream, const float* bias, const float* xyz,
                      const float* rb1, const float* rb5, int64_t R1, int64_t div1, const float* rbd, const float* qtab,
                      int64_t q_rps, const void* dqtab, int64_t dq_rps, const float* pts_tf, const float* cyc_ref,
                      float* xyz_out, float* cyc_out, int64_t S, int64_t M, const int32_t* run_start, void* stream);

/* ------------------------------------------------------------------------
 * Ray sampling and compositing  (nnutils/rendering.py)
 * ------------------------------------------------------------------------ */

/* rendering.py:64-89: z_vals (N,S) (depth- or disparity-linear, optional stratified jitter with the
 * caller's uniforms u (N,S)|NULL scaled by perturb) and xyz (N,S,3) = o + d z. */
int moda_sample_rays_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                         const float* u, float perturb, int32_t use_disp, int64_t N, int64_t S,
                         float* z_vals, float* xyz, void* stream);

/* xyz (N,S,3) = o + d z for given z_vals (rendering.py:112-113) */
int moda_points_fwd(const float* rays_o, const float* rays_d, const float* z_vals, int64_t N, int64_t S,
                    float* xyz, void* stream);

/* inference() tail (rendering.py:183-237): SDF->density, alpha, exclusive transmittance product, sums.
 *   rgbsigma (N,S,4) [rgb, raw sigma];  feat (N,S,F)|NULL;  noise (N,S)|NULL (already * noise_std);
 *   clip_bound (3)|NULL with xyz (N,S,3): alpha=0 where |xyz|>bound;  vis_pred (N,S)|NULL: alpha=0 where <0.5;
 *   cyc (N,S)|NULL -> cyc_out (N) = sum_S cyc*w (rendering.py:473).
 *   outputs: rgb (N,3), feat_out (N,F)|NULL, depth (N), sil (N) (excludes last sample), weights (N,S),
 *   visibility (N,S)|NULL, vis_out (N)|NULL = sum_S vis_pred*w (rendering.py:408).
 *   rgb_filter_scale > 0: opts.rgb_filter -- rgb = sum_{s<S-1} w * scale_rgb * sigmoid(-10 sigma_raw) * rgb_s
 *   (rendering.py:171, 225-230) with scale_rgb = rgb_filter_scale; <= 0: rgb = sum_s w rgb_s.
 *   Early ray termination -- opt-in, NOT reference behaviour (the reference composes all S samples, rendering.py:217-221):
 *   n_live (N) int32|NULL: samples s >= n_live[n] get weight 0 and their inputs are never read (they may be
 *   uncomputed, see moda_mlp_live_fwd);  term_tau in [0,1): with term_tau > 0 the samples whose incoming transmittance
 *   T_s < term_tau get weight 0 (at most term_tau of a ray's weight is dropped); n_used (N) int32|NULL receives the
 *   number of samples per ray that kept their weight.  NULL / 0 / NULL = the reference's arithmetic. */
int moda_composite_fwd(const float* rgbsigma, const float* feat, int32_t F, const float* z_vals,
                       const float* rays_d, const float* beta, const float* noise,
                       const float* xyz, const float* clip_bound, const float* vis_pred, const float* cyc,
                       float rgb_filter_scale, int64_t N, int64_t S,
                       float* rgb, float* feat_out, float* depth, float* sil, float* weights,
                       float* visibility, float* vis_out, float* cyc_out,
                       const int32_t* n_live, float term_tau, int32_t* n_used, void* stream);

/* sample_pdf (rendering.py:582-623): bins (N,n_bins), weights (N,n_bins-1) -> samples (N,n_importance);
 * u (N,n_importance) uniforms, or NULL for the deterministic linspace(0,1,n_importance) (det=True). */
int moda_sample_pdf_fwd(const float* bins, const float* weights, const float* u, int64_t N, int32_t n_bins,
                        int32_t n_importance, float* samples, void* stream);

/* out (N, La+Lb) = sort(cat(a (N,La), b (N,Lb)), -1)  (torch.sort of the merged depths, rendering.py:110) */
int moda_merge_sort_fwd(const float* a, int32_t La, const float* b, int32_t Lb, int64_t N, float* out, void* stream);

/* moda_merge_index_fwd (ABI 9, round 6): the same sorted depths z_out (N, La+Lb) PLUS their origin src (N, La+Lb) int32 = index into
 * cat(a, b) (< La: a coarse depth; equal keys: a first, then by index).  rendering.py:96-114 evaluates every network at the coarse
 * depths twice -- in the no-grad pre-pass (:96-104) and again inside the merged final pass (:116) -- the same pointwise functions at
 * the same points; with the origin known, the final pass evaluates the importance depths only and
 * moda_merge_rows_fwd: out (N, L, C)[n][p] = src[n][p] < La ? a (N, La, C)[n][src] : b (N, L-La, C)[n][src-La]
 * merges the pre-pass's results (warped positions, colour + density) back in. */
int moda_merge_index_fwd(const float* a, int32_t La, const float* b, int32_t Lb, int64_t N, float* z_out, int32_t* src, void* stream);
int moda_merge_rows_fwd(const int32_t* src, int64_t N, int32_t L, int32_t La, int32_t C, const float* a, const float* b, float* out,
                        void* stream);

/* vec_to_sim3 (geom_utils.py:187-199): vec (n,10) -> center (n,3), orient (n,3,3), scale (n,3) */
int moda_vec_to_sim3_fwd(const float* vec, int64_t n, float* center, float* orient, float* scale, void* stream);

/* ------------------------------------------------------------------------
 * Training path (exact fp32): what torch autograd runs for the reference (nn.Linear / ReLU / sigmoid backward,
 * nerf.py:147-198; Embedding backward, nerf.py:35-75)
 * ------------------------------------------------------------------------ */

/* C[M,N] (row-major, ldc) = act(op(A) op(B) + bias) with arbitrary element strides
 *   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn];  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   bias (N)|NULL; act 0 none / 1 relu / 2 sigmoid; mask_src (M,N; ldc)|NULL zeroes C where mask_src <= 0
 *   (ReLU backward fused); accumulate != 0: C += result by atomics, split_k > 1 splits K over blockIdx.z. */
int moda_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                  float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const float* bias, int32_t act,
                  const float* mask_src, int32_t accumulate, int32_t split_k, void* stream);

/* Extended, pipelined form of moda_gemm_f32 used by the whole-network training Functions.  Each operand needs ONE
 * unit stride: A either k-fast (sak == 1) or m-fast (sam == 1), B either k-fast (sbk == 1) or n-fast (sbn == 1).
 *   A2/sam2/K1      optional second k-fast source of A for k >= K1 (the skip layer's cat[input_xyz, h], nerf.py:174-175,
 *                   without materialising the concatenation); NULL = none.
 *   rowbias         (ceil(M / rows_per_bias), ld_rowbias): added as rowbias[m / rows_per_bias][n] -- the per-ray part of
 *                   a layer input (pose / env codes, direction embedding) folded through its weight columns.
 *   mask_src        (M,N; ld_mask)|NULL: result zeroed where mask_src <= 0 (ReLU backward of the layer below).
 *   accumulate      0: C = epi(acc);  1: atomic C += (split_k > 1 allowed; C pre-zeroed);  2: C = epi(C + acc). */
typedef struct moda_gemm_desc {
    const float* A; int64_t sam, sak;
    const float* A2; int64_t sam2; int64_t K1;
    const float* B; int64_t sbk, sbn;
    float* C; int64_t ldc;
    int64_t M, N, K;
    const float* bias;
    const float* rowbias; int64_t ld_rowbias; int64_t rows_per_bias;
    const float* mask_src; int64_t ld_mask;
    int32_t act, accumulate, split_k, reserved;   /* reserved: MODA_GEMM_* flags or 0 */
    float* a_sum;            /* m-fast A only: a_sum[m] += sum_k A(m,k) in the same pass (bias gradient next to dW); or NULL */
    void* mask_bits;         /* NULL, or a 1-bit-per-element sign map, row r at mask_bits + r * ld_bits bytes, bit (n & 7) of byte n >> 3
                              * = [element (r, n) > 0] (the bf16-native forms and the split-bf16 forms of gemm_x3.hip; MODA_ESHAPE otherwise):
                              *   dW form (A m-fast, accumulate 1): WRITTEN for B -- the map of B(k, n) > 0 over all (k, n), a by-product
                              *     of the pass that reads B anyway (B = a layer's saved activations);
                              *   dX form (A k-fast): READ instead of mask_src -- C(m, n) zeroed where bit (m, n) is clear.  The ReLU mask
                              *     of a 256-wide layer is then 8 MB instead of the 134 MB of activations the epilogue read before. */
    int64_t ld_bits;
} moda_gemm_desc;
/* Throughput mode of the training route: both operands rounded to bf16 (nearest even) on their way into the MFMA, products
 * and sums in fp32 (v_mfma_f32_32x32x16_bf16).  Without the flag the GEMM is exact fp32 (v_mfma_f32_32x32x2_f32).  The same
 * bit in moda_nerf_train_desc.reserved selects it for every GEMM of that network's forward and backward. */
#define MODA_GEMM_BF16 1
/* Storage types: the named operand's elements are bf16 in memory (pointers are still passed as float*, strides and leading
 * dimensions still count ELEMENTS); values are fp32 once loaded, C is rounded to nearest even at the store.  C as bf16 is
 * refused with accumulate == 1 (the split-K atomics are fp32).  A2, bias, rowbias and a_sum are always fp32. */
#define MODA_GEMM_A_BF16 2
#define MODA_GEMM_B_BF16 4
#define MODA_GEMM_C_BF16 8
#define MODA_GEMM_MASK_BF16 16
/* Parity-grade mode on the bf16 matrix cores ("split-bf16"): every fp32 operand value is split into hi = bf16(v) and
 * lo = bf16(v - hi) -- 16 significand bits together, 2^-17 relative instead of the 2^-9 of MODA_GEMM_BF16 -- and each product is
 * lo*hi + hi*lo + hi*hi (three v_mfma_f32_32x32x16_bf16, fp32 sums).  Results sit within ~1e-6 (relative to the largest output)
 * of the exact-fp32 GEMM; the training route in this mode is held to the SAME gradient bars as the exact mode.  Operands are
 * fp32 in memory: excludes MODA_GEMM_BF16 and the storage-type flags (MODA_EINVAL).  The three large forms of a Linear layer
 * (forward: A and B k-fast; dX: A k-fast, B n-fast; dW: A m-fast, B n-fast, accumulate 1) with 16-byte aligned operands run on
 * their own kernels (gemm_x3.hip: operands split once at staging time, bf16 LDS images, transposing reads); everything else on
 * the generic kernel with the same arithmetic.  In moda_nerf_train_desc.reserved it selects the mode for every GEMM of the
 * per-layer forward and the backward. */
#define MODA_GEMM_BF16X3 32
/* The same with THREE bf16 images per operand, hi + mid + lo = the fp32 value exactly, and six MFMAs per product (every term down
 * to 2^-18 of the product; the dropped ones are <= 2^-26): the accuracy class of the exact-fp32 GEMM -- differences are those of
 * the summation order -- at 16/6 of its matrix rate, on kernels bound by HBM.  The fast parity mode of the training route. */
#define MODA_GEMM_BF16X6 64
/* moda_nerf_train_desc.reserved, next to MODA_GEMM_BF16: the workspace of moda_nerf_train_fwd_fused keeps h / dd / fin as
 * bf16 and moda_nerf_train_bwd keeps dh / d_dir_encoding / d_final as bf16 in `scratch` (same element offsets and sizes as the
 * fp32 layout, the second half of each slot unused).  Must be the same in the forward and the backward call of one step;
 * moda_nerf_train_fwd (the per-layer fp32 forward) refuses it. */
#define MODA_TRAIN_BF16_STORE 2
/* moda_mlp_desc.reserved of moda_mlp_dump_fwd: dump_h / dump_dd receive bf16 elements (same element offsets). */
#define MODA_MLP_DUMP_BF16 1
#define MODA_MLP_ROWS_AT_RUNS 2   /* moda_mlp_desc.reserved, moda_mlp_warp_fwd with run_start: rb1 / rb5 hold valid rows only at
                                    run starts (moda_fold_rows with the same run_start): row run_start[set] is read (ABI 7) */
int moda_gemm_f32_ex(const moda_gemm_desc* d, void* stream);

/* One NeRF (Embedding + nerf.py:147-198) of the training route, every launch of its forward or backward from one call.
 *   xyz (M,3); code (R1,C1)|NULL the per-ray part of input_xyz; dir_src (Rd,Cd)|NULL the per-ray input_dir (R | M).
 *   params / grads: 2D+8 device pointers in module order -- (W_i, b_i) for the D xyz_encoding layers, then sigma,
 *   xyz_encoding_final, dir_encoding, rgb (weight, bias) -- in the reference's shapes.  Every parameter gradient is ADDED to
 *   what its buffer holds: zero-fill for a fresh gradient, or let several calls (one network evaluated more than once in a step)
 *   accumulate into one buffer.
 *   ws: moda_nerf_train_ws_floats(d) floats written by the forward (positional encoding, activations, packed weights,
 *   folded row biases) and read by the backward; scratch: moda_nerf_train_scratch_floats(d) floats.
 *   out (M, n_out [+1]) = [sigmoid(rgb) | sigma], or rgb alone for raw_feat, or sigma alone for sigma_only. */
typedef struct moda_nerf_train_desc {
    int32_t D, W, P, C1, Cd, n_out, raw_feat, sigma_only, n_freq, reserved;
    float window[16];
    int64_t M, R1, Rd;
} moda_nerf_train_desc;
int64_t moda_nerf_train_ws_floats(const moda_nerf_train_desc* d);
int64_t moda_nerf_train_scratch_floats(const moda_nerf_train_desc* d);
int moda_nerf_train_fwd(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                        const float* const* params, float* ws, float* out, void* stream);
/* The same forward in the throughput mode: ONE launch of the fused bf16 PE+MLP kernel writes every layer's activations into
 * `ws` (moda_mlp_dump_fwd) in place of one GEMM per layer; wstream / bias_block / bd_folded are the packed bf16 weight stream,
 * the bias block and the folded dir bias of moda_mlp_fwd.  W in {64, 128, 256}, not sigma_only (MODA_ESHAPE otherwise: use
 * moda_nerf_train_fwd).  The backward is moda_nerf_train_bwd, unchanged. */
int moda_nerf_train_fwd_fused(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                              const float* const* params, const void* wstream, const float* bias_block, const float* bd_folded,
                              float* ws, float* out, void* stream);
int moda_nerf_train_bwd(const moda_nerf_train_desc* d, const float* xyz, const float* code, const float* dir_src,
                        const float* const* params, const float* ws, const float* out, const float* g_out, float* scratch,
                        float* const* grads, float* d_xyz, float* d_code, float* d_dir, void* stream);

/* out[r, n] = sum_{s<S} X[(r*S + s)*ld + n]: per-ray sums over the S samples (gradient of a folded per-ray bias). */
int moda_segsum_f32(const float* X, int64_t R, int64_t S, int64_t N, int64_t ld, float* out, int64_t ldo, void* stream);

/* out[n] += sum_m X[m*ld + n]   (bias gradients) */
int moda_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* stream);

/* grad_x (M,C) from grad_out (M, C*(1+2F); row stride ldg) of moda_embed_fwd (same window / normalize arguments) */
/* moda_embed_jvp (round 5): the tangent of the encoding at constant x, out (M, ldo >= C (1 + 2 n_freq)) = J(x[m]) u[m] with u (M,C)
 * -- the transpose of moda_embed_bwd's product; the backward of the eikonal term's J^T g w.r.t. g (loss_utils.py:20-46). */
int moda_embed_jvp(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window, const float* u,
                   float* out, int64_t ldo, void* stream);
int moda_embed_bwd(const float* x, int64_t M, int32_t C, int32_t n_freq, const float* window, int32_t normalize,
                   const float* grad_out, int64_t ldg, float* grad_x, void* stream);

/* moda_sum_tensors (ABI 9, round 6): out (numel) = xs[0] + ... + xs[n-1], n <= 8, summed in argument order, 16-byte aligned
 * operands.  The reference lets PyTorch's autograd accumulate the gradients of a tensor that several nodes consume -- the warped
 * sample positions of rendering.py:319 feed the rest-pose skin network (:330), the forward warps (:338-360), the colour /
 * density network (:159), the feature network (:174-178) and the matching heads (:410-437) -- with one `add` launch per extra
 * consumer; here such a tensor is fanned out explicitly (autograd.FanOutFn) and its gradients meet in ONE launch. */
int moda_sum_tensors(const float* const* xs, int32_t n, int64_t numel, float* out, void* stream);

/* moda_affine3 (ABI 9, round 6): out (rows,3) = y + (x * scale[c]) * post + shift[c] (y, shift nullable; scale, shift (3,)), each
 * operation rounded on its own in this order: the lattice jitter `query + randn * bound * 0.05` of feat_match
 * (loss_utils.py:304-306) and the negatives `rand * 2 * bound - bound` of visibility_loss (loss_utils.py:137-138) bit for bit as
 * the eager expressions, one launch each instead of three. */
int moda_affine3(const float* x, const float* y, const float* scale, float post, const float* shift, int64_t rows, float* out,
                 void* stream);

/* dz = dy * act'(y): act 1 relu, 2 sigmoid */
int moda_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float* dz, void* stream);

/* Backward of moda_composite_fwd (rendering.py:183-237, torch autograd in the reference).  Takes the forward's inputs,
 * its saved weights / visibility, and the upstream gradients g_* (any may be NULL = zero); writes d_rgbsigma (N,S,4),
 * d_feat, d_cyc and ACCUMULATES into d_z (N,S), d_rays_d (N,3) (through |d|), d_beta (1). */
int moda_composite_bwd(const float* rgbsigma, const float* feat, int32_t F, const float* z_vals, const float* rays_d,
                       const float* beta, const float* noise, const float* xyz, const float* clip_bound,
                       const float* vis_pred, const float* cyc, const float* weights, const float* visibility,
                       float rgb_filter_scale, int64_t N, int64_t S, const float* g_rgb, const float* g_feat, const float* g_depth,
                       const float* g_sil, const float* g_weights, const float* g_cyc, float* d_rgbsigma, float* d_feat,
                       float* d_z, float* d_rays_d, float* d_beta, float* d_cyc, void* stream);

/* Backward of xyz = o + d z (rendering.py:88-89): accumulates d_rays_o (N,3), d_rays_d (N,3), d_z (N,S). */
int moda_points_bwd(const float* d_xyz, const float* z_vals, const float* rays_d, int64_t N, int64_t S,
                    float* d_rays_o, float* d_rays_d, float* d_z, void* stream);

/* The warp on prepared per-bone data -- prep (nsets,B,16) = [centre | R row-major | exp(scale) | 0] and the dual
 * quaternions q (N,B,8) that are blended as they are -- and its backward (skin (N,S,B) saved by the forward).
 * The backward writes d_pts (N,S,3), d_dskin (N,S,B), d_ref (N,S,3), d_q (N,B,8) and the per-ray d_prep_ray (N,B,16)
 * (to be summed over rays by the caller when the bones are shared), accumulates d_aux0 (1); d_bl (N,S,8) is scratch.
 * pts_tf (N,S,3)|NULL: when given, the weights are evaluated at pts and the blended transform is applied to pts_tf
 * (neu_dbs forward with a residual field, geom_utils.py:420-425); the backward then writes the transform's gradient to
 * d_pts_tf and only the weights' gradient to d_pts. */
int moda_warp_prepped_fwd(const float* prep, int32_t per_ray, const float* q, const float* pts, const float* pts_tf,
                          const float* dskin, int32_t dskin_bns, const float* skin_aux, int64_t N, int64_t S, int32_t B, float* xyz_out,
                          float* skin_out, const float* cyc_ref, float* cyc_out, void* stream);
int moda_warp_prepped_bwd(const float* prep, int32_t per_ray, const float* q, const float* pts, const float* pts_tf,
                          float* d_pts_tf, const float* skin,
                          const float* skin_aux, const float* cyc_ref, const float* g_out, const float* g_cyc,
                          const float* g_skin, int64_t N, int64_t S, int32_t B, float* d_pts, float* d_dskin,
                          float* d_prep_ray, float* d_q, float* d_aux0, float* d_ref, float* d_bl, void* stream);

/* Per-(ray, bone) preparation with its backward (the reference differentiates these through eager ops):
 *   moda_bone_prep:          bones (n,10) -> prep (n,16) = [centre | matrix(q/|q|) row-major | exp(log scale) | 0]
 *                            (vec_to_sim3, geom_utils.py:187-199); with g_prep (n,16) given it writes d_bones (n,10) instead.
 *   moda_bone_transform_bwd: backward of moda_bone_transform_fwd: g_out (N,B,10) -> d_rts (N,B,8) and the per-ray
 *                            partial d_bones_ray (N,B,10) (the caller sums it over rays).
 *   moda_dq_inverse_bwd:     backward of dq_inverse (dual_quat.py:87-94): g_out (n,8) -> d_dq (n,8). */
int moda_bone_prep(const float* bones, int64_t n, float* prep, const float* g_prep, float* d_bones, void* stream);
int moda_bone_transform_bwd(const float* bones, const float* rts, int64_t N, int32_t B, const float* g_out,
                            float* d_bones_ray, float* d_rts, void* stream);
int moda_dq_inverse_bwd(const float* dq, const float* g_out, int64_t n, float* d_dq, void* stream);

/* ------------------------------------------------------------------------
 * Correspondence heads of inference_deform (rendering.py:439-499)
 * ------------------------------------------------------------------------ */

/* obj_to_cam + pinhole_cam (geom_utils.py:567-581, 654-673) with K = mat2K(Kmatinv(Kinv)) (rendering.py:443-449):
 * xyz (N,S,3), rtk_vec (N,21) = [R 9 | T 3 | Kinv 9] -> out (N,S,3) = (u, v, Z).  Backward writes d_xyz, d_rtk_vec. */
int moda_project_fwd(const float* xyz, const float* rtk_vec, int64_t N, int64_t S, float* out, void* stream);
int moda_project_bwd(const float* xyz, const float* rtk_vec, const float* g_out, int64_t N, int64_t S, float* d_xyz,
                     float* d_rtk_vec, void* stream);

/* vrender_flo (geom_utils.py:1704-1743): weights (N,S), proj (N,S,3) from moda_project_fwd, xys (N,2) ->
 * flo (N,2), valid (N).  With g_flo (N,2) given it runs the backward instead and writes d_weights, d_proj. */
int moda_flow_render(const float* weights, const float* proj, const float* xys, float img_size, int64_t N, int64_t S,
                     float* flo, float* valid, const float* g_flo, float* d_weights, float* d_proj, void* stream);

/* compute_pts_exp (loss_utils.py:165-175): out (N,3) = sum_s w_s/(1e-9+sum w) pts_s; backward when g_out is given. */
int moda_pts_exp(const float* weights, const float* pts, int64_t N, int64_t S, float* out, const float* g_out,
                 float* d_weights, float* d_pts, void* stream);

/* ------------------------------------------------------------------------
 * Loss heads behind compositing (rendering.py:410-437, 475-477, 573-578)
 * ------------------------------------------------------------------------ */

/* F.normalize(x, 2, -1) on rows: y (M,F) = x / max(|x|, 1e-12); with g (M,F) given it writes the backward dx instead. */
int moda_normalize_rows(const float* x, int64_t M, int32_t F, float* y, const float* g, float* dx, void* stream);

/* feat_match (loss_utils.py:273-405), N pixels against G = 20^3 canonical grid points, F = 16 CSE channels.
 *   moda_match_matrix:  Kmat (N,G) = exp((<feats_n[n], vol_n[g]> - 1) * kappa[0]) on L2-normalised rows;
 *                       kappa = 1/0.03 is the Sinkhorn kernel K of loss_utils.py:340, kappa = |beta|+1e-9 the
 *                       softmax form of :331-332, :376 (softmax is shift-invariant).
 *                       The transposed copy KmatT (G,N) is the same call with the operands swapped.
 *   moda_match_sweep:   out (R) = epi(sum_c Mat[r,c] vec[c]) for Mat = Kmat (R=N, C=G) or KmatT (R=G, C=N).
 *                       epi mode 0: the sum; 1: p / (sum + 1e-8) (one Sinkhorn update, :363-369);
 *                       2: -sum * c^2 / p (reverse-mode step through that update, c = the saved a_t / b_t).
 *   moda_match_expect:  rowsum (N) = sum_g Kmat b;  pred (N,3) = sum_g (Kmat b / rowsum) query[g]  (:371-374, :389);
 *                       b (G)|NULL = column scaling of the last Sinkhorn iteration (NULL: ones, softmax form).
 *   moda_match_ecols:   (on KmatT) ubar (G) = -(sum_n e[n,g]) b[g] / p2 with e = prob (<g_pred, q> - <g_pred, pred>): seeds the
 *                       reverse sweep through the 20 Sinkhorn iterations.
 *   moda_match_dbar:    Dbar (N,G) = kappa (e + Kmat (sum_t A[t,n] Ubar[t,g] + sum_t Wbar[t,n] Bm[t,g])): gradient w.r.t.
 *                       the dot products; kappa_bar (1)|NULL accumulates the gradient w.r.t. kappa.
 *   moda_match_prob:    prob (N,G) = Kmat b / rowsum written out -- needed only by the back-correspondence term of
 *                       use_corr (corr_err = |prob prob^T - I|_2 per row, loss_utils.py:386-391).  Its gradient comes back
 *                       into moda_match_ecols / moda_match_dbar as g_prob (N,G) (transposed (G,N) for ecols) with
 *                       s_prob[n] = sum_g g_prob[n,g] prob[n,g]; both NULL when unused: e += prob (g_prob - s_prob). */
/* kmat_bf16 (ABI 5): the matrix Kmat / KmatT / Mat of these six entries holds bf16 elements (same pointer type, same element
 * offsets) -- the throughput mode of the training route, whose 78 Sinkhorn sweeps per step read it 78 times; every vector,
 * sum and result stays fp32.  0: fp32 (the parity mode). */
int moda_match_matrix(const float* feats_n, const float* vol_n, int64_t N, int64_t G, int32_t F, const float* kappa,
                      float* Kmat, int32_t kmat_bf16, void* stream);
/* moda_match_matrix_rows (round 5): the matrix of feat_match(init_pts=...) -- every pixel n against ITS OWN lattice's features,
 * Kmat (N,G) = exp((<feats_n[n], vol_n[n,g]> - 1) * kappa[0]) with vol_n (N,G,F) L2-normalised rows; fp32 (loss_utils.py:322-335). */
int moda_match_matrix_rows(const float* feats_n, const float* vol_n, int64_t N, int64_t G, int32_t F, const float* kappa,
                           float* Kmat, void* stream);
int moda_match_sweep(const float* Mat, int64_t R, int64_t C, const float* vec, int32_t mode, float p,
                     const float* c, float* out, int32_t kmat_bf16, void* stream);
/* moda_match_sinkhorn (ABI 8): ALL sweeps of the 20 Sinkhorn iterations (loss_utils.py:361-370) as ONE persistent launch --
 * forward (backward = 0): A (iters+1, N) with row 0 = a_0 given -> Bm (iters, G), rows 1.. of A; backward (= 1): the reverse sweep
 * from Ubar row iters-1 (moda_match_ecols) -> Wbar (iters-1, N), Ubar (iters, G), reading A and Bm.  Same arithmetic as the
 * moda_match_sweep chain (sums in another association).  One workgroup per CU keeps its rows of Kmat in LDS and of KmatT in
 * registers across the sweeps, with a flag-array grid barrier between them: bf16 matrices only, N % 512 == 0, N <= 2048,
 * G % 8 == 0, G <= 32 x CUs, N <= 8 x CUs; flags: (flags_len >= CUs + 1) int32 ZEROS -- flags[CUs] is raised if a workgroup
 * timed out waiting for the others (results invalid).  It must run ALONE on the device (every workgroup resident at once): call
 * it on the stream the step runs on, with nothing concurrent.  MODA_ESHAPE: shape not served, use the per-sweep launches. */
int moda_match_sinkhorn(const void* Kmat, const void* KmatT, int64_t N, int64_t G, int32_t iters, int32_t backward,
                        float* A, float* Bm, float* Ubar, float* Wbar, int32_t* flags, int32_t flags_len, void* stream);
int moda_match_expect(const float* Kmat, const float* b, const float* query, int64_t N, int64_t G, float* pred,
                      float* rowsum, int32_t kmat_bf16, void* stream);
int moda_match_prob(const float* Kmat, const float* b, const float* rowsum, int64_t N, int64_t G, float* prob, int32_t kmat_bf16,
                    void* stream);
int moda_match_ecols(const float* KmatT, const float* b, const float* rowsum, const float* g_pred, const float* pred,
                     const float* query, const float* g_probT, const float* s_prob, int64_t N, int64_t G, float p2,
                     float* ubar, int32_t kmat_bf16, void* stream);
int moda_match_dbar(const float* Kmat, const float* b, const float* rowsum, const float* g_pred, const float* pred,
                    const float* query, const float* A, const float* Ubar, int32_t T1, const float* Wbar, const float* Bm,
                    int32_t T2, const float* g_prob, const float* s_prob, int64_t N, int64_t G, const float* kappa,
                    float* Dbar, float* kappa_bar, int32_t kmat_bf16, void* stream);

/* visibility_loss terms (loss_utils.py:125-149): out[0] += scale * sum_i -logsigmoid(sign * x_i) * (w_i | 1);
 * with g_out (1) given it writes dx (n) = g_out * d(out)/dx instead. */
int moda_logsig_loss(const float* x, const float* w, int64_t n, float sign, float scale, float* out, const float* g_out,
                     float* dx, void* stream);

/* ------------------------------------------------------------------------
 * Per-frame feeders of the path
 * ------------------------------------------------------------------------ */

/* raycast (geom_utils.py:746-794): xys (bs,ns,2) pixels, Rmat (bs,3,3), Tmat (bs,3), Kinv (bs,3,3) ->
 * rays_d (bs,ns,3) = (Kinv [x,y,1])^T R,  rays_o (bs,ns,3) = -T^T R.  With g_rays_d (and optionally g_rays_o) given it
 * runs the backward instead and writes d_Rmat, d_Tmat, d_Kinv (gradients towards root pose and intrinsics). */
int moda_raycast(const float* xys, const float* Rmat, const float* Tmat, const float* Kinv, int64_t bs, int64_t ns,
                 float* rays_d, float* rays_o, const float* g_rays_d, const float* g_rays_o, float* d_Rmat, float* d_Tmat,
                 float* d_Kinv, void* stream);

/* DQ_RTHead's tail (nerf.py:260-279): rts (n,7) = [t | q] -> dq (n,8) = [q/|q|, 1/2 (0, 0.1 t) (x) q/|q|]; with g_dq given
 * it writes d_rts (n,7) instead. */
int moda_rt_to_dq(const float* rts, int64_t n, float* dq, const float* g_dq, float* d_rts, void* stream);

/* ------------------------------------------------------------------------
 * Dual-quaternion algebra  (nnutils/dual_quat.py), elementwise over n rows
 * ------------------------------------------------------------------------ */
#define MODA_DQ_QMUL        0  /* q_mul        (dual_quat.py:14-31)  a,b (n,4) -> (n,4) */
#define MODA_DQ_DQMUL       1  /* dq_mul       (dual_quat.py:33-49)  a,b (n,8) -> (n,8) */
#define MODA_DQ_NORMALIZE   2  /* dq_normalize (dual_quat.py:51-62)  a (n,8) */
#define MODA_DQ_QCONJ       3  /* dq_quaternion_conjugate (:65-74) */
#define MODA_DQ_CCONJ       4  /* dq_combined_conjugate   (:76-85) */
#define MODA_DQ_INVERSE     5  /* dq_inverse   (dual_quat.py:87-94) */
#define MODA_DQ_QNORMALIZE  6  /* q_normalize  (dual_quat.py:4-12)   a (n,4) */
/* flag (device int32, may be NULL): set to 1 if a normalisation met a zero norm (the reference asserts). */
int moda_dq_op(int32_t op, const float* a, const float* b, int64_t n, float* out, int32_t* flag, void* stream);

/* xyz_encoding_final folded into dir_encoding (nerf.py:184-187: a Linear without activation feeding a Linear):
 * prod (W/2, W) = Wdir[:, :W] @ Wfin  (Wdir (W/2, ldd) row-major, Wfin (W, W)), and, when bd_out is given,
 * bd_out (W/2) = bdir + Wdir[:, :W] @ bfin.  fp32 fmaf chains; one launch. */
int moda_fold_final(const float* Wdir, int64_t ldd, const float* Wfin, const float* bfin, const float* bdir, int64_t W,
                    float* prod, float* bd_out, void* stream);

/* The per-ray img / sil / flo loss terms of inference_deform with their batch-level statistics (nnutils/rendering.py:518-571:
 * img_loss_samp = mean_c (rgb - img_at)^2 * sil_at; sil_loss_samp = (sil - sil_at)^2 * balance * vis_at with the class balance of
 * :535-539 when `training`; flo_loss_samp = |flo - flo_at|^2 * cfd_at / mean(cfd_at[sil_flo]) * sil_at; sil_flo = sil_at > 0 &
 * valid == 1 & cfd_at != 0) in ONE launch.  rgb, img_at (N,3); flo, flo_at (N,2); the rest (N).  stats: 8 floats written by the
 * forward and read by the backward.  Backward when d_rgb / d_sil / d_flo are given (g_* may be NULL = zero). */
int moda_ray_loss(const float* rgb, const float* sil, const float* flo, const float* valid, const float* img_at,
                  const float* sil_at, const float* vis_at, const float* flo_at, const float* cfd_at, int64_t N,
                  int32_t training, float* img_loss, float* sil_loss, float* flo_loss, uint8_t* sil_flo, float* stats,
                  const float* g_img, const float* g_sil, const float* g_flo, float* d_rgb, float* d_sil, float* d_flo,
                  void* stream);

/* x[mask].mean() as a trainer's loss assembly forms it (nnutils/moda.py:540-640), without the boolean gather (no host sync):
 * out2[0] = sum_i sum_c x[i, c] [mask_i != 0] / out2[1], out2[1] = k * #selected; x (N, k), mask (N).  Backward when dx is given:
 * dx[i, c] = g[0] [mask_i != 0] / out2[1]. */
int moda_masked_mean(const float* x, const float* mask, int64_t N, int32_t k, float* out2, const float* g, float* dx,
                     void* stream);

/* Per-row distance of two (N, F) arrays -- the small reductions of the loss heads: mean_sq == 0: out[i] = ||a_i - b_i||_2
 * (feat_err nnutils/loss_utils.py:200, the reprojection error :216-221); mean_sq != 0: out[i] = mean_c (a_ic - b_ic)^2 (the
 * rendered-feature error, nnutils/rendering.py:573-577).  g == NULL: forward.  g (N) != NULL: backward -- da and / or db (N, F)
 * written: da = g (a - b) / ||a - b|| (0 where the norm is 0) or g 2 (a - b) / F; db = -da. */
int moda_row_dist(const float* a, const float* b, int64_t N, int32_t F, int32_t mean_sq, float* out, const float* g, float* da,
                  float* db, void* stream);

/* The weighted sum of a trainer's loss terms (nnutils/moda.py:540-705: `w * x[mask].mean()` per term, summed) in one launch each
 * way.  Term t: x (n, k) values; mask NULL / mask_kind 0 = every row, 1 = float (n), selected where > 0, 2 = uint8 / bool (n),
 * selected where != 0, 3 = float, selected where != 0; weight.  `terms` is a HOST array of n_terms <= 16 entries.
 * g == NULL, forward: out[0] = sum_t term_t, out[1 + t] = term_t = weight_t * sum(selected x) / den_t, out[1 + n_terms + t] =
 * den_t = k_t * #selected (a term with nothing selected is NaN, the mean of an empty selection there).
 * g != NULL, backward: for every term with dx != NULL, dx (n, k) = g[0] * weight_t / den_t on the selected rows, 0 elsewhere
 * (`out` as the forward call left it; x may be NULL). */
typedef struct {
    const float* x;
    const void* mask;
    float* dx;
    int64_t n;
    int32_t k, mask_kind;
    float weight;
    int32_t reserved;
} moda_loss_term;
int moda_loss_terms(const moda_loss_term* terms, int32_t n_terms, float* out, const float* g, void* stream);

/* S3IM, opts.s3im_loss (nnutils/loss_utils.py:575-702 S3IM.forward + SSIM(window 4, stride 4) / _ssim; called at
 * nnutils/rendering.py:528-532):  loss[0] = 1 - mean SSIM over the Gaussian 4x4 / stride 4 / padding 1 windows of the
 * (3, patch_h, patch_w_total) virtual patch whose pixel (h, w) holds row index[h * patch_w_total + w] % N of rgb * mask and of
 * tar * mask (rgb, tar (N,3); mask (N); index int32, patch_h * patch_w_total entries = [identity, permutations ...]).
 * g_loss == NULL: forward.  g_loss != NULL: backward, d_rgb (N,3) += g_loss[0] * d loss / d rgb (caller zero-fills). */
int moda_s3im(const float* rgb, const float* tar, const float* mask, int64_t N, const int32_t* index, int32_t patch_h,
              int32_t patch_w_total, float* loss, const float* g_loss, float* d_rgb, void* stream);

/* Diagnostic: one pass of workgroups that fill the whole LDS of every CU with `pattern` (tools/poison_check.py: a kernel whose
 * result depends on LDS it has not written shows up as a difference between two patterns). */
int moda_dbg_poison_lds(uint32_t pattern, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MODA_HIP_H */
