/*
 * ovo_hip.h -- C ABI of libovo_hip.so, the MI355X (gfx950) implementation of OVO's per-frame
 * open-vocabulary feature path.
 *
 * The reference (tberriel/OVO) is pure Python/PyTorch and has NO FFI of its own:
 * its boundary is a set of duck-typed Python classes.  This header is the boundary we add underneath
 * those classes; every entry point names the reference function(s) it replaces (paths relative to the
 * reference checkout).  The Python side (ovo_amd/) binds it with ctypes: raw device pointers from
 * tensor.data_ptr(), sizes, and the caller's hipStream_t -- no torch types cross this line.
 *
 * Conventions
 *   - every function returns 0 on success, a negative OVO_E_* code otherwise; ovo_hip_last_error()
 *     returns a thread-local message for the last failure.
 *   - pointers are DEVICE pointers unless the parameter is a small by-value struct or says "host".
 *   - nothing allocates: scratch comes from the caller (sizes from the *_workspace_bytes() queries).
 *   - all launches go to `stream`; no function synchronises the device.
 *   - row-major, C-contiguous arrays; "f32[N,3]" means N rows of 3 floats.
 */
#ifndef OVO_HIP_H
#define OVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVO_OK 0
#define OVO_E_ARG (-1)     /* bad argument (null pointer, negative size, unsupported shape) */
#define OVO_E_LAUNCH (-2)  /* hip launch / runtime error                                     */
#define OVO_E_UNSUPPORTED (-3)

typedef void *ovo_stream_t; /* hipStream_t */

const char *ovo_hip_last_error(void);
int ovo_hip_abi_version(void); /* bumped when a signature changes */

/* Optional profiler for bench.py's roofline figures: between start and stop every launch of a profiled kernel
 * family is bracketed by hipEvents on its own stream.  Kinds (n_kinds <= 10): 1 = fused attention (work = flops),
 * 2 = fused point-map tracking pass (bytes), 4..7 = MFMA GEMM with tile 128x128 / 128x64 / 64x128 / 64x64, 3 / 0 = the 256x256 /
 * 256x128 ping-pong GEMM, 8 = the weights-resident streaming GEMM (flops), 9 = the dense scatter-accumulate (scan + apply launches of
 * ovo_scatter_accum_touched; work = bytes = hits x (2 x 4 D + 12) + 2 n, the hit count read back from the device after the launch).
 * stop synchronises the device and returns, per kind, total milliseconds, total work and launch count. */
/* An empty one-thread kernel (`k_marker`): bench.py brackets its timed region with two of them so that a rocprofv3 kernel trace of the
 * same command can be cut to exactly that region (tools/kstats_region.py). */
int ovo_marker(int id, ovo_stream_t stream);
int ovo_profile_start(void);
int ovo_profile_stop(double *ms, double *work, int64_t *launches, int n_kinds);
/* per kind, the ALGORITHMIC bytes of the launches of the last profiled region (GEMM: A + W + C once, + residual / bias operands; attention:
 * q, k, v, o once) -- the denominator PMC traffic (profiles/pmc_traffic.json) is compared with. */
int ovo_profile_bytes(double *bytes, int n_kinds);

/* ---------------------------------------------------------------------------------------------
 * Camera / frustum parameters for one frame.  Filled on the host (8-corner and 6-plane math is tiny
 * and stays in torch-CPU, see DESIGN.md "bit-exactness").
 *   aabb   = min xyz, max xyz of the 8 frustum corners      geometry_utils.py:205-215
 *   planes = 6 rows (a,b,c,d), inside <=> a x + b y + c z + d <= 0      geometry_utils.py:163-202,233-249
 *   w2c    = inverse camera pose, row-major 4x4                   ovo.py:216, vanilla_mapper.py:59
 *   K      = intrinsics, row-major 3x3
 *   th     = |z - depth[v,u]| threshold in metres                 ovo.yaml:25 / vanilla_mapper.py:17
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float aabb[6];
    float planes[24];
    float w2c[16];
    float K[9];
    float th;
    int32_t h, w; /* depth image size */
} ovo_camera_t;

/* rgb_depth_ratio of ovo.py:218-221: pixel (u,v) of the depth frame -> pixel of the colour/seg frame:
 * u' = trunc((u + crop_edge) * r_w), v' = trunc((v + crop_edge) * r_h).  enabled = 0 -> identity. */
typedef struct {
    int32_t enabled;
    float r_h, r_w;
    int32_t crop_edge;
} ovo_ratio_t;

/* ---- a2: compute_frustum_point_ids (geometry_utils.py:252-276) ------------------------------
 * out_idx receives the ascending indices of points inside the AABB and all six planes;
 * *out_count (device int64) their number.  ws: ovo_compact_workspace_bytes(n). */
size_t ovo_compact_workspace_bytes(int64_t n);
int ovo_frustum_ids(const float *pts, int64_t n, const ovo_camera_t *cam, int64_t *out_idx,
                    int64_t *out_count, void *ws, size_t ws_bytes, ovo_stream_t stream);

/* ---- a3: project_3d_points / match_3d_points_to_2d_pixels (geometry_utils.py:26-89) ----------
 * pts has `stride` floats per point (3: w := 1, or 4 homogeneous).  No frustum test here: like the
 * reference it assumes the caller already culled.  ovo_project_points writes i32[n,2] (u,v).
 * ovo_match_points writes the ascending indices (into pts) and pixels of the points whose pixel is
 * inside the image, |z - depth| < th and depth != 0. */
int ovo_project_points(const float *pts, int64_t n, int stride, const ovo_camera_t *cam, int32_t *out_uv,
                       ovo_stream_t stream);
int ovo_match_points(const float *pts, int64_t n, int stride, const ovo_camera_t *cam, const float *depth,
                     int64_t *out_idx, int32_t *out_uv, int64_t *out_count, void *ws, size_t ws_bytes,
                     ovo_stream_t stream);

/* ---- a4: depth_filter (geometry_utils.py:92-96) ----------------------------------------------
 * 7x7 Gaussian (sigma 2.5, reflect padding, torchvision kernel), |d - blur| > th -> -1. */
int ovo_depth_filter(const float *depth, int h, int w, int ksize, float sigma, float th, float *out,
                     ovo_stream_t stream);

/* ---- a2+a3+a5 fused, tracking form (ovo.py:208-222) --------------------------------------------
 * One pass over the whole map: cull, project, depth-test, colour-frame remap, seg_map gather.
 *   point_seg  i16[n]: -2 = not matched, -1 = matched on an unlabelled pixel, >=0 = mask index.
 *   hist       i32[n_masks, hist_cols] (zeroed here): votes[mask][ins_id+1], column 0 = unassigned
 *              points; hist_cols must be > max instance id + 1            (ovo.py:257-264 inputs)
 *   counters   i64[2]: {points in frustum, points matched}  (zeroed here)
 * point_ins: i32[n] current instance id per point (-1 = none). */
int ovo_track_project(const float *pts, const int32_t *point_ins, int64_t n, const ovo_camera_t *cam,
                      const float *depth, const int32_t *seg_map, int seg_h, int seg_w, ovo_ratio_t ratio,
                      int16_t *point_seg, int32_t *hist, int n_masks, int hist_cols, int64_t *counters,
                      ovo_stream_t stream);

/* ---- a6 (device half): per-mask vote statistics (ovo.py:255-264) -------------------------------
 * stats i32[n_masks,4] = {matched points, of which already assigned, mode of their instance ids
 * (smallest id among ties, -1 if none), mask area in seg_map pixels}. */
int ovo_vote_stats(const int32_t *hist, int n_masks, int hist_cols, const int32_t *seg_map, int64_t seg_pixels,
                   int32_t *stats, ovo_stream_t stream);

/* ---- a6 (write half) + a8: ovo.py:228-229,280 -----------------------------------------------------
 * out_ins = point_ins, except points with point_seg = m >= 0, no instance yet and mask_target[m] > -1,
 * which receive mask_target[m].  In-place (out_ins == point_ins) is allowed.
 * new_count (device int64, optional) counts the re-labelled points. */
int ovo_assign_instances(const int32_t *point_ins, const int16_t *point_seg, int64_t n,
                         const int32_t *mask_target, int n_masks, int32_t *out_ins, int64_t *new_count,
                         ovo_stream_t stream);

/* ---- a9: VanillaMapper.map (vanilla_mapper.py:46-85) -------------------------------------------
 * Step 1 (only when the map is non-empty): mark depth pixels already explained by a map point.
 *   explained u8[h,w] is zeroed here, then set to 1 at every matched pixel.
 * Step 2: erode the valid mask (3x3, only when `erode`), subsample [::ds, ::ds], unproject and
 *   transform by c2w (row-major 4x4, by value in cam2world), append in row-major pixel order at row
 *   `base` of the capacity buffers: xyz f32[cap,3], ids i32[cap] (= first_id + k), ins i32[cap] (-1),
 *   rgb u8[cap,3].  *out_count (device int64) = number of appended points.
 *   ws: ovo_compact_workspace_bytes(ceil(h/ds)*ceil(w/ds)). */
int ovo_map_explained(const float *pts, int64_t n, const ovo_camera_t *cam, const float *depth,
                      uint8_t *explained, ovo_stream_t stream);
int ovo_map_backproject(const float *depth, const uint8_t *rgb, const uint8_t *explained, int h, int w,
                        int erode, int ds, const float *K9_host, const float *c2w16_host, int64_t base,
                        int32_t first_id, float *xyz, int32_t *ids, int32_t *ins, uint8_t *out_rgb,
                        int64_t *out_count, void *ws, size_t ws_bytes, ovo_stream_t stream);

/* ---- a20: multi-view descriptor fusion (instance3d.py:9-21,157-189) -----------------------------
 * store f32[R,D]: every per-(keyframe, instance) descriptor ever produced (append-only).
 * For update k, rows csr_rows[csr_off[k] .. csr_off[k+1]) are that instance's views in the order the
 * reference stacks them.  mode 0 = avg_pooling (mean, NOT re-normalised), 1 = l1_medoid,
 * 2 = cossim_medoid.  table f32[cap,D] row table_rows[k] receives the fused descriptor; out_view[k]
 * the medoid's position in the view list (-1 for avg; 0 for a single view). */
int ovo_fuse_views(const float *store, int D, const int32_t *csr_off, const int32_t *csr_rows, int n_updates,
                   int mode, float *table, const int32_t *table_rows, int32_t *out_view, ovo_stream_t stream);

/* avg_pooling as a running sum: update k adds the views csr_rows[csr_off[k] .. csr_off[k+1]) (NEW ones only) to sums f32[cap,D] row table_rows[k]
 * -- starting it when before[k] == 0 -- and writes table row = sum / (before[k] + new) (a single view is stored as it is).  The mean over all
 * of an instance's views without re-reading them on every update. */
int ovo_fuse_views_add(const float *store, int D, const int32_t *csr_off, const int32_t *csr_rows, const int32_t *before, int n_updates,
                       float *sums, float *table, const int32_t *table_rows, ovo_stream_t stream);

/* ---- dense ("voxel") fusion, BASELINE.json configs 3-4 -----------------------------------------
 * For every point with point_seg = m >= 0 and mask_row[m] >= 0:
 *   acc[i,:] += desc[mask_row[m],:]; cnt[i] += 1.      acc f32[n,D], cnt i32[n], desc f32[*,D]. */
int ovo_scatter_accum(const int16_t *point_seg, int64_t n, const int32_t *mask_row, int n_masks,
                      const float *desc, int D, float *acc, int32_t *cnt, ovo_stream_t stream);
/* The same pass, also emitting WHICH points it changed: touched i32[>= n] receives their indices (in any order), n_touched i32[1]
 * (zero on entry) their number.  Only those points can change class in the dense query of this keyframe
 * (ovo.py:473-492 semantics per point): feed the list to ovo_similarity_rows instead of re-querying the whole map.
 * n_next (optional i32[1]) is set to zero by this call: alternate two counters and no fill launch is ever needed.
 * touched / n_touched may both be NULL (no list).
 * Multi-GPU: with shard_count > 1, acc / cnt hold only THIS rank's block-cyclic shard of the points -- blocks of
 * shard_block points (a power of two), block b owned by rank b % shard_count at local block b / shard_count -- other ranks' points are
 * skipped and `touched` receives local row numbers.  Every rank applies every keyframe's descriptors (all-gathered, KBs) to its own
 * rows in keyframe order, so the shards equal the rows of a single accumulator bit for bit: the "merge" of the per-GPU accumulators
 * needs no floating-point reduce.  shard_count = 1, shard_rank = 0: the unsharded form. */
int ovo_scatter_accum_touched(const int16_t *point_seg, int64_t n, const int32_t *mask_row, int n_masks,
                              const float *desc, int D, float *acc, int32_t *cnt, int32_t *touched, int32_t *n_touched,
                              int32_t *n_next, int shard_rank, int shard_count, int shard_block, ovo_stream_t stream);

/* The dense fusion of one keyframe AND the re-query of the rows it changed in one launch (ABI 11): for every listed row whose mask has a
 * descriptor, acc[row] += desc[mask_row[point_seg[point]]], cnt[row] += 1, then (T != NULL) out_cls / out_conf[row] = the class and confidence
 * ovo_similarity would give the updated row (same arithmetic: bit-identical).  hits / n_hits: the list ovo_track_step_t.hits receives (local
 * row numbers under sharding; n_hits is read on the device, max_hits only sizes the launch).  This is ovo_scatter_accum_touched followed by
 * ovo_similarity_rows (instance3d.py:9-21 generalised per point; clip_utils.py:10-19) without the scan over point_seg and without reading
 * the rows twice.  OVO_E_UNSUPPORTED (nothing launched): Q > 16 or the text rows exceed 96 KB -- run the two calls. */
int ovo_scatter_accum_query(const int32_t *hits, const int32_t *n_hits, int64_t max_hits, const int16_t *point_seg, const int32_t *mask_row,
                            int n_masks, const float *desc, int D, float *acc, int32_t *cnt, int shard_rank, int shard_count, int shard_block,
                            const float *T, int Q, int siglip, float logit_scale, float logit_bias, float th, int64_t *out_cls,
                            float *out_conf, ovo_stream_t stream);

/* ---- a21 + a22: similarity query (clip_utils.py:10-19, ovo.py:487-491) ---------------------------
 * S[i,q] = row_scale(i) * sum_k F[i,k] T[q,k];  siglip: S = sigmoid(S * exp(logit_scale) + logit_bias).
 * F is f32 or f16 ([n,D], feat_dtype 0 = f32, 1 = f16, 2 = bf16); T f32[Q,D].
 * cnt (optional i32[n]): row_scale = 1/cnt (0 rows give 0) -- the dense accumulator form.
 * out_sim (optional) f32[n,Q]; out_cls (optional) i64[n] first-max argmax, -1 when conf <= th;
 * out_conf (optional) f32[n] (0 when conf <= th).  Exact f32 products at any Q, but F is re-read once per 16 queries:
 * meant for Q up to a few dozen (the instance table, the 10-prompt dense query). */
int ovo_similarity(const void *F, int feat_dtype, int64_t n, int D, const float *T, int Q, const int32_t *cnt,
                   int siglip, float logit_scale, float logit_bias, float th, float *out_sim,
                   int64_t *out_cls, float *out_conf, ovo_stream_t stream);
/* ovo_similarity over the rows named in rows[0 .. *n_rows) only (*n_rows <= max_rows is read on the device: no host sync).
 * out_cls / out_conf are indexed by the row id itself -- a class / confidence map that stays resident across keyframes and is
 * patched in place; per row the arithmetic is ovo_similarity's, so the patched map equals a full re-query bit for bit.
 * Needs D % 16 == 0 (the MFMA form). */
int ovo_similarity_rows(const void *F, int feat_dtype, const int32_t *rows, const int32_t *n_rows, int64_t max_rows, int D,
                        const float *T, int Q, const int32_t *cnt, int siglip, float logit_scale, float logit_bias, float th,
                        int64_t *out_cls, float *out_conf, ovo_stream_t stream);
/* Large vocabularies (BASELINE.json config 5, 1k texts x fp16 map): S f32[n,Q] = ovo_gemm(F, T) in f16/bf16 with fp32
 * accumulation, then this pass: optional SigLIP epilogue in place, first-max argmax / confidence / threshold per row
 * (same out_cls / out_conf meaning as ovo_similarity).  Q % 4 == 0. */
int ovo_row_argmax(float *S, int64_t n, int Q, int siglip, float logit_scale, float logit_bias, float th,
                   int64_t *out_cls, float *out_conf, ovo_stream_t stream);
/* (the fused form of this query, ovo_gemm_argmax / ovo_decode_best, is declared after ovo_gemm_t below) */

/* ---- a11: mask NMS intersections (segment_utils.py:218-230) --------------------------------------
 * bits u64[n, words] bit-packed masks (little-endian bit order, zero padded); inter i32[n,n]. */
int ovo_mask_intersections(const uint64_t *bits, int n, int64_t words, int32_t *inter, ovo_stream_t stream);
int ovo_pack_masks(const uint8_t *masks, int n, int64_t pixels, uint64_t *bits, int64_t words, ovo_stream_t stream);
/* The inverse (pixels % 16 == 0): masks u8 [n, pixels] = 0 / 1 per bit.  Multi-GPU: masks a rank's own SAM2 produced
 * for the keyframe it owns travel to the replicas bit-packed (1.2 MB for 32 masks of 640 x 480 instead of 9.8 MB). */
int ovo_unpack_masks(const uint64_t *bits, int n, int64_t pixels, int64_t words, uint8_t *masks, ovo_stream_t stream);

/* ---- a7: _fuse_masks_with_same_ins_id (ovo.py:284-324) -------------------------------------------------
 * masks u8/bool [n, pixels] (pixels % 16 == 0): masks[dst] |= masks[src] for each (dst, src) of pairs i32[n_pairs, 2];
 * ovo_mask_area: area[k] = number of set pixels of masks[rows[k]]. */
int ovo_mask_or(uint8_t *masks, int64_t pixels, const int32_t *pairs, int n_pairs, ovo_stream_t stream);
/* dst[k, :] = src[idx[k], :] for rows of row_bytes bytes (multiple of 16): the kept-mask reorder of ovo.py:322 and the
 * instance-table gather of ovo.py:513-527. */
int ovo_gather_rows(const void *src, int64_t row_bytes, const int32_t *idx, int n, void *dst, ovo_stream_t stream);
int ovo_mask_area(const uint8_t *masks, int64_t pixels, const int32_t *rows, int n_rows, int32_t *area, ovo_stream_t stream);

/* =============================================================================================
 * Encoder building blocks (a10, a12, a13): the reference calls these through third-party nn.Modules
 * (open_clip / perception_models ViT: clip_generator.py:112-122, textregion.py:141-142; sam2 Hiera:
 * mask_generator.py:113).  bf16 (or f16) activations and weights, fp32 accumulation, fp32 residual stream.
 * ============================================================================================= */

/* C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias[N]) + add[M,N]      (nn.Linear layout: W is [out, in])
 *   in_dtype : 1 = f16, 2 = bf16 (A and W);  out_dtype: 0 = f32, 1 = f16, 2 = bf16
 *   act      : 0 none, 1 GELU (erf), 2 QuickGELU x*sigmoid(1.702x) (open_clip "-qg" cards), 3 ReLU, 4 sigmoid (SAM2 decoder),
 *              5 tanh-GELU (SigLIP towers)
 *   add      : optional f32 [M,N] (residual stream / position embedding); may alias C when out_dtype = 0
 *   K % 32 == 0, N % 4 == 0, lda/ldw multiples of 8 elements, 16-byte aligned bases. */
typedef struct {
    const void *A; int64_t lda;
    const void *W; int64_t ldw;
    const float *bias;
    void *C; int64_t ldc;
    const float *add; int64_t ld_add;
    int32_t M, N, K;
    int32_t in_dtype, out_dtype, act;
    float alpha;
} ovo_gemm_t;
int ovo_gemm(const ovo_gemm_t *g, ovo_stream_t stream);
/* ovo_gemm with PE's rotary embedding (ovo_rope_qk below; perception_models Rope2D, textregion.py:141-142) fused into the
 * epilogue: columns [0, cols) of row m are rotated pairwise with cos/sin f32 [T, hd] at (m % T, column % hd), rows with
 * m % T < t0 (class token) untouched.  For the packed QKV projection: cols = 2 * width (q and k), hd = head_dim. */
typedef struct {
    const float *cos, *sin;
    int32_t T, hd, cols, t0;
} ovo_rope_t;
int ovo_gemm_rope(const ovo_gemm_t *g, const ovo_rope_t *rope, ovo_stream_t stream);
/* ovo_gemm with a row-periodic `add`: product row m adds add[m % add_rows] (ld_add columns apart).  The SAM2 decoder's k_proj(keys + pe) over
 * every prompt is keys . Wk^T + (pe . Wk^T)[pixel]: the per-pixel constant is added here instead of materialising (keys + pe). */
int ovo_gemm_periodic(const ovo_gemm_t *g, int64_t add_rows, ovo_stream_t stream);
/* ovo_gemm for Hiera's attention output projection (sam2 `window_unpartition` + residual, inside the encoder the reference
 * reaches at mask_generator.py:113): product row m is a token in WINDOW-major order -- windows of wh x ww tiling a B x H x W
 * token grid padded up to whole windows, M = B * ceil(H/wh) * ceil(W/ww) * wh * ww -- while C and add are addressed by the
 * token's SPATIAL row (b*H + y)*W + x; padding rows are dropped.  add may alias C (in-place residual). */
typedef struct {
    int32_t B, H, W, wh, ww;
} ovo_window_t;
int ovo_gemm_unwindow(const ovo_gemm_t *g, const ovo_window_t *win, ovo_stream_t stream);
/* ovo_gemm (win == NULL) / ovo_gemm_unwindow with f32 output that ALSO writes the LayerNorm of every result row (ABI v10):
 *   C[row] = A . W^T + bias (+ add);   ln_out bf16 [C rows, ld_ln] = LayerNorm(C[row]; ln_g, ln_b, eps)
 * -- Hiera stage 3's attention output projection + residual with the block's norm2 (the A operand of its MLP) taken from the accumulators: a workgroup
 * owns whole rows.  OVO_E_UNSUPPORTED -- nothing launched -- unless N = 448, K % 64 == 0, bf16 operands, f32 C, no activation (then ovo_gemm_unwindow +
 * ovo_gemm_f32a, which ovo_hiera_forward chains itself). */
int ovo_gemm_rowln(const ovo_gemm_t *g, const ovo_window_t *win, const float *ln_g, const float *ln_b, float eps, void *ln_out, int64_t ld_ln,
                   ovo_stream_t stream);
/* The LayerNorm FOLD of the ViT's batched forwards (ABI v12; no reference counterpart -- open_clip / PE run LayerNorm and Linear as two modules, transformer
 * blocks of `model.encode_image`, clip_generator.py:122 / textregion.py:142):  LN(x) . W^T + b = rstd (bf16(x) . W'^T - mean rowsum(W')) + b' with
 * W' = W . gamma and b' = b + W . beta (ovo_vit_layer_t.qkv_wf ...).  Three pieces, all OVO_E_UNSUPPORTED (nothing launched) outside the 256-row ping-pong
 * kernel's range (M >= 2048, N >= 256, N % 64 == 0, K % 64 == 0, bf16 operands, alpha = 1):
 *   ovo_gemm_fold_out:   ovo_gemm with f32 C += f32 add (the epilogue that writes the residual stream x) that ALSO writes xb bf16 [C rows, ld_xb] = bf16(C row)
 *                        and stats f32 [N / 64][ld_stats][2] = (sum, sum of squares) of each 64-column group (part) of each row, PART-major (ld_stats >= M rows
 *                        apart: both sides move a part's consecutive rows as whole cache lines);
 *   ovo_gemm_fold_stats: the same two outputs from an existing f32 matrix x [M, D] (the first LayerNorm of a forward): ONE part, stats [1][M][2];
 *   ovo_gemm_fold_in:    ovo_gemm (rope == NULL) / ovo_gemm_rope with 2-byte output whose A is xb, W = W', bias = b': the row's partials (`parts` per row,
 *                        <= 16, covering D columns) are summed in order, mean / rstd = rsqrt(var + eps) formed once per tile, and the epilogue applies the
 *                        line above before the activation (act 0 or 1) / rotary embedding (head_dim 64). */
int ovo_gemm_fold_out(const ovo_gemm_t *g, void *xb, int64_t ld_xb, float *stats, int64_t ld_stats, ovo_stream_t stream);
int ovo_gemm_fold_stats(const float *x, int64_t ldx, int M, int D, void *xb, int64_t ld_xb, float *stats, ovo_stream_t stream);
int ovo_gemm_fold_in(const ovo_gemm_t *g, const ovo_rope_t *rope, const float *stats, int64_t ld_stats, int parts, int D, const float *rowsum, float eps,
                     ovo_stream_t stream);
/* ovo_gemm whose A operand is the f32 residual stream itself (Hiera stages 1-2 inside SAM2AutomaticMaskGenerator.generate,
 * mask_generator.py:113; no reference counterpart -- there LayerNorm and the linear layer are two modules): g->A is ignored,
 * product row m reads x[src(m), 0..d) -- src(m) = m, or with `win` the SPATIAL token of window-major row m (a padding row
 * of the window grid reads zeros) -- through LayerNorm(gamma, beta, eps) (mode 1) or a plain cast to bf16 (mode 2) while the
 * operand is loaded; columns [d, K) are zeros; C is written in product order (g->add must be NULL).  The normalised bf16
 * copy of the stream is never written.  Only shapes the weights-resident streaming kernel covers (M >= 16384, K <= 256,
 * bf16, a handful of column-group widths): returns OVO_E_UNSUPPORTED otherwise, nothing launched -- the caller then
 * normalises into a buffer and calls ovo_gemm (ovo_hiera_forward does exactly that).
 * pool2x2 = 1 (Hiera's stage-change skip path maxpool(proj(LN(x))); needs `win` with even windows of width 2 / 4 / 8 that
 * tile the grid exactly, f32 output): the 2 x 2 max-pool over a window's tokens is taken in the epilogue and C holds the
 * B x H/2 x W/2 pooled tokens in SPATIAL order -- the projection of every single token is never written. */
int ovo_gemm_f32a(const ovo_gemm_t *g, const ovo_window_t *win, const float *x, int d, const float *gamma, const float *beta,
                  float eps, int mode, int pool2x2, ovo_stream_t stream);
/* The MLP half of a transformer block on the f32 residual stream, in place and in ONE launch (Hiera stages 1-2 inside
 * SAM2AutomaticMaskGenerator.generate, mask_generator.py:113; sam2 MultiScaleBlock `x = x + mlp(norm2(x))`):
 *     x[r, :] += W2 . GELU(W1 . LayerNorm(x[r, :]; ln_g, ln_b, eps) + b1) + b2
 * x f32 [rows, d] (16-byte aligned), W1 bf16 [hidden, ldw1 >= d] (columns >= d zero), W2 bf16 [d, ldw2 >= hidden], hidden = 4 d.
 * The hidden activations never reach memory (as two products they are 2 x rows x hidden x 2 bytes of HBM traffic).
 * Returns OVO_E_UNSUPPORTED -- nothing launched -- for widths without an instantiation ((d, ldw1) in (112, 128), (224, 256), (96, 128),
 * (192, 192), (144, 192)) or rows < 16384: run ovo_gemm_f32a + ovo_gemm then (ovo_hiera_forward does exactly that). */
int ovo_mlp_f32(float *x, int64_t rows, int d, const float *ln_g, const float *ln_b, float eps, const void *w1, int64_t ldw1,
                const float *b1, int hidden, const void *w2, int64_t ldw2, const float *b2, ovo_stream_t stream);
/* The attention half of a Hiera block of 8 x 8 (4 x 4) windows up to its output projection without the q | k | v tensor (sam2 `MultiScaleAttention` on
 * windowed tokens, reached at mask_generator.py:113; ABI v10): per window and head
 *     att[window-major row, head * hd + :] = softmax(q k^T) v,    q | k | v = LayerNorm(x; ln_g, ln_b, eps) . Wqkv^T + b
 * x f32 [B, H, W, d] (spatial order); qkv_w bf16 [3 d_out, ldw >= d rounded up to 128] (columns >= d zero; the q rows and q bias carry log2(e) / sqrt(head_dim):
 * the kernel takes exp2 of the scores as they are); att bf16 out [windows x window^2, ld_att] -- columns [0, d_out) written, the rest untouched.
 * pool = 1 (the stage-change block, q_stride 2): q is 2 x 2 max-pooled inside the window before the scores; att has 16 rows per window.
 * One launch per pair of heads.  OVO_E_UNSUPPORTED -- nothing launched -- unless H % window == W % window == 0, >= 512 windows and
 * (window, d, d_out, heads, pool) = (8, 112, 112, 2, 0), (8, 112, 224, 4, 1) or (4, 224, 224, 4, 0; ldw >= 224, an even
 * number >= 512 of windows): run ovo_gemm_f32a + ovo_attention then (ovo_hiera_forward does exactly that). */
int ovo_window_attention_f32(const float *x, int B, int H, int W, int window, int d, int d_out, int heads, int pool, const float *ln_g,
                             const float *ln_b, float eps, const void *qkv_w, int64_t ldw, const float *qkv_b, void *att, int ld_att,
                             ovo_stream_t stream);
/* The large-vocabulary query (BASELINE.json config 5) with the argmax FUSED into the GEMM epilogue: per row and wave one
 * 64-bit atomicMax on (order-preserving bits of the score << 32 | ~column) into best u64[M] (ZERO on entry; columns >=
 * n_valid -- vocabulary padding -- never win).  store_scores = 0 never writes the score matrix at all (g->C may be NULL).
 * ovo_decode_best turns `best` into classes / confidences with the threshold semantics of ovo.py:487-491. */
int ovo_gemm_argmax(const ovo_gemm_t *g, uint64_t *best, int store_scores, int n_valid, ovo_stream_t stream);
int ovo_decode_best(const uint64_t *best, int64_t n, float th, int64_t *out_cls, float *out_conf, ovo_stream_t stream);

/* O = softmax(Q K^T * scale) V per (batch, head); bf16 in/out, fp32 softmax and accumulation.
 * scale == 0 (ABI v9): Q arrives ALREADY multiplied by scale * log2(e) -- folded into the projection that produced it, in fp32,
 * before its bf16 rounding -- and the kernel applies no factor (scores leave the MFMA in log2 units).  With scale != 0 the
 * kernel multiplies the bf16 Q fragments itself, which rounds every query a second time.
 * Element (b, h, t, d) of X lives at X + b*x_sb + h*x_sh + t*x_st + d (strides in ELEMENTS, d contiguous), so
 * the packed QKV GEMM output [B, T, 3, H, hd] is consumed in place.  hd % 8 == 0, hd <= 128.
 * Windowed attention (Hiera) = windows folded into B; pooled queries = Tq < Tk. */
typedef struct {
    const void *q, *k, *v;
    void *o;
    int64_t q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
    int32_t B, H, Tq, Tk, hd;
    float scale;
    int32_t causal;                                   /* 1: key j is visible to query i only if j <= i (text towers); ABI v2 */
} ovo_attention_t;
int ovo_attention(const ovo_attention_t *a, ovo_stream_t stream);

/* y = LayerNorm(x) * gamma + beta over the last dim (biased variance, eps inside the sqrt).
 * x f32 rows at x + r*x_stride; y rows at y + r*y_stride, out_dtype 0 = f32, 2 = bf16.  d % 4 == 0. */
int ovo_layernorm(const float *x, int64_t x_stride, int64_t rows, int d, const float *gamma, const float *beta,
                  float eps, void *y, int64_t y_stride, int out_dtype, ovo_stream_t stream);

/* ViT token assembly: x[b, t, :] = (t < n_prefix ? prefix[t] : patch[b, t - n_prefix]) + pos[t], then an optional
 * LayerNorm (open_clip ln_pre).  patch f32 [B, P, d], prefix f32 [n_prefix, d] (class token) or NULL,
 * pos f32 [n_prefix + P, d] or NULL, x f32 [B, n_prefix + P, d]. */
int ovo_vit_embed(const float *patch, const float *prefix, int n_prefix, const float *pos, int B, int P, int d,
                  const float *gamma, const float *beta, float eps, float *x, ovo_stream_t stream);

/* Non-overlapping (stride == patch) or overlapping im2col of NCHW f32 images into bf16 rows
 * [B * oh * ow, kpad], column (c, ky, kx) = c*ksz*ksz + ky*ksz + kx, zero padded to kpad (kpad % 32 == 0),
 * zero outside the image (conv padding `pad`). */
int ovo_im2col(const float *img, int B, int C, int H, int W, int ksz, int stride, int pad, void *out, int kpad,
               ovo_stream_t stream);

/* Crop + resize (bilinear, align_corners = False, optional antialias exactly as torch's
 * F.interpolate(..., antialias=True); antialias = 2: antialiased bicubic) + per-channel normalise: out[c] = (resize(src[c]) * scale - mean[c]) / std[c].
 * src is CHW, u8 (src_dtype 3) or f32 (0), or an interleaved u8 [H, W, C] frame as the camera delivers it (src_dtype 4: no permute
 * pass before the encoders); the crop is rows [y0, y0+ch) x cols [x0, x0+cw). out f32 [C, oh, ow]. */
int ovo_resize_normalize(const void *src, int src_dtype, int C, int H, int W, int y0, int x0, int ch, int cw,
                         float *out, int oh, int ow, int antialias, float scale, const float *mean3_host,
                         const float *std3_host, ovo_stream_t stream);

/* The kept transforms of open_clip's preprocess (clip_utils.py:83-84: Resize + CenterCrop + Normalize; clip_generator.py:112-122 pushes whole
 * frames and mask crops through them): the output [C, oh, ow] is the WINDOW (top, left, oh, ow) of the crop's resize to virt_h x virt_w --
 * torchvision Resize(size) on the shorter side followed by CenterCrop(size) -- never materialising the part that is cropped away.
 * filter: 0 = bilinear, 1 = antialiased bilinear, 2 = antialiased bicubic (F.interpolate(mode="bicubic", antialias=True): Keys a = -0.5;
 * float input is not clamped, like torchvision's tensor path).  ovo_resize_normalize = this with the window covering the whole output. */
int ovo_resize_window_normalize(const void *src, int src_dtype, int C, int H, int W, int y0, int x0, int ch, int cw, float *out,
                                int oh, int ow, int virt_h, int virt_w, int top, int left, int filter, float scale,
                                const float *mean3_host, const float *std3_host, ovo_stream_t stream);
/* ovo_resize_normalize for n_src source images of one size / layout x n_crop crops each in ONE launch (ABI v10): out image i * n_crop + k
 * (f32 [C, oh, ow] each, contiguous) = crop k (crops_host[4 k ..] = y0, x0, h, w) of source i.  srcs_host / crops_host are HOST arrays. */
int ovo_resize_normalize_batch(const void *const *srcs_host, int n_src, int src_dtype, int C, int H, int W, const int32_t *crops_host, int n_crop,
                               float *out, int oh, int ow, int antialias, float scale, const float *mean3_host, const float *std3_host,
                               ovo_stream_t stream);

/* ---- a14: per-mask crops of the crop-mode descriptors (segment_utils.py:29-41 segmap2segimg, :43-94, :118-172) ----
 * ovo_mask_boxes: masks u8 [n, H, W] -> boxes i32 [n, 4] = (x, y, w, h) with the reference's w = x_max - x_min,
 *   h = y_max - y_min over INCLUSIVE edges (batched_mask_to_box + batched_box_xyxy_to_xywh), zeros for an empty mask.
 * ovo_mask_crops: image CHW (3 channels, u8 = dtype 3 or f32 = 0, range 0..255) -> out f32 [n, 3 or 6, R, R]:
 *   channels 0-2 the masked crop (zero background; zero-padded to a centred square when also_bbox = 0), channels 3-5
 *   (also_bbox = 1) the box grown by `margin` pixels; each resized to R x R exactly as torchvision F.resize does for a
 *   tensor (bilinear, antialias).  round_out = 1 rounds half-to-even like F.resize on a uint8 tensor. */
int ovo_mask_boxes(const uint8_t *masks, int n, int H, int W, int32_t *boxes_xywh, ovo_stream_t stream);
int ovo_mask_crops(const void *image, int img_dtype, int H, int W, const uint8_t *masks, const int32_t *boxes_xywh, int n,
                   int also_bbox, int margin, int R, int round_out, float *out, ovo_stream_t stream);

/* 2-D rotary embedding on q and k of a packed QKV buffer (perception_models Rope2D), in place.
 * qkv bf16 [B, T, 3, H, hd]; cos/sin f32 [T, hd]; pairs (2i, 2i+1) rotate together (interleaved form);
 * rows t < t0 (class token) are left alone. */
int ovo_rope_qk(void *qkv, int B, int T, int H, int hd, const float *cos_t, const float *sin_t, int t0,
                ovo_stream_t stream);

/* ---- a15: PETextRegion.get_features_mask (textregion.py:145-161) ------------------------------------
 * masks u8/bool [N, H, W] -> w bf16 [N, gpad] with w = 1 where the bilinear (align_corners = False) sample of the
 * mask at token cell (gy, gx) is > 0, else 0; columns >= gh*gw are zero; cnt f32[N] = number of ones. */
int ovo_feature_masks(const uint8_t *masks, int N, int H, int W, int gh, int gw, void *w, int gpad, float *cnt,
                      ovo_stream_t stream);

/* ---- a16: resize_features (textregion.py:9-28), written TRANSPOSED for the pooling GEMM --------------
 * tokens f32 [1 + nh*nw, tstride rows of d] (row 0 = global crop, then tiles); each crop has P*P patch tokens
 * starting at token `t0` (1 when a class token is present).  out bf16 [d, gpad]:
 * out[:, gy*gw + gx] = 0.5 * bilinear(global grid)(gy, gx) + tile(gy / P, gx / P) token (gy % P, gx % P). */
int ovo_stitch_tokens_t(const float *tokens, int tokens_per_crop, int t0, int d, int P, int nh, int nw, void *out,
                        int gpad, ovo_stream_t stream);

/* ---- a18: remove_global_patch (textregion.py:31-50), off in OVO's configuration (clip_generator.py:46) -------
 * The reference's patch_2_region_avg[t, n] = (1/|n|) sum_{t' in n} p_t . p_t' equals p_t . mean_{t' in n}(p_t'), so the
 * T x T patch similarity is never formed:  ovo_unit_tokens -> unit tokens in both GEMM layouts (u_t bf16 [d, gpad],
 * u bf16 [gpad, d]) from the stitched tokens x_t bf16 [d, gpad];  two ovo_gemm calls give r_t f32 [N, gpad];
 * ovo_global_patch_filter clears, in weights bf16 [N, gpad] ({0,1}), every token column whose mean score inside its
 * masks minus its mean score outside them is < th, and recounts cnt f32 [N]. */
int ovo_unit_tokens(const void *x_t, int d, int G, int gpad, void *u_t, void *u, ovo_stream_t stream);
int ovo_global_patch_filter(const float *r_t, void *weights, int N, int G, int gpad, float th, float *cnt, ovo_stream_t stream);

/* rows of f32 [N, d] scaled by 1 / cnt[row] -> bf16 [N, d] (masked mean);  y = x / ||x||_2 per row (f32). */
int ovo_scale_rows_bf16(const float *x, const float *cnt, int N, int d, void *y, ovo_stream_t stream);
int ovo_l2_normalize_rows(const float *x, int64_t N, int d, float *y, ovo_stream_t stream);
/* f32 -> bf16 (dtype 2) / f16 (dtype 1) conversion of n elements (n % 4 == 0). */
int ovo_cast_f32(const float *x, int64_t n, void *y, int dtype, ovo_stream_t stream);

/* ---- a12 / a13: ViT forward (open_clip VisionTransformer / perception_models pe.VisionTransformer) -----
 * Replaces `model.encode_image` (clip_generator.py:122) and `visual.forward_features(x, norm=True)`
 * (textregion.py:142).  The whole layer loop runs inside the library (one C call per forward, graph-capturable).
 * Weights: bf16 matrices in nn.Linear layout [out, in]; fp32 vectors (biases, LayerNorm, class / position
 * embeddings).  LayerScale, if a model has it, is folded into out_w/out_b and fc2_w/fc2_b at load time. */
typedef struct {
    int32_t image_size, patch, width, layers, heads, mlp_dim, out_dim;
    int32_t n_prefix;  /* 1 = class token, 0 = none                                             */
    int32_t act;       /* 1 = GELU, 2 = QuickGELU, 5 = tanh-GELU (SigLIP)                       */
    int32_t pre_ln;    /* ln_pre present                                                        */
    int32_t use_rope;  /* 2-D rotary embedding on q, k (PE)                                     */
    int32_t pool;      /* 0: all tokens after ln_post -> f32 [B, T, width] (forward_features);  */
                       /* 1: ln_post(class token) @ proj -> f32 [B, out_dim] (encode_image)     */
                       /* 2: SigLIP attention-pool head over ln_post(tokens) -> f32 [B, width]  */
    int32_t kpad;      /* 3*patch*patch rounded up to a multiple of 32                          */
    float ln_eps;
    int32_t q_prescaled; /* 1 = the q rows of every qkv_w / qkv_b (and map_q) already carry log2(e) / sqrt(head_dim), folded in  */
                         /* fp32 at load time BEFORE the bf16 rounding: the attention kernels then apply no factor to Q at all   */
                         /* (a factor applied to the bf16 Q inside the kernel is a second rounding of every query). ABI v9       */
} ovo_vit_config_t;

typedef struct {
    const float *ln1_g, *ln1_b;
    const void *qkv_w; const float *qkv_b;   /* [3*width, width], [3*width]  (q | k | v)  */
    const void *out_w; const float *out_b;   /* [width, width]                              */
    const float *ln2_g, *ln2_b;
    const void *fc1_w; const float *fc1_b;   /* [mlp_dim, width]                            */
    const void *fc2_w; const float *fc2_b;   /* [width, mlp_dim]                            */
    /* ABI v12, optional (all six or none; NULL = the LayerNorm kernels run): the two LayerNorms folded into the products that follow them,         */
    /* LN(x) . W^T + b = rstd (bf16(x) . W'^T - mean colsum(W')) + b'.  *_wf = bf16(W . gamma) (column k times gamma[k], in f32, ONE rounding),     */
    /* *_cs = f32 row sums of the ROUNDED W' (what the product really multiplies), *_bf = b + W . beta (f32).  Used for batched forwards only       */
    /* (>= 2048 token rows, width and mlp_dim multiples of 64, width <= 1024, act 0 / 1); otherwise the plain weights above are used.              */
    const void *qkv_wf; const float *qkv_bf, *qkv_cs;
    const void *fc1_wf; const float *fc1_bf, *fc1_cs;
} ovo_vit_layer_t;

typedef struct {
    const void *patch_w; const float *patch_b;        /* [width, kpad] (conv1 flattened, zero padded), bias or NULL */
    const float *prefix;                              /* [n_prefix, width] class embedding                           */
    const float *pos;                                 /* [n_prefix + P, width] or NULL                                */
    const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    const void *proj_w;                               /* [out_dim, width] (= proj^T) when pool = 1                    */
    const float *rope_cos, *rope_sin;                 /* [n_prefix + P, head_dim] when use_rope                       */
    const ovo_vit_layer_t *layers;                    /* HOST array of cfg.layers entries                            */
    /* pool = 2 (timm AttentionPoolLatent, SigLIP "map" pooling; open_clip TimmModel towers, clip_utils.py:53-75):    */
    const void *map_q;                                /* bf16 [width] = latent . Wq^T + bq (input independent)       */
    const void *map_kv_w; const float *map_kv_b;      /* [2*width, width] (k | v)                                    */
    const void *map_proj_w; const float *map_proj_b;  /* [width, width]                                              */
    const float *map_ln_g, *map_ln_b;
    const void *map_fc1_w; const float *map_fc1_b;    /* [mlp_dim, width]                                            */
    const void *map_fc2_w; const float *map_fc2_b;    /* [width, mlp_dim]                                            */
} ovo_vit_weights_t;

size_t ovo_vit_workspace_bytes(const ovo_vit_config_t *cfg, int B);
/* images f32 [B, 3, S, S] already resized + normalised (ovo_resize_normalize); out as described by cfg.pool. */
int ovo_vit_forward(const ovo_vit_config_t *cfg, const ovo_vit_weights_t *w, const float *images, int B, float *out,
                    void *ws, size_t ws_bytes, ovo_stream_t stream);

/* ---- a10: SAM2 image encoder (Hiera trunk + FPN neck) ------------------------------------------------
 * Replaces the encoder half of `SAM2AutomaticMaskGenerator.generate(image)` (mask_generator.py:113,
 * segment_utils.py:291-308; sam2 `forward_image`).  Architecture per the SAM2 paper / sam2 repo
 * (SURVEY.md App. A): 7x7/4 patch embed + (bicubic background + tiled window) position embedding, 4 stages of
 * windowed multi-head attention blocks with 2x2 max-pool query pooling at stage changes and a few global
 * blocks, FPN 1x1 laterals with nearest top-down on the coarse levels, optional decoder conv_s0 / conv_s1.
 * Outputs are NHWC f32: feat0 [B, S/4, S/4, c0], feat1 [B, S/8, S/8, c1], feat2 [B, S/16, S/16, fpn_dim]
 * with (c0, c1) = (32, 64) when hi_res else (fpn_dim, fpn_dim). */
typedef struct {
    int32_t image_size;            /* 1024 */
    int32_t dims[4], heads[4], blocks[4], window[4];
    int32_t n_global, global_blocks[8];
    int32_t fpn_dim;               /* 256 */
    int32_t hi_res;                /* apply conv_s0 (->32) / conv_s1 (->64) to the two fine levels */
    float ln_eps;                  /* 1e-6 */
    int32_t q_prescaled;           /* 1 = q rows of qkv_w / qkv_b carry log2(e) / sqrt(head_dim) (see ovo_vit_config_t); ABI v9 */
} ovo_hiera_config_t;

typedef struct {
    const float *ln1_g, *ln1_b;
    const void *qkv_w; const float *qkv_b;   /* [3*dim_out, pad64(dim)]      */
    const void *out_w; const float *out_b;   /* [dim_out, pad64(dim_out)]    */
    const float *ln2_g, *ln2_b;
    const void *fc1_w; const float *fc1_b;   /* [4*dim_out, pad64(dim_out)]  */
    const void *fc2_w; const float *fc2_b;   /* [dim_out, 4*dim_out]         */
    const void *res_w; const float *res_b;   /* [dim_out, pad64(dim)] at stage changes, else NULL */
} ovo_hiera_block_t;

typedef struct {
    const void *patch_w; const float *patch_b;   /* [dims[0], 192] (3*7*7 = 147 padded; pad64 = round up to a multiple of 64), [dims[0]] */
    const float *pos;                            /* [(S/4)^2, dims[0]] precomputed position embedding */
    const ovo_hiera_block_t *blocks;             /* HOST array, sum(blocks) entries */
    const void *neck_w[4]; const float *neck_b[4]; /* level i (fine -> coarse): [fpn_dim, pad64(dims[i])] */
    const void *s0_w; const float *s0_b;         /* [32, fpn_dim] */
    const void *s1_w; const float *s1_b;         /* [64, fpn_dim] */
} ovo_hiera_weights_t;

size_t ovo_hiera_workspace_bytes(const ovo_hiera_config_t *cfg, int B);
/* Hiera's patch embedding alone (sam2 `PatchEmbed`: Conv2d(3, E, 7, stride 4, padding 3) + the windowed position embedding; the first
 * step of SAM2's image encoder, mask_generator.py:113) as a direct convolution on the f32 image, no im2col matrix (ABI v10):
 *   x[b, oy * S/4 + ox, :] = conv(images[b])[:, oy, ox] + bias + pos[oy * S/4 + ox, :]
 * images f32 [B, 3, S, S]; patch_w bf16 [E, ldw], column (c * 7 + ky) * 7 + kx, ldw >= 147; bias f32 [E]; pos f32 [(S/4)^2, E]; x f32 out.
 * E in {96, 112, 144} and S % 128 == 0, else OVO_E_UNSUPPORTED (ovo_hiera_forward then runs ovo_im2col + ovo_gemm itself). */
int ovo_hiera_patch_embed(const float *images, int B, int S, int E, const void *patch_w, int ldw, const float *bias, const float *pos,
                          float *x, ovo_stream_t stream);
/* images f32 [B, 3, S, S], already resized + normalised. */
int ovo_hiera_forward(const ovo_hiera_config_t *cfg, const ovo_hiera_weights_t *w, const float *images, int B,
                      float *feat0, float *feat1, float *feat2, void *ws, size_t ws_bytes, ovo_stream_t stream);

/* =============================================================================================
 * f1: SAM2 mask decoder (the reference reaches it through sam2.automatic_mask_generator, segment_utils.py:291-308,
 * mask_generator.py:113).  Its matrix products are ovo_gemm / ovo_attention calls; these are the passes between them.
 * ============================================================================================= */

/* One pass over R rows of C channels (C % 4 == 0, C <= 2048):  v = x[r] (+ base[r % base_rows]);  if gamma: v = LN(v);
 * then any of  y f32[R,C] (may alias x),  y16 bf16[R,C],  ype16 bf16[R,C] = v + pe[r % pe_rows]  (NULL = skip).
 * This is every LayerNorm / residual / "+ positional code" / cast of the two-way transformer. */
int ovo_row_epilogue(const float *x, int64_t R, int C, const float *base, int64_t base_rows, const float *gamma, const float *beta,
                     float eps, const float *pe, int64_t pe_rows, float *y, void *y16, void *ype16, ovo_stream_t stream);

/* Upscaling stage 1: g bf16 [P*s*s, 4*C1] = embedding . W^T of ConvTranspose2d(C, C1, 2, stride 2) with GEMM column
 * (dy*2 + dx)*C1 + c;  out bf16 [P, 2s, 2s, C1] = GELU(LayerNorm2d(pixel_shuffle(g) + bias + feat))), feat f32 [2s, 2s, C1]. */
int ovo_sam_upscale_ln(const void *g, const float *bias, const float *feat, const float *gamma, const float *beta, float eps,
                       int64_t P, int s, int C1, void *out, ovo_stream_t stream);

/* Upscaling stage 2 fused with the hyper-network product: g bf16 [P*s2*s2, 4*C2] (same column order), feat f32 [2 s2, 2 s2, C2],
 * hyper f32 [P, n_mask, C2]  ->  out f32 [P, n_mask - first, 2 s2, 2 s2],  out[p,i] = sum_c hyper[p, first+i, c] *
 * GELU(pixel_shuffle(g) + bias + feat)[c]  (the 1 GB upscaled embedding of 256 prompts is never written). */
int ovo_sam_upscale_masks(const void *g, const float *bias, const float *feat, const float *hyper, int n_mask, int first, int64_t P,
                          int s2, int C2, float *out, ovo_stream_t stream);

/* The three HBM-bound products of the decoder's image side fused with the row passes above (samfuse.hip): the weights stay in LDS, the
 * product never reaches HBM.  Each returns OVO_E_UNSUPPORTED for widths other than SAM2's (hidden 256) and the test card's (128); the
 * caller then runs the unfused chain (ovo_gemm + the pass).
 * ovo_sam_proj_ln:   y = LayerNorm(res[m % res_rows] + A[M,K] . W[N,K]^T + bias) -> y32 f32 / y16 bf16 / ype16 bf16 (+ pe[m % pe_rows]);
 *                    the residual is f32 `res` or bf16 `res16` (at most one of them)
 *                    (attention out-projection + residual + norm4 of TwoWayAttentionBlock; (N, K) = (256, 128) | (128, 64)).
 * ovo_sam_up1_ln:    A bf16 [P s s, K] . W[4 C1, K]^T (+ bias, + feat, LayerNorm2d, GELU) -> out bf16 [P, 2s, 2s, C1]   (= ovo_gemm + ovo_sam_upscale_ln)
 * ovo_sam_up2_masks: A bf16 [P s2 s2, K] . W[4 C2, K]^T (+ bias, + feat, GELU, hyper-network dot) -> out f32 [P, n_mask - first, 2 s2, 2 s2]
 *                    (= ovo_gemm + ovo_sam_upscale_masks). */
/* ovo_sam_linear: C bf16 [M, N] (row stride ldc) = A bf16 [M, K] . W[N, K]^T + bias + add[m % add_rows] (f32, row stride ld_add; NULL = none):
 * the K | V and Q projections over the per-prompt keys ((K, N) = (256, 256 | 128) or (128, 128 | 64); else OVO_E_UNSUPPORTED -> ovo_gemm_periodic). */
int ovo_sam_linear(const void *A, const void *W, const float *bias, const float *add, int64_t add_rows, int ld_add, void *C, int ldc, int64_t M,
                   int N, int K, ovo_stream_t stream);
int ovo_sam_proj_ln(const void *A, const void *W, const float *bias, const float *res, const void *res16, int64_t res_rows, const float *gamma, const float *beta,
                    float eps, const float *pe, int64_t pe_rows, float *y32, void *y16, void *ype16, int64_t M, int N, int K, ovo_stream_t stream);
int ovo_sam_up1_ln(const void *A, const void *W, const float *bias, const float *feat, const float *gamma, const float *beta, float eps,
                   int64_t P, int s, int C1, int K, void *out, ovo_stream_t stream);
int ovo_sam_up2_masks(const void *A, const void *W, const float *bias, const float *feat, const float *hyper, int n_mask, int first, int64_t P,
                      int s2, int C2, int K, float *out, ovo_stream_t stream);

/* image -> token cross attention of the two-way transformer (head_dim 16, T <= 16 token keys per prompt):
 * q bf16 rows of 16 H channels at (prompt p, pixel s) = q + p * q_batch_stride + s * q_token_stride elements (0 batch stride = shared by every
 * prompt; a column block of a wider matrix is fine), k / v bf16 [P, T, 16 H] -> o bf16 [P, S, 16 H]. */
int ovo_sam_i2t_attention(const void *q, int64_t q_batch_stride, int q_token_stride, const void *k, const void *v, void *o, int64_t P, int S, int T,
                          int H, float scale, ovo_stream_t stream);
/* token -> image cross attention (head_dim 16): q bf16 [P, T, 16 H], k / v bf16 rows of 16 H channels at (prompt p, key s) = base +
 * p * kv_batch_stride + s * kv_token_stride elements (0 batch stride = keys shared by every prompt; k and v may be column blocks of one
 * wider matrix) -> o bf16 [P, T, 16 H].  OVO_E_UNSUPPORTED unless H * T divides 64 (then use ovo_attention). */
int ovo_sam_t2i_attention(const void *q, const void *k, const void *v, int64_t kv_batch_stride, int kv_token_stride, void *o, int64_t P, int S, int T,
                          int H, float scale, ovo_stream_t stream);

/* Automatic-mask-generator filters on the low-resolution logits f32 [n, h, w], evaluated on their H x W bilinear
 * upsampling (torch F.interpolate, align_corners = False) without materialising it:
 * ovo_amg_mask_stats: stats i32 [n, 7] = {#(v > thr + offset), #(v > thr - offset), #(v > thr), x_min, y_min, x_max, y_max}
 *   (stability score = stats[0] / stats[1]; the box is that of the binary mask v > thr; empty -> W, H, -1, -1).
 * ovo_amg_binarize: out u8 [n_sel, H, W] = upsampled logits[sel[k]] > thr. */
int ovo_amg_mask_stats(const float *logits, int n, int h, int w, int H, int W, float thr, float offset, int32_t *stats,
                       ovo_stream_t stream);
int ovo_amg_binarize(const float *logits, const int32_t *sel, int n_sel, int h, int w, int H, int W, float thr, uint8_t *out,
                     ovo_stream_t stream);
/* mask2segmap's painting (segment_utils.py:12-27) for masks already in descending-stability order:
 * seg i32 [pixels] = index of the first mask u8 [n, pixels] covering the pixel, -1 if none. */
int ovo_paint_segmap(const uint8_t *masks, int n, int64_t pixels, int32_t *seg, ovo_stream_t stream);

/* =============================================================================================
 * f3: loop-closure semantic update (ovo.py:366-424 update_map, instance_utils.py:5-35 same_instance / fuse_instances).
 * ============================================================================================= */

/* One pass over the map: cnt i32 [n_slots] points per instance id, sums f64 [n_slots, 3] of their coordinates
 * (centroid = sums / cnt: instance_utils' `obj_pcd.mean(axis=0)`; cnt > 0 replaces `points_ins_ids.unique()`).  Both are zeroed here. */
int ovo_instance_moments(const float *xyz, const int32_t *ins, int64_t n, int n_slots, double *sums, int32_t *cnt, ovo_stream_t stream);

/* instance_utils.py:20-27 without the KD-tree: for every candidate pair (a, b) (pairs i32 [n_pairs, 2], slots of the CSR
 * grouping pts_by_instance f32 [*, 3] / offsets i64 [n_slots + 1]) near_count[pair] = number of points of a that have a point of
 * b at distance < th  (=> p_dist = near_count / |a|).  max_points_a = largest |a| among the pairs (grid size). */
int ovo_near_fraction(const float *pts_by_instance, const int64_t *offsets, const int32_t *pairs, int n_pairs, int64_t max_points_a,
                      float th, int32_t *near_count, ovo_stream_t stream);

/* ins[i] = table[ins[i]] for ids in [0, n_slots): all `points_ins_ids[points_ins_ids == id2] = id1` of the merges in one pass. */
int ovo_remap_instances(int32_t *ins, int64_t n, const int32_t *table, int n_slots, ovo_stream_t stream);


/* =============================================================================================
 * The keyframe chain without host round trips (MI355X extension of a6 / a9; what the reference does with a `.sum()` / vstack
 * per mapped frame, vanilla_mapper.py:81-85, and >= 3 `.item()` syncs per mask, ovo.py:255-282).
 * The map's size, the next point id and the next instance id live in DEVICE memory; ovo_map_step and ovo_track_step read and
 * advance them there, size their launches from the caller's upper bound `n_upper`, take the instance decisions of ovo.py:255-282
 * and the mask fusion of ovo.py:284-309 on the device, and publish one small result block per call into PINNED host memory
 * (written by the last workgroup, sequence number last).  The host never has to wait between the launches of consecutive
 * keyframes: it reads the blocks when it needs them (ovo_host_wait), e.g. after queueing a whole round of keyframes.
 * ============================================================================================= */
typedef struct {
    float *xyz; int32_t *ids; int32_t *ins; uint8_t *rgb;   /* capacity buffers f32[cap,3], i32[cap], i32[cap], u8[cap,3] */
    int64_t cap;
    int64_t *state;      /* device i64[4] = {n points, next point id, error flags (bit 0: capacity overflow), ticket (zero)} */
    int64_t n, next_id;  /* n >= 0: the host's exact copy of state[0..1] (no call in flight) -- used instead of reading state,
                            which is re-seeded from it;  n < 0: read state */
} ovo_map_ref_t;

/* VanillaMapper.map (vanilla_mapper.py:46-85): explained-pixel test against the map (skipped while next point id == 0), erosion
 * (`erode`, also only then), [::ds, ::ds] subsample, unproject, ordered append at row n; state += appended.
 * result_host (optional, pinned i64[4]) = {seq, appended, n after, next id after}.  ws: ovo_compact_workspace_bytes(sub-sampled
 * pixels) + 8 bytes; explained u8[h*w] scratch.  map.cap >= n_upper + sub-sampled pixels. */
typedef struct {
    ovo_map_ref_t map;
    const float *depth; const uint8_t *rgb;   /* f32[h,w]; u8[h,w,3] or NULL */
    int32_t h, w;
    ovo_camera_t cam;                         /* frustum of the frame, th = the mapper's match distance */
    float K[9], c2w[16];
    int32_t ds, erode;
    int64_t n_upper;                          /* >= the map's size when the call executes */
    uint8_t *explained; void *ws; size_t ws_bytes;
    int64_t *result_host; int64_t seq;
} ovo_map_step_t;
int ovo_map_step(const ovo_map_step_t *a, ovo_stream_t stream);

/* OVO._match_and_track_instances (ovo.py:182-324) for one keyframe: [depth filter] -> ovo_track_project over the device-sized map
 * -> vote statistics -> decisions in mask order (matched: n_assigned > track_th -> mode id; new: n_fresh > track_th -> next
 * instance id, allocated in mask order) -> in-place assignment of map.ins -> masks of one instance OR-ed into its first mask +
 * that mask's fused area.  result (device copy inside ws, host copy in result_host, i32[8 + 6 n_masks]):
 *   [0] seq  [1] map points  [2] points in frustum  [3] points matched  [4] next instance id after  [5] before  [6..7] 0
 *   per mask: {matched points, of which assigned, mode id, seg-map area, target id (-1 none), fused area (-1: no other mask joined)}
 * point_seg i16[>= n_upper] as ovo_track_project writes it.  masks u8[n_masks, pixels] (pixels % 16 == 0) are fused IN PLACE;
 * NULL skips the fusion.  next_ins: device i32[1]; next_ins_host >= 0 = the host's exact copy (used instead), -1 = read it. */
typedef struct {
    ovo_map_ref_t map;
    const float *depth; int32_t filter_depth; float *depth_scratch;
    ovo_camera_t cam; ovo_ratio_t ratio;
    const int32_t *seg_map; int32_t seg_h, seg_w;
    uint8_t *masks; int32_t n_masks; int64_t pixels;
    int16_t *point_seg;
    void *ws; size_t ws_bytes;                /* ovo_track_workspace_bytes(n_masks, hist_cols) */
    int32_t hist_cols, track_th;
    int32_t *next_ins; int32_t next_ins_host;
    int64_t n_upper;
    int32_t *result_host; int32_t seq;
    /* ABI 11: the points this keyframe's masks cover (point_seg >= 0), listed by the tracking pass itself -- the row list of
     * ovo_scatter_accum_query, so that the dense fusion needs no scan over point_seg.  hits i32[>= n_upper + the frame's new points] (any
     * order), n_hits i32[1] (zeroed by the step); NULL: no list.  hit_shard_count > 1: only points of block-cyclic shard hit_shard_rank
     * (blocks of hit_shard_block points, a power of two) are listed, as LOCAL row numbers (the map of ovo_scatter_accum_touched). */
    int32_t *hits; int32_t *n_hits;
    int32_t hit_shard_rank, hit_shard_count, hit_shard_block;
} ovo_track_step_t;
size_t ovo_track_workspace_bytes(int n_masks, int hist_cols);
int ovo_track_step(const ovo_track_step_t *a, ovo_stream_t stream);

/* Both halves of ONE keyframe with their independent passes merged into shared launches (7 launches instead of 13; the map's size after
 * the append is read on the device by the tracking half in any case): same passes, same results, same two result blocks.  The map step
 * and the tracking step must describe the same map (`map.state` shared).  Shapes the merged launches do not cover fall back to the two calls. */
int ovo_keyframe_step(const ovo_map_step_t *map_step, const ovo_track_step_t *track_step, ovo_stream_t stream);

/* A whole ROUND of keyframes -- ovo_map_step + ovo_track_step of keyframe 0, then of keyframe 1, ... -- in ONE launch of a few dozen
 * persistent workgroups that walk through every pass, separated by grid-wide barriers.  Same passes, same results, same result
 * blocks; what changes is how often the chain waits for the dispatcher: as separate launches a keyframe is ~12 small dependent
 * kernels, each of which queues behind the encoders' GEMM workgroups on a busy GPU.  maps[k].depth == NULL: no map update for keyframe
 * k; tracks[k].n_masks == 0: no tracking.  OVO_E_UNSUPPORTED (n > 16, > 1024 masks, > 131072 sub-sampled pixels): use the two calls.
 * ctx: params_host = ovo_host_alloc(ovo_round_chain_params_bytes()); barrier = device u64[2], zero before the first call;
 * arrivals / next_slot start at 0 and are advanced here; at most 8 rounds may be in flight.  A barrier that times out (a bug, not
 * a load condition) sets result[6] of the tracking blocks instead of hanging the device. */
typedef struct {
    void *params_host;
    uint64_t *barrier;
    uint64_t arrivals;
    uint32_t next_slot;
    int32_t workgroups;      /* persistent workgroups per launch; 0 = 64 */
} ovo_round_chain_t;
size_t ovo_round_chain_params_bytes(void);
int ovo_round_chain(ovo_round_chain_t *ctx, const ovo_map_step_t *maps, const ovo_track_step_t *tracks, int n, ovo_stream_t stream);

/* Pinned, device-visible host memory for the result blocks, and the wait on a block's sequence word: spins (no runtime call) until
 * *flag == value, at most timeout_us microseconds (OVO_E_LAUNCH on timeout). */
void *ovo_host_alloc(size_t bytes);
void ovo_host_free(void *p);
int ovo_host_wait32(const int32_t *flag, int32_t value, int64_t timeout_us);
int ovo_host_wait64(const int64_t *flag, int64_t value, int64_t timeout_us);

#ifdef __cplusplus
}
#endif
#endif /* OVO_HIP_H */
