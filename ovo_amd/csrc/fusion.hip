// fusion.hip -- multi-view descriptor fusion (instance form) and dense per-point scatter-accumulate.
#include <float.h>

#include "common.h"

namespace {

// One workgroup (256 threads) per instance update.  V = number of views, D = descriptor length.
// mode 0: mean over views (sequential sum in view order, then one divide -- torch.mean's order for
//         a short strided reduction), 1: l1 medoid, 2: cosine medoid (instance3d.py:9-21).
__global__ void __launch_bounds__(256) k_fuse_views(const float *__restrict__ store, int D, const int32_t *__restrict__ csr_off,
                                                    const int32_t *__restrict__ csr_rows, int mode, float *__restrict__ table,
                                                    const int32_t *__restrict__ table_rows, int32_t *__restrict__ out_view) {
    const int k = blockIdx.x, t = threadIdx.x;
    const int lo = csr_off[k], V = csr_off[k + 1] - lo;
    float *dst = table + (int64_t)table_rows[k] * D;
    if (V <= 0) return;
    if (V == 1 || mode == 0) {
        for (int d = t; d < D; d += 256) {
            float s = store[(int64_t)csr_rows[lo] * D + d];
            for (int v = 1; v < V; ++v) s += store[(int64_t)csr_rows[lo + v] * D + d];
            dst[d] = V == 1 ? s : s / (float)V;
        }
        if (t == 0 && out_view) out_view[k] = V == 1 ? 0 : -1;
        return;
    }
    // medoids: score[i] = sum_j dist(i, j); pick argmin (l1) / argmax (cos), first index on ties
    __shared__ float red[256];
    __shared__ float s_best;
    __shared__ int s_arg;
    if (t == 0) { s_best = mode == 1 ? FLT_MAX : -FLT_MAX; s_arg = 0; }
    __syncthreads();
    for (int i = 0; i < V; ++i) {
        const float *a = store + (int64_t)csr_rows[lo + i] * D;
        float score = 0.f;
        for (int j = 0; j < V; ++j) {
            const float *b = store + (int64_t)csr_rows[lo + j] * D;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
            for (int d = t; d < D; d += 256) {
                const float x = a[d], y = b[d];
                if (mode == 1) p0 += fabsf(x - y);
                else { p0 += x * y; p1 += x * x; p2 += y * y; }
            }
            for (int r = 0; r < (mode == 1 ? 1 : 3); ++r) {
                red[t] = r == 0 ? p0 : (r == 1 ? p1 : p2);
                __syncthreads();
                for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
                const float tot = red[0];
                __syncthreads();
                if (r == 0) p0 = tot; else if (r == 1) p1 = tot; else p2 = tot;
            }
            if (mode == 1) score += p0;
            else score += p0 / (fmaxf(sqrtf(p1), 1e-8f) * fmaxf(sqrtf(p2), 1e-8f));   // torch.cosine_similarity eps
        }
        if (t == 0) {
            const bool better = mode == 1 ? score < s_best : score > s_best;
            if (better) { s_best = score; s_arg = i; }
        }
        __syncthreads();
    }
    const float *src = store + (int64_t)csr_rows[lo + s_arg] * D;
    for (int d = t; d < D; d += 256) dst[d] = src[d];
    if (t == 0 && out_view) out_view[k] = s_arg;
}

// avg_pooling kept as a RUNNING SUM per instance: an update adds the instance's NEW views (csr rows) to sums[slot] (or, with before[k] = 0, starts
// it) and writes table[slot] = sum / (before + new); a single view is stored as it is, like k_fuse_views.  The mean of all views
// (instance3d.py:9-21 `avg_pooling`, :170-178) without re-reading every view on every update -- with k_top_views = 10000 (ovo.yaml:49) an instance's
// view list, and the CSR the host had to build for it per keyframe, grows with the length of the sequence.
__global__ void __launch_bounds__(256) k_fuse_views_add(const float *__restrict__ store, int D, const int32_t *__restrict__ csr_off,
                                                        const int32_t *__restrict__ csr_rows, const int32_t *__restrict__ before,
                                                        float *__restrict__ sums, float *__restrict__ table, const int32_t *__restrict__ table_rows) {
    const int k = blockIdx.x, t = threadIdx.x;
    const int lo = csr_off[k], V = csr_off[k + 1] - lo, nb = before[k];
    if (V <= 0) return;
    float *acc = sums + (int64_t)table_rows[k] * D, *dst = table + (int64_t)table_rows[k] * D;
    for (int d = t; d < D; d += 256) {
        float s = nb > 0 ? acc[d] : 0.f;
        for (int v = 0; v < V; ++v) s += store[(int64_t)csr_rows[lo + v] * D + d];
        acc[d] = s;
        dst[d] = nb + V == 1 ? s : s / (float)(nb + V);
    }
}

// Dense fusion: one wave per point slot; lanes stride the descriptor with float4.
// HBM traffic per matched point: D*4 read-modify-write of acc (desc rows stay in L2).
__global__ void __launch_bounds__(256) k_scatter_accum(const int16_t *__restrict__ point_seg, int64_t n,
                                                       const int32_t *__restrict__ mask_row, int n_masks,
                                                       const float *__restrict__ desc, int D, float *__restrict__ acc,
                                                       int32_t *__restrict__ cnt, int32_t *__restrict__ touched = nullptr,
                                                       int32_t *__restrict__ n_touched = nullptr, int32_t *__restrict__ n_next = nullptr,
                                                       int shard_rank = 0, int shard_count = 1, int block_log2 = 0) {
    const int lane = threadIdx.x & 63;
    if (n_next && blockIdx.x == 0 && threadIdx.x == 0) *n_next = 0;   // the NEXT keyframe's counter (nobody reads it before that launch)
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int D4 = D >> 2;
    for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; base < n; base += waves * 64) {
        // each lane inspects one point of the 64-point chunk, then the wave serves the hits one by one
        const int64_t i = base + lane;
        int row = -1;
        int64_t li = i;                                            // row of acc / cnt: the point itself, or its slot in this rank's shard
        if (i < n) {
            const int s = point_seg[i];
            if (s >= 0 && s < n_masks) row = mask_row[s];
            if (shard_count > 1) {
                // block-cyclic point shards (blocks of 2^block_log2 points): rank r owns blocks r, r + R, ...; appended points spread evenly
                const int64_t blk = i >> block_log2;
                if (blk % shard_count != shard_rank) row = -1;
                li = ((blk / shard_count) << block_log2) | (i & ((1ll << block_log2) - 1));
            }
        }
        unsigned long long hits = __ballot(row >= 0);
        if (touched && hits) {                                     // compacted list of the rows this keyframe changed: one atomic per chunk
            int at = 0;
            if (lane == 0) at = atomicAdd(n_touched, __popcll(hits));
            at = __shfl(at, 0, 64);
            if (row >= 0) touched[at + __popcll(hits & ((1ull << lane) - 1ull))] = (int32_t)li;
        }
        while (hits) {
            const int src = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const int r = __shfl(row, src, 64);
            const int64_t p = __shfl(li, src, 64);
            const float4 *d4 = (const float4 *)(desc + (int64_t)r * D);
            float4 *a4 = (float4 *)(acc + p * D);
            for (int k = lane; k < D4; k += 64) {
                float4 a = a4[k];
                const float4 b = d4[k];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                a4[k] = a;
            }
            for (int k = (D4 << 2) + lane; k < D; k += 64) acc[p * D + k] += desc[(int64_t)r * D + k];
            if (lane == 0) cnt[p] += 1;
        }
    }
}

}  // namespace

extern "C" {

int ovo_fuse_views(const float *store, int D, const int32_t *csr_off, const int32_t *csr_rows, int n_updates, int mode,
                   float *table, const int32_t *table_rows, int32_t *out_view, ovo_stream_t stream) {
    OVO_REQUIRE(n_updates >= 0 && D > 0 && mode >= 0 && mode <= 2, "bad argument");
    if (n_updates == 0) return OVO_OK;
    OVO_REQUIRE(store && csr_off && csr_rows && table && table_rows, "null pointer");
    k_fuse_views<<<n_updates, 256, 0, (hipStream_t)stream>>>(store, D, csr_off, csr_rows, mode, table, table_rows, out_view);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// ovo_scatter_accum that also emits WHICH points it changed: touched i32[>= n] receives their indices (any order), n_touched i32[1]
// (zero on entry) their number -- the row list of ovo_similarity_rows, so that only those points are re-queried.  n_next (optional)
// is set to zero: with two counters used alternately every call prepares the next one's and no fill launch is needed.
// shard_count > 1: acc / cnt hold only this rank's block-cyclic shard of the points (blocks of shard_block points, a power of two;
// block b belongs to rank b % shard_count and sits at local block b / shard_count); points of other ranks are skipped and `touched`
// receives LOCAL row numbers.  Every rank applies every keyframe's descriptors to its own rows, in keyframe order: the shards hold,
// bit for bit, the rows a single accumulator would (no floating-point reduction across ranks).
int ovo_scatter_accum_touched(const int16_t *point_seg, int64_t n, const int32_t *mask_row, int n_masks, const float *desc,
                              int D, float *acc, int32_t *cnt, int32_t *touched, int32_t *n_touched, int32_t *n_next,
                              int shard_rank, int shard_count, int shard_block, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && n < (1ll << 31) && D > 0 && n_masks > 0, "bad argument");
    OVO_REQUIRE(shard_count >= 1 && shard_rank >= 0 && shard_rank < shard_count && shard_block > 0 && (shard_block & (shard_block - 1)) == 0,
                "bad shard description (shard_block must be a power of two)");
    int block_log2 = 0;
    while ((1 << block_log2) < shard_block) ++block_log2;
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(point_seg && mask_row && desc && acc && cnt && (touched == nullptr) == (n_touched == nullptr), "null pointer");
    OVO_REQUIRE((((uintptr_t)desc | (uintptr_t)acc) & 15) == 0 && D % 4 == 0, "acc/desc must be 16-byte aligned, D % 4 == 0");
    k_scatter_accum<<<ovo_grid((n + 63) / 64 * 64, 256), 256, 0, (hipStream_t)stream>>>(point_seg, n, mask_row, n_masks,
                                                                                       desc, D, acc, cnt, touched, n_touched, n_next,
                                                                                       shard_rank, shard_count, block_log2);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_fuse_views_add(const float *store, int D, const int32_t *csr_off, const int32_t *csr_rows, const int32_t *before, int n_updates,
                       float *sums, float *table, const int32_t *table_rows, ovo_stream_t stream) {
    OVO_REQUIRE(n_updates >= 0 && D > 0, "bad shape");
    if (n_updates == 0) return OVO_OK;
    OVO_REQUIRE(store && csr_off && csr_rows && before && sums && table && table_rows, "null pointer");
    k_fuse_views_add<<<n_updates, 256, 0, (hipStream_t)stream>>>(store, D, csr_off, csr_rows, before, sums, table, table_rows);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_scatter_accum(const int16_t *point_seg, int64_t n, const int32_t *mask_row, int n_masks, const float *desc,
                      int D, float *acc, int32_t *cnt, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && D > 0 && n_masks > 0, "bad argument");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(point_seg && mask_row && desc && acc && cnt, "null pointer");
    OVO_REQUIRE((((uintptr_t)desc | (uintptr_t)acc) & 15) == 0 && D % 4 == 0, "acc/desc must be 16-byte aligned, D % 4 == 0");
    k_scatter_accum<<<ovo_grid((n + 63) / 64 * 64, 256), 256, 0, (hipStream_t)stream>>>(point_seg, n, mask_row, n_masks,
                                                                                       desc, D, acc, cnt);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // extern "C"
