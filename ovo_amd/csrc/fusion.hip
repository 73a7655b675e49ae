// fusion.hip -- multi-view descriptor fusion (instance form) and dense per-point scatter-accumulate.
#include <float.h>

#include <type_traits>

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

// ---- dense fusion: acc[point] += desc[mask_row[seg(point)]], cnt[point] += 1 for every point a mask of this keyframe covers --------------------------
// HBM traffic per matched point: D*4 read-modify-write of acc (desc rows stay in L2); a point is hit at most once per launch (one segment per point).
//
// Round 5: the one-kernel form (each wave scans 64 points and serves its own hits one after the other, every hit a dependent load -> add -> store
// chain per 1 KB piece) ran at 2.0 TB/s of PMC bytes -- 98.6 us for 198 MB at the headline workload.  Hits are clustered: the
// frustum's points are consecutive map rows, so a few hundred of the 8192 waves held 64 hits each (64 x 4 dependent round trips to HBM) while most
// held none -- the launch lasted as long as its fullest wave.  Now:
//   k_scatter_scan  : the scan only -- segment id -> descriptor row, shard filter, the compacted list of hit rows (`touched`, one atomic per 64 points);
//   k_scatter_apply : the hits of the list dealt round-robin to 8192 waves (count read from the device), one 4 KB row per wave at a time, TWO rows
//                     in flight per wave with every 16-byte load of both issued before the first add (NV x 4 loads per lane).
// Without a `touched` list (ovo_scatter_accum: no workspace in its signature) the scan serves its own hits, with the same unrolled body.
template <int NV>
__device__ __forceinline__ void accum_rows(float *__restrict__ acc, const float *__restrict__ desc, int32_t *__restrict__ cnt, int D4, int D, int lane,
                                           int64_t p0, int r0, int64_t p1, int r1, bool two) {
    float4 *a0 = (float4 *)(acc + p0 * D), *a1 = (float4 *)(acc + p1 * D);
    const float4 *d0 = (const float4 *)(desc + (int64_t)r0 * D), *d1 = (const float4 *)(desc + (int64_t)r1 * D);
    float4 x0[NV], y0[NV], x1[NV], y1[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int k = v * 64 + lane;
        if (k < D4) { x0[v] = a0[k]; y0[v] = d0[k]; }
    }
    if (two) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = v * 64 + lane;
            if (k < D4) { x1[v] = a1[k]; y1[v] = d1[k]; }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int k = v * 64 + lane;
        if (k < D4) a0[k] = make_float4(x0[v].x + y0[v].x, x0[v].y + y0[v].y, x0[v].z + y0[v].z, x0[v].w + y0[v].w);
    }
    if (two) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = v * 64 + lane;
            if (k < D4) a1[k] = make_float4(x1[v].x + y1[v].x, x1[v].y + y1[v].y, x1[v].z + y1[v].z, x1[v].w + y1[v].w);
        }
    }
    if (lane == 0) { cnt[p0] += 1; if (two) cnt[p1] += 1; }
}
// rows wider than 5 x 64 float4 (D > 1280): the plain strided loop
__device__ __forceinline__ void accum_row_wide(float *__restrict__ acc, const float *__restrict__ desc, int32_t *__restrict__ cnt, int D4, int D, int lane,
                                               int64_t p, int r) {
    float4 *a4 = (float4 *)(acc + p * D);
    const float4 *d4 = (const float4 *)(desc + (int64_t)r * D);
    for (int k = lane; k < D4; k += 64) {
        float4 a = a4[k];
        const float4 b = d4[k];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a4[k] = a;
    }
    if (lane == 0) cnt[p] += 1;
}

template <int NV>
__global__ void __launch_bounds__(256) k_scatter_scan(const int16_t *__restrict__ point_seg, int64_t n,
                                                      const int32_t *__restrict__ mask_row, int n_masks,
                                                      const float *__restrict__ desc, int D, float *__restrict__ acc,
                                                      int32_t *__restrict__ cnt, int32_t *__restrict__ touched,
                                                      int32_t *__restrict__ n_touched, int32_t *__restrict__ n_next,
                                                      int shard_rank, int shard_count, int block_log2) {
    const int lane = threadIdx.x & 63;
    if (n_next && blockIdx.x == 0 && threadIdx.x == 0) *n_next = 0;   // the NEXT keyframe's counter (nobody reads it before that launch)
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int D4 = D >> 2;
    for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; base < n; base += waves * 64) {
        // each lane inspects one point of the 64-point chunk
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
        if (touched) {                                             // compacted list of the rows this keyframe changes: one atomic per chunk;
            if (hits) {                                            // k_scatter_apply does the arithmetic
                int at = 0;
                if (lane == 0) at = atomicAdd(n_touched, __popcll(hits));
                at = __shfl(at, 0, 64);
                if (row >= 0) touched[at + __popcll(hits & ((1ull << lane) - 1ull))] = (int32_t)li;
            }
            continue;
        }
        while (hits) {                                             // no list: the wave serves its hits itself, two at a time
            const int s0 = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const bool two = hits != 0;
            const int s1 = two ? __ffsll((long long)hits) - 1 : s0;
            if (two) hits &= hits - 1;
            const int r0 = __shfl(row, s0, 64), r1 = __shfl(row, s1, 64);
            const int64_t p0 = __shfl(li, s0, 64), p1 = __shfl(li, s1, 64);
            if (NV > 0) accum_rows<(NV > 0 ? NV : 1)>(acc, desc, cnt, D4, D, lane, p0, r0, p1, r1, two);
            else { accum_row_wide(acc, desc, cnt, D4, D, lane, p0, r0); if (two) accum_row_wide(acc, desc, cnt, D4, D, lane, p1, r1); }
        }
    }
}

template <int NV>
__global__ void __launch_bounds__(256) k_scatter_apply(const int32_t *__restrict__ touched, const int32_t *__restrict__ n_touched,
                                                       const int16_t *__restrict__ point_seg, const int32_t *__restrict__ mask_row,
                                                       const float *__restrict__ desc, int D, float *__restrict__ acc, int32_t *__restrict__ cnt,
                                                       int shard_rank, int shard_count, int block_log2) {
    const int lane = threadIdx.x & 63;
    const int W = gridDim.x * (blockDim.x >> 6), w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int count = *n_touched, D4 = D >> 2;
    // the global point of local row li (the inverse of the scan's shard map): its segment names the descriptor row
    auto desc_row = [&](int li) {
        int64_t i = li;
        if (shard_count > 1) {
            const int64_t lb = (int64_t)li >> block_log2;
            i = ((lb * shard_count + shard_rank) << block_log2) | (li & ((1 << block_log2) - 1));
        }
        return mask_row[point_seg[i]];
    };
    for (int h = w; h < count; h += 2 * W) {
        const bool two = h + W < count;
        const int p0 = __builtin_amdgcn_readfirstlane(touched[h]), p1 = __builtin_amdgcn_readfirstlane(touched[two ? h + W : h]);
        const int r0 = __builtin_amdgcn_readfirstlane(desc_row(p0)), r1 = __builtin_amdgcn_readfirstlane(desc_row(p1));
        if (NV > 0) accum_rows<(NV > 0 ? NV : 1)>(acc, desc, cnt, D4, D, lane, p0, r0, p1, r1, two);
        else { accum_row_wide(acc, desc, cnt, D4, D, lane, p0, r0); if (two) accum_row_wide(acc, desc, cnt, D4, D, lane, p1, r1); }
    }
}

// ---- round 6: the dense fusion of one keyframe and the re-query of the rows it changed in ONE pass, from the tracking pass's own hit list --------------
// ovo_scatter_accum_touched + ovo_similarity_rows were three launches per keyframe: the scan (33 us to find, in 2.6 MB of point_seg, the points
// k_track_project had just had in registers), the apply (read + write of every hit row) and the query (the same rows read a third time).  Here a wave owns
// 16 hits and walks their rows in steps of 32 columns:
//   * MEMORY side, "load layout": lane l handles piece l & 7 (16 bytes) of row l >> 3 and of row 8 + (l >> 3) -- a wave instruction covers 8 rows x one whole
//     128-byte line; accumulator pieces + descriptor pieces (L2-resident: <= 128 rows) are added in that layout and stored back as whole lines
//     (non-temporal).  (First form, measured: the lanes loaded straight in the MFMA's layout, 64 bytes per row and instruction, as k_similarity_mfma's f32
//     loader -- 78.6 us for 200 MB in the bench's isolated pass = 2.4 TB/s, whether 2 or 8 steps were in flight: half-line requests, not latency.)
//   * ARITHMETIC side: the sums cross a 2 KB per-wave LDS slab (XOR-swizzled 16-byte pieces, conflict-free both ways) into the layout of the exact-f32 MFMA
//     (v_mfma_f32_16x16x4_f32; lane (rr, g) holds columns k0 + 4 g .. and k0 + 16 + 4 g .. of row rr) against the text rows in LDS -- the SAME instruction
//     sequence per row as k_similarity_mfma<0>, so class / confidence are bit-identical to querying the stored row afterwards
//     (tests/test_gpu_features.py::test_scatter_query_fused_vs_two_passes), and the accumulator row is the x + y of k_scatter_apply.
// HBM bytes per hit: 2 D 4 (row in and out) instead of 3 D 4 + the scan.
typedef __attribute__((ext_vector_type(4))) float sq_f32x4;
constexpr int SQ_WAVES = 8, SQ_SLAB = 2048;                          // waves per workgroup; bytes of one staging slab (16 rows x 128 B), two per wave
__global__ void __launch_bounds__(512) k_scatter_query(const int32_t *__restrict__ hits, const int32_t *__restrict__ n_hits,
                                                       const int16_t *__restrict__ point_seg, const int32_t *__restrict__ mask_row, int n_masks,
                                                       const float *__restrict__ desc, int D, float *__restrict__ acc, int32_t *__restrict__ cnt,
                                                       int shard_rank, int shard_count, int block_log2, const float *__restrict__ T, int Q, int siglip,
                                                       float scale_exp, float bias, float th, long long *__restrict__ out_cls, float *__restrict__ out_conf,
                                                       int32_t *__restrict__ n_live) {
    extern __shared__ __attribute__((aligned(16))) char sq_smem[];                  // staging slabs [8 waves][2][2 KB], then the text rows [Q <= 16][D]
    float *sT = (float *)(sq_smem + SQ_WAVES * 2 * SQ_SLAB);
    const int lane = threadIdx.x & 63, rr = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6, WPB = blockDim.x >> 6;
    const int count = *n_hits;
    const int groups = (count + 15) >> 4;
    if ((int)blockIdx.x >= groups) return;                                          // (workgroup-uniform: before the barrier below)
    if (T) {
        for (int i = threadIdx.x * 4; i < Q * D; i += blockDim.x * 4) *(float4 *)(sT + i) = *(const float4 *)(T + i);
        __syncthreads();
    }
    const float *tq = sT + (rr < Q ? rr : Q - 1) * D;
    const int D32 = D & ~31;
    char *slab = sq_smem + wv * 2 * SQ_SLAB;
    // load layout: rows ra = l >> 3 and rb = 8 + ra, piece pc = l & 7;  slab address of (row, piece) = row * 128 + ((piece ^ ((row >> 1) & 7)) << 4)
    const int ra = lane >> 3, rb = 8 + ra, pc = lane & 7;
    const int wr_a = ra * 128 + ((pc ^ ((ra >> 1) & 7)) << 4), wr_b = rb * 128 + ((pc ^ ((rb >> 1) & 7)) << 4);
    const int rd_0 = rr * 128 + ((g ^ ((rr >> 1) & 7)) << 4), rd_1 = rr * 128 + (((4 + g) ^ ((rr >> 1) & 7)) << 4);
    // groups are dealt wave-slot-major: slot w of every workgroup before slot w + 1 of any, so a short list still spreads over all CUs
    for (int grp = wv * gridDim.x + blockIdx.x; grp < groups; grp += WPB * gridDim.x) {
        const int c = grp * 16 + rr;
        int li = 0, row = -1;
        if (c < count) {
            li = hits[c];
            int64_t i = li;
            if (shard_count > 1) {
                const int64_t lb = (int64_t)li >> block_log2;
                i = ((lb * shard_count + shard_rank) << block_log2) | (li & ((1 << block_log2) - 1));
            }
            const int sgm = point_seg[i];
            if (sgm >= 0 && sgm < n_masks) row = mask_row[sgm];
        }
        const bool live = row >= 0;
        const unsigned long long lm = __ballot(live);
        if (!lm) continue;
        if (n_live && lane == 0) atomicAdd(n_live, (int)__popcll(lm & 0xffffull));   // (profiled passes only: rows really changed)
        // the two rows this lane moves (lanes 0 .. 15 hold candidates 0 .. 15)
        const int li_a = __shfl(li, ra, 64), li_b = __shfl(li, rb, 64), row_a = __shfl(row, ra, 64), row_b = __shfl(row, rb, 64);
        const bool live_a = row_a >= 0, live_b = row_b >= 0;
        // (rows that are not live read row 0 of both tables -- an unconditional 16-byte load -- and store nothing: with the load inside a select the compiler
        //  made four guarded dword loads of each, 75 us per launch)
        float *ap_a = acc + (int64_t)(live_a ? li_a : 0) * D + pc * 4, *ap_b = acc + (int64_t)(live_b ? li_b : 0) * D + pc * 4;
        const float *dp_a = desc + (int64_t)(live_a ? row_a : 0) * D + pc * 4, *dp_b = desc + (int64_t)(live_b ? row_b : 0) * D + pc * 4;
        sq_f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // A BLOCK of U steps is fetched whole before its first add, and the next block is in flight while this one is added, stored and multiplied
        constexpr int U = 4;
        auto fetch = [&](float4 (&a)[2 * U], float4 (&d)[2 * U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 32 * u < D32 ? k0 + 32 * u : 0;                       // (steps past the row: re-read its start, never used)
                a[2 * u] = *(const float4 *)(ap_a + k);
                a[2 * u + 1] = *(const float4 *)(ap_b + k);
                d[2 * u] = *(const float4 *)(dp_a + k);
                d[2 * u + 1] = *(const float4 *)(dp_b + k);
            }
        };
        auto mfma8 = [&](int k, const sq_f32x4 &a0, const sq_f32x4 &a1) {
            const float4 t0 = *(const float4 *)(tq + k + g * 4), t1 = *(const float4 *)(tq + k + 16 + g * 4);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.x, a0[0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.y, a0[1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.z, a0[2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.w, a0[3], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.x, a1[0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.y, a1[1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.z, a1[2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.w, a1[3], s, 0, 0, 0);
        };
        // ALL: every one of the wave's 16 rows is live (the usual case) -- its stores are unconditional.  A store under a per-lane condition is a basic
        // block of its own, and the wait-count pass then drains EVERYTHING in flight (s_waitcnt vmcnt(0)) before it: the next block's loads with it.
        auto finish = [&](auto ALL_, float4 (&a)[2 * U], float4 (&d)[2 * U], int k0) {
            constexpr bool ALL = decltype(ALL_)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 32 * u;
                if (k >= D32) break;
                const sq_f32x4 xa = {a[2 * u].x + d[2 * u].x, a[2 * u].y + d[2 * u].y, a[2 * u].z + d[2 * u].z, a[2 * u].w + d[2 * u].w};
                const sq_f32x4 xb = {a[2 * u + 1].x + d[2 * u + 1].x, a[2 * u + 1].y + d[2 * u + 1].y, a[2 * u + 1].z + d[2 * u + 1].z, a[2 * u + 1].w + d[2 * u + 1].w};
                if (ALL || live_a) __builtin_nontemporal_store(xa, (sq_f32x4 *)(ap_a + k));
                if (ALL || live_b) __builtin_nontemporal_store(xb, (sq_f32x4 *)(ap_b + k));
                if (T) {
                    char *sl = slab + (u & 1) * SQ_SLAB;                                 // (the step before last read the other slab: same wave, LDS in order)
                    *(sq_f32x4 *)(sl + wr_a) = xa;
                    *(sq_f32x4 *)(sl + wr_b) = xb;
                    __builtin_amdgcn_wave_barrier();
                    const sq_f32x4 a0 = *(const sq_f32x4 *)(sl + rd_0), a1 = *(const sq_f32x4 *)(sl + rd_1);
                    mfma8(k, a0, a1);
                }
            }
        };
        auto walk = [&](auto ALL_) {
            float4 xa[2 * U], xd[2 * U], ya[2 * U], yd[2 * U];
            fetch(xa, xd, 0);
            for (int k0 = 0; k0 < D32; k0 += 64 * U) {                                 // two blocks per trip: static buffer roles
                fetch(ya, yd, k0 + 32 * U);                                            // (past the row's end: a re-read of its start, never used)
                finish(ALL_, xa, xd, k0);
                fetch(xa, xd, k0 + 64 * U);
                finish(ALL_, ya, yd, k0 + 32 * U);
            }
        };
        if ((lm & 0xffffull) == 0xffffull) walk(std::integral_constant<bool, true>{});
        else walk(std::integral_constant<bool, false>{});
        if (D32 < D) {                                                               // one 16-wide tail step: pieces 0 .. 3 of every row
            sq_f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = xa;
            if (pc < 4) {
                if (live_a) { const float4 v = *(const float4 *)(ap_a + D32), w = *(const float4 *)(dp_a + D32); xa = sq_f32x4{v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w}; *(sq_f32x4 *)(ap_a + D32) = xa; }
                if (live_b) { const float4 v = *(const float4 *)(ap_b + D32), w = *(const float4 *)(dp_b + D32); xb = sq_f32x4{v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w}; *(sq_f32x4 *)(ap_b + D32) = xb; }
            }
            if (T) {
                *(sq_f32x4 *)(slab + wr_a) = xa;
                *(sq_f32x4 *)(slab + wr_b) = xb;
                __builtin_amdgcn_wave_barrier();
                const sq_f32x4 a0 = *(const sq_f32x4 *)(slab + rd_0);
                const float4 t0 = *(const float4 *)(tq + D32 + g * 4);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.x, a0[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.y, a0[1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.z, a0[2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.w, a0[3], s, 0, 0, 0);
            }
        }
        int cn = 0;
        if (live) { cn = cnt[li] + 1; }
        __builtin_amdgcn_wave_barrier();
        if (live && g == 0) cnt[li] = cn;                                            // (the four lanes of a row read the same old count above)
        if (!T) continue;
        // lane (row rr, g): s[r] = S[4 g + r][row] -- the finish of k_similarity_mfma (mean = sum / count, first maximum, threshold)
        const float rs = cn > 0 ? 1.0f / (float)cn : 0.f;
        float best = -3.0e38f;
        int arg = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 4 * g + r;
            float v = s[r] * rs;
            if (siglip) v = 1.0f / (1.0f + __expf(-(v * scale_exp + bias)));
            if (q < Q && v > best) { best = v; arg = q; }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (g == 0 && live) {
            if (best <= th) { best = 0.f; arg = -1; }
            out_conf[li] = best;
            out_cls[li] = arg;
        }
    }
}

int launch_scatter(const int16_t *point_seg, int64_t n, const int32_t *mask_row, int n_masks, const float *desc, int D, float *acc, int32_t *cnt,
                   int32_t *touched, int32_t *n_touched, int32_t *n_next, int shard_rank, int shard_count, int block_log2, hipStream_t s) {
    const int nv = (D / 4 + 63) / 64;
    const int g_scan = ovo_grid((n + 63) / 64 * 64, 256);
    const bool prof = touched && ovo_prof_enabled();
    if (prof) ovo_prof_begin(9, 0.0, s);                         // bytes follow from the hit count, read back after the end event (ovo_prof_count)
#define SCAN(NV) k_scatter_scan<NV><<<g_scan, 256, 0, s>>>(point_seg, n, mask_row, n_masks, desc, D, acc, cnt, touched, n_touched, n_next, shard_rank, shard_count, block_log2)
#define APPLY(NV) k_scatter_apply<NV><<<2048, 256, 0, s>>>(touched, n_touched, point_seg, mask_row, desc, D, acc, cnt, shard_rank, shard_count, block_log2)
    switch (nv) {
    case 1: SCAN(1); if (touched) APPLY(1); break;
    case 2: SCAN(2); if (touched) APPLY(2); break;
    case 3: SCAN(3); if (touched) APPLY(3); break;
    case 4: SCAN(4); if (touched) APPLY(4); break;
    case 5: SCAN(5); if (touched) APPLY(5); break;
    default: SCAN(0); if (touched) APPLY(0); break;
    }
#undef SCAN
#undef APPLY
    if (prof) { ovo_prof_end(s); ovo_prof_count(n_touched, 2.0 * D * 4.0 + 12.0, 2.0 * (double)n, s); }   // per hit: the row read + written, its list entry, cnt; + the segment ids
    return OVO_OK;
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
    launch_scatter(point_seg, n, mask_row, n_masks, desc, D, acc, cnt, touched, n_touched, n_next, shard_rank, shard_count, block_log2, (hipStream_t)stream);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// ovo_scatter_accum_touched + ovo_similarity_rows in one launch, driven by the hit list of the tracking pass (ovo_track_step_t.hits; ABI 11).
// hits[0 .. *n_hits) name the (local) rows whose point_seg >= 0; rows whose mask has no descriptor (mask_row < 0) are skipped.  T == NULL: accumulate only.
int ovo_scatter_accum_query(const int32_t *hits, const int32_t *n_hits, int64_t max_hits, const int16_t *point_seg, const int32_t *mask_row, int n_masks,
                            const float *desc, int D, float *acc, int32_t *cnt, int shard_rank, int shard_count, int shard_block,
                            const float *T, int Q, int siglip, float logit_scale, float logit_bias, float th, int64_t *out_cls, float *out_conf,
                            ovo_stream_t stream) {
    OVO_REQUIRE(max_hits >= 0 && D > 0 && n_masks > 0, "bad argument");
    OVO_REQUIRE(shard_count >= 1 && shard_rank >= 0 && shard_rank < shard_count && shard_block > 0 && (shard_block & (shard_block - 1)) == 0,
                "bad shard description (shard_block must be a power of two)");
    if (max_hits == 0) return OVO_OK;
    OVO_REQUIRE(hits && n_hits && point_seg && mask_row && desc && acc && cnt, "null pointer");
    OVO_REQUIRE((((uintptr_t)desc | (uintptr_t)acc) & 15) == 0 && D % 16 == 0, "acc/desc must be 16-byte aligned, D % 16 == 0");
    size_t lds = 0;
    if (T) {
        OVO_REQUIRE(Q > 0 && out_cls && out_conf, "query without outputs");
        if (Q > 16 || (size_t)Q * D * sizeof(float) > 96 * 1024) return OVO_E_UNSUPPORTED;       // the caller runs the two passes
        OVO_REQUIRE(((uintptr_t)T & 15) == 0, "T must be 16-byte aligned");
        lds = (size_t)Q * D * sizeof(float) + SQ_WAVES * 2 * SQ_SLAB;
    }
    int block_log2 = 0;
    while ((1 << block_log2) < shard_block) ++block_log2;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        OVO_HIP(hipFuncSetAttribute((const void *)k_scatter_query, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024 + SQ_WAVES * 2 * SQ_SLAB));
        attr_done = true;
    }
    const int64_t groups = (max_hits + 15) / 16;
    const int grid = (int)(groups < 256 ? groups : 256);
    const bool prof = ovo_prof_enabled();
    int32_t *n_live = nullptr;
    if (prof) {                                                  // rows really changed (listed rows whose mask has a descriptor): counted by the kernel itself
        static int32_t *live_ring = nullptr;
        static unsigned live_at = 0;
        if (!live_ring && hipMalloc((void **)&live_ring, 4096 * sizeof(int32_t)) != hipSuccess) live_ring = nullptr;
        if (live_ring) { n_live = live_ring + (live_at++ & 4095); OVO_HIP(hipMemsetAsync(n_live, 0, sizeof(int32_t), s)); }
        ovo_prof_begin(9, 0.0, s);                               // bytes follow from that count, read back after the end event (ovo_prof_count)
    }
    k_scatter_query<<<grid, 512, lds, s>>>(hits, n_hits, point_seg, mask_row, n_masks, desc, D, acc, cnt, shard_rank, shard_count, block_log2, T, Q, siglip,
                                           expf(logit_scale), logit_bias, th, (long long *)out_cls, out_conf, n_live);
    if (prof) { ovo_prof_end(s); ovo_prof_count(n_live, 2.0 * D * 4.0 + 12.0, 0.0, s); }      // per changed row: the row read + written, its list entry, its segment id, cnt
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
    launch_scatter(point_seg, n, mask_row, n_masks, desc, D, acc, cnt, nullptr, nullptr, nullptr, 0, 1, 0, (hipStream_t)stream);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // extern "C"
