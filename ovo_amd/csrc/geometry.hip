// geometry.hip -- frustum cull, projection, depth matching, back-projection, instance voting.
//
// Bandwidth-bound integer/float32 work over the point map (12 B per point read).  Numerics are pinned:
// every 3- or 4-term product is a left-to-right FMA chain, divisions are IEEE, rounding to pixels is
// half-to-even -- the exact sequence oracle/ovo_oracle.c uses, which is bit-identical to the reference's
// torch-CPU einsum (tests/golden/geometry_*.npz).  This file is compiled with -ffp-contract=off.
#include <limits.h>
#include <string.h>

#include "common.h"
#include "depth_filter.h"

namespace {

__device__ __forceinline__ float dot3(const float *m, float x, float y, float z) {
    float a = __fmul_rn(m[0], x);
    a = __fmaf_rn(m[1], y, a);
    a = __fmaf_rn(m[2], z, a);
    return a;
}
__device__ __forceinline__ float dot4(const float *m, float x, float y, float z, float w) {
    float a = __fmul_rn(m[0], x);
    a = __fmaf_rn(m[1], y, a);
    a = __fmaf_rn(m[2], z, a);
    a = __fmaf_rn(m[3], w, a);
    return a;
}
// tensor.int() of the reference's CPU path: truncation, INT_MIN when unrepresentable.
__device__ __forceinline__ int f2i(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

__device__ __forceinline__ bool in_frustum(const ovo_camera_t &c, float x, float y, float z) {
    if (!(x >= c.aabb[0] && x <= c.aabb[3] && y >= c.aabb[1] && y <= c.aabb[4] && z >= c.aabb[2] && z <= c.aabb[5]))
        return false;
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 6; ++p) ok = ok && (dot4(c.planes + 4 * p, x, y, z, 1.0f) <= 0.0f);
    return ok;
}

__device__ __forceinline__ void project(const ovo_camera_t &c, float x, float y, float z, float w, float &zc,
                                        int &u, int &v) {
    float lx = dot4(c.w2c, x, y, z, w), ly = dot4(c.w2c + 4, x, y, z, w);
    float lz = dot4(c.w2c + 8, x, y, z, w), lw = dot4(c.w2c + 12, x, y, z, w);
    zc = lz;
    float cx = __fdiv_rn(lx, lw), cy = __fdiv_rn(ly, lw), cz = __fdiv_rn(lz, lw);
    float pu = dot3(c.K, cx, cy, cz), pv = dot3(c.K + 3, cx, cy, cz), pw = dot3(c.K + 6, cx, cy, cz);
    u = f2i(rintf(__fdiv_rn(pu, pw)));
    v = f2i(rintf(__fdiv_rn(pv, pw)));
}

__device__ __forceinline__ bool depth_match(const ovo_camera_t &c, const float *__restrict__ depth, float zc, int u,
                                            int v) {
    if (!(u < c.w && v < c.h && u >= 0 && v >= 0)) return false;
    float d = depth[(int64_t)v * c.w + u];
    return (fabsf(__fsub_rn(zc, d)) < c.th) && (d != 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Ordered stream compaction: (A) one ballot word per 64 items, (B) single-block exclusive scan of the
// word popcounts, (C) emit at offset + rank-in-word.  Output order == input order, which the reference's
// boolean-mask indexing guarantees and pcd_ids depend on.
// ws layout: u64 words[n_words] | i64 offs[n_words] | i64 sums[chunks]
struct CompactWs {
    unsigned long long *words;
    long long *offs;
    long long *sums;           // one total per SCAN_CHUNK words
    int64_t n_words;
};

__host__ CompactWs carve(void *ws, int64_t n) {
    CompactWs c;
    c.n_words = (n + 63) / 64;
    c.words = (unsigned long long *)ws;
    c.offs = (long long *)(c.words + c.n_words);
    c.sums = c.offs + c.n_words;
    return c;
}

// (B) as three small launches (a single workgroup walking 156k words took 330 us of a 370 us frustum cull at 10 M points):
//   k_scan_sums   one workgroup per SCAN_CHUNK words: total popcount of its chunk            -> sums[b]
//   k_scan_bases  one workgroup: exclusive scan of the (few hundred) chunk totals, in place    -> sums[b] = base of chunk b, *total
//   k_scan_words  one workgroup per chunk: exclusive scan of its words' popcounts + its base   -> offs[w]
constexpr int SCAN_CHUNK = 2048;     // words per workgroup (256 threads x 8 consecutive words)

__device__ __forceinline__ long long block_exclusive_scan(long long v, long long *sh, long long &block_total) {
    // 256 threads: wave-level inclusive scan by shuffles, then the 4 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    long long before = 0;
    for (int k = 0; k < wave; ++k) before += sh[k];
    block_total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return before + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_sums(const unsigned long long *__restrict__ words, int64_t n_words, long long *__restrict__ sums) {
    __shared__ long long sh[4];
    const int64_t w0 = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
    long long c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (w0 + k < n_words) c += __popcll(words[w0 + k]);
    long long total;
    block_exclusive_scan(c, sh, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_bases(long long *__restrict__ sums, int64_t n_chunks, long long *__restrict__ total) {
    __shared__ long long sh[4];
    long long run = 0;
    for (int64_t c0 = 0; c0 < n_chunks; c0 += 256) {             // 256 chunk totals per trip (10 M points: one trip)
        const int64_t c = c0 + threadIdx.x;
        const long long v = c < n_chunks ? sums[c] : 0;
        long long tot;
        const long long ex = block_exclusive_scan(v, sh, tot);
        if (c < n_chunks) sums[c] = run + ex;
        run += tot;
    }
    if (threadIdx.x == 0 && total) *total = run;
}

__global__ void __launch_bounds__(256) k_scan_words(const unsigned long long *__restrict__ words, long long *__restrict__ offs,
                                                    int64_t n_words, const long long *__restrict__ bases) {
    __shared__ long long sh[4];
    const int64_t w0 = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
    int pc[8];
    long long c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        pc[k] = w0 + k < n_words ? __popcll(words[w0 + k]) : 0;
        c += pc[k];
    }
    long long total;
    long long run = bases[blockIdx.x] + block_exclusive_scan(c, sh, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (w0 + k < n_words) offs[w0 + k] = run;
        run += pc[k];
    }
}

// the three steps in one workgroup when the words fit one chunk (<= 131 072 items: the back-projection of a 640 x 480 frame)
__device__ void dev_scan_one(const unsigned long long *words, long long *offs, int64_t n_words, long long *total_out) {
    __shared__ long long sh[4];
    const int64_t w0 = (int64_t)threadIdx.x * 8;
    int pc[8];
    long long c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        pc[k] = w0 + k < n_words ? __popcll(words[w0 + k]) : 0;
        c += pc[k];
    }
    long long total;
    long long run = block_exclusive_scan(c, sh, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (w0 + k < n_words) offs[w0 + k] = run;
        run += pc[k];
    }
    if (threadIdx.x == 0 && total_out) *total_out = total;
}
__global__ void __launch_bounds__(256) k_scan_one(const unsigned long long *__restrict__ words, long long *__restrict__ offs, int64_t n_words,
                                                  long long *__restrict__ total_out) {
    dev_scan_one(words, offs, n_words, total_out);
}

// the scan of one compaction: words -> offs, *total = number of set bits
__host__ void scan_words(const unsigned long long *words, long long *offs, long long *sums, int64_t n_words, long long *total, hipStream_t s) {
    const int64_t chunks = (n_words + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (chunks <= 1) {
        k_scan_one<<<1, 256, 0, s>>>(words, offs, n_words, total);
        return;
    }
    k_scan_sums<<<(unsigned)chunks, 256, 0, s>>>(words, n_words, sums);
    k_scan_bases<<<1, 256, 0, s>>>(sums, chunks, total);
    k_scan_words<<<(unsigned)chunks, 256, 0, s>>>(words, offs, n_words, sums);
}

// A workgroup's place in the grid that shares a pass: the launch's own (blockIdx, gridDim) for the one-pass kernels, the persistent
// workgroup's (index, count) inside k_round_chain, where one launch walks through every pass of a round of keyframes.
struct Blk { int bid, nblk; };
__device__ __forceinline__ Blk this_block() { return Blk{(int)blockIdx.x, (int)gridDim.x}; }

template <typename Load, typename Pred>
__device__ __forceinline__ void flag_words(int64_t n, unsigned long long *words, Load load, Pred pred) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n_words = (n + 63) >> 6;
    // four words (4 x 64 items) per wave and trip, the items' loads issued before any arithmetic (bytes in flight, see k_track_project)
    for (int64_t wd0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); wd0 < n_words; wd0 += 4 * waves) {
        float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = (wd0 + k * waves) * 64 + lane;
            if (i < n) p[k] = load(i);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t wd = wd0 + k * waves;
            if (wd >= n_words) break;
            const int64_t i = wd * 64 + lane;
            const bool ok = i < n && pred(p[k]);
            const unsigned long long m = __ballot(ok);
            if (lane == 0) words[wd] = m;
        }
    }
}

template <typename Pred>
__device__ __forceinline__ void flag_words_idx(Blk b, int64_t n, unsigned long long *words, Pred pred) {     // pred(index): small inputs
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)b.nblk * (blockDim.x >> 6);
    const int64_t n_words = (n + 63) >> 6;
    for (int64_t wd = (int64_t)b.bid * (blockDim.x >> 6) + (threadIdx.x >> 6); wd < n_words; wd += waves) {
        const int64_t i = wd * 64 + lane;
        const bool p = i < n && pred(i);
        const unsigned long long m = __ballot(p);
        if (lane == 0) words[wd] = m;
    }
}

template <typename Emit>
__device__ __forceinline__ void emit_words(Blk b, int64_t n, const unsigned long long *words, const long long *offs, Emit emit) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)b.nblk * (blockDim.x >> 6);
    const int64_t n_words = (n + 63) >> 6;
    for (int64_t wd = (int64_t)b.bid * (blockDim.x >> 6) + (threadIdx.x >> 6); wd < n_words; wd += waves) {
        const unsigned long long m = words[wd];
        if ((m >> lane) & 1ull) {
            const long long pos = offs[wd] + __popcll(m & ((1ull << lane) - 1ull));
            emit(wd * 64 + lane, pos);
        }
    }
}

// ---- a2 ----
__global__ void __launch_bounds__(256) k_frustum_flag(const float *__restrict__ pts, int64_t n, ovo_camera_t cam,
                                                      unsigned long long *words) {
    flag_words(n, words, [&](int64_t i) { return make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 1.0f); },
               [&](float4 p) { return in_frustum(cam, p.x, p.y, p.z); });
}
__global__ void __launch_bounds__(256) k_frustum_emit(int64_t n, const unsigned long long *words, const long long *offs,
                                                      int64_t *out_idx) {
    emit_words(this_block(), n, words, offs, [&](int64_t i, long long pos) { out_idx[pos] = i; });
}

// ---- a3 ----
__global__ void __launch_bounds__(256) k_project(const float *__restrict__ pts, int64_t n, int stride, ovo_camera_t cam,
                                                 int32_t *__restrict__ out_uv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float *p = pts + i * stride;
        float zc; int u, v;
        project(cam, p[0], p[1], p[2], stride == 4 ? p[3] : 1.0f, zc, u, v);
        out_uv[2 * i] = u;
        out_uv[2 * i + 1] = v;
    }
}
__global__ void __launch_bounds__(256) k_match_flag(const float *__restrict__ pts, int64_t n, int stride, ovo_camera_t cam,
                                                    const float *__restrict__ depth, unsigned long long *words) {
    flag_words(n, words, [&](int64_t i) {
        const float *p = pts + i * stride;
        return make_float4(p[0], p[1], p[2], stride == 4 ? p[3] : 1.0f);
    }, [&](float4 p) {
        float zc; int u, v;
        project(cam, p.x, p.y, p.z, p.w, zc, u, v);
        return depth_match(cam, depth, zc, u, v);
    });
}
__global__ void __launch_bounds__(256) k_match_emit(const float *__restrict__ pts, int64_t n, int stride, ovo_camera_t cam,
                                                    const unsigned long long *words, const long long *offs,
                                                    int64_t *out_idx, int32_t *out_uv) {
    emit_words(this_block(), n, words, offs, [&](int64_t i, long long pos) {
        const float *p = pts + i * stride;
        float zc; int u, v;
        project(cam, p[0], p[1], p[2], stride == 4 ? p[3] : 1.0f, zc, u, v);
        out_idx[pos] = i;
        out_uv[2 * pos] = u;
        out_uv[2 * pos + 1] = v;
    });
}

// ---- fused tracking pass (a2+a3+a5 + vote histogram) ----
// Wave-aggregated histogram: neighbouring map points come from neighbouring pixels and mostly share the
// (mask, instance) cell, so one atomic per distinct cell per wave instead of one per point.
__device__ __forceinline__ void wave_hist_add(int32_t *hist, int cell, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(cell, leader, 64);
        const unsigned long long same = __ballot(active && cell == k);
        if (lane == leader) atomicAdd(hist + k, (int)__popcll(same));
        todo &= ~same;
    }
}

// The survivors of the cheap frustum test are compacted before the expensive part runs.  PMC at 10 M points (profiles/r02_geom_sq_counters.txt):
// with every lane walking the whole chain the pass issued 4.5x the VALU instructions of the plain frustum flag pass -- nearly every wave
// holds a few in-frustum lanes, so the projection (five IEEE divisions, strict-rounding FMA chains) ran for all of them at ~8 % lane use.
// Round 2 compacted per WORKGROUP (LDS queue, an LDS atomic per wave, four __syncthreads per 1024 points): VALU halved, but the waves
// then waited (SQ_ACTIVE_INST_ANY 18 % of SQ_WAVE_CYCLES).  Round 3: per WAVE -- a wave owns a private LDS ring, a lane's slot is
// tail + mbcnt(ballot), no atomic and no workgroup barrier anywhere in the loop; whenever 64 survivors are queued the wave runs the
// chain on them with all lanes active.
struct WaveQueue {                      // one per wave: 256 slots (a trip adds at most 4 x 64, a full batch of 64 leaves first)
    float x[256], y[256], z[256];
    int i[256];
};

__device__ __forceinline__ void wq_push(WaveQueue &q, int &tail, bool keep, float x, float y, float z, int i) {
    const unsigned long long m = __ballot(keep);
    if (keep) {
        const int slot = (tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & 255;
        q.x[slot] = x; q.y[slot] = y; q.z[slot] = z; q.i[slot] = i;
    }
    tail += (int)__popcll(m);
}

// The points a keyframe's masks cover (seg >= 0), listed while the tracking pass still has them in registers (round 6): the dense scatter-reduce used to
// find them again with a scan over point_seg (k_scatter_scan: 33 us for 2.6 MB, launch + latency).  hits == NULL: no list.  shard_count > 1: only the
// points of this rank's block-cyclic shard are listed, as LOCAL row numbers (the map of ovo_scatter_accum_touched).
struct HitSink { int32_t *hits; int32_t *n_hits; int shard_rank, shard_count, block_log2; };
// A wave collects its hits in 256 LDS slots and reserves list space with ONE returning atomic per flush (when the slots fill, and at the end of its
// pass): the first form reserved per batch of 64 survivors, and the wave waited ~1.5 us for every one of those round trips -- k_track_project 20.6 ->
// 46 us in the bench's isolated pass.
constexpr int HIT_SLOTS = 256;
__device__ __forceinline__ void hits_flush(const HitSink &sink, const int32_t *hbuf, int &hcount) {
    const int lane = threadIdx.x & 63;
    if (hcount > 0) {
        int at = 0;
        if (lane == 0) at = atomicAdd(sink.n_hits, hcount);
        at = __shfl(at, 0, 64);
        for (int k = lane; k < hcount; k += 64) sink.hits[at + k] = hbuf[k];
        __builtin_amdgcn_wave_barrier();
    }
    hcount = 0;
}

constexpr int CNT_SLOTS = 32, CNT_STRIDE = 16;                     // counter slots of a tracking step; u64 words between two slots (128 bytes)

// The chain of one batch of survivors (lane < count active): project, depth-test, colour-frame remap, seg lookup, vote.
__device__ __forceinline__ void track_batch(const WaveQueue &q, int head, int count, const ovo_camera_t &cam, const float *__restrict__ depth,
                                            const int32_t *__restrict__ point_ins, const int32_t *__restrict__ seg_map, int seg_h, int seg_w,
                                            const ovo_ratio_t &ratio, int16_t *__restrict__ point_seg, int32_t *__restrict__ hist, int n_masks,
                                            int hist_cols, long long &n_match, const HitSink &sink, int32_t *hbuf, int &hcount) {
    const int lane = threadIdx.x & 63;
    int cell = 0;
    bool vote = false;
    int hit_row = -1;
    if (lane < count) {
        const int slot = (head + lane) & 255;
        const float x = q.x[slot], y = q.y[slot], z = q.z[slot];
        const int64_t i = q.i[slot];
        int seg = -2;
        float zc; int u, v;
        project(cam, x, y, z, 1.0f, zc, u, v);
        if (depth_match(cam, depth, zc, u, v)) {
            ++n_match;
            if (ratio.enabled) {
                u = f2i(__fmul_rn((float)(u + ratio.crop_edge), ratio.r_w));
                v = f2i(__fmul_rn((float)(v + ratio.crop_edge), ratio.r_h));
            }
            seg = -1;
            if (u >= 0 && v >= 0 && u < seg_w && v < seg_h) seg = seg_map[(int64_t)v * seg_w + u];
            if (seg >= n_masks) seg = -1;
            if (seg >= 0) {
                int ins = point_ins[i];
                if (ins + 1 >= hist_cols) ins = -1;   // never taken: the host sizes hist_cols > max id + 1
                cell = seg * hist_cols + (ins < 0 ? 0 : ins + 1);
                vote = true;
                hit_row = (int)i;
                if (sink.shard_count > 1) {
                    const int64_t blk = i >> sink.block_log2;
                    hit_row = blk % sink.shard_count == sink.shard_rank ? (int)(((blk / sink.shard_count) << sink.block_log2) | (i & ((1ll << sink.block_log2) - 1))) : -1;
                }
            }
        }
        point_seg[i] = (int16_t)seg;
    }
    wave_hist_add(hist, cell, vote);
    if (sink.hits) {                                                // into the wave's LDS slots (hits_flush reserves the list space)
        const unsigned long long m = __ballot(hit_row >= 0);
        if (m) {
            const int c = (int)__popcll(m);
            if (hcount + c > HIT_SLOTS) hits_flush(sink, hbuf, hcount);
            if (hit_row >= 0) hbuf[hcount + (int)__popcll(m & ((1ull << lane) - 1ull))] = hit_row;
            __builtin_amdgcn_wave_barrier();
            hcount += c;
        }
    }
}

__device__ void dev_track_project(Blk b, const float *__restrict__ pts, const int32_t *__restrict__ point_ins,
                                  int64_t n, const ovo_camera_t &cam, const float *__restrict__ depth,
                                  const int32_t *__restrict__ seg_map, int seg_h, int seg_w,
                                  ovo_ratio_t ratio, int16_t *__restrict__ point_seg,
                                  int32_t *__restrict__ hist, int n_masks, int hist_cols,
                                  unsigned long long *__restrict__ counters, int cnt_slots, const HitSink &sink) {
    __shared__ WaveQueue s_q[4];
    __shared__ int32_t s_hits[4][HIT_SLOTS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveQueue &q = s_q[wave];
    int32_t *hbuf = s_hits[wave];
    int hcount = 0;
    long long n_in = 0, n_match = 0;
    int head = 0, tail = 0;                                         // wave-uniform ring positions (mod 256 at use)
    const int64_t step = (int64_t)b.nblk * blockDim.x;
    // trip: 4 points per lane (loads first: bytes in flight), the frustum test, survivors -> the wave's ring; full batches of 64 run the chain
    for (int64_t i0 = (int64_t)b.bid * blockDim.x + threadIdx.x; i0 - lane < n; i0 += 4 * step) {     // wave-uniform trip count
        float px[4], py[4], pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * step;
            if (i < n) { px[k] = pts[3 * i]; py[k] = pts[3 * i + 1]; pz[k] = pts[3 * i + 2]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * step;
            const bool in = i < n && in_frustum(cam, px[k], py[k], pz[k]);
            if (i < n && !in) point_seg[i] = (int16_t)-2;
            wq_push(q, tail, in, px[k], py[k], pz[k], (int)i);
            __builtin_amdgcn_wave_barrier();                        // the ring is private to the wave: LDS operations of one wave stay in order
            if (tail - head >= 64) {                                // at most 63 + 64 queued: the 256-slot ring never wraps onto live entries
                track_batch(q, head, 64, cam, depth, point_ins, seg_map, seg_h, seg_w, ratio, point_seg, hist, n_masks, hist_cols, n_match, sink, hbuf, hcount);
                head += 64; n_in += 1;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (tail - head > 0) {
        n_in += lane < tail - head;
        track_batch(q, head, tail - head, cam, depth, point_ins, seg_map, seg_h, seg_w, ratio, point_seg, hist, n_masks, hist_cols, n_match, sink, hbuf, hcount);
    }
    if (sink.hits) hits_flush(sink, hbuf, hcount);
    // the two counters: wave reduction, then across the workgroup's waves through LDS -> one atomic pair per workgroup
    // (every wave of the grid adding to the same two addresses serialises in the L2 atomic unit: 16k same-address atomics)
    for (int o = 32; o > 0; o >>= 1) {
        n_in += __shfl_xor(n_in, o, 64);
        n_match += __shfl_xor(n_match, o, 64);
    }
    __shared__ long long s_cnt[2][4];
    if (lane == 0) { s_cnt[0][wave] = n_in; s_cnt[1][wave] = n_match; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const long long t = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
        if (t) atomicAdd(counters + (size_t)(b.bid % cnt_slots) * CNT_STRIDE + threadIdx.x, (unsigned long long)t);
    }
    __syncthreads();
}

// Round 6: the counter pair is SPREAD over CNT_SLOTS slots a cache line apart (workgroup b adds to slot b % CNT_SLOTS; the publisher sums them): 1024-4096
// workgroups ending on the same two addresses serialised in one L2 atomic unit (~3 ns each).  cnt_slots = 1: the caller's own i64[2] (ovo_track_project).
// Grid of the tracking pass: 1024 workgroups, not the usual 2048.  Every workgroup ends with one atomic pair on the same two counters and a
// ragged last batch of its waves' rings; with twice the workgroups those fixed costs outweigh the extra waves in flight (tools/geom_bench.py,
// 1 M / 5 M / 10 M points: 33.4 / 47.0 / 62.3 us at 2048, 23.0 / 38.4 / 54.1 at 1024, 22.1 / 49.9 / 81.3 at 512; without the counter atomics 4096
// workgroups would take 48.7 us at 10 M -- the same-address atomics cost ~3 ns each).
constexpr int TRACK_GRID_CAP_DEFAULT = 1024;
static int track_grid_cap() {
    static int cap = getenv("OVO_TRACK_GRID_CAP") ? atoi(getenv("OVO_TRACK_GRID_CAP")) : TRACK_GRID_CAP_DEFAULT;
    if (ovo_knobs_dynamic()) cap = getenv("OVO_TRACK_GRID_CAP") ? atoi(getenv("OVO_TRACK_GRID_CAP")) : TRACK_GRID_CAP_DEFAULT;
    return cap > 0 ? cap : TRACK_GRID_CAP_DEFAULT;
}
#define TRACK_GRID_CAP track_grid_cap()
__global__ void __launch_bounds__(256) k_track_project(const float *__restrict__ pts, const int32_t *__restrict__ point_ins,
                                                       int64_t n, ovo_camera_t cam, const float *__restrict__ depth,
                                                       const int32_t *__restrict__ seg_map, int seg_h, int seg_w,
                                                       ovo_ratio_t ratio, int16_t *__restrict__ point_seg,
                                                       int32_t *__restrict__ hist, int n_masks, int hist_cols,
                                                       unsigned long long *__restrict__ counters, int cnt_slots, const long long *__restrict__ n_dev,
                                                       HitSink sink) {
    if (n_dev) n = *n_dev;                                         // device-resident map size (ovo_track_step): no host round trip
    dev_track_project(this_block(), pts, point_ins, n, cam, depth, seg_map, seg_h, seg_w, ratio, point_seg, hist, n_masks, hist_cols, counters, cnt_slots, sink);
}

// ---- a6: per-mask statistics ----
__device__ void dev_seg_area(Blk b, const int32_t *__restrict__ seg_map, int64_t pixels, int n_masks, int32_t *__restrict__ stats, int *area) {
    for (int i = threadIdx.x; i < n_masks; i += blockDim.x) area[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)b.bid * blockDim.x + threadIdx.x; i < pixels; i += (int64_t)b.nblk * blockDim.x) {
        const int s = seg_map[i];
        if (s >= 0 && s < n_masks) atomicAdd(area + s, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_masks; i += blockDim.x)
        if (area[i]) atomicAdd(stats + 4 * i + 3, area[i]);
}
__global__ void __launch_bounds__(256) k_seg_area(const int32_t *__restrict__ seg_map, int64_t pixels, int n_masks,
                                                  int32_t *__restrict__ stats) {
    extern __shared__ int area[];
    dev_seg_area(this_block(), seg_map, pixels, n_masks, stats, area);
}

__global__ void __launch_bounds__(256) k_vote_stats(const int32_t *__restrict__ hist, int hist_cols, int32_t *__restrict__ stats) {
    // one block per mask: total, assigned, mode with smallest-id tie break (torch.mode on CPU)
    __shared__ long long s_cnt[256];
    __shared__ int s_best[256], s_arg[256];
    const int m = blockIdx.x, t = threadIdx.x;
    const int32_t *row = hist + (int64_t)m * hist_cols;
    long long assigned = 0;
    int best = 0, arg = -1;
    for (int c = 1 + t; c < hist_cols; c += 256) {
        const int v = row[c];
        assigned += v;
        if (v > best) { best = v; arg = c - 1; }     // ascending c per thread => first hit is the smallest id
    }
    s_cnt[t] = assigned; s_best[t] = best; s_arg[t] = arg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            s_cnt[t] += s_cnt[t + o];
            const int b2 = s_best[t + o], a2 = s_arg[t + o];
            if (b2 > s_best[t] || (b2 == s_best[t] && b2 > 0 && (s_arg[t] < 0 || a2 < s_arg[t]))) {
                s_best[t] = b2; s_arg[t] = a2;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        stats[4 * m + 0] = (int)(s_cnt[0] + row[0]);
        stats[4 * m + 1] = (int)s_cnt[0];
        stats[4 * m + 2] = s_arg[0];
    }
}

__global__ void __launch_bounds__(256) k_assign(const int32_t *__restrict__ point_ins, const int16_t *__restrict__ point_seg,
                                                int64_t n, const int32_t *__restrict__ mask_target, int n_masks,
                                                int32_t *__restrict__ out_ins, unsigned long long *__restrict__ new_count,
                                                const long long *__restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int ins = point_ins[i];
        const int s = point_seg[i];
        if (s >= 0 && s < n_masks && ins == -1) {
            const int tgt = mask_target[s];
            if (tgt > -1) { ins = tgt; ++c; }
        }
        out_ins[i] = ins;
    }
    if (new_count) {
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(new_count, (unsigned long long)c);
    }
}

// k_assign with the targets read from the device-side decision block (row stride 6, column 4), in place
__device__ __forceinline__ void dev_assign_res(Blk b, int32_t *__restrict__ point_ins, const int16_t *__restrict__ point_seg, int64_t n,
                                               const int32_t *res, int n_masks) {
    for (int64_t i = (int64_t)b.bid * blockDim.x + threadIdx.x; i < n; i += (int64_t)b.nblk * blockDim.x) {
        const int s = point_seg[i];
        if (s >= 0 && s < n_masks && point_ins[i] == -1) {
            const int tgt = res[8 + 6 * s + 4];
            if (tgt > -1) point_ins[i] = tgt;
        }
    }
}
__global__ void __launch_bounds__(256) k_assign_res(int32_t *__restrict__ point_ins, const int16_t *__restrict__ point_seg, int64_t n,
                                                    const int32_t *__restrict__ res, int n_masks, const long long *__restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    dev_assign_res(this_block(), point_ins, point_seg, n, res, n_masks);
}

// ---- a9 ----
__device__ void dev_map_explained(Blk b, const float *__restrict__ pts, int64_t n, const ovo_camera_t &cam,
                                  const float *__restrict__ depth, uint8_t *__restrict__ explained) {
    __shared__ WaveQueue s_q[4];
    const int lane = threadIdx.x & 63;
    WaveQueue &q = s_q[threadIdx.x >> 6];
    int head = 0, tail = 0;
    const int64_t step = (int64_t)b.nblk * blockDim.x;
    auto batch = [&](int count) {
        if (lane < count) {
            const int slot = (head + lane) & 255;
            float zc; int u, v;
            project(cam, q.x[slot], q.y[slot], q.z[slot], 1.0f, zc, u, v);
            if (depth_match(cam, depth, zc, u, v)) explained[(int64_t)v * cam.w + u] = 1;
        }
    };
    for (int64_t i0 = (int64_t)b.bid * blockDim.x + threadIdx.x; i0 - lane < n; i0 += 4 * step) {     // as dev_track_project
        float px[4], py[4], pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * step;
            if (i < n) { px[k] = pts[3 * i]; py[k] = pts[3 * i + 1]; pz[k] = pts[3 * i + 2]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * step;
            wq_push(q, tail, i < n && in_frustum(cam, px[k], py[k], pz[k]), px[k], py[k], pz[k], 0);
            __builtin_amdgcn_wave_barrier();
            if (tail - head >= 64) { batch(64); head += 64; __builtin_amdgcn_wave_barrier(); }
        }
    }
    if (tail - head > 0) batch(tail - head);
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_map_explained(const float *__restrict__ pts, int64_t n, ovo_camera_t cam,
                                                       const float *__restrict__ depth, uint8_t *__restrict__ explained,
                                                       const long long *__restrict__ state) {
    if (state) {                                                   // device-resident map state {n, next point id} (ovo_map_step)
        n = state[0];
        if (state[1] <= 0) return;                                 // vanilla_mapper.py:56 `if self.max_id > 0`
    }
    dev_map_explained(this_block(), pts, n, cam, depth, explained);
}

struct BackprojArgs {
    float K[9];
    float c2w[16];
    int h, w, ds, ws_w, erode;
};

__device__ __forceinline__ bool px_valid(const float *depth, const uint8_t *explained, int64_t i) {
    return depth[i] > 0.0f && !(explained && explained[i]);
}

__device__ __forceinline__ void dev_backproj_flag(Blk b, const float *__restrict__ depth, const uint8_t *explained, const BackprojArgs &a,
                                                  int64_t n_sub, unsigned long long *words) {
    flag_words_idx(b, n_sub, words, [&](int64_t k) {
        const int y = (int)(k / a.ws_w) * a.ds, x = (int)(k % a.ws_w) * a.ds;
        if (!px_valid(depth, explained, (int64_t)y * a.w + x)) return false;
        if (a.erode) {
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= a.h || xx < 0 || xx >= a.w) continue;
                    if (!px_valid(depth, explained, (int64_t)yy * a.w + xx)) return false;
                }
        }
        return true;
    });
}
__global__ void __launch_bounds__(256) k_backproj_flag(const float *__restrict__ depth, const uint8_t *__restrict__ explained,
                                                       BackprojArgs a, int64_t n_sub, unsigned long long *words,
                                                       const long long *__restrict__ state) {
    if (state && state[1] <= 0) {                                  // empty map so far: no explained-pixel test, no erosion (vanilla_mapper.py:56,62)
        explained = nullptr;
        a.erode = 0;
    }
    dev_backproj_flag(this_block(), depth, explained, a, n_sub, words);
}

// What the LAST workgroup of k_backproj_emit does for ovo_map_step: advance the device-resident map state by the number of appended
// points and publish it to the host's pinned result block (sequence number last, after a system-scope fence).
struct MapCommit {
    long long *state;            // device i64[4] {n, next point id, error flags, ticket}; NULL = plain ovo_map_backproject
    const long long *total;      // number of appended points (the scan's total)
    volatile long long *result;  // pinned host i64[4] {seq, appended, n after, next id after} or NULL
    long long seq, cap;
    long long n_host, id_host;   // the host's exact copy of the state when it has one (n_host >= 0), else read `state`
};

__device__ __forceinline__ void dev_backproj_emit(Blk b, const float *__restrict__ depth, const uint8_t *__restrict__ rgb, const BackprojArgs &a,
                                                  int64_t n_sub, const unsigned long long *words, const long long *offs,
                                                  int64_t base, int32_t first_id, int64_t cap, float *xyz, int32_t *ids, int32_t *ins,
                                                  uint8_t *out_rgb) {
    emit_words(b, n_sub, words, offs, [&](int64_t k, long long pos) {
        const int y = (int)(k / a.ws_w) * a.ds, x = (int)(k % a.ws_w) * a.ds;
        const int64_t px = (int64_t)y * a.w + x;
        const float d = depth[px];
        const float x3 = __fdiv_rn(__fmul_rn(__fsub_rn((float)x, a.K[2]), d), a.K[0]);
        const float y3 = __fdiv_rn(__fmul_rn(__fsub_rn((float)y, a.K[5]), d), a.K[4]);
        const int64_t r = base + pos;
        if (r >= cap) return;                                      // never taken: the host reserves n_upper + n_sub rows (flagged by the commit)
        xyz[3 * r + 0] = dot4(a.c2w, x3, y3, d, 1.0f);
        xyz[3 * r + 1] = dot4(a.c2w + 4, x3, y3, d, 1.0f);
        xyz[3 * r + 2] = dot4(a.c2w + 8, x3, y3, d, 1.0f);
        ids[r] = first_id + (int32_t)pos;
        ins[r] = -1;
        if (rgb) {
            out_rgb[3 * r + 0] = rgb[3 * px + 0];
            out_rgb[3 * r + 1] = rgb[3 * px + 1];
            out_rgb[3 * r + 2] = rgb[3 * px + 2];
        }
    });
}

__global__ void __launch_bounds__(256) k_backproj_emit(const float *__restrict__ depth, const uint8_t *__restrict__ rgb, BackprojArgs a,
                                                       int64_t n_sub, const unsigned long long *words, const long long *offs,
                                                       int64_t base, int32_t first_id, float *xyz, int32_t *ids, int32_t *ins,
                                                       uint8_t *out_rgb, MapCommit mc) {
    if (mc.state) {
        base = mc.n_host >= 0 ? mc.n_host : mc.state[0];
        first_id = (int32_t)(mc.n_host >= 0 ? mc.id_host : mc.state[1]);
    }
    dev_backproj_emit(this_block(), depth, rgb, a, n_sub, words, offs, base, first_id, mc.state ? mc.cap : LLONG_MAX, xyz, ids, ins, out_rgb);
    if (!mc.state) return;
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd((unsigned long long *)(mc.state + 3), 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;                       // every other workgroup has read `base` before it took its ticket
    long long m = *mc.total;
    if (base + m > mc.cap) { m = mc.cap - base; mc.state[2] |= 1; }
    mc.state[0] = base + m;
    mc.state[1] = (long long)first_id + m;
    mc.state[3] = 0;
    if (mc.result) {
        mc.result[1] = m; mc.result[2] = base + m; mc.result[3] = (long long)first_id + m;
        __threadfence_system();
        mc.result[0] = mc.seq;
    }
}

// ---- a6 decisions on the device (ovo.py:255-282): mask -> instance targets, instance-id allocation in mask order, the first mask of
// every instance (the row its other masks are OR-ed into, ovo.py:284-303).  One workgroup; runs as the tail of the vote statistics.
// res rows i32[n_masks, 6] = {matched points, already assigned, mode id, seg-map area, target (-1 none), fused area (-1: not fused)}.
struct Decide {
    int32_t *res;                // device result block: 8 header ints + 6 per mask
    int32_t *dst;                // i32[n_masks]: first mask with the same target (-1: no target)
    int32_t *next_ins;           // device i32[1] next instance id
    int32_t next_host;           // the host's exact copy (>= 0) or -1: read next_ins
    int32_t track_th, n_masks;
};

__device__ __forceinline__ int ld_agent(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void decide_masks(const Decide d, const int32_t *stats) {
    __shared__ int s_scan[256];
    __shared__ int s_base;
    const int t = threadIdx.x;
    if (t == 0) s_base = d.next_host >= 0 ? d.next_host : ld_agent(d.next_ins);
    __syncthreads();
    const int first_new = s_base;
    for (int m0 = 0; m0 < d.n_masks; m0 += 256) {
        const int m = m0 + t;
        int kind = 0, mode = -1;
        if (m < d.n_masks) {
            const int n_pts = ld_agent(stats + 4 * m), n_as = ld_agent(stats + 4 * m + 1);
            mode = ld_agent(stats + 4 * m + 2);
            if (n_pts > d.track_th) kind = n_as > d.track_th ? 1 : (n_pts - n_as > d.track_th ? 2 : 0);
        }
        s_scan[t] = kind == 2;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                         // inclusive scan of the "new instance" flags, mask order
            const int v = t >= o ? s_scan[t - o] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int before = s_scan[t] - (kind == 2);
        if (m < d.n_masks) {
            const int target = kind == 1 ? mode : (kind == 2 ? s_base + before : -1);
            int32_t *r = d.res + 8 + 6 * m;
            r[0] = ld_agent(stats + 4 * m); r[1] = ld_agent(stats + 4 * m + 1); r[2] = mode; r[3] = ld_agent(stats + 4 * m + 3); r[4] = target; r[5] = -1;
        }
        __syncthreads();
        if (t == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (t == 0) { *d.next_ins = s_base; d.res[4] = s_base; d.res[5] = first_new; d.res[6] = 0; d.res[7] = 0; }
    __threadfence_block();
    __syncthreads();
    for (int m = t; m < d.n_masks; m += 256) {                      // first mask of each target (targets of new instances are unique)
        const int tg = d.res[8 + 6 * m + 4];
        int first = tg < 0 ? -1 : m;
        if (tg >= 0 && tg < first_new)
            for (int k = 0; k < m; ++k)
                if (d.res[8 + 6 * k + 4] == tg) { first = k; break; }
        d.dst[m] = first;
    }
}

// one workgroup, one mask: total, assigned, mode with smallest-id tie break (k_vote_stats), thread 0 writes the row
__device__ void dev_vote_row(int m, const int32_t *hist, int hist_cols, int32_t *stats) {
    __shared__ long long s_cnt[256];
    __shared__ int s_best[256], s_arg[256];
    const int t = threadIdx.x;
    const int32_t *row = hist + (int64_t)m * hist_cols;
    long long assigned = 0;
    int best = 0, arg = -1;
    for (int c = 1 + t; c < hist_cols; c += 256) {
        const int v = ld_agent(row + c);
        assigned += v;
        if (v > best) { best = v; arg = c - 1; }
    }
    s_cnt[t] = assigned; s_best[t] = best; s_arg[t] = arg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            s_cnt[t] += s_cnt[t + o];
            const int b2 = s_best[t + o], a2 = s_arg[t + o];
            if (b2 > s_best[t] || (b2 == s_best[t] && b2 > 0 && (s_arg[t] < 0 || a2 < s_arg[t]))) {
                s_best[t] = b2; s_arg[t] = a2;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        stats[4 * m + 0] = (int)(s_cnt[0] + ld_agent(row));
        stats[4 * m + 1] = (int)s_cnt[0];
        stats[4 * m + 2] = s_arg[0];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_vote_decide(const int32_t *__restrict__ hist, int hist_cols, int32_t *__restrict__ stats,
                                                     unsigned int *__restrict__ ticket, Decide d) {
    __shared__ int s_last;
    dev_vote_row(blockIdx.x, hist, hist_cols, stats);
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    decide_masks(d, (const int32_t *)stats);                        // the last workgroup: every mask's statistics are in
}

// ---- a7 on the device (ovo.py:284-309): masks[first] |= masks[m] for every other mask m of the same instance, the fused area of
// `first` for the top-k view heap; then the LAST workgroup publishes the keyframe's result block to pinned host memory.
struct Publish {
    const int32_t *res;          // device result block
    volatile int32_t *host;      // pinned host copy (same layout), host[0] = seq is written last
    const unsigned long long *counters;   // {in frustum, matched} x cnt_slots slots, CNT_STRIDE words apart
    int cnt_slots;
    const long long *n_dev; long long n_host;
    unsigned int *ticket;
    int32_t seq, n_ints;
};

__device__ __forceinline__ int nonzero_bytes(unsigned int w) {
    w |= w >> 4; w |= w >> 2; w |= w >> 1;
    return __popc(w & 0x01010101u);
}

// masks[d] |= its followers, for the pixel chunk (part of parts) of row d; the row's fused area accumulates in res[d][5] (from -1)
__device__ void dev_fuse_row(int d, int part, int parts, uint4 *masks, long long px16, int n_masks, const int32_t *dst, int32_t *res) {
    __shared__ int s_follow[256], s_nf;
    if (ld_agent(dst + d) != d) return;                            // workgroup-uniform
    __syncthreads();
    // the row's followers in ascending order: wave 0 tests 64 masks per step -- their `dst` loads are in flight together -- and a ballot keeps
    // the order (one thread walking the row with dependent agent-scope loads cost ~1 us per mask: 37 us per keyframe with 32 masks)
    if (threadIdx.x < 64) {
        int c = 0;
        for (int base = d + 1; base < n_masks; base += 64) {
            const int m = base + (int)threadIdx.x;
            const bool f = m < n_masks && ld_agent(dst + m) == d;
            const unsigned long long bal = __ballot(f);
            if (f) {
                const int pos = c + __popcll(bal & ((1ull << threadIdx.x) - 1ull));
                if (pos < 256) s_follow[pos] = m;
            }
            c += __popcll(bal);
        }
        if (threadIdx.x == 0) s_nf = c;
    }
    __syncthreads();
    const int n_follow = s_nf;
    if (!n_follow) return;
    int area = 0;
    for (long long i = (long long)part * blockDim.x + threadIdx.x; i < px16; i += (long long)parts * blockDim.x) {
        uint4 a = masks[d * px16 + i];
        if (n_follow <= 256) {
            for (int k = 0; k < n_follow; ++k) {
                const uint4 b = masks[s_follow[k] * px16 + i];
                a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
            }
        } else {
            for (int m = d + 1; m < n_masks; ++m)
                if (ld_agent(dst + m) == d) {
                    const uint4 b = masks[m * px16 + i];
                    a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
                }
        }
        masks[d * px16 + i] = a;
        area += nonzero_bytes(a.x) + nonzero_bytes(a.y) + nonzero_bytes(a.z) + nonzero_bytes(a.w);
    }
    for (int o = 32; o > 0; o >>= 1) area += __shfl_xor(area, o, 64);
    if ((threadIdx.x & 63) == 0) {
        if (part == 0 && threadIdx.x == 0) area += 1;              // the row starts at -1 ("not fused")
        if (area) atomicAdd(res + 8 + 6 * d + 5, area);
    }
}

// the keyframe's result block -> pinned host memory, sequence word last (one workgroup)
__device__ void dev_publish(int32_t *res, const Publish &pb, long long n_points) {
    if (threadIdx.x < 64) {                                         // the counter slots, summed by the first wave
        long long c0 = 0, c1 = 0;
        if ((int)threadIdx.x < pb.cnt_slots) {
            c0 = (long long)__hip_atomic_load(pb.counters + (size_t)threadIdx.x * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c1 = (long long)__hip_atomic_load(pb.counters + (size_t)threadIdx.x * CNT_STRIDE + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
        if (threadIdx.x == 0) { res[1] = (int32_t)n_points; res[2] = (int32_t)c0; res[3] = (int32_t)c1; }
    }
    __syncthreads();
    if (!pb.host) return;
    for (int i = 1 + threadIdx.x; i < pb.n_ints; i += 256) pb.host[i] = ld_agent(res + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) pb.host[0] = pb.seq;
}

__global__ void __launch_bounds__(256) k_fuse_publish(uint4 *__restrict__ masks, long long px16, int n_masks, const int32_t *__restrict__ dst,
                                                      int32_t *__restrict__ res, Publish pb) {
    const int d = blockIdx.y;
    if (masks) dev_fuse_row(d, blockIdx.x, gridDim.x, masks, px16, n_masks, dst, res);
    // two-level ticket (per mask row, then one per row): a thousand workgroups on ONE address serialise in the L2 atomic unit
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = 0;
        if (atomicAdd(pb.ticket + 1 + d, 1u) == gridDim.x - 1) s_last = atomicAdd(pb.ticket, 1u) == gridDim.y - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    dev_publish(res, pb, pb.n_host >= 0 ? pb.n_host : *pb.n_dev);
}

// =================================================================================================
// ovo_keyframe_step: the two halves of a keyframe's chain with their independent passes MERGED into shared launches -- 7 launches instead
// of 13.  On a GPU whose CUs are filled by the encoders' GEMM workgroups every dependent launch of the chain waits for the dispatcher
// (~45 us each, profiles/r03_round_emulation.txt); the passes themselves are the device functions above, so results do not change.
//   k_kf_zero      explained[h w] and the tracking scratch (votes, statistics, counters, tickets)
//   k_kf_phase1    workgroup ranges: explained-pixel pass over the map | depth high-pass filter | mask areas of the seg map
//   k_backproj_flag
//   k_kf_emit      every workgroup rebuilds the (<= 2048-word) scan in LDS, emits its words; the last one commits the map state
//   k_track_project
//   k_vote_decide
//   k_kf_finish    workgroup ranges: in-place assignment | mask fusion; the last one publishes the result block
constexpr int CHAIN_MAX_MASKS_V = 1024;
struct KfZero { uint4 *a; long long a16; uint4 *b; long long b16; int32_t *c; };
__global__ void __launch_bounds__(256) k_kf_zero(KfZero z) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (z.c && i0 == 0) *z.c = 0;                                  // the keyframe's hit counter
    for (long long i = i0; i < z.a16; i += step) z.a[i] = make_uint4(0, 0, 0, 0);
    for (long long i = i0; i < z.b16; i += step) z.b[i] = make_uint4(0, 0, 0, 0);
}

struct KfPhase1 {
    int g_expl, g_filt, g_area;
    const float *xyz; long long n_host; const long long *state; ovo_camera_t cam_map; const float *depth_map; uint8_t *explained;
    const float *depth_t; int fh, fw; float filter_th; float *depth_f;
    const int32_t *seg_map; long long seg_pixels; int n_masks; int32_t *stats;
};
__global__ void __launch_bounds__(256) k_kf_phase1(KfPhase1 a, BlurTaps taps) {
    __shared__ int s_area[CHAIN_MAX_MASKS_V];
    int bid = blockIdx.x;
    if (bid < a.g_expl) {
        long long n = a.n_host;
        if (n < 0) { n = a.state[0]; if (a.state[1] <= 0) return; }
        dev_map_explained(Blk{bid, a.g_expl}, a.xyz, n, a.cam_map, a.depth_map, a.explained);
        return;
    }
    bid -= a.g_expl;
    if (bid < a.g_filt) {
        const int n = a.fh * a.fw;
        for (int i = bid * 256 + threadIdx.x; i < n; i += a.g_filt * 256) a.depth_f[i] = depth_filter_pixel(a.depth_t, a.fh, a.fw, taps, a.filter_th, i % a.fw, i / a.fw);
        return;
    }
    bid -= a.g_filt;
    dev_seg_area(Blk{bid, a.g_area}, a.seg_map, a.seg_pixels, a.n_masks, a.stats, s_area);
}

__global__ void __launch_bounds__(256) k_kf_emit(const float *__restrict__ depth, const uint8_t *__restrict__ rgb, BackprojArgs a, int64_t n_sub,
                                                 const unsigned long long *words, float *xyz, int32_t *ids, int32_t *ins, uint8_t *out_rgb, MapCommit mc) {
    __shared__ int s_offs[SCAN_CHUNK + 1];
    __shared__ long long sh[4];
    const int n_words = (int)((n_sub + 63) >> 6);
    // the whole scan, in every workgroup: <= 2048 words (16 KB from L2) cost less than a launch boundary
    const int w0 = threadIdx.x * 8;
    int pc[8];
    long long c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { pc[k] = w0 + k < n_words ? __popcll(words[w0 + k]) : 0; c += pc[k]; }
    long long total;
    long long run = block_exclusive_scan(c, sh, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (w0 + k < n_words) s_offs[w0 + k] = (int)run; run += pc[k]; }
    __syncthreads();
    const long long base = mc.n_host >= 0 ? mc.n_host : mc.state[0];
    const int32_t first_id = (int32_t)(mc.n_host >= 0 ? mc.id_host : mc.state[1]);
    const int lane = threadIdx.x & 63;
    for (int wd = blockIdx.x * 4 + (threadIdx.x >> 6); wd < n_words; wd += gridDim.x * 4) {
        const unsigned long long m = words[wd];
        if ((m >> lane) & 1ull) {
            const long long pos = s_offs[wd] + __popcll(m & ((1ull << lane) - 1ull));
            const int64_t k = (int64_t)wd * 64 + lane;
            const int y = (int)(k / a.ws_w) * a.ds, x = (int)(k % a.ws_w) * a.ds;
            const int64_t px = (int64_t)y * a.w + x;
            const float d = depth[px];
            const float x3 = __fdiv_rn(__fmul_rn(__fsub_rn((float)x, a.K[2]), d), a.K[0]);
            const float y3 = __fdiv_rn(__fmul_rn(__fsub_rn((float)y, a.K[5]), d), a.K[4]);
            const int64_t r = base + pos;
            if (r < mc.cap) {
                xyz[3 * r + 0] = dot4(a.c2w, x3, y3, d, 1.0f);
                xyz[3 * r + 1] = dot4(a.c2w + 4, x3, y3, d, 1.0f);
                xyz[3 * r + 2] = dot4(a.c2w + 8, x3, y3, d, 1.0f);
                ids[r] = first_id + (int32_t)pos;
                ins[r] = -1;
                if (rgb) { out_rgb[3 * r + 0] = rgb[3 * px + 0]; out_rgb[3 * r + 1] = rgb[3 * px + 1]; out_rgb[3 * r + 2] = rgb[3 * px + 2]; }
            }
        }
    }
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd((unsigned long long *)(mc.state + 3), 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    long long m = total;
    if (base + m > mc.cap) { m = mc.cap - base; mc.state[2] |= 1; }
    mc.state[0] = base + m;
    mc.state[1] = (long long)first_id + m;
    mc.state[3] = 0;
    if (mc.result) {
        mc.result[1] = m; mc.result[2] = base + m; mc.result[3] = (long long)first_id + m;
        __threadfence_system();
        mc.result[0] = mc.seq;
    }
}

struct KfFinish { int g_assign, gx; int32_t *ins; const int16_t *point_seg; long long n_host; const long long *n_dev; uint4 *masks; long long px16; int n_masks;
                  const int32_t *dst; int32_t *res; };
__global__ void __launch_bounds__(256) k_kf_finish(KfFinish a, Publish pb) {
    int bid = blockIdx.x;
    if (bid < a.g_assign) {
        const long long n = a.n_host >= 0 ? a.n_host : *a.n_dev;
        dev_assign_res(Blk{bid, a.g_assign}, a.ins, a.point_seg, n, a.res, a.n_masks);
    } else if (a.masks) {
        bid -= a.g_assign;
        dev_fuse_row(bid / a.gx, bid % a.gx, a.gx, a.masks, a.px16, a.n_masks, a.dst, a.res);
    }
    // two-level ticket, as k_fuse_publish: workgroup b reports to slot 1 + b % n_masks, the last of a slot to the kernel's ticket -- a couple of
    // thousand workgroups on ONE address serialise in the L2 atomic unit (~3 ns each, behind a fence each)
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned slots = a.n_masks > 0 ? (unsigned)a.n_masks : 1u, slot = blockIdx.x % slots;
        const unsigned in_slot = (gridDim.x - 1 - slot) / slots + 1;
        s_last = 0;
        if (atomicAdd(pb.ticket + 1 + slot, 1u) == in_slot - 1) {
            __threadfence();
            const unsigned used = gridDim.x < slots ? gridDim.x : slots;
            s_last = atomicAdd(pb.ticket, 1u) == used - 1;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    dev_publish(a.res, pb, pb.n_host >= 0 ? pb.n_host : *pb.n_dev);
}

#ifdef OVO_EXPERIMENTAL   // python -m ovo_amd.build --experimental: measured slower than the per-pass launches, kept for the record
// =================================================================================================
// k_round_chain: the map + tracking chains of a whole ROUND of keyframes in ONE launch.  A few dozen persistent workgroups walk
// through every pass of every keyframe, separated by grid-wide barriers (an atomic arrival counter, spin on an agent-scope load).
// Why: as separate launches a chain is ~12 small DEPENDENT kernels, and on a GPU whose CUs are filled by the encoders' GEMM workgroups
// every one of them waits for the dispatcher to get round to its queue (~45 us each: 0.6 ms per keyframe instead of 0.14 on an idle
// GPU, the serial term of an 8-GPU round).  Here the workgroups wait for CUs once per round and then keep them.
// The passes are the device functions the one-pass kernels above are made of (same arithmetic, same results).
struct ChainKf {
    // map update (vanilla_mapper.py:46-85)
    float *xyz; int32_t *ids, *ins; uint8_t *rgb_out; long long cap; long long *state; long long n_host, id_host;
    const float *depth; const uint8_t *rgb; ovo_camera_t cam_map; BackprojArgs bp; long long n_sub;
    uint8_t *explained; unsigned long long *words; long long *offs; long long *total;
    volatile long long *map_result; long long map_seq;
    int do_map, erode;
    // tracking (ovo.py:182-324)
    int do_track, filter; const float *depth_t; float *depth_f; float filter_th;
    ovo_camera_t cam; ovo_ratio_t ratio; const int32_t *seg_map; int seg_h, seg_w;
    uint4 *masks; int n_masks; long long px16; int16_t *point_seg;
    int32_t *hist, *stats; unsigned long long *counters; int32_t *dst, *res; long long zero_bytes; int hist_cols, track_th;
    int32_t *next_ins; int next_host; volatile int32_t *result; int seq;
};
constexpr int CHAIN_MAX_KF = 16, CHAIN_MAX_MASKS = 1024, CHAIN_BARRIERS = 10;

__device__ __forceinline__ void dev_depth_filter(Blk b, const float *__restrict__ depth, int h, int w, const BlurTaps &taps, float th, float *out) {
    const int n = h * w;
    for (int i = b.bid * 256 + threadIdx.x; i < n; i += b.nblk * 256) out[i] = depth_filter_pixel(depth, h, w, taps, th, i % w, i / w);
}

struct GridBar { unsigned long long *count; unsigned long long base; unsigned int nblk; unsigned int *abort; };

__device__ __forceinline__ void grid_sync(const GridBar &g, unsigned long long &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(g.count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = g.base + gen * g.nblk;
        unsigned long long spins = 0;
        while (__hip_atomic_load(g.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1ull << 26)) { *g.abort = 1; break; }   // ~ seconds: a workgroup never arrived; leave instead of hanging the GPU
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // drop this CU's stale L1 lines of what the others wrote
}

__global__ void __launch_bounds__(256) k_round_chain(const ChainKf *__restrict__ params, int n_kf, GridBar gb, BlurTaps taps) {
    __shared__ ChainKf kf;
    __shared__ int s_area[CHAIN_MAX_MASKS];
    const Blk b{(int)blockIdx.x, (int)gridDim.x};
    unsigned long long gen = 0;
    for (int k = 0; k < n_kf; ++k) {
        if (__hip_atomic_load(gb.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // a barrier timed out: leave, the host sees the flag
        __syncthreads();
        for (int i = threadIdx.x; i < (int)(sizeof(ChainKf) / 4); i += blockDim.x) ((int32_t *)&kf)[i] = ((const int32_t *)(params + k))[i];
        __syncthreads();
        // ---- pass A: zero the scratch, high-pass filter of the depth (geometry_utils.py:92-96)
        const long long n_old = kf.n_host >= 0 ? kf.n_host : __hip_atomic_load(kf.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long id_old = kf.n_host >= 0 ? kf.id_host : __hip_atomic_load(kf.state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool nonempty = id_old > 0;
        const int hw = kf.cam_map.h * kf.cam_map.w;
        if (kf.do_map) {
            for (int i = b.bid * 256 + threadIdx.x; i < hw / 16; i += b.nblk * 256) ((uint4 *)kf.explained)[i] = make_uint4(0, 0, 0, 0);
            if (b.bid == 0 && threadIdx.x < hw % 16) kf.explained[hw - 1 - threadIdx.x] = 0;
        }
        if (kf.do_track) {
            for (long long i = b.bid * 256 + threadIdx.x; i < kf.zero_bytes / 4; i += b.nblk * 256) kf.hist[i] = 0;
            if (kf.filter) dev_depth_filter(b, kf.depth_t, kf.cam.h, kf.cam.w, taps, kf.filter_th, kf.depth_f);
        }
        grid_sync(gb, gen);                                                                                       // 1
        // ---- pass B: explained pixels (vanilla_mapper.py:56-61); mask areas of the seg map
        if (kf.do_map && nonempty) dev_map_explained(b, kf.xyz, n_old, kf.cam_map, kf.depth, kf.explained);
        if (kf.do_track) dev_seg_area(b, kf.seg_map, (int64_t)kf.seg_h * kf.seg_w, kf.n_masks, kf.stats, s_area);
        grid_sync(gb, gen);                                                                                       // 2
        // ---- pass C..E: erode + subsample -> ordered append (vanilla_mapper.py:62-85)
        long long m_new = 0;
        if (kf.do_map) {
            BackprojArgs bp = kf.bp;
            bp.erode = kf.erode && nonempty;
            dev_backproj_flag(b, kf.depth, nonempty ? kf.explained : nullptr, bp, kf.n_sub, kf.words);
        }
        grid_sync(gb, gen);                                                                                       // 3
        if (kf.do_map && b.bid == 0) dev_scan_one(kf.words, kf.offs, (kf.n_sub + 63) >> 6, kf.total);
        grid_sync(gb, gen);                                                                                       // 4
        if (kf.do_map) {
            m_new = __hip_atomic_load(kf.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_old + m_new > kf.cap) m_new = kf.cap - n_old;
            dev_backproj_emit(b, kf.depth, kf.rgb, kf.bp, kf.n_sub, kf.words, kf.offs, n_old, (int32_t)id_old, kf.cap, kf.xyz, kf.ids, kf.ins, kf.rgb_out);
        }
        const long long n_now = n_old + m_new;
        grid_sync(gb, gen);                                                                                       // 5
        // ---- pass F: cull / project / depth-test / seg lookup / votes over the whole map (ovo.py:208-222)
        if (kf.do_track && n_now > 0)
            dev_track_project(b, kf.xyz, kf.ins, n_now, kf.cam, kf.filter ? kf.depth_f : kf.depth_t, kf.seg_map, kf.seg_h, kf.seg_w, kf.ratio,
                              kf.point_seg, kf.hist, kf.n_masks, kf.hist_cols, kf.counters, CNT_SLOTS, HitSink{nullptr, nullptr, 0, 1, 0});
        grid_sync(gb, gen);                                                                                       // 6
        // ---- pass G: per-mask vote statistics, then the decisions in mask order (ovo.py:255-282)
        if (kf.do_track)
            for (int m = b.bid; m < kf.n_masks; m += b.nblk) dev_vote_row(m, kf.hist, kf.hist_cols, kf.stats);
        grid_sync(gb, gen);                                                                                       // 7
        if (kf.do_track && b.bid == 0) {
            Decide d;
            d.res = kf.res; d.dst = kf.dst; d.next_ins = kf.next_ins; d.next_host = kf.next_host; d.track_th = kf.track_th; d.n_masks = kf.n_masks;
            decide_masks(d, kf.stats);
        }
        grid_sync(gb, gen);                                                                                       // 8
        // ---- pass H: assignment in place (ovo.py:228-229,280); masks of one instance OR-ed into its first mask (ovo.py:284-309)
        if (kf.do_track) {
            if (n_now > 0) dev_assign_res(b, kf.ins, kf.point_seg, n_now, kf.res, kf.n_masks);
            if (kf.masks)
                for (int d = 0; d < kf.n_masks; ++d) dev_fuse_row(d, b.bid, b.nblk, kf.masks, kf.px16, kf.n_masks, kf.dst, kf.res);
        }
        grid_sync(gb, gen);                                                                                       // 9
        // ---- pass I: commit the map state, publish both result blocks
        if (b.bid == 0) {
            if (kf.do_map && threadIdx.x == 0) {
                if (n_old + __hip_atomic_load(kf.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > kf.cap) kf.state[2] |= 1;
                kf.state[0] = n_now;
                kf.state[1] = id_old + m_new;
                if (kf.map_result) {
                    kf.map_result[1] = m_new; kf.map_result[2] = n_now; kf.map_result[3] = id_old + m_new;
                    __threadfence_system();
                    kf.map_result[0] = kf.map_seq;
                }
            }
            if (kf.do_track) {
                Publish pb;
                pb.res = kf.res; pb.host = kf.result; pb.counters = kf.counters; pb.cnt_slots = CNT_SLOTS; pb.n_dev = nullptr; pb.n_host = n_now; pb.ticket = nullptr;
                pb.seq = kf.seq; pb.n_ints = 8 + 6 * kf.n_masks;
                if (threadIdx.x == 0) kf.res[6] = (int32_t)__hip_atomic_load(gb.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dev_publish(kf.res, pb, n_now);
            }
        }
        grid_sync(gb, gen);                                                                                       // 10: the next keyframe reads the state
    }
}
#endif  // OVO_EXPERIMENTAL

}  // namespace

// =================================================================================================
extern "C" {

size_t ovo_compact_workspace_bytes(int64_t n) {
    const int64_t w = (n + 63) / 64 + 1;
    return (size_t)w * 16 + (size_t)(w / 2048 + 2) * 8;           // words + offsets + one total per 2048-word scan chunk
}

int ovo_frustum_ids(const float *pts, int64_t n, const ovo_camera_t *cam, int64_t *out_idx, int64_t *out_count,
                    void *ws, size_t ws_bytes, ovo_stream_t stream) {
    OVO_REQUIRE(cam && out_count && n >= 0, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { OVO_HIP(hipMemsetAsync(out_count, 0, 8, s)); return OVO_OK; }
    OVO_REQUIRE(pts && out_idx && ws && ws_bytes >= ovo_compact_workspace_bytes(n), "workspace too small");
    CompactWs c = carve(ws, n);
    const int g = ovo_grid(n, 256);
    k_frustum_flag<<<g, 256, 0, s>>>(pts, n, *cam, c.words);
    scan_words(c.words, c.offs, c.sums, c.n_words, (long long *)out_count, s);
    k_frustum_emit<<<g, 256, 0, s>>>(n, c.words, c.offs, out_idx);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_project_points(const float *pts, int64_t n, int stride, const ovo_camera_t *cam, int32_t *out_uv,
                       ovo_stream_t stream) {
    OVO_REQUIRE(cam && n >= 0 && (stride == 3 || stride == 4), "bad argument");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(pts && out_uv, "null pointer");
    k_project<<<ovo_grid(n, 256), 256, 0, (hipStream_t)stream>>>(pts, n, stride, *cam, out_uv);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_match_points(const float *pts, int64_t n, int stride, const ovo_camera_t *cam, const float *depth,
                     int64_t *out_idx, int32_t *out_uv, int64_t *out_count, void *ws, size_t ws_bytes,
                     ovo_stream_t stream) {
    OVO_REQUIRE(cam && out_count && n >= 0 && (stride == 3 || stride == 4), "bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { OVO_HIP(hipMemsetAsync(out_count, 0, 8, s)); return OVO_OK; }
    OVO_REQUIRE(pts && depth && out_idx && out_uv && ws && ws_bytes >= ovo_compact_workspace_bytes(n), "null / workspace");
    CompactWs c = carve(ws, n);
    const int g = ovo_grid(n, 256);
    k_match_flag<<<g, 256, 0, s>>>(pts, n, stride, *cam, depth, c.words);
    scan_words(c.words, c.offs, c.sums, c.n_words, (long long *)out_count, s);
    k_match_emit<<<g, 256, 0, s>>>(pts, n, stride, *cam, c.words, c.offs, out_idx, out_uv);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_track_project(const float *pts, const int32_t *point_ins, int64_t n, const ovo_camera_t *cam,
                      const float *depth, const int32_t *seg_map, int seg_h, int seg_w, ovo_ratio_t ratio,
                      int16_t *point_seg, int32_t *hist, int n_masks, int hist_cols, int64_t *counters,
                      ovo_stream_t stream) {
    OVO_REQUIRE(cam && depth && seg_map && hist && counters && n >= 0, "null argument");
    OVO_REQUIRE(n_masks > 0 && n_masks < 32767 && hist_cols >= 1, "bad mask / histogram shape");
    hipStream_t s = (hipStream_t)stream;
    OVO_HIP(hipMemsetAsync(hist, 0, (size_t)n_masks * hist_cols * sizeof(int32_t), s));
    OVO_HIP(hipMemsetAsync(counters, 0, 16, s));
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(pts && point_ins && point_seg, "null pointer");
    const bool prof = ovo_prof_enabled();
    if (prof) ovo_prof_begin(2, 14.0 * (double)n, s);          // 12 B xyz read + 2 B mask id written per map point
    k_track_project<<<ovo_grid(n, 256, TRACK_GRID_CAP), 256, 0, s>>>(pts, point_ins, n, *cam, depth, seg_map, seg_h, seg_w, ratio,
                                                      point_seg, hist, n_masks, hist_cols,
                                                      (unsigned long long *)counters, 1, nullptr, HitSink{nullptr, nullptr, 0, 1, 0});
    if (prof) ovo_prof_end(s);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_vote_stats(const int32_t *hist, int n_masks, int hist_cols, const int32_t *seg_map, int64_t seg_pixels,
                   int32_t *stats, ovo_stream_t stream) {
    OVO_REQUIRE(hist && seg_map && stats && n_masks > 0 && hist_cols >= 1, "bad argument");
    OVO_REQUIRE(n_masks <= 8192, "more than 8192 masks");
    hipStream_t s = (hipStream_t)stream;
    OVO_HIP(hipMemsetAsync(stats, 0, (size_t)n_masks * 4 * sizeof(int32_t), s));
    k_seg_area<<<ovo_grid(seg_pixels, 256, 256), 256, n_masks * sizeof(int), s>>>(seg_map, seg_pixels, n_masks, stats);
    k_vote_stats<<<n_masks, 256, 0, s>>>(hist, hist_cols, stats);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_assign_instances(const int32_t *point_ins, const int16_t *point_seg, int64_t n, const int32_t *mask_target,
                         int n_masks, int32_t *out_ins, int64_t *new_count, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && mask_target && n_masks > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (new_count) OVO_HIP(hipMemsetAsync(new_count, 0, 8, s));
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(point_ins && point_seg && out_ins, "null pointer");
    k_assign<<<ovo_grid(n, 256), 256, 0, s>>>(point_ins, point_seg, n, mask_target, n_masks, out_ins,
                                               (unsigned long long *)new_count, nullptr);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_map_explained(const float *pts, int64_t n, const ovo_camera_t *cam, const float *depth, uint8_t *explained,
                      ovo_stream_t stream) {
    OVO_REQUIRE(cam && depth && explained && n >= 0, "null argument");
    hipStream_t s = (hipStream_t)stream;
    OVO_HIP(hipMemsetAsync(explained, 0, (size_t)cam->h * cam->w, s));
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(pts, "null pointer");
    k_map_explained<<<ovo_grid(n, 256), 256, 0, s>>>(pts, n, *cam, depth, explained, nullptr);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_map_backproject(const float *depth, const uint8_t *rgb, const uint8_t *explained, int h, int w, int erode,
                        int ds, const float *K9_host, const float *c2w16_host, int64_t base, int32_t first_id,
                        float *xyz, int32_t *ids, int32_t *ins, uint8_t *out_rgb, int64_t *out_count, void *ws,
                        size_t ws_bytes, ovo_stream_t stream) {
    OVO_REQUIRE(depth && K9_host && c2w16_host && xyz && ids && ins && out_count && ws, "null argument");
    OVO_REQUIRE(h > 0 && w > 0 && ds >= 1 && base >= 0, "bad shape");
    BackprojArgs a;
    for (int i = 0; i < 9; ++i) a.K[i] = K9_host[i];
    for (int i = 0; i < 16; ++i) a.c2w[i] = c2w16_host[i];
    a.h = h; a.w = w; a.ds = ds; a.erode = erode;
    a.ws_w = (w + ds - 1) / ds;
    const int64_t n_sub = (int64_t)((h + ds - 1) / ds) * a.ws_w;
    OVO_REQUIRE(ws_bytes >= ovo_compact_workspace_bytes(n_sub), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    CompactWs c = carve(ws, n_sub);
    const int g = ovo_grid(n_sub, 256);
    k_backproj_flag<<<g, 256, 0, s>>>(depth, explained, a, n_sub, c.words, nullptr);
    scan_words(c.words, c.offs, c.sums, c.n_words, (long long *)out_count, s);
    MapCommit none = {};
    k_backproj_emit<<<g, 256, 0, s>>>(depth, rgb, a, n_sub, c.words, c.offs, base, first_id, xyz, ids, ins, out_rgb, none);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}


// ---- the keyframe chain without host round trips (ovo_map_step, ovo_track_step) ------------------------------------------------
size_t ovo_track_workspace_bytes(int n_masks, int hist_cols) {
    // hist | stats[4 n] | counters (CNT_SLOTS x 128 bytes) | tickets (2 + n x u32) | dst[n] | result block [8 + 6 n]
    return ((size_t)n_masks * hist_cols + 4 * (size_t)n_masks + 2 * CNT_SLOTS * CNT_STRIDE + 2 + n_masks + n_masks + 8 + 6 * (size_t)n_masks + 4) * sizeof(int32_t);
}

int ovo_map_step(const ovo_map_step_t *a, ovo_stream_t stream) {
    OVO_REQUIRE(a && a->depth && a->map.xyz && a->map.ids && a->map.ins && a->map.state && a->explained && a->ws, "null argument");
    OVO_REQUIRE(a->h > 0 && a->w > 0 && a->ds >= 1 && a->n_upper >= 0, "bad shape");
    const int64_t ws_w = (a->w + a->ds - 1) / a->ds;
    const int64_t n_sub = (int64_t)((a->h + a->ds - 1) / a->ds) * ws_w;
    OVO_REQUIRE(a->ws_bytes >= ovo_compact_workspace_bytes(n_sub) + 8, "workspace too small");
    OVO_REQUIRE(a->map.cap >= a->n_upper + n_sub, "map capacity below n_upper + one frame of points");
    OVO_REQUIRE(a->map.n < 0 || a->map.n <= a->n_upper, "n_upper below the known point count");
    hipStream_t s = (hipStream_t)stream;
    const bool known = a->map.n >= 0;
    const bool maybe_nonempty = known ? a->map.next_id > 0 : true;
    OVO_HIP(hipMemsetAsync(a->explained, 0, (size_t)a->h * a->w, s));
    if (maybe_nonempty && a->n_upper > 0) {
        if (known)
            k_map_explained<<<ovo_grid(a->map.n, 256), 256, 0, s>>>(a->map.xyz, a->map.n, a->cam, a->depth, a->explained, nullptr);
        else
            k_map_explained<<<ovo_grid(a->n_upper, 256), 256, 0, s>>>(a->map.xyz, 0, a->cam, a->depth, a->explained,
                                                                      (const long long *)a->map.state);
    }
    BackprojArgs b;
    for (int i = 0; i < 9; ++i) b.K[i] = a->K[i];
    for (int i = 0; i < 16; ++i) b.c2w[i] = a->c2w[i];
    b.h = a->h; b.w = a->w; b.ds = a->ds; b.ws_w = (int)ws_w;
    b.erode = a->erode && maybe_nonempty;
    long long *total = (long long *)a->ws;
    CompactWs c = carve((char *)a->ws + 8, n_sub);
    const int g = ovo_grid(n_sub, 256);
    k_backproj_flag<<<g, 256, 0, s>>>(a->depth, maybe_nonempty ? a->explained : nullptr, b, n_sub, c.words,
                                      known ? nullptr : (const long long *)a->map.state);
    scan_words(c.words, c.offs, c.sums, c.n_words, total, s);
    MapCommit mc;
    mc.state = (long long *)a->map.state; mc.total = total; mc.result = (volatile long long *)a->result_host; mc.seq = a->seq;
    mc.cap = a->map.cap; mc.n_host = known ? a->map.n : -1; mc.id_host = a->map.next_id;
    k_backproj_emit<<<g, 256, 0, s>>>(a->depth, a->rgb, b, n_sub, c.words, c.offs, 0, 0, a->map.xyz, a->map.ids, a->map.ins,
                                      a->map.rgb, mc);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// the hit list of a tracking step (ovo_track_step_t.hits; ABI 11), validated
static int make_sink(const ovo_track_step_t *t, HitSink &sink) {
    sink = HitSink{nullptr, nullptr, 0, 1, 0};
    if (!t->hits) return OVO_OK;
    OVO_REQUIRE(t->n_hits, "hits without n_hits");
    const int count = t->hit_shard_count < 1 ? 1 : t->hit_shard_count;
    OVO_REQUIRE(t->hit_shard_rank >= 0 && t->hit_shard_rank < count, "bad hit shard rank");
    int block_log2 = 0;
    if (count > 1) {
        OVO_REQUIRE(t->hit_shard_block > 0 && (t->hit_shard_block & (t->hit_shard_block - 1)) == 0, "hit_shard_block must be a power of two");
        while ((1 << block_log2) < t->hit_shard_block) ++block_log2;
    }
    sink = HitSink{t->hits, t->n_hits, t->hit_shard_rank, count, block_log2};
    return OVO_OK;
}

int ovo_track_step(const ovo_track_step_t *a, ovo_stream_t stream) {
    OVO_REQUIRE(a && a->depth && a->seg_map && a->point_seg && a->ws && a->map.state && a->next_ins, "null argument");
    OVO_REQUIRE(a->n_upper == 0 || (a->map.xyz && a->map.ins), "null map");
    OVO_REQUIRE(a->n_masks > 0 && a->n_masks <= 8192 && a->hist_cols >= 1, "bad mask / histogram shape");
    OVO_REQUIRE(a->ws_bytes >= ovo_track_workspace_bytes(a->n_masks, a->hist_cols), "workspace too small");
    OVO_REQUIRE(!a->masks || (a->pixels > 0 && a->pixels % 16 == 0 && ((uintptr_t)a->masks & 15) == 0), "masks: pixels must be a multiple of 16");
    OVO_REQUIRE(a->map.n < 0 || a->map.n <= a->n_upper, "n_upper below the known point count");
    hipStream_t s = (hipStream_t)stream;
    const int nm = a->n_masks;
    int32_t *hist = (int32_t *)a->ws;
    int32_t *stats = hist + (size_t)nm * a->hist_cols;
    unsigned long long *counters = (unsigned long long *)(stats + 4 * (size_t)nm + ((((size_t)nm * a->hist_cols) & 1) ? 1 : 0));   // 8-byte aligned
    unsigned int *tickets = (unsigned int *)(counters + CNT_SLOTS * CNT_STRIDE);
    int32_t *dst = (int32_t *)(tickets + 2 + nm);
    int32_t *res = dst + nm;
    const size_t zero_bytes = (size_t)((char *)dst - (char *)hist);
    OVO_HIP(hipMemsetAsync(hist, 0, zero_bytes, s));
    HitSink sink;
    { const int rc = make_sink(a, sink); if (rc != OVO_OK) return rc; }
    if (sink.hits) OVO_HIP(hipMemsetAsync(sink.n_hits, 0, sizeof(int32_t), s));
    const bool known = a->map.n >= 0;
    const long long *n_dev = known ? nullptr : (const long long *)a->map.state;
    const float *depth = a->depth;
    if (a->filter_depth) {
        OVO_REQUIRE(a->depth_scratch, "depth_scratch needed for the depth filter");
        const int rc = ovo_depth_filter(a->depth, a->cam.h, a->cam.w, 7, 2.5f, 0.05f, a->depth_scratch, stream);
        if (rc != OVO_OK) return rc;
        depth = a->depth_scratch;
    }
    const int64_t n_grid = known ? a->map.n : a->n_upper;
    if (n_grid > 0) {
        const bool prof = ovo_prof_enabled();
        if (prof) ovo_prof_begin(2, 14.0 * (double)n_grid, s);
        k_track_project<<<ovo_grid(n_grid, 256, TRACK_GRID_CAP), 256, 0, s>>>(a->map.xyz, a->map.ins, known ? a->map.n : 0, a->cam, depth, a->seg_map, a->seg_h,
                                                           a->seg_w, a->ratio, a->point_seg, hist, nm, a->hist_cols, counters, CNT_SLOTS, n_dev, sink);
        if (prof) ovo_prof_end(s);
    }
    const int64_t seg_pixels = (int64_t)a->seg_h * a->seg_w;
    k_seg_area<<<ovo_grid(seg_pixels, 256, 256), 256, nm * sizeof(int), s>>>(a->seg_map, seg_pixels, nm, stats);
    Decide d;
    d.res = res; d.dst = dst; d.next_ins = a->next_ins; d.next_host = a->next_ins_host; d.track_th = a->track_th; d.n_masks = nm;
    k_vote_decide<<<nm, 256, 0, s>>>(hist, a->hist_cols, stats, tickets, d);
    if (n_grid > 0)
        k_assign_res<<<ovo_grid(n_grid, 256, 256), 256, 0, s>>>(a->map.ins, a->point_seg, known ? a->map.n : 0, res, nm, n_dev);
    Publish pb;
    pb.res = res; pb.host = (volatile int32_t *)a->result_host; pb.counters = counters; pb.cnt_slots = CNT_SLOTS; pb.n_dev = (const long long *)a->map.state;
    pb.n_host = known ? a->map.n : -1; pb.ticket = tickets + 1; pb.seq = a->seq; pb.n_ints = 8 + 6 * nm;
    const long long px16 = a->masks ? a->pixels / 16 : 0;
    dim3 grid(a->masks ? ovo_grid(px16, 256, 8) : 1, nm);              // (few, fat workgroups: see k_kf_finish's launch)
    k_fuse_publish<<<grid, 256, 0, s>>>((uint4 *)a->masks, px16, nm, dst, res, pb);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}


// ---- a whole round of keyframes in one launch (k_round_chain) -------------------------------------------------------------------
#ifndef OVO_EXPERIMENTAL
// production builds carry no one-launch form: 0 bytes = "not built" (RoundLauncher then goes keyframe by keyframe: ovo_keyframe_step)
size_t ovo_round_chain_params_bytes(void) { return 0; }
int ovo_round_chain(ovo_round_chain_t *, const ovo_map_step_t *, const ovo_track_step_t *, int, ovo_stream_t) {
    ovo_set_error("ovo_round_chain: not in this build (python -m ovo_amd.build --force --experimental)");
    return OVO_E_UNSUPPORTED;
}
#else
size_t ovo_round_chain_params_bytes(void) { return (size_t)8 * CHAIN_MAX_KF * sizeof(ChainKf); }       // a ring of 8 rounds

int ovo_round_chain(ovo_round_chain_t *ctx, const ovo_map_step_t *maps, const ovo_track_step_t *tracks, int n, ovo_stream_t stream) {
    OVO_REQUIRE(ctx && ctx->params_host && ctx->barrier && maps && tracks && n > 0, "null argument");
    static_assert(sizeof(ChainKf) % 4 == 0, "ChainKf is copied word by word");
    if (n > CHAIN_MAX_KF) { ovo_set_error("ovo_round_chain: more than %d keyframes", CHAIN_MAX_KF); return OVO_E_UNSUPPORTED; }
    ChainKf *slot = (ChainKf *)ctx->params_host + (size_t)(ctx->next_slot % 8) * CHAIN_MAX_KF;
    for (int k = 0; k < n; ++k) {
        const ovo_map_step_t *a = maps + k;
        const ovo_track_step_t *t = tracks + k;
        ChainKf c;
        memset(&c, 0, sizeof(c));
        c.do_map = a->depth != nullptr;
        c.do_track = t->n_masks > 0;
        const ovo_map_ref_t &mr = c.do_map ? a->map : t->map;
        OVO_REQUIRE(c.do_map || c.do_track, "a keyframe with neither map update nor tracking");
        OVO_REQUIRE(mr.xyz && mr.ins && mr.state, "null map");
        c.xyz = mr.xyz; c.ids = mr.ids; c.ins = mr.ins; c.rgb_out = mr.rgb; c.cap = mr.cap; c.state = (long long *)mr.state;
        c.n_host = mr.n; c.id_host = mr.next_id;
        if (c.do_map) {
            OVO_REQUIRE(a->map.ids && a->explained && a->ws && a->h > 0 && a->w > 0 && a->ds >= 1, "bad map step");
            const int64_t ws_w = (a->w + a->ds - 1) / a->ds;
            c.n_sub = (int64_t)((a->h + a->ds - 1) / a->ds) * ws_w;
            if (((c.n_sub + 63) >> 6) > SCAN_CHUNK) { ovo_set_error("ovo_round_chain: frame too large for the one-workgroup scan"); return OVO_E_UNSUPPORTED; }
            OVO_REQUIRE(a->ws_bytes >= ovo_compact_workspace_bytes(c.n_sub) + 8, "workspace too small");
            OVO_REQUIRE(a->map.cap >= a->n_upper + c.n_sub, "map capacity below n_upper + one frame of points");
            c.depth = a->depth; c.rgb = a->rgb; c.cam_map = a->cam;
            for (int i = 0; i < 9; ++i) c.bp.K[i] = a->K[i];
            for (int i = 0; i < 16; ++i) c.bp.c2w[i] = a->c2w[i];
            c.bp.h = a->h; c.bp.w = a->w; c.bp.ds = a->ds; c.bp.ws_w = (int)ws_w; c.bp.erode = a->erode;
            c.erode = a->erode;
            c.explained = a->explained;
            c.total = (long long *)a->ws;
            CompactWs cw = carve((char *)a->ws + 8, c.n_sub);
            c.words = cw.words; c.offs = cw.offs;
            c.map_result = (volatile long long *)a->result_host; c.map_seq = a->seq;
        }
        if (c.do_track) {
            if (t->n_masks > CHAIN_MAX_MASKS) { ovo_set_error("ovo_round_chain: more than %d masks", CHAIN_MAX_MASKS); return OVO_E_UNSUPPORTED; }
            if (t->hits) { ovo_set_error("ovo_round_chain: no hit list in the one-launch form"); return OVO_E_UNSUPPORTED; }
            OVO_REQUIRE(t->depth && t->seg_map && t->point_seg && t->ws && t->next_ins && t->hist_cols >= 1, "bad track step");
            OVO_REQUIRE(t->ws_bytes >= ovo_track_workspace_bytes(t->n_masks, t->hist_cols), "workspace too small");
            OVO_REQUIRE(!t->masks || (t->pixels > 0 && t->pixels % 16 == 0 && ((uintptr_t)t->masks & 15) == 0), "masks: pixels must be a multiple of 16");
            const int nm = t->n_masks;
            if (!c.do_map) c.cam_map = t->cam;
            c.depth_t = t->depth;
            c.filter = t->filter_depth; c.depth_f = t->depth_scratch; c.filter_th = 0.05f;
            OVO_REQUIRE(!c.filter || c.depth_f, "depth_scratch needed for the depth filter");
            c.cam = t->cam; c.ratio = t->ratio; c.seg_map = t->seg_map; c.seg_h = t->seg_h; c.seg_w = t->seg_w;
            c.masks = (uint4 *)t->masks; c.n_masks = nm; c.px16 = t->masks ? t->pixels / 16 : 0; c.point_seg = t->point_seg;
            c.hist = (int32_t *)t->ws;
            c.stats = c.hist + (size_t)nm * t->hist_cols;
            c.counters = (unsigned long long *)(c.stats + 4 * (size_t)nm + ((((size_t)nm * t->hist_cols) & 1) ? 1 : 0));
            unsigned int *tickets = (unsigned int *)(c.counters + CNT_SLOTS * CNT_STRIDE);
            c.dst = (int32_t *)(tickets + 2 + nm);
            c.res = c.dst + nm;
            c.zero_bytes = (long long)((char *)c.dst - (char *)c.hist);
            c.hist_cols = t->hist_cols; c.track_th = t->track_th;
            c.next_ins = t->next_ins; c.next_host = t->next_ins_host;
            c.result = (volatile int32_t *)t->result_host; c.seq = t->seq;
        }
        slot[k] = c;
    }
    const unsigned nblk = ctx->workgroups > 0 ? (unsigned)ctx->workgroups : 64u;
    GridBar gb;
    gb.count = (unsigned long long *)ctx->barrier; gb.abort = (unsigned int *)(ctx->barrier + 1); gb.base = ctx->arrivals; gb.nblk = nblk;
    const BlurTaps taps = make_blur_taps(7, 2.5f);
    k_round_chain<<<nblk, 256, 0, (hipStream_t)stream>>>(slot, n, gb, taps);
    OVO_CHECK_LAUNCH();
    ctx->arrivals += (uint64_t)nblk * CHAIN_BARRIERS * n;
    ctx->next_slot += 1;
    return OVO_OK;
}
#endif  // OVO_EXPERIMENTAL


int ovo_keyframe_step(const ovo_map_step_t *a, const ovo_track_step_t *t, ovo_stream_t stream) {
    OVO_REQUIRE(a && t && a->depth && t->depth && t->n_masks > 0, "both halves are needed (use ovo_map_step / ovo_track_step for one)");
    const int64_t ws_w = (a->w + a->ds - 1) / a->ds;
    const int64_t n_sub = (int64_t)((a->h + a->ds - 1) / a->ds) * ws_w;
    if (((n_sub + 63) >> 6) > SCAN_CHUNK || t->n_masks > CHAIN_MAX_MASKS_V || ((size_t)a->h * a->w) % 16 != 0) {      // shapes the merged launches do not cover
        const int rc = ovo_map_step(a, stream);
        return rc != OVO_OK ? rc : ovo_track_step(t, stream);
    }
    OVO_REQUIRE(a->map.xyz && a->map.ids && a->map.ins && a->map.state && a->explained && a->ws && t->seg_map && t->point_seg && t->ws && t->next_ins, "null argument");
    OVO_REQUIRE(a->h > 0 && a->w > 0 && a->ds >= 1 && a->n_upper >= 0 && t->hist_cols >= 1, "bad shape");
    OVO_REQUIRE(a->ws_bytes >= ovo_compact_workspace_bytes(n_sub) + 8 && t->ws_bytes >= ovo_track_workspace_bytes(t->n_masks, t->hist_cols), "workspace too small");
    OVO_REQUIRE(a->map.cap >= a->n_upper + n_sub, "map capacity below n_upper + one frame of points");
    OVO_REQUIRE(!t->masks || (t->pixels > 0 && t->pixels % 16 == 0 && ((uintptr_t)t->masks & 15) == 0), "masks: pixels must be a multiple of 16");
    OVO_REQUIRE(!t->filter_depth || t->depth_scratch, "depth_scratch needed for the depth filter");
    OVO_REQUIRE(((uintptr_t)a->explained & 15) == 0 && ((uintptr_t)t->ws & 15) == 0, "misaligned scratch");
    hipStream_t s = (hipStream_t)stream;
    const int nm = t->n_masks;
    int32_t *hist = (int32_t *)t->ws;
    int32_t *stats = hist + (size_t)nm * t->hist_cols;
    unsigned long long *counters = (unsigned long long *)(stats + 4 * (size_t)nm + ((((size_t)nm * t->hist_cols) & 1) ? 1 : 0));
    unsigned int *tickets = (unsigned int *)(counters + CNT_SLOTS * CNT_STRIDE);
    int32_t *dst = (int32_t *)(tickets + 2 + nm);
    int32_t *res = dst + nm;
    const size_t zero_bytes = ((size_t)((char *)dst - (char *)hist) + 15) & ~(size_t)15;        // (dst / res are rewritten by the decisions anyway)
    // ---- 1: zero
    KfZero z;
    HitSink sink;
    { const int rc = make_sink(t, sink); if (rc != OVO_OK) return rc; }
    z.a = (uint4 *)a->explained; z.a16 = (long long)((size_t)a->h * a->w / 16); z.b = (uint4 *)hist; z.b16 = (long long)(zero_bytes / 16); z.c = sink.n_hits;
    k_kf_zero<<<64, 256, 0, s>>>(z);
    // ---- 2: the three independent first passes
    const bool known = a->map.n >= 0;
    const bool maybe_nonempty = known ? a->map.next_id > 0 : true;
    KfPhase1 p1;
    memset(&p1, 0, sizeof(p1));
    p1.g_expl = (maybe_nonempty && a->n_upper > 0) ? ovo_grid(known ? a->map.n : a->n_upper, 256) : 0;
    p1.g_filt = t->filter_depth ? ovo_grid((int64_t)t->cam.h * t->cam.w, 256, 512) : 0;
    const int64_t seg_pixels = (int64_t)t->seg_h * t->seg_w;
    p1.g_area = ovo_grid(seg_pixels, 256, 256);
    p1.xyz = a->map.xyz; p1.n_host = known ? a->map.n : -1; p1.state = (const long long *)a->map.state; p1.cam_map = a->cam; p1.depth_map = a->depth;
    p1.explained = a->explained;
    p1.depth_t = t->depth; p1.fh = t->cam.h; p1.fw = t->cam.w; p1.filter_th = 0.05f; p1.depth_f = t->depth_scratch;
    p1.seg_map = t->seg_map; p1.seg_pixels = seg_pixels; p1.n_masks = nm; p1.stats = stats;
    k_kf_phase1<<<p1.g_expl + p1.g_filt + p1.g_area, 256, 0, s>>>(p1, make_blur_taps(7, 2.5f));
    // ---- 3, 4: erode + subsample -> ordered append, state commit
    BackprojArgs b;
    for (int i = 0; i < 9; ++i) b.K[i] = a->K[i];
    for (int i = 0; i < 16; ++i) b.c2w[i] = a->c2w[i];
    b.h = a->h; b.w = a->w; b.ds = a->ds; b.ws_w = (int)ws_w;
    b.erode = a->erode && maybe_nonempty;
    CompactWs c = carve((char *)a->ws + 8, n_sub);
    const int g = ovo_grid(n_sub, 256);
    k_backproj_flag<<<g, 256, 0, s>>>(a->depth, maybe_nonempty ? a->explained : nullptr, b, n_sub, c.words, known ? nullptr : (const long long *)a->map.state);
    MapCommit mc;
    mc.state = (long long *)a->map.state; mc.total = nullptr; mc.result = (volatile long long *)a->result_host; mc.seq = a->seq;
    mc.cap = a->map.cap; mc.n_host = known ? a->map.n : -1; mc.id_host = a->map.next_id;
    // (every workgroup rebuilds the scan and ends on the commit ticket: 128 of them emit a 640 x 480 frame in 9 us, 1200 in 11.5)
    k_kf_emit<<<g < 128 ? g : 128, 256, 0, s>>>(a->depth, a->rgb, b, n_sub, c.words, a->map.xyz, a->map.ids, a->map.ins, a->map.rgb, mc);
    // ---- 5: the tracking pass (the map's size after the append is device-resident in any case)
    const float *depth = t->filter_depth ? t->depth_scratch : t->depth;
    const int64_t n_grid = a->n_upper + n_sub;
    const bool prof = ovo_prof_enabled();
    if (prof) ovo_prof_begin(2, 14.0 * (double)n_grid, s);
    k_track_project<<<ovo_grid(n_grid, 256, TRACK_GRID_CAP), 256, 0, s>>>(t->map.xyz, t->map.ins, 0, t->cam, depth, t->seg_map, t->seg_h, t->seg_w, t->ratio, t->point_seg, hist, nm,
                                                       t->hist_cols, counters, CNT_SLOTS, (const long long *)t->map.state, sink);
    if (prof) ovo_prof_end(s);
    // ---- 6: vote statistics + decisions
    Decide d;
    d.res = res; d.dst = dst; d.next_ins = t->next_ins; d.next_host = t->next_ins_host; d.track_th = t->track_th; d.n_masks = nm;
    k_vote_decide<<<nm, 256, 0, s>>>(hist, t->hist_cols, stats, tickets, d);
    // ---- 7: assignment | mask fusion, publish
    KfFinish f;
    // few, fat workgroups: every one of them pays a fixed chain of dependent loads (map size, targets / dst), a fence and its ticket, and beyond the
    // resident capacity they queue behind each other -- 1 M points, 32 masks of 640 x 480, idle GPU: 1024 + 32 x 32 workgroups 50.5 us, 256 + 32 x 8
    // 21.8, 128 + 32 x 4 22.9, 4096 + 32 x 32 118 (rocprofv3 over tools/round_profile.py with NOSAM=1)
    f.g_assign = ovo_grid(n_grid, 256, 256); f.gx = t->masks ? ovo_grid(t->pixels / 16, 256, 8) : 1;
    f.ins = t->map.ins; f.point_seg = t->point_seg; f.n_host = -1; f.n_dev = (const long long *)t->map.state;
    f.masks = (uint4 *)t->masks; f.px16 = t->masks ? t->pixels / 16 : 0; f.n_masks = nm; f.dst = dst; f.res = res;
    Publish pb;
    pb.res = res; pb.host = (volatile int32_t *)t->result_host; pb.counters = counters; pb.cnt_slots = CNT_SLOTS; pb.n_dev = (const long long *)t->map.state; pb.n_host = -1;
    pb.ticket = tickets + 1; pb.seq = t->seq; pb.n_ints = 8 + 6 * nm;
    k_kf_finish<<<f.g_assign + (t->masks ? f.gx * nm : 0), 256, 0, s>>>(f, pb);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // extern "C"
