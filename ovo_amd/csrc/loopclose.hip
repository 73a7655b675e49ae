// loopclose.hip -- device passes of the loop-closure semantic update (reference ovo.py:366-424 +
// instance_utils.py:5-35).  The reference walks all instance pairs in Python, slices the map per instance and asks
// Open3D's KD-tree for nearest-neighbour distances; here:
//   k_instance_moments : one pass over the map -> per-instance point count and coordinate sums (centroids, presence)
//   k_near_fraction    : for each candidate pair (a, b): how many points of a have a point of b closer than th
//                        (the only thing the reference uses the NN distances for: `(dists < th).mean()`), brute force
//                        over b with early exit, b's points streamed through LDS
//   k_remap_instances  : ins[i] = table[ins[i]] after the merges
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_instance_moments(const float *__restrict__ xyz, const int32_t *__restrict__ ins, long long n, int n_slots,
                                                          double *__restrict__ sums, int32_t *__restrict__ cnt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int id = ins[i];
        if (id < 0 || id >= n_slots) continue;
        atomicAdd(&cnt[id], 1);
        atomicAdd(&sums[3 * id + 0], (double)xyz[3 * i + 0]);
        atomicAdd(&sums[3 * id + 1], (double)xyz[3 * i + 1]);
        atomicAdd(&sums[3 * id + 2], (double)xyz[3 * i + 2]);
    }
}

// pts: the map points grouped by instance (CSR: rows [off[s], off[s+1]) belong to slot s); pairs i32 [n_pairs, 2] = (slot a, slot b)
__global__ void __launch_bounds__(256) k_near_fraction(const float *__restrict__ pts, const int64_t *__restrict__ off, const int32_t *__restrict__ pairs,
                                                       float th2, int32_t *__restrict__ near) {
    __shared__ float sb[256 * 3];
    const int pair = blockIdx.y, a = pairs[2 * pair], b = pairs[2 * pair + 1];
    const long long a0 = off[a], na = off[a + 1] - a0, b0 = off[b], nb = off[b + 1] - b0;
    const long long ia = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((long long)blockIdx.x * blockDim.x >= na) return;
    const bool live = ia < na;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = pts[3 * (a0 + ia)]; py = pts[3 * (a0 + ia) + 1]; pz = pts[3 * (a0 + ia) + 2]; }
    bool hit = !live;                                            // dead lanes never keep the block alive
    for (long long t = 0; t < nb; t += 256) {
        __syncthreads();
        const long long j = t + threadIdx.x;
        if (j < nb) { sb[3 * threadIdx.x] = pts[3 * (b0 + j)]; sb[3 * threadIdx.x + 1] = pts[3 * (b0 + j) + 1]; sb[3 * threadIdx.x + 2] = pts[3 * (b0 + j) + 2]; }
        __syncthreads();
        const int m = (int)(nb - t < 256 ? nb - t : 256);
        if (!hit) {
            for (int k = 0; k < m; ++k) {
                const float dx = px - sb[3 * k], dy = py - sb[3 * k + 1], dz = pz - sb[3 * k + 2];
                if (__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx))) < th2) { hit = true; break; }
            }
        }
        if (__syncthreads_and(hit)) break;                       // every point of this block already has a close neighbour
    }
    if (live && hit) atomicAdd(&near[pair], 1);
}

__global__ void __launch_bounds__(256) k_remap_instances(int32_t *__restrict__ ins, long long n, const int32_t *__restrict__ table, int n_slots) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int id = ins[i];
        if (id >= 0 && id < n_slots) ins[i] = table[id];
    }
}

}  // namespace

extern "C" int ovo_instance_moments(const float *xyz, const int32_t *ins, int64_t n, int n_slots, double *sums, int32_t *cnt, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && n_slots > 0 && sums && cnt, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    OVO_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 3 * n_slots, st));
    OVO_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * n_slots, st));
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(xyz && ins, "null pointer");
    k_instance_moments<<<ovo_grid(n, 256), 256, 0, st>>>(xyz, ins, n, n_slots, sums, cnt);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_near_fraction(const float *pts_by_instance, const int64_t *offsets, const int32_t *pairs, int n_pairs, int64_t max_points_a,
                                 float th, int32_t *near_count, ovo_stream_t stream) {
    OVO_REQUIRE(n_pairs >= 0 && n_pairs <= 65535 && max_points_a >= 0 && th >= 0.f, "bad argument");
    if (n_pairs == 0) return OVO_OK;
    OVO_REQUIRE(pts_by_instance && offsets && pairs && near_count, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    OVO_HIP(hipMemsetAsync(near_count, 0, sizeof(int32_t) * n_pairs, st));
    if (max_points_a == 0) return OVO_OK;
    dim3 grid((unsigned)((max_points_a + 255) / 256), (unsigned)n_pairs);
    k_near_fraction<<<grid, 256, 0, st>>>(pts_by_instance, offsets, pairs, th * th, near_count);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_remap_instances(int32_t *ins, int64_t n, const int32_t *table, int n_slots, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && n_slots > 0 && table, "bad argument");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(ins, "null pointer");
    k_remap_instances<<<ovo_grid(n, 256), 256, 0, (hipStream_t)stream>>>(ins, n, table, n_slots);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
