// image.hip -- per-frame image-space helpers: depth high-pass filter (geometry_utils.py:92-96),
// mask bit-packing and pairwise mask intersections for NMS (segment_utils.py:195-230).
#include "common.h"
#include "depth_filter.h"

namespace {

__global__ void __launch_bounds__(256) k_depth_filter(const float *__restrict__ depth, int h, int w, BlurTaps taps, float th,
                                                      float *__restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    out[(int64_t)y * w + x] = depth_filter_pixel(depth, h, w, taps, th, x, y);
}

__global__ void __launch_bounds__(256) k_pack_masks(const uint8_t *__restrict__ masks, int64_t pixels, unsigned long long *__restrict__ bits,
                                                    int64_t words) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.y;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t wd = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); wd < words; wd += waves) {
        const int64_t px = wd * 64 + lane;
        const bool b = px < pixels && masks[(int64_t)m * pixels + px] != 0;
        const unsigned long long v = __ballot(b);
        if (lane == 0) bits[(int64_t)m * words + wd] = v;
    }
}

// the inverse of k_pack_masks: one byte (0 / 1) per bit, 16 pixels (= 16 bits of a word) per thread
__global__ void __launch_bounds__(256) k_unpack_masks(const unsigned long long *__restrict__ bits, int64_t words, int64_t pixels,
                                                      uint8_t *__restrict__ masks) {
    const int m = blockIdx.y;
    const int64_t px16 = pixels / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < px16; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int b = (unsigned int)(bits[(int64_t)m * words + (i >> 2)] >> ((i & 3) * 16)) & 0xffffu;
        uint4 o;
        unsigned int *op = &o.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int n4 = (b >> (4 * k)) & 0xfu;
            op[k] = (n4 & 1u) | ((n4 & 2u) << 7) | ((n4 & 4u) << 14) | ((n4 & 8u) << 21);
        }
        ((uint4 *)(masks + (int64_t)m * pixels))[i] = o;
    }
}

// inter[i][j] = popcount(mask_i & mask_j); block = one i and a strip of 4 j's (one wave each).
__global__ void __launch_bounds__(256) k_mask_inter(const unsigned long long *__restrict__ bits, int n, int64_t words,
                                                    int32_t *__restrict__ inter) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n || j < i) return;
    const unsigned long long *a = bits + (int64_t)i * words, *b = bits + (int64_t)j * words;
    int c = 0;
    for (int64_t k = lane; k < words; k += 64) c += __popcll(a[k] & b[k]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) { inter[(int64_t)i * n + j] = c; inter[(int64_t)j * n + i] = c; }
}

// masks[dst] |= masks[src] for every (dst, src) pair (a mask is never both), 16 pixels per thread
__global__ void __launch_bounds__(256) k_mask_or(uint4 *__restrict__ masks, long long px16, const int32_t *__restrict__ pairs, int n_pairs) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < px16; i += (long long)gridDim.x * blockDim.x)
        for (int p = 0; p < n_pairs; ++p) {
            const long long d = pairs[2 * p] * px16 + i, s = pairs[2 * p + 1] * px16 + i;
            uint4 a = masks[d];
            const uint4 b = masks[s];
            a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
            masks[d] = a;
        }
}

// area[k] = number of non-zero bytes of masks[rows[k]]
__global__ void __launch_bounds__(256) k_mask_area(const uint8_t *__restrict__ masks, long long pixels, const int32_t *__restrict__ rows,
                                                   int32_t *__restrict__ area) {
    const uint8_t *m = masks + (long long)rows[blockIdx.y] * pixels;
    int c = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x) c += m[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(area + blockIdx.y, c);
}

// dst[k] = src[idx[k]] for rows of row16 16-byte units
__global__ void __launch_bounds__(256) k_gather_rows(const uint4 *__restrict__ src, long long row16, const int32_t *__restrict__ idx,
                                                     uint4 *__restrict__ dst) {
    const uint4 *s = src + (long long)idx[blockIdx.y] * row16;
    uint4 *d = dst + (long long)blockIdx.y * row16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < row16; i += (long long)gridDim.x * blockDim.x) d[i] = s[i];
}

}  // namespace

extern "C" {

int ovo_gather_rows(const void *src, int64_t row_bytes, const int32_t *idx, int n, void *dst, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && row_bytes > 0 && row_bytes % 16 == 0, "row_bytes must be a multiple of 16");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(src && idx && dst && n <= 65535 && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0), "null / misaligned pointer");
    dim3 grid(ovo_grid(row_bytes / 16, 256, 64), n);
    k_gather_rows<<<grid, 256, 0, (hipStream_t)stream>>>((const uint4 *)src, row_bytes / 16, idx, (uint4 *)dst);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_mask_or(uint8_t *masks, int64_t pixels, const int32_t *pairs, int n_pairs, ovo_stream_t stream) {
    OVO_REQUIRE(n_pairs >= 0 && pixels > 0 && pixels % 16 == 0, "pixels must be a multiple of 16");
    if (n_pairs == 0) return OVO_OK;
    OVO_REQUIRE(masks && pairs && ((uintptr_t)masks & 15) == 0, "null / misaligned pointer");
    k_mask_or<<<ovo_grid(pixels / 16, 256, 512), 256, 0, (hipStream_t)stream>>>((uint4 *)masks, pixels / 16, pairs, n_pairs);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_mask_area(const uint8_t *masks, int64_t pixels, const int32_t *rows, int n_rows, int32_t *area, ovo_stream_t stream) {
    OVO_REQUIRE(n_rows >= 0 && pixels > 0, "bad shape");
    if (n_rows == 0) return OVO_OK;
    OVO_REQUIRE(masks && rows && area && n_rows <= 65535, "null pointer");
    OVO_HIP(hipMemsetAsync(area, 0, (size_t)n_rows * sizeof(int32_t), (hipStream_t)stream));
    dim3 grid(ovo_grid(pixels, 256, 64), n_rows);
    k_mask_area<<<grid, 256, 0, (hipStream_t)stream>>>(masks, pixels, rows, area);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_depth_filter(const float *depth, int h, int w, int ksize, float sigma, float th, float *out, ovo_stream_t stream) {
    OVO_REQUIRE(depth && out && h > 0 && w > 0, "null / empty image");
    OVO_REQUIRE(ksize >= 1 && ksize <= 15 && (ksize & 1) && sigma > 0.f, "ksize must be odd and <= 15");
    OVO_REQUIRE(h > ksize / 2 && w > ksize / 2, "image smaller than the reflect padding");
    const BlurTaps t = make_blur_taps(ksize, sigma);
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    k_depth_filter<<<grid, 256, 0, (hipStream_t)stream>>>(depth, h, w, t, th, out);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_pack_masks(const uint8_t *masks, int n, int64_t pixels, uint64_t *bits, int64_t words, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && pixels > 0 && words * 64 >= pixels, "bad shape");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(masks && bits && n <= 65535, "null pointer / too many masks");
    dim3 grid(ovo_grid(words * 64, 256, 64), n);
    k_pack_masks<<<grid, 256, 0, (hipStream_t)stream>>>(masks, pixels, (unsigned long long *)bits, words);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_unpack_masks(const uint64_t *bits, int n, int64_t pixels, int64_t words, uint8_t *masks, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && pixels > 0 && pixels % 16 == 0 && words * 64 >= pixels, "pixels must be a multiple of 16");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(bits && masks && n <= 65535 && ((uintptr_t)masks & 15) == 0, "null / misaligned pointer");
    dim3 grid(ovo_grid(pixels / 16, 256, 64), n);
    k_unpack_masks<<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned long long *)bits, words, pixels, masks);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_mask_intersections(const uint64_t *bits, int n, int64_t words, int32_t *inter, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && words > 0, "bad shape");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(bits && inter && n <= 65535, "null pointer / too many masks");
    dim3 grid((n + 3) / 4, n);
    k_mask_inter<<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned long long *)bits, n, words, inter);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // extern "C"
