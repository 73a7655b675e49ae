// depth_filter.h -- the 7x7 Gaussian high-pass of geometry_utils.py:92-96, shared by its own kernel (image.hip) and the round chain (geometry.hip).
#pragma once
#include <math.h>

#include "common.h"

struct BlurTaps { float w[15 * 15]; int k; };

// torchvision _get_gaussian_kernel1d: pdf at linspace(-(k-1)/2, (k-1)/2, k), normalised, in fp32; the 2-D kernel is the outer product
static inline BlurTaps make_blur_taps(int ksize, float sigma) {
    BlurTaps t;
    t.k = ksize;
    float k1[15], sum = 0.f;
    const float half = (ksize - 1) * 0.5f;
    for (int i = 0; i < ksize; ++i) {
        const float x = ksize == 1 ? 0.f : -half + (2.f * half) * (float)i / (float)(ksize - 1);
        const float r = x / sigma;
        k1[i] = expf(-0.5f * r * r);
        sum += k1[i];
    }
    for (int i = 0; i < ksize; ++i) k1[i] /= sum;
    for (int a = 0; a < ksize; ++a)
        for (int b = 0; b < ksize; ++b) t.w[a * ksize + b] = k1[a] * k1[b];
    return t;
}

__device__ __forceinline__ int reflect_index(int i, int n) {   // torch 'reflect' padding (no edge repeat)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// the reference's 7 x 7 kernel with the loops unrolled (tap weights become scalar operands) and no reflection arithmetic away from the border:
// the same fmaf chain in the same (dy, dx) order -- same bits -- at a third of the time of the run-time loops below (25 -> 8 us on 640 x 480)
template <int K>
__device__ __forceinline__ float depth_filter_pixel_k(const float *__restrict__ depth, int h, int w, const BlurTaps &taps, float th, int x, int y) {
    constexpr int P = K >> 1;
    float acc = 0.f;
    if (x >= P && x < w - P && y >= P && y < h - P) {
        const float *c = depth + (int64_t)(y - P) * w + (x - P);
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int dx = 0; dx < K; ++dx) acc = fmaf(taps.w[dy * K + dx], c[dy * w + dx], acc);
    } else {
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            int yy = y + dy - P;
            yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                int xx = x + dx - P;
                xx = xx < 0 ? -xx : xx; xx = xx >= w ? 2 * w - 2 - xx : xx;
                acc = fmaf(taps.w[dy * K + dx], depth[(int64_t)yy * w + xx], acc);
            }
        }
    }
    const float d = depth[(int64_t)y * w + x];
    return fabsf(d - acc) > th ? -1.0f : d;
}

__device__ __forceinline__ float depth_filter_pixel(const float *__restrict__ depth, int h, int w, const BlurTaps &taps, float th, int x, int y) {
    if (taps.k == 7 && h > 7 && w > 7) return depth_filter_pixel_k<7>(depth, h, w, taps, th, x, y);
    const int k = taps.k, p = k >> 1;
    float acc = 0.f;
    for (int dy = 0; dy < k; ++dy) {
        const int yy = reflect_index(y + dy - p, h);
        for (int dx = 0; dx < k; ++dx) {
            const int xx = reflect_index(x + dx - p, w);
            acc = fmaf(taps.w[dy * k + dx], depth[(int64_t)yy * w + xx], acc);
        }
    }
    const float d = depth[(int64_t)y * w + x];
    return fabsf(d - acc) > th ? -1.0f : d;
}
