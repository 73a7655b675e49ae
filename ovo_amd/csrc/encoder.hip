// encoder.hip -- the bandwidth-bound glue of the ViT / Hiera forward passes and of TextRegion region pooling:
// LayerNorm, token assembly, im2col, crop+resize+normalise, rotary embedding, mask -> token-grid resampling,
// multi-resolution token stitching, row scaling / L2 normalisation.  One wave per row wherever a row is
// reduced (64-lane shuffles, no LDS), 16-byte accesses along the contiguous dimension.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float x) {      // v_cvt_pk_bf16_f32 (RNE)
    const __bf16 h = (__bf16)x;
    return *(const uint16_t *)&h;
}
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    return make_uint2((uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16), (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16));
}

// ---- LayerNorm: one wave per row.  Rows of up to 2048 columns are read ONCE into registers (NV float4 per lane, all loads in flight together);
// the sum, the centred sum of squares and the output walk the registers in the order the three-pass form walks memory, so the bits are the same.
// (the three-pass form over an L1-resident row, kept for wider rows, moved 3.1 TB/s on the ViT's 13 848 x 1024 rows: three dependent rounds of loads)
__device__ __forceinline__ void ln_row_wide(const float *__restrict__ x, int d, const float *__restrict__ gamma,
                                            const float *__restrict__ beta, float eps, void *__restrict__ y, int out_bf16, int lane,
                                            const float *__restrict__ extra_add) {
    const int d4 = d >> 2;
    float s = 0.f;
    for (int i = lane; i < d4; i += 64) {
        float4 v = ((const float4 *)x)[i];
        if (extra_add) { const float4 e = ((const float4 *)extra_add)[i]; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = lane; i < d4; i += 64) {
        float4 v = ((const float4 *)x)[i];
        if (extra_add) { const float4 e = ((const float4 *)extra_add)[i]; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e2 = v.w - mean;
        q += (a * a + b * b) + (c * c + e2 * e2);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    for (int i = lane; i < d4; i += 64) {
        float4 v = ((const float4 *)x)[i];
        if (extra_add) { const float4 e = ((const float4 *)extra_add)[i]; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        const float4 g = ((const float4 *)gamma)[i], b = ((const float4 *)beta)[i];
        const float o0 = (v.x - mean) * rstd * g.x + b.x, o1 = (v.y - mean) * rstd * g.y + b.y;
        const float o2 = (v.z - mean) * rstd * g.z + b.z, o3 = (v.w - mean) * rstd * g.w + b.w;
        if (out_bf16) ((uint2 *)y)[i] = pack4_bf16(o0, o1, o2, o3);
        else ((float4 *)y)[i] = make_float4(o0, o1, o2, o3);
    }
}

template <int NV>
__device__ __forceinline__ void ln_row_regs(const float *__restrict__ x, int d, const float *__restrict__ gamma,
                                            const float *__restrict__ beta, float eps, void *__restrict__ y, int out_bf16, int lane,
                                            const float *__restrict__ extra_add) {
    const int d4 = d >> 2;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < d4 ? ((const float4 *)x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (extra_add) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < d4) { const float4 e = ((const float4 *)extra_add)[i]; v[j].x += e.x; v[j].y += e.y; v[j].z += e.z; v[j].w += e.w; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane + 64 * j < d4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane + 64 * j < d4) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e2 = v[j].w - mean;
            q += (a * a + b * b) + (c * c + e2 * e2);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < d4) {
            const float4 g = ((const float4 *)gamma)[i], b = ((const float4 *)beta)[i];
            const float o0 = (v[j].x - mean) * rstd * g.x + b.x, o1 = (v[j].y - mean) * rstd * g.y + b.y;
            const float o2 = (v[j].z - mean) * rstd * g.z + b.z, o3 = (v[j].w - mean) * rstd * g.w + b.w;
            if (out_bf16) ((uint2 *)y)[i] = pack4_bf16(o0, o1, o2, o3);
            else ((float4 *)y)[i] = make_float4(o0, o1, o2, o3);
        }
    }
}

__device__ __forceinline__ void ln_row(const float *__restrict__ x, int d, const float *__restrict__ gamma,
                                       const float *__restrict__ beta, float eps, void *__restrict__ y, int out_bf16, int lane,
                                       const float *__restrict__ extra_add = nullptr) {
    const int nv = ((d >> 2) + 63) >> 6;                         // (wave-uniform)
    if (nv <= 2) ln_row_regs<2>(x, d, gamma, beta, eps, y, out_bf16, lane, extra_add);
    else if (nv <= 4) ln_row_regs<4>(x, d, gamma, beta, eps, y, out_bf16, lane, extra_add);
    else if (nv <= 8) ln_row_regs<8>(x, d, gamma, beta, eps, y, out_bf16, lane, extra_add);
    else ln_row_wide(x, d, gamma, beta, eps, y, out_bf16, lane, extra_add);
}

__global__ void __launch_bounds__(256) k_layernorm(const float *__restrict__ x, long long xs, long long rows, int d,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                   char *__restrict__ y, long long ys, int out_bf16) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += waves)
        ln_row(x + r * xs, d, gamma, beta, eps, y + r * ys * (out_bf16 ? 2 : 4), out_bf16, lane);
}

__global__ void __launch_bounds__(256) k_vit_embed(const float *__restrict__ patch, const float *__restrict__ prefix, int n_prefix,
                                                   const float *__restrict__ pos, int B, int P, int d, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float eps, float *__restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int T = n_prefix + P;
    const long long rows = (long long)B * T, waves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += waves) {
        const int b = (int)(r / T), t = (int)(r % T);
        const float *src = t < n_prefix ? prefix + (long long)t * d : patch + ((long long)b * P + (t - n_prefix)) * d;
        const float *pe = pos ? pos + (long long)t * d : nullptr;
        float *dst = x + r * d;
        if (gamma) ln_row(src, d, gamma, beta, eps, dst, 0, lane, pe);
        else
            for (int i = lane; i < (d >> 2); i += 64) {
                float4 v = ((const float4 *)src)[i];
                if (pe) { const float4 e = ((const float4 *)pe)[i]; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
                ((float4 *)dst)[i] = v;
            }
    }
}

// ---- im2col ----
__global__ void __launch_bounds__(256) k_im2col(const float *__restrict__ img, int B, int C, int H, int W, int ksz, int stride, int pad,
                                                int oh, int ow, uint16_t *__restrict__ out, int kpad) {
    // eight consecutive patch columns per thread, one 16-byte store (kpad % 8 == 0): the one-element-per-thread form wrote 2 bytes per
    // lane and moved 0.7 TB/s on the 1024^2 SAM2 input (four frames: 223 us).  Index arithmetic in 32 bits (the launch checks the sizes), the
    // (channel, ky, kx) of the eight columns advanced incrementally: five 64-bit and twenty-four 32-bit divisions per thread were ~1100 VALU
    // instructions for 16 bytes of output (137 us per four frames = 1 TB/s)
    const uint32_t k8 = (uint32_t)kpad >> 3;
    const uint32_t total = (uint32_t)B * oh * ow * k8;
    const int kk = ksz * ksz, kreal = C * kk;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t row = i / k8, kb = (i - row * k8) * 8;
        const uint32_t t1 = row / (uint32_t)ow, ox = row - t1 * (uint32_t)ow;
        const uint32_t b = t1 / (uint32_t)oh, oy = t1 - b * (uint32_t)oh;
        int c = (int)(kb / (uint32_t)kk), rem = (int)kb - c * kk, ky = rem / ksz, kx = rem - ky * ksz;
        const int y0 = (int)oy * stride - pad, x0 = (int)ox * stride - pad;
        uint16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = 0.f;
            if ((int)kb + e < kreal) {
                const int y = y0 + ky, x = x0 + kx;
                if (y >= 0 && y < H && x >= 0 && x < W) f = img[(((long long)b * C + c) * H + y) * W + x];
            }
            v[e] = f2bf(f);
            if (++kx == ksz) { kx = 0; if (++ky == ksz) { ky = 0; ++c; } }
        }
        uint4 p;
        p.x = v[0] | ((uint32_t)v[1] << 16); p.y = v[2] | ((uint32_t)v[3] << 16); p.z = v[4] | ((uint32_t)v[5] << 16); p.w = v[6] | ((uint32_t)v[7] << 16);
        *(uint4 *)(out + (long long)row * kpad + kb) = p;
    }
}

// ---- crop + resize + normalise ----
struct ResizeArgs {
    const void *src; int src_u8, hwc; int C, H, W, y0, x0, ch, cw, oh, ow, aa;     // hwc: interleaved [H, W, C] source (a camera frame as it arrives)
    int vh, vw, top, left;         // the output is the window (top, left, oh, ow) of a virtual vh x vw resize (Resize + CenterCrop)
    float scale, mean[4], std[4];
};
__device__ __forceinline__ float src_px(const ResizeArgs &a, int c, int y, int x) {
    const long long i = a.hwc ? ((long long)(a.y0 + y) * a.W + (a.x0 + x)) * a.C + c : ((long long)c * a.H + (a.y0 + y)) * a.W + (a.x0 + x);
    return a.src_u8 ? (float)((const uint8_t *)a.src)[i] : ((const float *)a.src)[i];
}
// all channels of one source pixel.  An interleaved u8 frame (the camera frame as it arrives) keeps a pixel's 3 channels in consecutive bytes: ONE unaligned
// 4-byte load instead of three byte loads -- the resize kernels are bound by their gather instructions (27 byte loads per output pixel of SAM2's 1024^2
// input, ~60 for a TextRegion crop), not by arithmetic.  The very last pixel of the frame would read one byte past the buffer: it takes the byte loads.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ void src_px_all(const ResizeArgs &a, int y, int x, float (&v)[4]) {
    if (a.hwc && a.src_u8 && a.C == 3) {
        const long long i = ((long long)(a.y0 + y) * a.W + (a.x0 + x)) * 3;
        const uint8_t *p = (const uint8_t *)a.src + i;
        if (i + 4 <= (long long)a.H * a.W * 3) {
            const uint32_t w = *(const u32_unaligned *)p;
            v[0] = (float)(w & 0xffu); v[1] = (float)((w >> 8) & 0xffu); v[2] = (float)((w >> 16) & 0xffu);
        } else { v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2]; }
        return;
    }
    for (int c = 0; c < a.C; ++c) v[c] = src_px(a, c, y, x);
}
__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }
// Keys cubic with a = -0.5: the kernel of torch's antialiased bicubic (_upsample_bicubic2d_aa; the plain bicubic uses -0.75)
__device__ __forceinline__ float cubic_aa(float x) {
    x = fabsf(x);
    if (x < 1.f) return ((1.5f * x - 2.5f) * x) * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * -0.5f;
    return 0.f;
}
__device__ __forceinline__ float aa_filter(int kind, float x) { return kind == 2 ? cubic_aa(x) : tri(x); }

// One thread per output pixel; the tap weights do not depend on the channel, so they are computed once (the first version recomputed them,
// divisions included, per channel and per tap: 30 us for a 1024^2 output that is 13 MB of stores) and the channels run innermost.
template <int T>
__device__ __forceinline__ void aa_taps(const ResizeArgs &a, int ymin, int xmin, int ny, int nx, float cy, float cx, float ivy, float ivx, float wy_tot,
                                        float wx_tot, float (&v)[4]) {
    float wy[T], wx[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        wy[i] = i < ny ? aa_filter(a.aa, ((float)(ymin + i) - cy + 0.5f) * ivy) / wy_tot : 0.f;
        wx[i] = i < nx ? aa_filter(a.aa, ((float)(xmin + i) - cx + 0.5f) * ivx) / wx_tot : 0.f;
    }
#pragma unroll
    for (int iy = 0; iy < T; ++iy) {
        if (iy >= ny) break;
        float rowacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ix = 0; ix < T; ++ix) {
            if (ix >= nx) break;
            float px[4];
            src_px_all(a, ymin + iy, xmin + ix, px);
            for (int c = 0; c < a.C; ++c) rowacc[c] += wx[ix] * px[c];
        }
        for (int c = 0; c < a.C; ++c) v[c] += wy[iy] * rowacc[c];
    }
}

#define OVO_RS_TAPS 10                                  // taps per axis held in registers: down-scaling up to ~4.5x; beyond that the generic loop
__device__ __forceinline__ void resize_norm_body(const ResizeArgs &a, float *__restrict__ out) {
    const int wx_ = blockIdx.x * 64 + (threadIdx.x & 63), wy_ = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (wx_ >= a.ow || wy_ >= a.oh) return;
    const int ox = wx_ + a.left, oy = wy_ + a.top;     // position in the virtual vh x vw output
    const float sy = (float)a.ch / (float)a.vh, sx = (float)a.cw / (float)a.vw;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (!a.aa) {                                        // torch upsample_bilinear2d, align_corners = False
        float fy = sy * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        float fx = sx * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < a.ch - 1 ? 1 : 0), x1 = x0 + (x0 < a.cw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        float p00[4], p01[4], p10[4], p11[4];
        src_px_all(a, y0, x0, p00); src_px_all(a, y0, x1, p01); src_px_all(a, y1, x0, p10); src_px_all(a, y1, x1, p11);
        for (int c = 0; c < a.C; ++c)
            v[c] = (1.f - ly) * ((1.f - lx) * p00[c] + lx * p01[c]) + ly * ((1.f - lx) * p10[c] + lx * p11[c]);
    } else {                                            // torch _upsample_bilinear2d_aa / _upsample_bicubic2d_aa (separable triangle / cubic filter)
        const float half = a.aa == 2 ? 2.f : 1.f;       // interp_size / 2
        const float supy = (sy >= 1.f ? sy : 1.f) * half, supx = (sx >= 1.f ? sx : 1.f) * half;
        const float ivy = sy >= 1.f ? 1.f / sy : 1.f, ivx = sx >= 1.f ? 1.f / sx : 1.f;
        const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
        int ymin = (int)(cy - supy + 0.5f); ymin = ymin < 0 ? 0 : ymin;
        int ymax = (int)(cy + supy + 0.5f); ymax = ymax > a.ch ? a.ch : ymax;
        int xmin = (int)(cx - supx + 0.5f); xmin = xmin < 0 ? 0 : xmin;
        int xmax = (int)(cx + supx + 0.5f); xmax = xmax > a.cw ? a.cw : xmax;
        float wy_tot = 0.f, wx_tot = 0.f;
        for (int y = ymin; y < ymax; ++y) wy_tot += aa_filter(a.aa, ((float)y - cy + 0.5f) * ivy);
        for (int x = xmin; x < xmax; ++x) wx_tot += aa_filter(a.aa, ((float)x - cx + 0.5f) * ivx);
        const int ny = ymax - ymin, nx = xmax - xmin;
        // taps held in registers: 3 per axis when up-scaling (support 1: a 640 x 480 frame to SAM2's 1024^2 -- the 10-tap form spent most of its
        // time on taps that do not exist: 60 -> 20 us), 6 up to ~2.5x down-scaling (the 336^2 ViT crops of that frame), 10 up to ~4.5x; the bound
        // comes from the scale factors, so a launch takes one branch.  Same weights, same accumulation order in every form.
        const int bound = (int)(2.f * (supy > supx ? supy : supx)) + 1;
        if (bound <= 3 && ny <= 3 && nx <= 3) aa_taps<3>(a, ymin, xmin, ny, nx, cy, cx, ivy, ivx, wy_tot, wx_tot, v);
        else if (bound <= 6 && ny <= 6 && nx <= 6) aa_taps<6>(a, ymin, xmin, ny, nx, cy, cx, ivy, ivx, wy_tot, wx_tot, v);
        else if (ny <= OVO_RS_TAPS && nx <= OVO_RS_TAPS) {
            aa_taps<OVO_RS_TAPS>(a, ymin, xmin, ny, nx, cy, cx, ivy, ivx, wy_tot, wx_tot, v);
        } else {
            for (int c = 0; c < a.C; ++c) {
                float acc = 0.f;
                for (int y = ymin; y < ymax; ++y) {
                    const float wyy = aa_filter(a.aa, ((float)y - cy + 0.5f) * ivy) / wy_tot;
                    float rowacc = 0.f;
                    for (int x = xmin; x < xmax; ++x) rowacc += (aa_filter(a.aa, ((float)x - cx + 0.5f) * ivx) / wx_tot) * src_px(a, c, y, x);
                    acc += wyy * rowacc;
                }
                v[c] = acc;
            }
        }
    }
    for (int c = 0; c < a.C; ++c) out[((long long)c * a.oh + wy_) * a.ow + wx_] = (v[c] * a.scale - a.mean[c]) / a.std[c];
}
__global__ void __launch_bounds__(256) k_resize_norm(ResizeArgs a, float *__restrict__ out) { resize_norm_body(a, out); }

// Several (source image, crop) pairs of one geometry in ONE launch (blockIdx.z): the look-ahead encoders resize 12 frames x 2 TextRegion crops and 12 SAM2
// inputs per group -- 36 launches of 12-18 us each, most of it launch ramp for 0.3 M output pixels.  Same body, same arithmetic per pixel.
constexpr int RESIZE_BATCH_MAX = 64;
struct ResizeBatch { const void *src[RESIZE_BATCH_MAX]; int y0[RESIZE_BATCH_MAX], x0[RESIZE_BATCH_MAX], ch[RESIZE_BATCH_MAX], cw[RESIZE_BATCH_MAX]; };
__global__ void __launch_bounds__(256) k_resize_norm_batch(ResizeArgs a, ResizeBatch b, float *__restrict__ out) {
    const int z = blockIdx.z;
    a.src = b.src[z]; a.y0 = b.y0[z]; a.x0 = b.x0[z]; a.ch = b.ch[z]; a.cw = b.cw[z];
    resize_norm_body(a, out + (long long)z * a.C * a.oh * a.ow);
}

// The two resizes every keyframe pays -- an interleaved u8 camera frame (3 channels) through the antialiased triangle filter to SAM2's 1024^2 (3 taps per
// axis) and to the TextRegion crops' 336^2 (<= 6) -- without the generic kernel's control flow: T taps per axis held in registers, taps beyond the filter's
// support carry a ZERO weight and a clamped address instead of a branch (x + 0 * p = x: the sums are the generic kernel's, bit for bit), the source layout
// and filter are compile-time.  The generic body (run-time layout / filter / tap-count branches around every tap, tap arrays in scratch) spent ~1000
// instructions per output pixel: the two launches of a 14-frame group 419 + 122 us.
template <int T>
__global__ void __launch_bounds__(256) k_resize_tri_hwc3(ResizeArgs a, ResizeBatch b, float *__restrict__ out) {
    const int z = blockIdx.z;
    const uint8_t *src = (const uint8_t *)b.src[z];
    const int y0c = b.y0[z], x0c = b.x0[z], ch = b.ch[z], cw = b.cw[z];
    const int wx_ = blockIdx.x * 64 + (threadIdx.x & 63), wy_ = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (wx_ >= a.ow || wy_ >= a.oh) return;
    const int ox = wx_ + a.left, oy = wy_ + a.top;
    const float sy = (float)ch / (float)a.vh, sx = (float)cw / (float)a.vw;
    const float supy = sy >= 1.f ? sy : 1.f, supx = sx >= 1.f ? sx : 1.f;
    const float ivy = sy >= 1.f ? 1.f / sy : 1.f, ivx = sx >= 1.f ? 1.f / sx : 1.f;
    const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
    int ymin = (int)(cy - supy + 0.5f); ymin = ymin < 0 ? 0 : ymin;
    int ymax = (int)(cy + supy + 0.5f); ymax = ymax > ch ? ch : ymax;
    int xmin = (int)(cx - supx + 0.5f); xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(cx + supx + 0.5f); xmax = xmax > cw ? cw : xmax;
    float wy[T], wx[T], wy_tot = 0.f, wx_tot = 0.f;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        wy[i] = ymin + i < ymax ? tri(((float)(ymin + i) - cy + 0.5f) * ivy) : 0.f;
        wx[i] = xmin + i < xmax ? tri(((float)(xmin + i) - cx + 0.5f) * ivx) : 0.f;
        wy_tot += wy[i]; wx_tot += wx[i];
    }
#pragma unroll
    for (int i = 0; i < T; ++i) { wy[i] = wy[i] / wy_tot; wx[i] = wx[i] / wx_tot; }
    const long long last = (long long)a.H * a.W * 3 - 4;          // byte offset up to which a 4-byte load stays inside the frame
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int iy = 0; iy < T; ++iy) {
        int y = ymin + iy; y = y < ch - 1 ? y : ch - 1;
        const long long row = (long long)(y0c + y) * a.W + x0c;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int ix = 0; ix < T; ++ix) {
            int x = xmin + ix; x = x < cw - 1 ? x : cw - 1;
            const long long off = (row + x) * 3;
            uint32_t w;
            if (off <= last) w = *(const u32_unaligned *)(src + off);
            else w = (uint32_t)src[off] | ((uint32_t)src[off + 1] << 8) | ((uint32_t)src[off + 2] << 16);
            r0 += wx[ix] * (float)(w & 0xffu); r1 += wx[ix] * (float)((w >> 8) & 0xffu); r2 += wx[ix] * (float)((w >> 16) & 0xffu);
        }
        v0 += wy[iy] * r0; v1 += wy[iy] * r1; v2 += wy[iy] * r2;
    }
    float *o = out + (long long)z * 3 * a.oh * a.ow + (long long)wy_ * a.ow + wx_;
    const long long plane = (long long)a.oh * a.ow;
    o[0] = (v0 * a.scale - a.mean[0]) / a.std[0];
    o[plane] = (v1 * a.scale - a.mean[1]) / a.std[1];
    o[2 * plane] = (v2 * a.scale - a.mean[2]) / a.std[2];
}
// taps per axis the triangle filter can reach for a crop of ch x cw pixels resized to vh x vw (the generic kernel's own bound)
inline int resize_tap_bound(int ch, int cw, int vh, int vw) {
    const float sy = (float)ch / (float)vh, sx = (float)cw / (float)vw;
    const float supy = sy >= 1.f ? sy : 1.f, supx = sx >= 1.f ? sx : 1.f;
    return (int)(2.f * (supy > supx ? supy : supx)) + 1;
}
// one launch over nz (source, crop) pairs: the fast kernel when it covers them, else the generic one
inline void launch_resize_batch(const ResizeArgs &a, const ResizeBatch &b, int nz, float *out, hipStream_t s) {
    dim3 grid((a.ow + 63) / 64, (a.oh + 3) / 4, nz);
    int bound = 0;
    for (int z = 0; z < nz; ++z) { const int t = resize_tap_bound(b.ch[z], b.cw[z], a.vh, a.vw); bound = t > bound ? t : bound; }
    static const bool generic_once = getenv("OVO_RESIZE_GENERIC") != nullptr;              // measurement / tests
    const bool generic = ovo_knobs_dynamic() ? getenv("OVO_RESIZE_GENERIC") != nullptr : generic_once;
    if (!generic && a.hwc && a.src_u8 && a.C == 3 && a.aa == 1 && bound <= 3) k_resize_tri_hwc3<3><<<grid, 256, 0, s>>>(a, b, out);
    else if (!generic && a.hwc && a.src_u8 && a.C == 3 && a.aa == 1 && bound <= 6) k_resize_tri_hwc3<6><<<grid, 256, 0, s>>>(a, b, out);
    else k_resize_norm_batch<<<grid, 256, 0, s>>>(a, b, out);
}

// ---- a14: per-mask crops for the crop-mode descriptors (segment_utils.py:29-41, 118-170) ----
// k_mask_boxes: XYXY box of every mask (inclusive max edges) -> XYWH with the reference's w = x2 - x1, h = y2 - y1 (the last
// column / row is NOT part of the crop: segment_utils.py:88-94 subtracts inclusive edges); empty mask -> 0,0,0,0.
__global__ void __launch_bounds__(256) k_mask_boxes(const uint8_t *__restrict__ masks, int H, int W, int32_t *__restrict__ boxes) {
    __shared__ int s[4];
    if (threadIdx.x == 0) { s[0] = W; s[1] = H; s[2] = -1; s[3] = -1; }
    __syncthreads();
    const uint8_t *m = masks + (long long)blockIdx.x * H * W;
    int x0 = W, y0 = H, x1 = -1, y1 = -1;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        if (m[i]) {
            const int y = i / W, x = i - y * W;
            x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        }
    }
    if (x1 >= 0) { atomicMin(&s[0], x0); atomicMin(&s[1], y0); atomicMax(&s[2], x1); atomicMax(&s[3], y1); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *b = boxes + 4 * blockIdx.x;
        if (s[2] < 0) { b[0] = b[1] = b[2] = b[3] = 0; }
        else { b[0] = s[0]; b[1] = s[1]; b[2] = s[2] - s[0]; b[3] = s[3] - s[1]; }
    }
}

// k_mask_crops: out[i, part] = antialiased-bilinear resize to R x R (torchvision F.resize on a tensor) of
//   part 0: the masked crop (zero background) -- zero-padded to a centred square first when there is no bbox part ("vanilla")
//   part 1: the box crop grown by `margin` px (clamped at 0 on the left / top, sliced at the image edge on the right / bottom)
// One thread per output pixel, all 3 channels; the source "canvas" is virtual (never materialised).
struct CropArgs {
    const void *img; int img_u8, H, W;
    const uint8_t *masks; const int32_t *boxes;
    int n, also_bbox, margin, R, round_out;
};
__global__ void __launch_bounds__(256) k_mask_crops(CropArgs a, float *__restrict__ out) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int parts = a.also_bbox ? 2 : 1, i = blockIdx.z / parts, part = blockIdx.z % parts;
    if (ox >= a.R || oy >= a.R) return;
    const int32_t *b = a.boxes + 4 * i;
    int bx = b[0], by = b[1], bw = b[2], bh = b[3];
    int ch, cw, offy = 0, offx = 0;                      // canvas size, placement of the crop inside it
    const uint8_t *mask = nullptr;
    if (part == 0) {
        mask = a.masks + (long long)i * a.H * a.W;
        if (a.also_bbox) { ch = bh; cw = bw; }
        else {                                           // pad_img (segment_utils.py:141-150)
            ch = cw = bh > bw ? bh : bw;
            if (bh > bw) offx = (bh - bw) / 2; else offy = (bw - bh) / 2;
        }
    } else {                                             // increase_bbox_by_margin + slicing (segment_utils.py:152-172, 136-139)
        bx -= a.margin; by -= a.margin; bw += 2 * a.margin; bh += 2 * a.margin;
        if (bx < 0) { bw += bx; bx = 0; }
        if (by < 0) { bh += by; by = 0; }
        if (bx + bw > a.W) bw = a.W - bx;
        if (by + bh > a.H) bh = a.H - by;
        ch = bh; cw = bw;
    }
    float *o = out + ((long long)(i * parts + part) * 3) * a.R * a.R + (long long)oy * a.R + ox;
    const long long plane = (long long)a.R * a.R;
    if (ch <= 0 || cw <= 0) { o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f; return; }   // degenerate box (the reference raises)
    const float sy = (float)ch / (float)a.R, sx = (float)cw / (float)a.R;
    const float supy = sy >= 1.f ? sy : 1.f, supx = sx >= 1.f ? sx : 1.f;
    const float ivy = sy >= 1.f ? 1.f / sy : 1.f, ivx = sx >= 1.f ? 1.f / sx : 1.f;
    const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
    int ymin = (int)(cy - supy + 0.5f); ymin = ymin < 0 ? 0 : ymin;
    int ymax = (int)(cy + supy + 0.5f); ymax = ymax > ch ? ch : ymax;
    int xmin = (int)(cx - supx + 0.5f); xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(cx + supx + 0.5f); xmax = xmax > cw ? cw : xmax;
    float wy_tot = 0.f, wx_tot = 0.f;
    for (int y = ymin; y < ymax; ++y) wy_tot += tri(((float)y - cy + 0.5f) * ivy);
    for (int x = xmin; x < xmax; ++x) wx_tot += tri(((float)x - cx + 0.5f) * ivx);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int y = ymin; y < ymax; ++y) {
        const float wy = tri(((float)y - cy + 0.5f) * ivy) / wy_tot;
        const int yy = y - offy;
        float row[3] = {0.f, 0.f, 0.f};
        if (yy >= 0 && yy < bh) {
            for (int x = xmin; x < xmax; ++x) {
                const int xx = x - offx;
                if (xx < 0 || xx >= bw) continue;
                const long long pix = (long long)(by + yy) * a.W + (bx + xx);
                if (mask && !mask[pix]) continue;
                const float wx = tri(((float)x - cx + 0.5f) * ivx) / wx_tot;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const long long idx = (long long)c * a.H * a.W + pix;
                    row[c] += wx * (a.img_u8 ? (float)((const uint8_t *)a.img)[idx] : ((const float *)a.img)[idx]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += wy * row[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = a.round_out ? rintf(acc[c]) : acc[c];
}

// ---- rotary embedding on q, k of a packed [B, T, 3, H, hd] bf16 buffer ----
__global__ void __launch_bounds__(256) k_rope_qk(uint32_t *__restrict__ qkv, int B, int T, int H, int hd, const float *__restrict__ cos_t,
                                                 const float *__restrict__ sin_t, int t0) {
    const int hp = hd >> 1;                                    // pairs per head
    const long long total = (long long)B * T * 2 * H * hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hp);
        long long r = i / hp;
        const int h = (int)(r % H); r /= H;
        const int which = (int)(r % 2); r /= 2;                 // 0 = q, 1 = k
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        if (t < t0) continue;
        uint32_t *w = qkv + ((((long long)b * T + t) * 3 + which) * H + h) * hp + p;
        const uint32_t raw = *w;
        const float x0 = __uint_as_float(raw << 16), x1 = __uint_as_float(raw & 0xffff0000u);
        const float c0 = cos_t[(long long)t * hd + 2 * p], c1 = cos_t[(long long)t * hd + 2 * p + 1];
        const float s0 = sin_t[(long long)t * hd + 2 * p], s1 = sin_t[(long long)t * hd + 2 * p + 1];
        const float y0 = x0 * c0 - x1 * s0, y1 = x1 * c1 + x0 * s1;
        *w = (uint32_t)f2bf(y0) | ((uint32_t)f2bf(y1) << 16);
    }
}

// ---- a15: masks -> token-grid weights ----
__global__ void __launch_bounds__(256) k_feature_masks(const uint8_t *__restrict__ masks, int N, int H, int W, int gh, int gw,
                                                       uint16_t *__restrict__ w, int gpad, float *__restrict__ cnt) {
    const int n = blockIdx.x;
    const uint8_t *m = masks + (long long)n * H * W;
    const float sy = (float)H / (float)gh, sx = (float)W / (float)gw;
    int local = 0;
    for (int g = threadIdx.x; g < gpad; g += blockDim.x) {
        float v = 0.f;
        if (g < gh * gw) {
            const int gy = g / gw, gx = g % gw;
            float fy = sy * ((float)gy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
            float fx = sx * ((float)gx + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float a = m[(long long)y0 * W + x0] ? 1.f : 0.f, b = m[(long long)y0 * W + x1] ? 1.f : 0.f;
            const float c = m[(long long)y1 * W + x0] ? 1.f : 0.f, d = m[(long long)y1 * W + x1] ? 1.f : 0.f;
            v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
        }
        const bool on = v > 0.f;
        w[(long long)n * gpad + g] = on ? 0x3f80 : 0;      // bf16 1.0 / 0.0
        local += on;
    }
    __shared__ int red[256];
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) cnt[n] = (float)red[0];
}

// ---- a16: multi-resolution stitch, transposed output [d, gpad] ----
__global__ void __launch_bounds__(256) k_stitch_t(const float *__restrict__ tokens, int tpc, int t0, int d, int P, int nh, int nw,
                                                  uint16_t *__restrict__ out, int gpad) {
    const int gh = P * nh, gw = P * nw, G = gh * gw;
    const long long total = (long long)d * gpad;
    const float sy = (float)P / (float)gh, sx = (float)P / (float)gw;
    // thread -> (g fastest across lanes for coalesced writes; reads of a token row stride d are served by L2)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % gpad), c = (int)(i / gpad);
        float v = 0.f;
        if (g < G) {
            const int gy = g / gw, gx = g % gw;
            float fy = sy * ((float)gy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
            float fx = sx * ((float)gx + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < P - 1 ? 1 : 0), x1 = x0 + (x0 < P - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float *glob = tokens + (long long)t0 * d + c;                        // crop 0
            const float a = glob[(long long)(y0 * P + x0) * d], b = glob[(long long)(y0 * P + x1) * d];
            const float e = glob[(long long)(y1 * P + x0) * d], f = glob[(long long)(y1 * P + x1) * d];
            const float up = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * e + lx * f);
            const int tile = 1 + (gy / P) * nw + (gx / P);
            const float loc = tokens[((long long)tile * tpc + t0 + (gy % P) * P + (gx % P)) * d + c];
            v = 0.5f * up + loc;
        }
        out[i] = f2bf(v);
    }
}

// ---- a18: remove_global_patch (textregion.py:31-50) ----
// k_unit_tokens: per token g the L2 norm over the d channels of x_t [d, gpad] (lanes run along g: coalesced), then the
// unit token in BOTH layouts the two GEMMs of the score need: u_t [d, gpad] and u [gpad, d]; padding tokens are zero.
__global__ void __launch_bounds__(256) k_unit_tokens(const uint16_t *__restrict__ x_t, int d, int G, int gpad, uint16_t *__restrict__ u_t,
                                                     uint16_t *__restrict__ u) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gpad) return;
    float ss = 0.f;
    if (g < G)
        for (int c = 0; c < d; ++c) { const float v = bf2f(x_t[(long long)c * gpad + g]); ss = fmaf(v, v, ss); }
    const float inv = g < G ? 1.0f / sqrtf(ss) : 0.f;
    for (int c = 0; c < d; ++c) {
        const uint16_t q = g < G ? f2bf(bf2f(x_t[(long long)c * gpad + g]) * inv) : (uint16_t)0;
        u_t[(long long)c * gpad + g] = q;
        u[(long long)g * d + c] = q;
    }
}

// k_global_patch: r_t f32 [N, gpad] holds, per mask n and token g, cos(token g, mean unit token of mask n) -- the
// reference's patch_2_region_avg without its T x T similarity matrix (sum_t' in n of p_g.p_t' = p_g . sum_t' p_t').
// A token whose mean score over the masks it belongs to does not exceed its mean score over the others by `th` is a
// "global" patch: its column is cleared in every mask.
__global__ void __launch_bounds__(256) k_global_patch(const float *__restrict__ r_t, uint16_t *__restrict__ w, int N, int G, int gpad,
                                                      float th) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float belong = 0.f, outside = 0.f, nb = 0.f, no = 0.f;
    for (int n = 0; n < N; ++n) {
        const bool in = w[(long long)n * gpad + g] != 0;
        const float r = r_t[(long long)n * gpad + g];
        belong += in ? r : r * 0.f;                    // r * 0 keeps a NaN score (empty mask: 0/0 in the reference) poisonous
        outside += in ? r * 0.f : r;
        nb += in ? 1.f : 0.f;
        no += in ? 0.f : 1.f;
    }
    const float diff = belong / (nb + 1e-9f) - outside / (no + 1e-9f);
    if (diff < th)
        for (int n = 0; n < N; ++n) w[(long long)n * gpad + g] = 0;
}

__global__ void __launch_bounds__(256) k_row_count(const uint16_t *__restrict__ w, int gpad, float *__restrict__ cnt) {
    int local = 0;
    for (int g = threadIdx.x; g < gpad; g += blockDim.x) local += w[(long long)blockIdx.x * gpad + g] != 0;
    __shared__ int red[256];
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) cnt[blockIdx.x] = (float)red[0];
}

__global__ void __launch_bounds__(256) k_scale_rows(const float *__restrict__ x, const float *__restrict__ cnt, int N, int d,
                                                    uint16_t *__restrict__ y) {
    const long long total = (long long)N * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        y[i] = f2bf(x[i] / cnt[i / d]);
}

__global__ void __launch_bounds__(256) k_l2norm(const float *__restrict__ x, long long N, int d, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < N; r += waves) {
        const float *row = x + r * d;
        float s = 0.f;
        for (int i = lane; i < d; i += 64) s += row[i] * row[i];
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);      // F.normalize eps
        for (int i = lane; i < d; i += 64) y[r * d + i] = row[i] * inv;
    }
}

__global__ void __launch_bounds__(256) k_cast(const float4 *__restrict__ x, long long n4, uint2 *__restrict__ y, int dtype) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        if (dtype == 2) y[i] = pack4_bf16(v.x, v.y, v.z, v.w);
        else {
            const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
            y[i] = make_uint2(*(const uint32_t *)&a, *(const uint32_t *)&b);
        }
    }
}

}  // namespace

extern "C" {

int ovo_layernorm(const float *x, int64_t x_stride, int64_t rows, int d, const float *gamma, const float *beta, float eps,
                  void *y, int64_t y_stride, int out_dtype, ovo_stream_t stream) {
    OVO_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0, "d must be a multiple of 4");
    OVO_REQUIRE(out_dtype == 0 || out_dtype == 2, "out_dtype: 0 = f32, 2 = bf16");
    if (rows == 0) return OVO_OK;
    OVO_REQUIRE(x && gamma && beta && y && x_stride % 4 == 0 && y_stride % 4 == 0, "null pointer / misaligned stride");
    k_layernorm<<<ovo_grid(rows * 64, 256), 256, 0, (hipStream_t)stream>>>(x, x_stride, rows, d, gamma, beta, eps, (char *)y,
                                                                          y_stride, out_dtype == 2);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_vit_embed(const float *patch, const float *prefix, int n_prefix, const float *pos, int B, int P, int d,
                  const float *gamma, const float *beta, float eps, float *x, ovo_stream_t stream) {
    OVO_REQUIRE(patch && x && B > 0 && P > 0 && d > 0 && d % 4 == 0 && n_prefix >= 0, "bad argument");
    OVO_REQUIRE(n_prefix == 0 || prefix, "prefix tokens missing");
    OVO_REQUIRE((gamma == nullptr) == (beta == nullptr), "gamma and beta go together");
    k_vit_embed<<<ovo_grid((long long)B * (n_prefix + P) * 64, 256), 256, 0, (hipStream_t)stream>>>(patch, prefix, n_prefix, pos, B, P, d,
                                                                                                    gamma, beta, eps, x);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_im2col(const float *img, int B, int C, int H, int W, int ksz, int stride, int pad, void *out, int kpad,
               ovo_stream_t stream) {
    OVO_REQUIRE(img && out && B > 0 && C > 0 && H > 0 && W > 0 && ksz > 0 && stride > 0 && pad >= 0, "bad argument");
    OVO_REQUIRE(kpad >= C * ksz * ksz && kpad % 32 == 0, "kpad must cover C*k*k and be a multiple of 32");
    const int oh = (H + 2 * pad - ksz) / stride + 1, ow = (W + 2 * pad - ksz) / stride + 1;
    OVO_REQUIRE(oh > 0 && ow > 0, "empty output");
    OVO_REQUIRE(kpad % 8 == 0 && ((uintptr_t)out & 15) == 0, "kpad must be a multiple of 8 and the output 16-byte aligned");
    // (the kernel's 32-bit grid-stride index must not wrap: total + one grid stride -- at most 2048 x 256 threads -- stays below 2^32)
    OVO_REQUIRE((long long)B * oh * ow * (kpad / 8) < (1ll << 32) - (1ll << 22), "more than 2^32 16-byte pieces: split the batch");
    k_im2col<<<ovo_grid((long long)B * oh * ow * (kpad / 8), 256), 256, 0, (hipStream_t)stream>>>(img, B, C, H, W, ksz, stride, pad, oh, ow,
                                                                                                 (uint16_t *)out, kpad);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_resize_window_normalize(const void *src, int src_dtype, int C, int H, int W, int y0, int x0, int ch, int cw, float *out,
                                int oh, int ow, int virt_h, int virt_w, int top, int left, int filter, float scale,
                                const float *mean3_host, const float *std3_host, ovo_stream_t stream) {
    OVO_REQUIRE(src && out && (src_dtype == 0 || src_dtype == 3 || src_dtype == 4), "src_dtype: 0 = f32 [C,H,W], 3 = u8 [C,H,W], 4 = u8 [H,W,C]");
    OVO_REQUIRE(C >= 1 && C <= 4 && ch > 0 && cw > 0 && oh > 0 && ow > 0, "bad shape");
    OVO_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + ch <= H && x0 + cw <= W, "crop outside the image");
    OVO_REQUIRE(top >= 0 && left >= 0 && top + oh <= virt_h && left + ow <= virt_w, "window outside the resized image");
    OVO_REQUIRE(filter >= 0 && filter <= 2, "filter: 0 = bilinear, 1 = antialiased bilinear, 2 = antialiased bicubic");
    ResizeArgs a;
    a.src = src; a.src_u8 = src_dtype >= 3; a.hwc = src_dtype == 4; a.C = C; a.H = H; a.W = W; a.y0 = y0; a.x0 = x0; a.ch = ch; a.cw = cw;
    a.oh = oh; a.ow = ow; a.aa = filter; a.scale = scale;
    a.vh = virt_h; a.vw = virt_w; a.top = top; a.left = left;
    for (int c = 0; c < 4; ++c) { a.mean[c] = mean3_host && c < C ? mean3_host[c] : 0.f; a.std[c] = std3_host && c < C ? std3_host[c] : 1.f; }
    if (a.hwc && a.src_u8 && C == 3 && filter == 1) {               // the camera-frame case: the branch-free kernel (a batch of one)
        ResizeBatch b = {};
        b.src[0] = src; b.y0[0] = y0; b.x0[0] = x0; b.ch[0] = ch; b.cw[0] = cw;
        launch_resize_batch(a, b, 1, out, (hipStream_t)stream);
    } else {
        dim3 grid((ow + 63) / 64, (oh + 3) / 4);
        k_resize_norm<<<grid, 256, 0, (hipStream_t)stream>>>(a, out);
    }
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_resize_normalize(const void *src, int src_dtype, int C, int H, int W, int y0, int x0, int ch, int cw, float *out,
                         int oh, int ow, int antialias, float scale, const float *mean3_host, const float *std3_host,
                         ovo_stream_t stream) {
    return ovo_resize_window_normalize(src, src_dtype, C, H, W, y0, x0, ch, cw, out, oh, ow, oh, ow, 0, 0, antialias, scale, mean3_host,
                                       std3_host, stream);
}

int ovo_resize_normalize_batch(const void *const *srcs_host, int n_src, int src_dtype, int C, int H, int W, const int32_t *crops_host, int n_crop, float *out,
                               int oh, int ow, int antialias, float scale, const float *mean3_host, const float *std3_host, ovo_stream_t stream) {
    OVO_REQUIRE(srcs_host && crops_host && out && n_src > 0 && n_crop > 0, "null / empty argument");
    OVO_REQUIRE(src_dtype == 0 || src_dtype == 3 || src_dtype == 4, "src_dtype: 0 = f32 [C,H,W], 3 = u8 [C,H,W], 4 = u8 [H,W,C]");
    OVO_REQUIRE(C >= 1 && C <= 4 && oh > 0 && ow > 0 && antialias >= 0 && antialias <= 2, "bad shape");
    ResizeArgs a;
    a.src = nullptr; a.src_u8 = src_dtype >= 3; a.hwc = src_dtype == 4; a.C = C; a.H = H; a.W = W; a.y0 = a.x0 = 0; a.ch = H; a.cw = W;
    a.oh = oh; a.ow = ow; a.aa = antialias; a.scale = scale; a.vh = oh; a.vw = ow; a.top = 0; a.left = 0;
    for (int c = 0; c < 4; ++c) { a.mean[c] = mean3_host && c < C ? mean3_host[c] : 0.f; a.std[c] = std3_host && c < C ? std3_host[c] : 1.f; }
    for (int k = 0; k < n_crop; ++k) {
        const int32_t *r = crops_host + 4 * k;
        OVO_REQUIRE(r[2] > 0 && r[3] > 0 && r[0] >= 0 && r[1] >= 0 && r[0] + r[2] <= H && r[1] + r[3] <= W, "crop outside the image");
    }
    // output image i * n_crop + k = crop k of source i; as many launches of up to RESIZE_BATCH_MAX pairs as it takes
    const int total = n_src * n_crop;
    for (int z0 = 0; z0 < total; z0 += RESIZE_BATCH_MAX) {
        const int nz = total - z0 < RESIZE_BATCH_MAX ? total - z0 : RESIZE_BATCH_MAX;
        ResizeBatch b;
        for (int z = 0; z < nz; ++z) {
            const int i = (z0 + z) / n_crop, k = (z0 + z) % n_crop;
            OVO_REQUIRE(srcs_host[i], "null source image");
            b.src[z] = srcs_host[i]; b.y0[z] = crops_host[4 * k]; b.x0[z] = crops_host[4 * k + 1]; b.ch[z] = crops_host[4 * k + 2]; b.cw[z] = crops_host[4 * k + 3];
        }
        launch_resize_batch(a, b, nz, out + (long long)z0 * C * oh * ow, (hipStream_t)stream);
    }
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_mask_boxes(const uint8_t *masks, int n, int H, int W, int32_t *boxes_xywh, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && H > 0 && W > 0, "bad shape");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(masks && boxes_xywh, "null pointer");
    k_mask_boxes<<<n, 256, 0, (hipStream_t)stream>>>(masks, H, W, boxes_xywh);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_mask_crops(const void *image, int img_dtype, int H, int W, const uint8_t *masks, const int32_t *boxes_xywh, int n,
                   int also_bbox, int margin, int R, int round_out, float *out, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && H > 0 && W > 0 && R > 0 && margin >= 0, "bad shape");
    OVO_REQUIRE(img_dtype == 0 || img_dtype == 3, "img_dtype: 0 = f32, 3 = u8");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(image && masks && boxes_xywh && out, "null pointer");
    CropArgs a;
    a.img = image; a.img_u8 = img_dtype == 3; a.H = H; a.W = W; a.masks = masks; a.boxes = boxes_xywh; a.n = n;
    a.also_bbox = also_bbox != 0; a.margin = margin; a.R = R; a.round_out = round_out != 0;
    dim3 grid((R + 63) / 64, (R + 3) / 4, n * (a.also_bbox ? 2 : 1));
    k_mask_crops<<<grid, 256, 0, (hipStream_t)stream>>>(a, out);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_rope_qk(void *qkv, int B, int T, int H, int hd, const float *cos_t, const float *sin_t, int t0, ovo_stream_t stream) {
    OVO_REQUIRE(qkv && cos_t && sin_t && B > 0 && T > 0 && H > 0 && hd > 0 && hd % 2 == 0 && t0 >= 0, "bad argument");
    k_rope_qk<<<ovo_grid((long long)B * T * H * hd, 256), 256, 0, (hipStream_t)stream>>>((uint32_t *)qkv, B, T, H, hd, cos_t, sin_t, t0);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_feature_masks(const uint8_t *masks, int N, int H, int W, int gh, int gw, void *w, int gpad, float *cnt,
                      ovo_stream_t stream) {
    OVO_REQUIRE(N >= 0 && H > 0 && W > 0 && gh > 0 && gw > 0 && gpad >= gh * gw && gpad % 32 == 0, "bad shape");
    if (N == 0) return OVO_OK;
    OVO_REQUIRE(masks && w && cnt, "null pointer");
    k_feature_masks<<<N, 256, 0, (hipStream_t)stream>>>(masks, N, H, W, gh, gw, (uint16_t *)w, gpad, cnt);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_stitch_tokens_t(const float *tokens, int tokens_per_crop, int t0, int d, int P, int nh, int nw, void *out, int gpad,
                        ovo_stream_t stream) {
    OVO_REQUIRE(tokens && out && d > 0 && P > 0 && nh > 0 && nw > 0 && t0 >= 0, "bad argument");
    OVO_REQUIRE(tokens_per_crop >= t0 + P * P && gpad >= P * P * nh * nw && gpad % 32 == 0, "bad token / grid shape");
    k_stitch_t<<<ovo_grid((long long)d * gpad, 256), 256, 0, (hipStream_t)stream>>>(tokens, tokens_per_crop, t0, d, P, nh, nw,
                                                                                   (uint16_t *)out, gpad);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_unit_tokens(const void *x_t, int d, int G, int gpad, void *u_t, void *u, ovo_stream_t stream) {
    OVO_REQUIRE(x_t && u_t && u && d > 0 && G > 0 && gpad >= G, "bad argument");
    k_unit_tokens<<<(gpad + 255) / 256, 256, 0, (hipStream_t)stream>>>((const uint16_t *)x_t, d, G, gpad, (uint16_t *)u_t, (uint16_t *)u);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_global_patch_filter(const float *r_t, void *weights, int N, int G, int gpad, float th, float *cnt, ovo_stream_t stream) {
    OVO_REQUIRE(N >= 0 && G > 0 && gpad >= G, "bad shape");
    if (N == 0) return OVO_OK;
    OVO_REQUIRE(r_t && weights && cnt, "null pointer");
    k_global_patch<<<(G + 255) / 256, 256, 0, (hipStream_t)stream>>>(r_t, (uint16_t *)weights, N, G, gpad, th);
    k_row_count<<<N, 256, 0, (hipStream_t)stream>>>((const uint16_t *)weights, gpad, cnt);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_scale_rows_bf16(const float *x, const float *cnt, int N, int d, void *y, ovo_stream_t stream) {
    OVO_REQUIRE(N >= 0 && d > 0, "bad shape");
    if (N == 0) return OVO_OK;
    OVO_REQUIRE(x && cnt && y, "null pointer");
    k_scale_rows<<<ovo_grid((long long)N * d, 256), 256, 0, (hipStream_t)stream>>>(x, cnt, N, d, (uint16_t *)y);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_l2_normalize_rows(const float *x, int64_t N, int d, float *y, ovo_stream_t stream) {
    OVO_REQUIRE(N >= 0 && d > 0, "bad shape");
    if (N == 0) return OVO_OK;
    OVO_REQUIRE(x && y, "null pointer");
    k_l2norm<<<ovo_grid(N * 64, 256), 256, 0, (hipStream_t)stream>>>(x, N, d, y);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

int ovo_cast_f32(const float *x, int64_t n, void *y, int dtype, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && n % 4 == 0 && (dtype == 1 || dtype == 2), "n % 4 == 0, dtype 1 (f16) or 2 (bf16)");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(x && y && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 7) == 0), "null / misaligned pointer");
    k_cast<<<ovo_grid(n / 4, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)x, n / 4, (uint2 *)y, dtype);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // extern "C"
