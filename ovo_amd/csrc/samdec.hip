// samdec.hip -- glue kernels of the SAM2 mask decoder.  The matrix products of the decoder run on
// ovo_gemm / ovo_attention; this file holds what sits between them:
//   * k_row_epilogue  : [+ broadcast base] -> [LayerNorm] -> f32 / bf16 / bf16(+ positional code) copies, one pass
//   * k_upscale_ln    : ConvTranspose2d(2x2, stride 2) as GEMM -> pixel shuffle + bias + skip feature + LayerNorm2d + GELU
//   * k_upscale_masks : second ConvTranspose2d + skip + GELU fused with the hyper-network product -> mask logits
// All tensors are token-major (NHWC): the image embedding of the encoder is consumed as it is produced.
#include "common.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float x) {      // v_cvt_pk_bf16_f32 (RNE)
    const __bf16 h = (__bf16)x;
    return *(const uint16_t *)&h;
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    return make_uint2((uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16), (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16));
}
// GELU (erf form) with the polynomial erf of gemm.hip (|error| <= 8.6e-6 absolute)
__device__ __forceinline__ float gelu_poly(float x) {
    const float z = x * 0.70710678118654752f;
    const float zc = fminf(fmaxf(z, -3.5f), 3.5f);
    const float s = fmaf(zc * zc, 0.16326530612244897f, -1.0f);
    float p = -3.398861796e-03f;
    p = fmaf(p, s, 8.621919328e-03f); p = fmaf(p, s, -8.698635955e-03f); p = fmaf(p, s, 1.271555869e-02f);
    p = fmaf(p, s, -2.870869786e-02f); p = fmaf(p, s, 4.709060027e-02f); p = fmaf(p, s, -6.528488840e-02f);
    p = fmaf(p, s, 8.795614477e-02f); p = fmaf(p, s, -1.145324569e-01f); p = fmaf(p, s, 1.467849556e-01f);
    p = fmaf(p, s, -2.007044758e-01f); p = fmaf(p, s, 4.038725490e-01f);
    const float hx = 0.5f * x;
    return fmaf(hx, p * zc, hx);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {                // gelu_poly on two values: v_pk_fma_f32 / v_pk_mul_f32
    const f32x2 z = x * 0.70710678118654752f;
    const f32x2 zc = __builtin_elementwise_min(__builtin_elementwise_max(z, (f32x2)(-3.5f)), (f32x2)(3.5f));
    const f32x2 s = __builtin_elementwise_fma(zc * zc, (f32x2)(0.16326530612244897f), (f32x2)(-1.0f));
    f32x2 p = (f32x2)(-3.398861796e-03f);
    p = __builtin_elementwise_fma(p, s, (f32x2)(8.621919328e-03f)); p = __builtin_elementwise_fma(p, s, (f32x2)(-8.698635955e-03f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(1.271555869e-02f)); p = __builtin_elementwise_fma(p, s, (f32x2)(-2.870869786e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(4.709060027e-02f)); p = __builtin_elementwise_fma(p, s, (f32x2)(-6.528488840e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(8.795614477e-02f)); p = __builtin_elementwise_fma(p, s, (f32x2)(-1.145324569e-01f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(1.467849556e-01f)); p = __builtin_elementwise_fma(p, s, (f32x2)(-2.007044758e-01f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(4.038725490e-01f));
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, p * zc, hx);
}

// ---- one wave per row of C <= 1024 channels (C % 4 == 0) ----
struct RowArgs {
    const float *x; const float *base; long long base_rows;
    const float *gamma, *beta; float eps;
    const float *pe; long long pe_rows;
    float *y; uint16_t *y16; uint16_t *ype16;
    long long R; int C;
};
template <int NV>
__global__ void __launch_bounds__(256) k_row_epilogue(RowArgs a) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += waves) {
        float4 v[NV];
        const int nv = (a.C + 255) / 256;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nv && c < a.C) {
                v[i] = *(const float4 *)(a.x + r * a.C + c);
                if (a.base) {
                    const float4 b = *(const float4 *)(a.base + (r % a.base_rows) * a.C + c);
                    v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;
                }
                sum += v[i].x + v[i].y + v[i].z + v[i].w;
            }
        }
        if (a.gamma) {
            const float mean = wave_sum(sum) / (float)a.C;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (i < nv && c < a.C) {
                    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                    sq += dx * dx + dy * dy + dz * dz + dw * dw;
                }
            }
            const float rstd = rsqrtf(wave_sum(sq) / (float)a.C + a.eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (i < nv && c < a.C) {
                    const float4 g = *(const float4 *)(a.gamma + c), b = *(const float4 *)(a.beta + c);
                    v[i].x = (v[i].x - mean) * rstd * g.x + b.x; v[i].y = (v[i].y - mean) * rstd * g.y + b.y;
                    v[i].z = (v[i].z - mean) * rstd * g.z + b.z; v[i].w = (v[i].w - mean) * rstd * g.w + b.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (i < nv && c < a.C) {
                if (a.y) *(float4 *)(a.y + r * a.C + c) = v[i];
                if (a.y16) *(uint2 *)(a.y16 + r * a.C + c) = pack4(v[i].x, v[i].y, v[i].z, v[i].w);
                if (a.ype16) {
                    const float4 p = *(const float4 *)(a.pe + (r % a.pe_rows) * a.C + c);
                    *(uint2 *)(a.ype16 + r * a.C + c) = pack4(v[i].x + p.x, v[i].y + p.y, v[i].z + p.z, v[i].w + p.w);
                }
            }
        }
    }
}

// ---- upscale stage 1: g bf16 [P * s * s, 4 * C1] with column = (dy * 2 + dx) * C1 + c  ->  out bf16 [P, 2s, 2s, C1] ----
// LPP lanes per output pixel, 4 channels per lane (C1 = 4 * LPP)
template <int LPP>
__global__ void __launch_bounds__(256) k_upscale_ln(const uint16_t *__restrict__ g, const float *__restrict__ bias, const float *__restrict__ feat,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, long long P, int s,
                                                    uint16_t *__restrict__ out) {
    constexpr int C1 = 4 * LPP, PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    const long long pix = (long long)blockIdx.x * PPB + threadIdx.x / LPP, side = 2 * s, total = P * side * side;
    if (pix >= total) return;
    const long long p = pix / (side * side);
    const int rem = (int)(pix % (side * side)), y = rem / (int)side, x = rem % (int)side;
    const long long row = p * s * s + (long long)(y >> 1) * s + (x >> 1);
    const int col = (((y & 1) << 1) | (x & 1)) * C1 + sub * 4;
    const uint2 raw = *(const uint2 *)(g + row * (4 * C1) + col);
    const float4 b = *(const float4 *)(bias + sub * 4), f = *(const float4 *)(feat + (long long)rem * C1 + sub * 4);
    float v0 = bf2f((uint16_t)(raw.x & 0xffff)) + b.x + f.x, v1 = bf2f((uint16_t)(raw.x >> 16)) + b.y + f.y;
    float v2 = bf2f((uint16_t)(raw.y & 0xffff)) + b.z + f.z, v3 = bf2f((uint16_t)(raw.y >> 16)) + b.w + f.w;
    float sum = v0 + v1 + v2 + v3;
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)C1;
    const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
    float sq = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)C1 + eps);
    const float4 gm = *(const float4 *)(gamma + sub * 4), bt = *(const float4 *)(beta + sub * 4);
    *(uint2 *)(out + pix * C1 + sub * 4) = pack4(gelu_poly(d0 * rstd * gm.x + bt.x), gelu_poly(d1 * rstd * gm.y + bt.y),
                                                 gelu_poly(d2 * rstd * gm.z + bt.z), gelu_poly(d3 * rstd * gm.w + bt.w));
}

// ---- upscale stage 2 + hyper-network product: g bf16 [P * s2 * s2, 4 * C2]  ->  masks f32 [P, n_out, 2 s2, 2 s2] ----
// one thread per output pixel; the prompt's n_out x C2 hyper-network rows sit in LDS (blockIdx.y = prompt)
__global__ void __launch_bounds__(256) k_upscale_masks(const uint16_t *__restrict__ g, const float *__restrict__ bias, const float *__restrict__ feat,
                                                       const float *__restrict__ hyper, int n_mask, int first, int n_out, int s2, int C2,
                                                       float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [n_out][C2] hyper rows, then [C2] bias
    const long long p = blockIdx.y;
    for (int i = threadIdx.x; i < n_out * C2; i += blockDim.x) sh[i] = hyper[(p * n_mask + first) * C2 + i];
    for (int i = threadIdx.x; i < C2; i += blockDim.x) sh[n_out * C2 + i] = bias[i];
    __syncthreads();
    const int side = 2 * s2;
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= side * side) return;
    const int y = rem / side, x = rem % side;
    const long long row = p * s2 * s2 + (long long)(y >> 1) * s2 + (x >> 1);
    const uint16_t *src = g + row * (4 * C2) + (((y & 1) << 1) | (x & 1)) * C2;
    const float *ft = feat + (long long)rem * C2;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C2; c += 8) {                            // 16-byte loads, GELU on packed f32 (two values per VALU slot)
        const uint4 raw = *(const uint4 *)(src + c);
        const float4 fa = *(const float4 *)(ft + c), fb = *(const float4 *)(ft + c + 4);
        const float4 ba = *(const float4 *)(sh + n_out * C2 + c), bb = *(const float4 *)(sh + n_out * C2 + c + 4);
        const f32x2 v01 = gelu2(f32x2{__uint_as_float(raw.x << 16) + ba.x + fa.x, __uint_as_float(raw.x & 0xffff0000u) + ba.y + fa.y});
        const f32x2 v23 = gelu2(f32x2{__uint_as_float(raw.y << 16) + ba.z + fa.z, __uint_as_float(raw.y & 0xffff0000u) + ba.w + fa.w});
        const f32x2 v45 = gelu2(f32x2{__uint_as_float(raw.z << 16) + bb.x + fb.x, __uint_as_float(raw.z & 0xffff0000u) + bb.y + fb.y});
        const f32x2 v67 = gelu2(f32x2{__uint_as_float(raw.w << 16) + bb.z + fb.z, __uint_as_float(raw.w & 0xffff0000u) + bb.w + fb.w});
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n_out) {
                const float4 ha = *(const float4 *)(sh + i * C2 + c), hb = *(const float4 *)(sh + i * C2 + c + 4);
                acc[i] = fmaf(ha.x, v01.x, fmaf(ha.y, v01.y, fmaf(ha.z, v23.x, fmaf(ha.w, v23.y, acc[i]))));
                acc[i] = fmaf(hb.x, v45.x, fmaf(hb.y, v45.y, fmaf(hb.z, v67.x, fmaf(hb.w, v67.y, acc[i]))));
            }
    }
    for (int i = 0; i < n_out; ++i) out[((p * n_out + i) * side + y) * side + x] = acc[i];
}

// ---- image -> token cross attention of the two-way transformer: S image queries x T (<= 16) token keys, head_dim 16 ----
// The generic flash kernel pads T = 8 keys to a 64-key tile and head_dim 16 to 64 (5 TFLOP/s, 0.9 ms per call at 256 clicks);
// here one thread owns one (pixel, head): 8 dot products of length 16 against the prompt's keys in LDS, softmax in registers,
// 16 outputs.  HBM-bound: 256 B read + 256 B written per pixel.  Lanes run head-fastest, so a wave touches 8 whole pixels.
__global__ void __launch_bounds__(256) k_i2t_attention(const uint16_t *__restrict__ q, long long q_sb, int q_st, const uint16_t *__restrict__ k,
                                                       const uint16_t *__restrict__ v, uint16_t *__restrict__ o, int S, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float kv[];   // [T][ci] keys, then [T][ci] values (ci = 16 H)
    const int ci = 16 * H;
    const long long p = blockIdx.y;
    for (int i = threadIdx.x; i < T * ci; i += blockDim.x) {
        kv[i] = bf2f(k[p * T * ci + i]) * scale;
        kv[T * ci + i] = bf2f(v[p * T * ci + i]);
    }
    __syncthreads();
    const int per_iter = blockDim.x / H, head = threadIdx.x % H;
    const int chunk = (S + gridDim.x - 1) / gridDim.x;
    const int s_end = min(S, (int)(blockIdx.x + 1) * chunk);
    for (int s = blockIdx.x * chunk + threadIdx.x / H; s < s_end; s += per_iter) {
        const uint16_t *qp = q + p * q_sb + (long long)s * q_st + head * 16;
        const uint4 r0 = *(const uint4 *)qp, r1 = *(const uint4 *)(qp + 8);
        float x[16];
        const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
        float sc[16], mx = -3.0e38f;                             // fully unrolled over 16 slots: sc[] stays in registers
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            sc[t] = -3.0e38f;
            if (t < T) {
                const float4 *kt = (const float4 *)(kv + t * ci + head * 16);       // 4 x ds_read_b128 (the kernel was LDS-issue bound
                float a = 0.f;                                                       // on 16 scalar reads per key)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float4 kk = kt[d];
                    a = fmaf(x[4 * d], kk.x, fmaf(x[4 * d + 1], kk.y, fmaf(x[4 * d + 2], kk.z, fmaf(x[4 * d + 3], kk.w, a))));
                }
                sc[t] = a;
                mx = fmaxf(mx, a);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) { sc[t] = t < T ? __expf(sc[t] - mx) : 0.f; den += sc[t]; }
        const float inv = 1.0f / den;
        float out[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) out[d] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < T) {
                const float4 *vt = (const float4 *)(kv + (T + t) * ci + head * 16);
                const float pt = sc[t] * inv;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float4 vv = vt[d];
                    out[4 * d] = fmaf(pt, vv.x, out[4 * d]); out[4 * d + 1] = fmaf(pt, vv.y, out[4 * d + 1]);
                    out[4 * d + 2] = fmaf(pt, vv.z, out[4 * d + 2]); out[4 * d + 3] = fmaf(pt, vv.w, out[4 * d + 3]);
                }
            }
        }
        uint16_t *op = o + (p * S + s) * ci + head * 16;
        const uint2 a = pack4(out[0], out[1], out[2], out[3]), b = pack4(out[4], out[5], out[6], out[7]);
        const uint2 c = pack4(out[8], out[9], out[10], out[11]), e = pack4(out[12], out[13], out[14], out[15]);
        *(uint4 *)op = make_uint4(a.x, a.y, b.x, b.y);
        *(uint4 *)(op + 8) = make_uint4(c.x, c.y, e.x, e.y);
    }
}

// ---- automatic mask generator: low-res logits -> full-resolution statistics / binary masks ----
// value of the H x W bilinear upsampling (torch upsample_bilinear2d, align_corners = False) of one h x w logit map
__device__ __forceinline__ float up_sample(const float *__restrict__ m, int h, int w, float sy, float sx, int Y, int X) {
    float fy = sy * ((float)Y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = sx * ((float)X + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    return (1.f - ly) * ((1.f - lx) * m[y0 * w + x0] + lx * m[y0 * w + x1]) + ly * ((1.f - lx) * m[y1 * w + x0] + lx * m[y1 * w + x1]);
}

__global__ void k_amg_init(int32_t *__restrict__ stats, int n, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t *s = stats + 7 * i;
    s[0] = s[1] = s[2] = 0; s[3] = W; s[4] = H; s[5] = -1; s[6] = -1;
}

#define AMG_RB 16
// stats[i] = {#(v > thr + off), #(v > thr - off), #(v > thr), x_min, y_min, x_max, y_max of (v > thr)} over the H x W upsampling
__global__ void __launch_bounds__(256) k_amg_stats(const float *__restrict__ logits, int h, int w, int H, int W, float thr, float off,
                                                   int32_t *__restrict__ stats) {
    // A block owns a band of AMG_RB output rows of one candidate and stages the few logit rows they interpolate in LDS: the
    // 4 taps of a sample then come from LDS instead of 4 dependent global loads (0.95 -> see DESIGN.md; same arithmetic as
    // up_sample / k_amg_binarize, so the statistics and the binary masks agree bit for bit).
    extern __shared__ __attribute__((aligned(16))) float rows_s[];
    const float *m = logits + (long long)blockIdx.y * h * w;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int Y0 = blockIdx.x * AMG_RB, Y1 = min(H, Y0 + AMG_RB);
    float f0 = sy * ((float)Y0 + 0.5f) - 0.5f; f0 = f0 < 0.f ? 0.f : f0;
    float f1 = sy * ((float)(Y1 - 1) + 0.5f) - 0.5f; f1 = f1 < 0.f ? 0.f : f1;
    const int r0 = (int)f0, r1 = min((int)f1 + 1, h - 1);
    for (int i = threadIdx.x; i < (r1 - r0 + 1) * w; i += blockDim.x) rows_s[i] = m[r0 * w + i];
    __syncthreads();
    const float *ms = rows_s - r0 * w;                           // ms[y * w + x] == m[y * w + x] for the staged rows
    int hi = 0, lo = 0, ar = 0, x0 = W, y0 = H, x1 = -1, y1 = -1;
    for (int i = threadIdx.x; i < (Y1 - Y0) * W; i += blockDim.x) {
        const int Y = Y0 + i / W, X = i % W;
        const float v = up_sample(ms, h, w, sy, sx, Y, X);
        hi += v > thr + off; lo += v > thr - off;
        if (v > thr) { ++ar; x0 = X < x0 ? X : x0; x1 = X > x1 ? X : x1; y0 = Y < y0 ? Y : y0; y1 = Y > y1 ? Y : y1; }
    }
    __shared__ int sh[7];
    if (threadIdx.x == 0) { sh[0] = sh[1] = sh[2] = 0; sh[3] = W; sh[4] = H; sh[5] = -1; sh[6] = -1; }
    __syncthreads();
    if (lo) { atomicAdd(&sh[0], hi); atomicAdd(&sh[1], lo); }
    if (ar) { atomicAdd(&sh[2], ar); atomicMin(&sh[3], x0); atomicMin(&sh[4], y0); atomicMax(&sh[5], x1); atomicMax(&sh[6], y1); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *s = stats + 7 * blockIdx.y;
        if (sh[1]) { atomicAdd(&s[0], sh[0]); atomicAdd(&s[1], sh[1]); }
        if (sh[2]) { atomicAdd(&s[2], sh[2]); atomicMin(&s[3], sh[3]); atomicMin(&s[4], sh[4]); atomicMax(&s[5], sh[5]); atomicMax(&s[6], sh[6]); }
    }
}

__global__ void __launch_bounds__(256) k_amg_binarize(const float *__restrict__ logits, const int32_t *__restrict__ sel, int h, int w, int H, int W,
                                                      float thr, uint8_t *__restrict__ out) {
    const float *m = logits + (long long)sel[blockIdx.y] * h * w;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    uint8_t *o = out + (long long)blockIdx.y * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int Y = i / W, X = i - Y * W;
        o[i] = up_sample(m, h, w, sy, sx, Y, X) > thr;
    }
}

// seg[p] = first mask (in the given order) that covers pixel p, -1 if none: mask2segmap's "most stable mask wins" painting
__global__ void __launch_bounds__(256) k_paint_segmap(const uint8_t *__restrict__ masks, int n, long long pixels, int32_t *__restrict__ seg) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (long long)gridDim.x * blockDim.x) {
        int id = -1;
        for (int i = 0; i < n; ++i)
            if (masks[i * pixels + p]) { id = i; break; }
        seg[p] = id;
    }
}

}  // namespace

extern "C" int ovo_sam_i2t_attention(const void *q, int64_t q_batch_stride, int q_token_stride, const void *k, const void *v, void *o, int64_t P, int S,
                                     int T, int H, float scale, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && P <= 65535 && S > 0 && T > 0 && T <= 16 && H > 0 && 256 % H == 0, "T <= 16 tokens, H a divisor of 256");
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(q && k && v && o && q_batch_stride % 8 == 0 && q_token_stride % 8 == 0 && q_token_stride >= 16 * H, "null / misaligned argument");
    int bx = (S + 511) / 512;
    k_i2t_attention<<<dim3(bx, (unsigned)P), 256, (size_t)2 * T * 16 * H * sizeof(float), (hipStream_t)stream>>>(
        (const uint16_t *)q, q_batch_stride, q_token_stride, (const uint16_t *)k, (const uint16_t *)v, (uint16_t *)o, S, T, H, scale);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_paint_segmap(const uint8_t *masks, int n, int64_t pixels, int32_t *seg, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && pixels > 0 && seg && (n == 0 || masks), "bad argument");
    k_paint_segmap<<<ovo_grid(pixels, 256), 256, 0, (hipStream_t)stream>>>(masks, n, pixels, seg);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_amg_mask_stats(const float *logits, int n, int h, int w, int H, int W, float thr, float offset, int32_t *stats,
                                  ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && n <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "bad shape");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(logits && stats, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    k_amg_init<<<(n + 255) / 256, 256, 0, st>>>(stats, n, H, W);
    const int src_rows = (int)((float)AMG_RB * (float)h / (float)H) + 3;      // logit rows a band of AMG_RB output rows can touch
    const size_t lds = (size_t)src_rows * w * sizeof(float);
    OVO_REQUIRE(lds <= 64 * 1024, "logit rows of one output band exceed 64 KiB of LDS (down-sampling by a large factor is not supported)");
    k_amg_stats<<<dim3((H + AMG_RB - 1) / AMG_RB, n), 256, lds, st>>>(logits, h, w, H, W, thr, offset, stats);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_amg_binarize(const float *logits, const int32_t *sel, int n_sel, int h, int w, int H, int W, float thr, uint8_t *out,
                                ovo_stream_t stream) {
    OVO_REQUIRE(n_sel >= 0 && n_sel <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "bad shape");
    if (n_sel == 0) return OVO_OK;
    OVO_REQUIRE(logits && sel && out, "null pointer");
    int bx = (H * W + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : bx;
    k_amg_binarize<<<dim3(bx, n_sel), 256, 0, (hipStream_t)stream>>>(logits, sel, h, w, H, W, thr, out);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_row_epilogue(const float *x, int64_t R, int C, const float *base, int64_t base_rows, const float *gamma,
                                const float *beta, float eps, const float *pe, int64_t pe_rows, float *y, void *y16, void *ype16,
                                ovo_stream_t stream) {
    OVO_REQUIRE(R >= 0 && C > 0 && C % 4 == 0 && C <= 2048, "C must be a multiple of 4, <= 2048");
    if (R == 0) return OVO_OK;
    OVO_REQUIRE(x && (y || y16 || ype16), "null pointer");
    OVO_REQUIRE((gamma == nullptr) == (beta == nullptr), "gamma and beta go together");
    OVO_REQUIRE((!base || base_rows > 0) && (!ype16 || (pe && pe_rows > 0)), "broadcast sources need their row counts");
    RowArgs a;
    a.x = x; a.base = base; a.base_rows = base_rows; a.gamma = gamma; a.beta = beta; a.eps = eps; a.pe = pe; a.pe_rows = pe_rows;
    a.y = y; a.y16 = (uint16_t *)y16; a.ype16 = (uint16_t *)ype16; a.R = R; a.C = C;
    if (C <= 1024) k_row_epilogue<4><<<ovo_grid(R, 4, 256 * 16), 256, 0, (hipStream_t)stream>>>(a);
    else k_row_epilogue<8><<<ovo_grid(R, 4, 256 * 16), 256, 0, (hipStream_t)stream>>>(a);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_upscale_ln(const void *g, const float *bias, const float *feat, const float *gamma, const float *beta, float eps,
                                  int64_t P, int s, int C1, void *out, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && s > 0, "bad shape");
    OVO_REQUIRE(C1 == 16 || C1 == 32 || C1 == 64 || C1 == 128 || C1 == 256, "C1 must be 16, 32, 64, 128 or 256");
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(g && bias && feat && gamma && beta && out, "null pointer");
    const long long total = P * 4LL * s * s;
    hipStream_t st = (hipStream_t)stream;
#define GO(L) k_upscale_ln<L><<<(unsigned)((total + 256 / L - 1) / (256 / L)), 256, 0, st>>>((const uint16_t *)g, bias, feat, gamma, beta, eps, P, s, (uint16_t *)out)
    if (C1 == 16) GO(4); else if (C1 == 32) GO(8); else if (C1 == 64) GO(16); else if (C1 == 128) GO(32); else GO(64);
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_upscale_masks(const void *g, const float *bias, const float *feat, const float *hyper, int n_mask, int first,
                                     int64_t P, int s2, int C2, float *out, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && P <= 65535 && s2 > 0 && C2 > 0 && C2 % 8 == 0 && C2 <= 256, "C2 must be a multiple of 8, <= 256");
    OVO_REQUIRE(n_mask > 0 && first >= 0 && first < n_mask && n_mask - first <= 4, "at most 4 mask tokens");
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(g && bias && feat && hyper && out, "null pointer");
    const int n_out = n_mask - first, side = 2 * s2;
    dim3 grid((side * side + 255) / 256, (unsigned)P);
    k_upscale_masks<<<grid, 256, (size_t)(n_out + 1) * C2 * sizeof(float), (hipStream_t)stream>>>((const uint16_t *)g, bias, feat, hyper, n_mask,
                                                                                                  first, n_out, s2, C2, out);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
