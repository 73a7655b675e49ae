// hiera.hip -- SAM2 image encoder forward (Hiera trunk + FPN neck) as one C call.
//
// Tokens live in an fp32 residual stream x[B, H, W, C] (NHWC).  Per block:
//   k_ln_window   LayerNorm + window partition (+ zero rows for the padding windows, zero K-padding columns)
//   ovo_gemm      QKV (bf16, MFMA)                         [rows, 3*dim_out]
//   k_qpool       2x2 max-pool of q inside each window     (stage-change blocks only)
//   ovo_attention per-window (or global) fused attention, windows folded into the batch dimension
//   ovo_gemm      output projection -> fp32 rows in window order
//   (epilogue of that GEMM, ovo_gemm_unwindow)  window order -> spatial order, + residual (the pooled projected skip at stage changes)
//   k_ln_window(identity) -> FC1 GEMM(+GELU) -> FC2 GEMM(+bias, += x)
// All GEMM operands have K padded to a multiple of 64 with zeros (dims 112 / 224 of hiera_b+, 144 / 288 of hiera_l; the 7x7x3 patch: 192).
#include "gemm_common.h"

namespace {

__device__ __forceinline__ uint16_t f2bf(float x) {      // v_cvt_pk_bf16_f32 (RNE)
    const __bf16 h = (__bf16)x;
    return *(const uint16_t *)&h;
}
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

struct Grid {           // window partition geometry of an [B, H, W] token grid
    int B, H, W, ws;    // ws == 0: one window = whole image (global attention)
    int nwh, nww, wh, ww;   // windows per image (rows, cols) and window height / width
    long long rows;     // B * nwh * nww * wh * ww (including padding rows)
};
__host__ __device__ inline Grid make_grid(int B, int H, int W, int ws) {
    Grid g; g.B = B; g.H = H; g.W = W; g.ws = ws;
    if (ws > 0) { g.wh = g.ww = ws; g.nwh = (H + ws - 1) / ws; g.nww = (W + ws - 1) / ws; }
    else { g.wh = H; g.ww = W; g.nwh = g.nww = 1; }
    g.rows = (long long)B * g.nwh * g.nww * g.wh * g.ww;
    return g;
}
// window-order row of spatial token (b, y, x)
__device__ __forceinline__ long long row_of(const Grid &g, int b, int y, int x) {
    const int wy = y / g.wh, wx = x / g.ww;
    return ((((long long)b * g.nwh + wy) * g.nww + wx) * g.wh + (y - wy * g.wh)) * g.ww + (x - wx * g.ww);
}

// LayerNorm of x[b, y, x, :d] written as bf16 row `r` (window order) of width kp; padding rows / columns = 0.
// LPR lanes cooperate on one row (16 for the narrow early stages: 4 rows per wave, 64 for wide rows); d % 4 == 0,
// kp % 4 == 0; float4 reads, 8-byte bf16 writes.
template <int LPR, int NV>
__global__ void __launch_bounds__(256) k_ln_window(const float *__restrict__ x, Grid g, int d, int kp, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float eps, uint16_t *__restrict__ out) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63, sl = lane % LPR, sub = lane / LPR;
    const uint32_t waves = gridDim.x * 4u;
    const int d4 = d >> 2, kp4 = kp >> 2;
    const uint32_t per_win = (uint32_t)(g.wh * g.ww), n_rows = (uint32_t)g.rows;       // (the launch checks rows < 2^31)
    const uint32_t n_iter = (n_rows + RPW - 1) / RPW;
    for (uint32_t it = blockIdx.x * 4u + (threadIdx.x >> 6); it < n_iter; it += waves) {
        // row -> (image, window, position): four 32-bit divisions, on the SCALAR unit when the whole wave works on one row.  (The 64-bit
        // long-long form of these seven quotients / remainders was ~700 VALU instructions per row: the kernel issued more than it loaded)
        uint32_t r = it * RPW + sub;
        if (LPR == 64) r = __builtin_amdgcn_readfirstlane(r);
        bool real = r < n_rows;
        const float *src = x;
        if (real) {
            const uint32_t win = r / per_win, in = r - win * per_win;
            const uint32_t ly = in / (uint32_t)g.ww, lx = in - ly * (uint32_t)g.ww;
            const uint32_t q2 = win / (uint32_t)g.nww, wx = win - q2 * (uint32_t)g.nww;
            const uint32_t b = q2 / (uint32_t)g.nwh, wy = q2 - b * (uint32_t)g.nwh;
            const int y = (int)(wy * g.wh + ly), xx = (int)(wx * g.ww + lx);
            if (y >= g.H || xx >= g.W) {                // padding row of a partial window: zeros
                uint2 *o = (uint2 *)(out + (long long)r * kp);
                for (int i = sl; i < kp4; i += LPR) o[i] = make_uint2(0, 0);
                real = false;
            } else src = x + (((long long)b * g.H + y) * g.W + xx) * d;
        }
        // the row is read ONCE into registers (NV float4 per lane, d <= 4 LPR NV); sums and output walk them in the order the three-pass form
        // walked memory (same bits).  Three dependent rounds of loads per row moved 2.9 TB/s on stage 3's 58 800 x 448 rows
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = sl + LPR * j;
            v[j] = (real && i < d4) ? ((const float4 *)src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (real && sl + LPR * j < d4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (real && sl + LPR * j < d4) {
                const float a = v[j].x - mean, b2 = v[j].y - mean, c2 = v[j].z - mean, e2 = v[j].w - mean;
                q += (a * a + b2 * b2) + (c2 * c2 + e2 * e2);
            }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q / (float)d + eps);
        if (!real) continue;
        uint2 *o = (uint2 *)(out + (long long)r * kp);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = sl + LPR * j;
            if (i < d4) {
                const float4 gm = ((const float4 *)gamma)[i], bt = ((const float4 *)beta)[i];
                const uint32_t lo = (uint32_t)f2bf((v[j].x - mean) * rstd * gm.x + bt.x) | ((uint32_t)f2bf((v[j].y - mean) * rstd * gm.y + bt.y) << 16);
                const uint32_t hi = (uint32_t)f2bf((v[j].z - mean) * rstd * gm.z + bt.z) | ((uint32_t)f2bf((v[j].w - mean) * rstd * gm.w + bt.w) << 16);
                o[i] = make_uint2(lo, hi);
            }
        }
        for (int i = d4 + sl; i < kp4; i += LPR) o[i] = make_uint2(0, 0);      // K-padding columns
    }
}

void launch_ln_window(const float *x, const Grid &g, int d, int kp, const float *gamma, const float *beta, float eps, uint16_t *out,
                      hipStream_t hs) {
    const int d4 = d >> 2;
    if (d <= 128) k_ln_window<16, 2><<<ovo_grid(g.rows * 16, 256), 256, 0, hs>>>(x, g, d, kp, gamma, beta, eps, out);
    else if (d <= 256) k_ln_window<16, 4><<<ovo_grid(g.rows * 16, 256), 256, 0, hs>>>(x, g, d, kp, gamma, beta, eps, out);
    else if (d4 <= 128) k_ln_window<64, 2><<<ovo_grid(g.rows * 64, 256), 256, 0, hs>>>(x, g, d, kp, gamma, beta, eps, out);
    else if (d4 <= 256) k_ln_window<64, 4><<<ovo_grid(g.rows * 64, 256), 256, 0, hs>>>(x, g, d, kp, gamma, beta, eps, out);
    else k_ln_window<64, 8><<<ovo_grid(g.rows * 64, 256), 256, 0, hs>>>(x, g, d, kp, gamma, beta, eps, out);
}

// q of a packed qkv buffer [rows, 3*C] (window order, window wh x ww) -> pooled q [rows/4, C]: 2x2 max.
// Eight channels per thread (16-byte loads and stores) and 32-bit index arithmetic when C % 8 == 0 and the pooled tensor has < 2^32 elements; the
// one-element-per-thread form with its three 64-bit divisions per 2-byte output (kept for odd widths) took 46-85 us on stage 1's 786 432 x 112 q.
__global__ void __launch_bounds__(256) k_qpool8(const uint16_t *__restrict__ qkv, uint32_t total8, int wh, int ww, int C, uint16_t *__restrict__ qp) {
    const uint32_t oh = wh / 2, ow = ww / 2, c8n = C / 8;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += gridDim.x * blockDim.x) {
        const uint32_t t1 = i / c8n, c = (i - t1 * c8n) * 8;
        const uint32_t t2 = t1 / ow, ox = t1 - t2 * ow;
        const uint32_t win = t2 / oh, oy = t2 - win * oh;
        const uint16_t *base = qkv + ((long long)win * wh * ww) * 3 * C + c;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const uint4 v = *(const uint4 *)(base + ((long long)(2 * oy + dy) * ww + (2 * ox + dx)) * 3 * C);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], __uint_as_float(w[e] << 16));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], __uint_as_float(w[e] & 0xffff0000u));
                }
            }
        uint4 o;                                                 // (a maximum of bf16 values is one of them: the upper halves are the result)
        o.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u);
        o.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u);
        o.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xffff0000u);
        o.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xffff0000u);
        *(uint4 *)(qp + (long long)t1 * C + c) = o;
    }
}
__global__ void __launch_bounds__(256) k_qpool(const uint16_t *__restrict__ qkv, long long n_windows, int wh, int ww, int C,
                                               uint16_t *__restrict__ qp) {
    const int oh = wh / 2, ow = ww / 2;
    const long long total = n_windows * oh * ow * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ox = (int)(t % ow); t /= ow;
        const int oy = (int)(t % oh);
        const long long win = t / oh;
        const uint16_t *base = qkv + (win * wh * ww) * 3 * C + c;
        float m = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) m = fmaxf(m, bf2f(base[((long long)(2 * oy + dy) * ww + (2 * ox + dx)) * 3 * C]));
        qp[i] = f2bf(m);
    }
}

// skip path at a stage change: out[b, y2, x2, :] = max over the 2x2 block of rows[row_of(b, 2y2+dy, 2x2+dx), :]
__global__ void __launch_bounds__(256) k_pool_unwindow(const float *__restrict__ rows, Grid g, int C, float *__restrict__ out) {
    const int oh = g.H / 2, ow = g.W / 2;
    const long long total = (long long)g.B * oh * ow * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int x = (int)(t % ow); t /= ow;
        const int y = (int)(t % oh);
        const int b = (int)(t / oh);
        float m = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) m = fmaxf(m, rows[row_of(g, b, 2 * y + dy, 2 * x + dx) * C + c]);
        out[i] = m;
    }
}

// f32 [rows, C] -> bf16 [rows, kp] with zero padding columns
__global__ void __launch_bounds__(256) k_cast_pad(const float *__restrict__ x, long long rows, int C, int kp, uint16_t *__restrict__ y) {
    if (C % 4 == 0 && kp % 4 == 0) {                        // four columns per thread: 16-byte loads, 8-byte stores
        const int k4 = kp >> 2;
        const long long total = rows * k4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % k4) * 4;
            const long long r = i / k4;
            uint2 p = make_uint2(0u, 0u);
            if (c < C) {
                const float4 v = *(const float4 *)(x + r * C + c);
                p.x = f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
                p.y = f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
            }
            *(uint2 *)(y + r * kp + c) = p;
        }
        return;
    }
    const long long total = rows * kp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % kp);
        y[i] = c < C ? f2bf(x[(i / kp) * C + c]) : (uint16_t)0;
    }
}

// fine[b, y, x, :] += coarse[b, y/2, x/2, :]   (nearest 2x top-down)
__global__ void __launch_bounds__(256) k_topdown_add(float *__restrict__ fine, const float *__restrict__ coarse, int B, int H, int W, int C) {
    const int c4 = C >> 2;
    const long long total = (long long)B * H * W * c4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        long long t = i / c4;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const float4 u = ((const float4 *)coarse)[(((long long)b * (H / 2) + y / 2) * (W / 2) + x / 2) * c4 + c];
        float4 v = ((float4 *)fine)[i];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        ((float4 *)fine)[i] = v;
    }
}

inline int padk(int v) { return (v + 63) / 64 * 64; }   // K of every GEMM operand: a multiple of 64 keeps them on the 8-wave BK = 64 kernels
                                                         // ((16384,1344,224 -> 256): 34.8 -> 29.1 us although 14 % of the products are zeros)
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Plan {
    int n_blocks;
    int dim_in[64], dim_out[64], heads[64], ws[64], pool[64], Hin[64];
    int stage_end[64];        // stage index if the block closes a stage, else -1
};

int make_plan(const ovo_hiera_config_t &c, Plan &p) {
    int idx = 0, H = c.image_size / 4;
    for (int s = 0; s < 4; ++s)
        for (int b = 0; b < c.blocks[s]; ++b) {
            if (idx >= 64) return -1;
            const bool first = s > 0 && b == 0;
            p.dim_in[idx] = first ? c.dims[s - 1] : c.dims[s];
            p.dim_out[idx] = c.dims[s];
            p.heads[idx] = c.heads[s];
            int ws = first ? c.window[s - 1] : c.window[s];
            for (int k = 0; k < c.n_global; ++k) if (c.global_blocks[k] == idx) ws = 0;
            p.ws[idx] = ws;
            p.pool[idx] = first;
            p.Hin[idx] = H;
            if (first) H /= 2;
            p.stage_end[idx] = b == c.blocks[s] - 1 ? s : -1;
            ++idx;
        }
    p.n_blocks = idx;
    return 0;
}

struct Ws {
    uint16_t *col; float *x, *xr, *tmp; uint16_t *h, *qkv, *qp, *att, *u, *cast; float *lat[4];
    size_t bytes;
};

Ws carve(const ovo_hiera_config_t &c, const Plan &p, int B, void *base) {
    size_t max_x = 0, max_h = 0, max_qkv = 0, max_att = 0, max_u = 0, max_tmp = 0, max_cast = 0;
    for (int i = 0; i < p.n_blocks; ++i) {
        const int H = p.Hin[i], Ho = p.pool[i] ? H / 2 : H;
        const Grid g = make_grid(B, H, H, p.ws[i]);
        const size_t tok_in = (size_t)B * H * H, tok_out = (size_t)B * Ho * Ho;
        const size_t rows_out = p.pool[i] ? (size_t)g.rows / 4 : (size_t)g.rows;
        max_x = max_x > tok_in * p.dim_in[i] ? max_x : tok_in * p.dim_in[i];
        max_x = max_x > tok_out * p.dim_out[i] ? max_x : tok_out * p.dim_out[i];
        size_t v = (size_t)g.rows * padk(p.dim_in[i]); max_h = max_h > v ? max_h : v;
        v = tok_out * padk(p.dim_out[i]); max_h = max_h > v ? max_h : v;
        v = (size_t)g.rows * 3 * p.dim_out[i]; max_qkv = max_qkv > v ? max_qkv : v;
        v = rows_out * padk(p.dim_out[i]); max_att = max_att > v ? max_att : v;
        v = tok_out * 4 * p.dim_out[i]; max_u = max_u > v ? max_u : v;
        v = (size_t)g.rows * p.dim_out[i]; max_tmp = max_tmp > v ? max_tmp : v;
        v = tok_out * padk(p.dim_out[i]); max_cast = max_cast > v ? max_cast : v;
    }
    const size_t T0 = (size_t)B * (c.image_size / 4) * (c.image_size / 4);
    max_cast = max_cast > T0 * c.fpn_dim ? max_cast : T0 * c.fpn_dim;
    Ws w;
    char *b0 = (char *)base;
    size_t off = 0;
    auto take = [&](size_t n) { char *r = b0 ? b0 + off : nullptr; off += align256(n); return r; };
    w.col = (uint16_t *)take(T0 * 192 * 2);
    w.x = (float *)take(max_x * 4);
    w.xr = (float *)take(max_x * 4);
    w.tmp = (float *)take(max_tmp * 4);
    w.h = (uint16_t *)take(max_h * 2);
    w.qkv = (uint16_t *)take(max_qkv * 2);
    w.qp = (uint16_t *)take(max_att * 2);
    w.att = (uint16_t *)take(max_att * 2);
    w.u = (uint16_t *)take(max_u * 2);
    w.cast = (uint16_t *)take(max_cast * 2);
    for (int s = 0; s < 4; ++s) { const size_t t = T0 >> (2 * s); w.lat[s] = (float *)take(t * c.fpn_dim * 4); }
    w.bytes = off;
    return w;
}

int gemm(const void *A, long long lda, const void *W, long long ldw, const float *bias, void *C, long long ldc, int out_dtype,
         const float *add, long long ld_add, long long M, int N, int K, int act, ovo_stream_t s) {
    ovo_gemm_t g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.add = add; g.ld_add = ld_add;
    g.M = (int)M; g.N = N; g.K = K; g.in_dtype = 2; g.out_dtype = out_dtype; g.act = act; g.alpha = 1.0f;
    return ovo_gemm(&g, s);
}

// C = act(LN-or-cast(x rows) . W^T + bias) through the streaming GEMM's f32 A path when it covers the shape; otherwise the two-pass form:
// normalise / cast into `h` (window order when g.ws > 0), then ovo_gemm.  `h_done`: a previous call already filled `h` for this x.
int gemm_from_f32(const float *x, const Grid &g, int d, int kp, const float *gamma, const float *beta, float eps, int mode, uint16_t *h, bool &h_done,
                  const void *W, const float *bias, void *C, long long ldc, int out_dtype, int N, int act, ovo_stream_t s) {
    ovo_gemm_t q;
    q.A = h; q.lda = kp; q.W = W; q.ldw = kp; q.bias = bias; q.C = C; q.ldc = ldc; q.add = nullptr; q.ld_add = 0;
    q.M = (int)g.rows; q.N = N; q.K = kp; q.in_dtype = 2; q.out_dtype = out_dtype; q.act = act; q.alpha = 1.0f;
    if (!h_done) {
        const ovo_window_t w = {g.B, g.H, g.W, g.wh, g.ww};
        const int rc = ovo_gemm_detail::gemm_f32a_stream(&q, g.ws > 0 ? &w : nullptr, x, d, gamma, beta, eps, mode, 0, s);
        if (rc != OVO_E_UNSUPPORTED) return rc;
        OVO_REQUIRE(d <= 2048 && g.rows < (1ll << 31), "LayerNorm rows of more than 2048 columns / more than 2^31 rows");
        if (mode == 1) launch_ln_window(x, g, d, kp, gamma, beta, eps, h, (hipStream_t)s);
        else k_cast_pad<<<ovo_grid(g.rows * kp, 256), 256, 0, (hipStream_t)s>>>(x, g.rows, d, kp, h);
        h_done = true;
    }
    return ovo_gemm(&q, s);
}

// ---- patch embedding as a direct 7 x 7 / stride-4 convolution (round 5) ----
// x[b, (oy, ox), :] = conv7x7s4p3(image[b]) + bias + pos[(oy, ox), :], f32, straight from the f32 image: no im2col matrix (302 MB written and read back
// per 12 frames at 1024^2) and no per-image GEMM launches.  A workgroup (8 waves, one output row each) walks over tiles of 8 x 32 output positions: a tile's 35 x 131 x 3 input
// pixels go to LDS as bf16 (the rounding the im2col pass applied; float4 loads from the 16-byte aligned column 3 on), the weights [E, 192] once
// per workgroup, re-ordered while they are copied so that a K-step of 32 is four (channel, ky) rows of 8 columns each -- the pixel left of the
// window (a zero weight) and the 7 of kx: a lane's 8 consecutive k of the MFMA's B operand are then 8 consecutive pixels of one input row
// starting at an 8-byte aligned LDS address -- two ds_read_b64 -- and of its A operand 16 bytes of a weight row.  out^T = W . patches^T, so a
// lane ends up with 4 consecutive channels of one token: bias + position embedding + store are float4.
// Bound by the 4 E bytes per token it writes (352 MB per 12 frames of hiera_b+).
constexpr int PE_TH = 8, PE_TW = 32, PE_IH = 4 * PE_TH + 3, PE_RS = 136, PE_WS = 200;
template <int NT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_patch_embed7(const float *__restrict__ img, int S, const uint16_t *__restrict__ w, int ldw,
                                                      const float *__restrict__ bias, const float *__restrict__ pos, float *__restrict__ out, int B,
                                                      int n_tiles) {
    using namespace ovo_gemm_detail;
    constexpr int E = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char pe_lds[];
    uint16_t *s_img = (uint16_t *)pe_lds;                               // [3 * PE_IH][PE_RS]: input column x of the tile sits at index x - x_first + 1
    uint16_t *s_w = s_img + 3 * PE_IH * PE_RS;                          // [E][PE_WS]: k' = (channel * 7 + ky) * 8 + (kx + 1)
    const int S4 = S >> 2, tiles_x = S4 / PE_TW, n_pos = tiles_x * (S4 / PE_TH);
    for (int i = threadIdx.x; i < E * 24; i += 512) {                   // one (channel, ky) row of one output channel per thread: 7 weights
        const int e = i / 24, r = i - 24 * e;
        uint4 p = make_uint4(0u, 0u, 0u, 0u);
        if (r < 21) {
            const uint16_t *q = w + (long long)e * ldw + 7 * r;
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6];
            p.x = w0 << 16; p.y = w1 | (w2 << 16); p.z = w3 | (w4 << 16); p.w = w5 | (w6 << 16);
        }
        *(uint4 *)(s_w + e * PE_WS + 8 * r) = p;
    }
    for (int i = threadIdx.x; i < 3 * PE_IH; i += 512) s_img[i * PE_RS] = 0;     // index 0 of every row: multiplied by the zero weight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l16 = lane & 15;
    constexpr int NV = (3 * PE_IH * 32 + 511) / 512;
    float4 v[NV];
    float h;
    // a tile's pixels: columns 3 .. 130 of each row are 32 aligned float4, columns 0 .. 2 (left halo) scalars -- into registers, one tile ahead
    auto load_tile = [&](int t) {                                       // (branch-free: clamped addresses, then a select -- seven loads in flight)
        const int tt = t % n_pos, b = t / n_pos;
        const int oy0 = (tt / tiles_x) * PE_TH, ox0 = (tt % tiles_x) * PE_TW;
        const int y_first = 4 * oy0 - 3, x_first = 4 * ox0 - 3;
        const float *src = img + (long long)b * 3 * S * S;
        const float *col0 = src + 4 * ox0 + 4 * (threadIdx.x & 31);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int row = (threadIdx.x >> 5) + 16 * j;
            if (row > 3 * PE_IH - 1) row = 3 * PE_IH - 1;
            const int c = row / PE_IH, y = y_first + (row - c * PE_IH);
            const int yc = y < 0 ? 0 : (y > S - 1 ? S - 1 : y);
            const float4 u = *(const float4 *)(col0 + (c * S + yc) * S);
            const bool ok = y == yc;
            v[j].x = ok ? u.x : 0.f; v[j].y = ok ? u.y : 0.f; v[j].z = ok ? u.z : 0.f; v[j].w = ok ? u.w : 0.f;
        }
        {
            int row = threadIdx.x / 3;
            const int col = threadIdx.x - 3 * row;
            if (row > 3 * PE_IH - 1) row = 3 * PE_IH - 1;
            const int c = row / PE_IH, y = y_first + (row - c * PE_IH), x = x_first + col;
            const int yc = y < 0 ? 0 : (y > S - 1 ? S - 1 : y), xc = x < 0 ? 0 : x;
            const float u = src[(c * S + yc) * S + xc];
            h = (y == yc && x == xc) ? u : 0.f;
        }
    };
    if ((int)blockIdx.x < n_tiles) load_tile(blockIdx.x);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        // task t = (image, tile position), position fastest: a workgroup's tasks are gridDim.x apart, so with the grid a multiple of the positions
        // per image it keeps ONE position and walks through the images -- the tile's position-embedding rows come from its own L1 / L2 after the first
        // (image-fastest, the twelve readers of a position sat on different XCDs: PMC fetch 591 MB per launch for 151 MB of pixels + 29 MB of table;
        // the launch time did not move, 193 -> 191 us: it is bound by its 352 MB of stores)
        const int tt = t % n_pos, b = t / n_pos;
        const int oy0 = (tt / tiles_x) * PE_TH, ox0 = (tt % tiles_x) * PE_TW;
        __syncthreads();                                                // the previous tile's products have read s_img
        // accumulators start at bias + position embedding: the loads fly while the tile is written to LDS
        // lane: channels 16 n + 4 g .. + 3 of token (oy0 + wave, ox0 + 16 m + l16): a wave owns one output row of the tile
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long long tok = (long long)(oy0 + wave) * S4 + ox0 + 16 * m + l16;
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = *(const f32x4 *)(pos + tok * E + 16 * n + 4 * g);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int row = (threadIdx.x >> 5) + 16 * j;
            if (16 * j + 15 < 3 * PE_IH || row < 3 * PE_IH) {
                uint2 p;
                p.x = f2bf(v[j].x) | ((uint32_t)f2bf(v[j].y) << 16); p.y = f2bf(v[j].z) | ((uint32_t)f2bf(v[j].w) << 16);
                *(uint2 *)(s_img + row * PE_RS + 4 + 4 * (threadIdx.x & 31)) = p;
            }
        }
        {
            const int i = threadIdx.x, row = i / 3, col = i - 3 * row;
            if (row < 3 * PE_IH) s_img[row * PE_RS + 1 + col] = f2bf(h);
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f32x4 bb = *(const f32x4 *)(bias + 16 * n + 4 * g);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] += bb;
        }
        if (t + (int)gridDim.x < n_tiles) load_tile(t + gridDim.x);    // the next tile's pixels fly under this tile's products and stores
#pragma unroll 1                                                       // (unrolled, the scheduler hoists all 6 steps' fragments)
        for (int s = 0; s < 6; ++s) {
            int r = 4 * s + g;
            if (r > 20) r = 20;                                         // rows 21-23: zero weights, any finite pixels
            const int c = r / 7, ky = r - 7 * c;
            bf16x8 pb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = wave, ox = 16 * m + l16;
                const uint2 *q = (const uint2 *)(s_img + (c * PE_IH + 4 * oy + ky) * PE_RS + 4 * ox);
                const uint2 lo = q[0], hi = q[1];
                uint4 u; u.x = lo.x; u.y = lo.y; u.z = hi.x; u.w = hi.y;
                pb[m] = *(const bf16x8 *)&u;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 wa = *(const bf16x8 *)(s_w + (16 * n + l16) * PE_WS + 32 * s + 8 * g);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m][n] = Mfma<bf16x8>::run(wa, pb[m], acc[m][n]);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long long tok = (long long)(oy0 + wave) * S4 + ox0 + 16 * m + l16;
            float *dst = out + ((long long)b * S4 * S4 + tok) * E;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float4 o;
                o.x = acc[m][n][0]; o.y = acc[m][n][1]; o.z = acc[m][n][2]; o.w = acc[m][n][3];
                *(float4 *)(dst + 16 * n + 4 * g) = o;
            }
        }
    }
}

template <int NT>
int launch_patch_embed7(const float *img, int S, const void *w, int ldw, const float *bias, const float *pos, float *out, int B, hipStream_t s) {
    const size_t lds = (size_t)(3 * PE_IH * PE_RS + 16 * NT * PE_WS) * 2;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_patch_embed7<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ovo_set_error("ovo_hiera_forward: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        set = true;
    }
    const int S4 = S / 4, n_tiles = B * (S4 / PE_TH) * (S4 / PE_TW);
    const bool prof = ovo_prof_enabled();
    if (prof) {
        const double tok = (double)B * S4 * S4;
        ovo_prof_begin(8, 2.0 * tok * 147.0 * 16 * NT, s); ovo_prof_shape((int)tok, 16 * NT, 147); ovo_prof_flags(1 | 2 | 256);
        ovo_prof_bytes(12.0 * B * S * S + 4.0 * 16 * NT * (tok + (double)S4 * S4));
    }
    k_patch_embed7<NT><<<n_tiles < 512 ? n_tiles : 512, 512, lds, s>>>(img, S, (const uint16_t *)w, ldw, bias, pos, out, B, n_tiles);
    if (prof) ovo_prof_end(s);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// E = 96 (hiera_t / s), 112 (b+), 144 (l); anything else (or a grid that does not tile by 8 x 32) keeps the im2col + GEMM form
int patch_embed_direct(const float *img, int S, int E, const void *w, int ldw, const float *bias, const float *pos, float *out, int B, hipStream_t s) {
    static const bool off_once = getenv("OVO_HIERA_PATCH_GEMM") != nullptr;               // measurement / tests: the im2col + GEMM form
    if (ovo_knobs_dynamic() ? getenv("OVO_HIERA_PATCH_GEMM") != nullptr : off_once) return OVO_E_UNSUPPORTED;
    if (S % 128 != 0 || ldw < 147) return OVO_E_UNSUPPORTED;
    switch (E) {
        case 96: return launch_patch_embed7<6>(img, S, w, ldw, bias, pos, out, B, s);
        case 112: return launch_patch_embed7<7>(img, S, w, ldw, bias, pos, out, B, s);
        case 144: return launch_patch_embed7<9>(img, S, w, ldw, bias, pos, out, B, s);
        default: return OVO_E_UNSUPPORTED;
    }
}

}  // namespace

#define TRY(call)                        \
    do {                                 \
        const int rc__ = (call);         \
        if (rc__ != OVO_OK) return rc__; \
    } while (0)
#define LAUNCHED() OVO_CHECK_LAUNCH()

extern "C" {

size_t ovo_hiera_workspace_bytes(const ovo_hiera_config_t *cfg, int B) {
    if (!cfg || B <= 0) return 0;
    Plan p;
    if (make_plan(*cfg, p) != 0) return 0;
    return carve(*cfg, p, B, nullptr).bytes;
}

int ovo_hiera_patch_embed(const float *images, int B, int S, int E, const void *patch_w, int ldw, const float *bias, const float *pos,
                          float *x, ovo_stream_t stream) {
    OVO_REQUIRE(images && patch_w && bias && pos && x && B > 0 && S > 0 && E > 0, "bad argument");
    OVO_REQUIRE((long long)B * (S / 4) * (S / 4) < (1ll << 31) / 256, "too many tokens for one launch");
    return patch_embed_direct(images, S, E, patch_w, ldw, bias, pos, x, B, (hipStream_t)stream);
}

int ovo_hiera_forward(const ovo_hiera_config_t *cfg, const ovo_hiera_weights_t *w, const float *images, int B, float *feat0,
                      float *feat1, float *feat2, void *ws, size_t ws_bytes, ovo_stream_t stream) {
    OVO_REQUIRE(cfg && w && images && feat0 && feat1 && feat2 && ws && B > 0, "null argument");
    const ovo_hiera_config_t &c = *cfg;
    OVO_REQUIRE(c.image_size % 32 == 0 && c.fpn_dim % 32 == 0 && c.n_global >= 0 && c.n_global <= 8, "bad config");
    for (int s = 0; s < 4; ++s)
        OVO_REQUIRE(c.dims[s] % 8 == 0 && c.heads[s] > 0 && c.dims[s] % c.heads[s] == 0 && (c.dims[s] / c.heads[s]) % 8 == 0 &&
                        c.blocks[s] > 0 && c.window[s] > 0, "bad stage config");
    Plan p;
    OVO_REQUIRE(make_plan(c, p) == 0, "too many blocks");
    OVO_REQUIRE(w->patch_w && w->patch_b && w->pos && w->blocks, "missing weights");
    Ws k = carve(c, p, B, ws);
    OVO_REQUIRE(ws_bytes >= k.bytes, "workspace too small");
    hipStream_t hs = (hipStream_t)stream;
    const int S4 = c.image_size / 4;
    const long long T0 = (long long)S4 * S4;

    // patch embedding (+ position embedding through the GEMM epilogue), one image at a time (pos has no batch dim)
    const int pe_rc = patch_embed_direct(images, c.image_size, c.dims[0], w->patch_w, 192, w->patch_b, w->pos, k.x, B, hs);
    if (pe_rc == OVO_E_UNSUPPORTED) {
        TRY(ovo_im2col(images, B, 3, c.image_size, c.image_size, 7, 4, 3, k.col, 192, stream));
        for (int b = 0; b < B; ++b)
            TRY(gemm(k.col + (size_t)b * T0 * 192, 192, w->patch_w, 192, w->patch_b, k.x + (size_t)b * T0 * c.dims[0], c.dims[0], 0, w->pos,
                     c.dims[0], T0, c.dims[0], 192, 0, stream));
    } else
        TRY(pe_rc);

    float *x = k.x, *spare = k.xr;
    long long att_rows = -1;
    int att_kout = 0;
    for (int i = 0; i < p.n_blocks; ++i) {
        const ovo_hiera_block_t &L = w->blocks[i];
        const int din = p.dim_in[i], dout = p.dim_out[i], H = p.Hin[i], Ho = p.pool[i] ? H / 2 : H;
        const int kin = padk(din), kout = padk(dout), hd = dout / p.heads[i];
        const Grid g = make_grid(B, H, H, p.ws[i]);
        OVO_REQUIRE(!p.pool[i] || (g.wh % 2 == 0 && H % 2 == 0), "query pooling needs even windows");
        const long long tok_out = (long long)B * Ho * Ho;

        // LayerNorm 1 happens inside the A-operand load of its consumers where the streaming GEMM runs them (stages 1-2), else as a pass into k.h
        bool h_done = false;
        const float *residual = x;
        if (din != dout) {                                   // skip = maxpool(proj(LN(x)))
            OVO_REQUIRE(L.res_w && L.res_b && p.pool[i], "stage-change block without projection weights");
            // projection + 2 x 2 max-pool in one launch where the streaming GEMM covers it (pooled rows straight into `spare`) ...
            ovo_gemm_t q;
            q.A = nullptr; q.lda = kin; q.W = L.res_w; q.ldw = kin; q.bias = L.res_b; q.C = spare; q.ldc = dout; q.add = nullptr; q.ld_add = 0;
            q.M = (int)g.rows; q.N = dout; q.K = kin; q.in_dtype = 2; q.out_dtype = 0; q.act = 0; q.alpha = 1.0f;
            const ovo_window_t wd = {g.B, g.H, g.W, g.wh, g.ww};
            const int rc = g.ws > 0 ? ovo_gemm_detail::gemm_f32a_stream(&q, &wd, x, din, L.ln1_g, L.ln1_b, c.ln_eps, 1, 1, stream) : OVO_E_UNSUPPORTED;
            if (rc == OVO_E_UNSUPPORTED) {                   // ... else every token's projection into k.tmp, then the pool pass
                TRY(gemm_from_f32(x, g, din, kin, L.ln1_g, L.ln1_b, c.ln_eps, 1, k.h, h_done, L.res_w, L.res_b, k.tmp, dout, 0, dout, 0, stream));
                k_pool_unwindow<<<ovo_grid(tok_out * dout, 256), 256, 0, hs>>>(k.tmp, g, dout, spare);
            } else if (rc != OVO_OK) {
                return rc;
            }
            residual = spare;
        }
        const long long n_win = (long long)B * g.nwh * g.nww;
        const int tk = g.wh * g.ww, tq = p.pool[i] ? tk / 4 : tk;
        // K-padding columns of the attention output must be zero; the attention kernels only write the real ones, so they STAY zero from block to
        // block while rows x row width do not change (the blocks of one stage): one fill per layout instead of one per block (3 of 5 for hiera_b+)
        if (kout != dout && (att_rows != (long long)n_win * tq || att_kout != kout)) {
            OVO_HIP(hipMemsetAsync(k.att, 0, (size_t)n_win * tq * kout * 2, hs));
            att_rows = (long long)n_win * tq; att_kout = kout;
        } else if (kout == dout) {
            att_rows = -1;                                       // every column written: the next padded layout starts from a fresh fill
        }
        // stage 1 of hiera_b+ and the stage-change block after it (round 5): LayerNorm -> QKV (-> 2 x 2 pool of q) -> window attention in one pass over x
        // per pair of heads, q | k | v never written (winattn.hip)
        int fused_attn = OVO_E_UNSUPPORTED;
        if (g.ws > 0 && !h_done && cfg->q_prescaled)
            fused_attn = ovo_gemm_detail::win_attn_launch(x, B, H, H, g.ws, din, dout, p.heads[i], p.pool[i] ? 1 : 0, L.ln1_g, L.ln1_b, c.ln_eps, L.qkv_w, kin,
                                                          L.qkv_b, k.att, kout, hs);
        if (fused_attn != OVO_OK && fused_attn != OVO_E_UNSUPPORTED) return fused_attn;
        if (fused_attn == OVO_E_UNSUPPORTED) {
        // QKV; at a stage change the q columns are pooled 2 x 2 in the product's epilogue where the streaming GEMM runs it (else k_qpool below)
        bool q_pooled = false;
        if (p.pool[i] && g.ws > 0 && !h_done) {
            ovo_gemm_t q;
            q.A = nullptr; q.lda = kin; q.W = L.qkv_w; q.ldw = kin; q.bias = L.qkv_b; q.C = k.qkv; q.ldc = 3 * dout; q.add = nullptr; q.ld_add = 0;
            q.M = (int)g.rows; q.N = 3 * dout; q.K = kin; q.in_dtype = 2; q.out_dtype = 2; q.act = 0; q.alpha = 1.0f;
            const ovo_window_t wd = {g.B, g.H, g.W, g.wh, g.ww};
            const int rc = ovo_gemm_detail::gemm_f32a_stream(&q, &wd, x, din, L.ln1_g, L.ln1_b, c.ln_eps, 1, 0, stream, k.qp, dout);
            if (rc == OVO_OK) q_pooled = true;
            else if (rc != OVO_E_UNSUPPORTED) return rc;
        }
        if (!q_pooled)
            TRY(gemm_from_f32(x, g, din, kin, L.ln1_g, L.ln1_b, c.ln_eps, 1, k.h, h_done, L.qkv_w, L.qkv_b, k.qkv, 3 * dout, 2, 3 * dout, 0, stream));
        ovo_attention_t a = {};
        a.k = k.qkv + dout; a.v = k.qkv + 2 * dout; a.o = k.att;
        a.k_sb = a.v_sb = (int64_t)tk * 3 * dout; a.k_sh = a.v_sh = hd; a.k_st = a.v_st = 3 * dout;
        if (p.pool[i]) {
            if (!q_pooled) {
                const long long total8 = n_win * tq * (dout / 8);
                if (dout % 8 == 0 && total8 < (1ll << 32) - (1ll << 22) && (((uintptr_t)k.qkv | (uintptr_t)k.qp) & 15) == 0)
                    k_qpool8<<<ovo_grid(total8, 256, 256 * 16), 256, 0, hs>>>(k.qkv, (uint32_t)total8, g.wh, g.ww, dout, k.qp);
                else k_qpool<<<ovo_grid(n_win * tq * dout, 256), 256, 0, hs>>>(k.qkv, n_win, g.wh, g.ww, dout, k.qp);
            }
            a.q = k.qp; a.q_sb = (int64_t)tq * dout; a.q_sh = hd; a.q_st = dout;
        } else {
            a.q = k.qkv; a.q_sb = a.k_sb; a.q_sh = hd; a.q_st = 3 * dout;
        }
        a.o_sb = (int64_t)tq * kout; a.o_sh = hd; a.o_st = kout;
        a.B = (int)n_win; a.H = p.heads[i]; a.Tq = tq; a.Tk = tk; a.hd = hd; a.scale = cfg->q_prescaled ? 0.0f : 1.0f / sqrtf((float)hd);
        TRY(ovo_attention(&a, stream));
        }
        // output projection; its epilogue also takes the rows from window order (pooled window size) back to spatial order and
        // adds the residual (ovo_gemm_unwindow; k_unwindow_add was a pass of its own).  In place: same-dim blocks add onto x; at a
        // stage change the old x is dead (only LN1 read it) and the pooled skip lives in `spare`, so the smaller new stream is
        // written over the old buffer.
        const Grid go = make_grid(B, Ho, Ho, p.ws[i] > 0 ? (p.pool[i] ? p.ws[i] / 2 : p.ws[i]) : 0);
        bool ln2_done = false;
        {
            ovo_gemm_t og;
            og.A = k.att; og.lda = kout; og.W = L.out_w; og.ldw = kout; og.bias = L.out_b; og.C = x; og.ldc = dout; og.add = residual; og.ld_add = dout;
            og.M = (int)(n_win * tq); og.N = dout; og.K = kout; og.in_dtype = 2; og.out_dtype = 0; og.act = 0; og.alpha = 1.0f;
            const ovo_window_t ow = {go.B, go.H, go.W, go.wh, go.ww};
            // stage 3 (N = 448), OPT-IN (OVO_HIERA_PROJ_LN=1): a full-row 128 x 448 tile, norm2 of the result rows from its accumulators straight into k.h, the
            // MLP below then skips its LayerNorm pass.  MEASURED AND LEFT OFF (tools/rowln_bench.py, profiles/r05c_rowln_bench.txt): 87 us against 64 (128 x 64
            // tiles) + 28 (LayerNorm pass) alone, but 25 us per block SLOWER inside the forward (15.97 vs 15.57 ms per 12 frames): one 512-thread workgroup
            // per CU with a two-stage ring of 72 KB K-tiles does not overlap its neighbours the way the small tiles and the copy-rate LayerNorm pass do.
            static const bool rowln_once = getenv("OVO_HIERA_PROJ_LN") != nullptr;
            const bool rowln = ovo_knobs_dynamic() ? getenv("OVO_HIERA_PROJ_LN") != nullptr : rowln_once;
            int rc = OVO_E_UNSUPPORTED;
            if (rowln && dout == 448 && kout == dout)
                rc = ovo_gemm_detail::gemm_unwindow_rowln(&og, &ow, L.ln2_g, L.ln2_b, c.ln_eps, k.h, kout, stream);
            if (rc == OVO_OK) ln2_done = true;
            else if (rc != OVO_E_UNSUPPORTED) return rc;
            else TRY(ovo_gemm_unwindow(&og, &ow, stream));
        }
        // MLP
        const Grid gi = make_grid(B, Ho, Ho, 0);
        h_done = false;
        // Row chunks (round 5): the hidden activations of stages 1-2 (12 frames: 786 432 x 448 and 196 608 x 896 bf16 = 704 / 352 MB) were written by
        // FC1 and read back by FC2 through HBM.  The MLP is row-wise, so FC1 / FC2 alternate over chunks of rows whose hidden block fits the 256 MB
        // Infinity Cache with room to spare, every chunk through the SAME hidden buffer: FC2 reads what FC1 just left in the cache and the next
        // chunk overwrites those lines before they are written back.  MEASURED AND LEFT OFF (bench.py, one box, 12-frame groups): one pass 409.4
        // frames/s, 96 MB chunks 404.0, 48 MB 398.2, 24 MB 387.1 -- the write stream of the FC1 chunks is not absorbed by the cache (the launches
        // stay write-bound) and every extra launch pair adds its ramp.  OVO_HIERA_MLP_CHUNK_MB = chunk size (default 0 = one pass over all rows).
        static const long long chunk_mb = getenv("OVO_HIERA_MLP_CHUNK_MB") ? atoll(getenv("OVO_HIERA_MLP_CHUNK_MB")) : 0;
        const long long hid_row = (long long)4 * dout * 2;
        long long chunk_rows = chunk_mb > 0 ? (chunk_mb << 20) / hid_row / 4096 * 4096 : 0;
        if (chunk_rows <= 0 || tok_out < 2 * chunk_rows || tok_out * hid_row <= (200ll << 20)) chunk_rows = tok_out;   // (a hidden block the cache holds anyway: one pass)
        // stages 1-2 (hidden width 4 dout, tall streams): LayerNorm -> FC1 -> GELU -> FC2 -> + residual in one pass, the hidden row in registers (mlp_stream.hip)
        const int fused = ovo_gemm_detail::mlp_stream_launch(x, tok_out, dout, L.ln2_g, L.ln2_b, c.ln_eps, L.fc1_w, kout, L.fc1_b, 4 * dout, L.fc2_w, 4 * dout, L.fc2_b, hs);
        if (fused != OVO_OK && fused != OVO_E_UNSUPPORTED) return fused;
        for (long long r0 = fused == OVO_OK ? tok_out : 0; r0 < tok_out; r0 += chunk_rows) {
            const long long nr = tok_out - r0 < chunk_rows ? tok_out - r0 : chunk_rows;
            const Grid gc = chunk_rows == tok_out ? gi : make_grid(1, (int)nr, 1, 0);
            bool h_ready = chunk_rows == tok_out ? (h_done || ln2_done) : false;      // (a chunked pass re-normalises its rows: ln2_done is for the one-pass form only)
            float *xc = x + r0 * dout;
            TRY(gemm_from_f32(xc, gc, dout, kout, L.ln2_g, L.ln2_b, c.ln_eps, 1, k.h, h_ready, L.fc1_w, L.fc1_b, k.u, 4 * dout, 2, 4 * dout, 1, stream));
            TRY(gemm(k.u, 4 * dout, L.fc2_w, 4 * dout, L.fc2_b, xc, dout, 0, xc, dout, nr, dout, 4 * dout, 0, stream));
        }
        h_done = false;
        LAUNCHED();

        if (p.stage_end[i] >= 0) {                           // FPN lateral 1x1 conv of this stage's output
            const int s = p.stage_end[i];
            OVO_REQUIRE(w->neck_w[s] && w->neck_b[s], "missing neck weights");
            bool cast_done = false;
            TRY(gemm_from_f32(x, gi, dout, kout, nullptr, nullptr, 0.f, 2, k.cast, cast_done, w->neck_w[s], w->neck_b[s], k.lat[s], c.fpn_dim, 0,
                              c.fpn_dim, 0, stream));
        }
    }
    // top-down on the coarse levels: level 2 += up(level 3); levels 0 and 1 are laterals only
    const int S16 = c.image_size / 16;
    k_topdown_add<<<ovo_grid((long long)B * S16 * S16 * (c.fpn_dim / 4), 256), 256, 0, hs>>>(k.lat[2], k.lat[3], B, S16, S16, c.fpn_dim);
    LAUNCHED();
    OVO_HIP(hipMemcpyAsync(feat2, k.lat[2], (size_t)B * S16 * S16 * c.fpn_dim * 4, hipMemcpyDeviceToDevice, hs));
    const long long t0 = (long long)B * T0, t1 = t0 / 4;
    if (c.hi_res) {
        OVO_REQUIRE(w->s0_w && w->s0_b && w->s1_w && w->s1_b, "missing conv_s0 / conv_s1 weights");
        bool cast_done = false;
        const Grid g0 = make_grid(B, S4, S4, 0), g1 = make_grid(B, S4 / 2, S4 / 2, 0);
        TRY(gemm_from_f32(k.lat[0], g0, c.fpn_dim, c.fpn_dim, nullptr, nullptr, 0.f, 2, k.cast, cast_done, w->s0_w, w->s0_b, feat0, 32, 0, 32, 0, stream));
        cast_done = false;
        TRY(gemm_from_f32(k.lat[1], g1, c.fpn_dim, c.fpn_dim, nullptr, nullptr, 0.f, 2, k.cast, cast_done, w->s1_w, w->s1_b, feat1, 64, 0, 64, 0, stream));
    } else {
        OVO_HIP(hipMemcpyAsync(feat0, k.lat[0], (size_t)t0 * c.fpn_dim * 4, hipMemcpyDeviceToDevice, hs));
        OVO_HIP(hipMemcpyAsync(feat1, k.lat[1], (size_t)t1 * c.fpn_dim * 4, hipMemcpyDeviceToDevice, hs));
    }
    LAUNCHED();
    return OVO_OK;
}

}  // extern "C"
