// samfuse.hip -- the "skinny" matrix products of the SAM2 mask decoder fused with the row-wise passes that follow them.
//
// At 256 clicks the image side of the two-way transformer is 1 M rows (click x pixel) of 256 channels: every product over it is an HBM
// stream (K = 64..256, N = 128..256: 30-130 flops per byte moved), and the unfused chain wrote each product to HBM only for a
// row pass (residual + LayerNorm, pixel shuffle + LayerNorm2d + GELU, GELU + hyper-network dot) to read it back.  Here the weight matrix
// (16-128 KB) stays in LDS for the life of a workgroup, a wave multiplies 16 rows x ALL N columns on MFMA
// (v_mfma_f32_16x16x32_bf16, activation fragments straight from global memory: a row is 128-512 contiguous bytes), and the row pass runs
// on the accumulators: lane (fr, fq) holds row fr, columns 16 j + 4 fq + r -- a row sits in the 4 lanes that share fr, so row
// statistics are an in-lane sum and two xor-shuffles.  Nothing but the final tensors touches HBM:
//   * k_proj_ln     : y = LayerNorm(res + A . W^T + bias)  -> f32 / bf16 / bf16(+ positional code)   (attention out-projection + norm4)
//   * k_up1_ln      : ConvTranspose2d(2x2, s2) + skip + LayerNorm2d + GELU -> bf16 [P, 2s, 2s, C1]   (output_upscaling.0-2)
//   * k_up2_masks   : ConvTranspose2d(2x2, s2) + skip + GELU + hyper-network product -> mask logits  (output_upscaling.3-4 + hypernets)
// Reference path: sam2 MaskDecoder.predict_masks / TwoWayAttentionBlock, reached at segment_utils.py:291-308, mask_generator.py:113.
#include <stdlib.h>

#include "skinny.h"

namespace {

using ovo_gemm_detail::bf16x8;
using ovo_gemm_detail::f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ uint32_t pack2(float a, float b) {     // v_cvt_pk_bf16_f32 (RNE)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return *(const uint32_t *)&h;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {                // erf GELU, polynomial erf of gemm_common.h (|error| <= 8.6e-6), packed f32
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
__device__ __forceinline__ float quad_sum(float v) {             // over the 4 lanes that hold one row (same lane & 15)
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

using ovo_skinny::Skinny;
using ovo_skinny::store_pair16;

// ---- y = LayerNorm(res[m % res_rows] + A . W^T + bias) -------------------------------------------------------------------------
struct ProjLnArgs {
    const uint16_t *A, *W;
    const float *bias, *res; const uint16_t *res16; int res_rows;   // residual: f32 `res` or bf16 `res16` (at most one)
    const float *gamma, *beta; float eps;
    const float *pe; int pe_rows;
    float *y32; uint16_t *y16, *ype16;
    int M;
};
template <int K, int N>
__global__ void __launch_bounds__(512, 4) k_proj_ln(ProjLnArgs a) {
    using S = Skinny<K, N>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *gs = (float *)(smem + S::W_BYTES), *bs = gs + N, *cs = bs + N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    S::load_w(smem, a.W, K, tid, 512);
    for (int i = tid; i < N; i += 512) { gs[i] = a.gamma[i]; bs[i] = a.beta[i]; cs[i] = a.bias ? a.bias[i] : 0.f; }
    __syncthreads();
    const int blocks = (a.M + 15) / 16;
    const float *gl = gs + fq * 4, *bl = bs + fq * 4, *cl = cs + fq * 4;     // lane base + constant: the column offset rides in the ds_read offset field
    for (int b = blockIdx.x * 8 + wave; b < blocks; b += gridDim.x * 8) {
        const int m = b * 16 + fr, mc = m < a.M ? m : a.M - 1;
        bf16x8 af[S::KS];
        S::load_a(af, a.A, K, mc, fq);
        f32x4 acc[S::NT];
        const long long rrow = (a.res || a.res16) ? (long long)(mc % a.res_rows) * N : 0;
        const float *rp = a.res ? a.res + rrow : nullptr;
        const uint16_t *rp16 = a.res16 ? a.res16 + rrow : nullptr;
#pragma unroll
        for (int j = 0; j < S::NT; ++j) {                        // the residual and the bias seed the accumulators
            const int c = j * 16 + fq * 4;
            f32x4 v = *(const f32x4 *)(cl + j * 16);
            if (rp) v += *(const f32x4 *)(rp + c);
            if (rp16) {
                const uint2 r = *(const uint2 *)(rp16 + c);
                v += f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
            }
            acc[j] = v;
            if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // bounds the loads in flight (registers)
        }
        S::mma(acc, af, smem, fr, fq);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < S::NT; ++j) sum += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
        const float mean = quad_sum(sum) * (1.0f / N);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < S::NT; ++j) {
            acc[j] -= mean;
            sq += (acc[j][0] * acc[j][0] + acc[j][1] * acc[j][1]) + (acc[j][2] * acc[j][2] + acc[j][3] * acc[j][3]);
        }
        const float rstd = rsqrtf(quad_sum(sq) * (1.0f / N) + a.eps);
        if (m >= a.M) continue;
        const float *pp = a.ype16 ? a.pe + (long long)(m % a.pe_rows) * N : nullptr;
        static_assert(S::NT % 2 == 0, "column tiles are stored in pairs");
#pragma unroll
        for (int j = 0; j < S::NT; j += 2) {
            uint2 o16[2], pe16[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = (j + h) * 16 + fq * 4;
                const f32x4 v = acc[j + h] * rstd * *(const f32x4 *)(gl + (j + h) * 16) + *(const f32x4 *)(bl + (j + h) * 16);
                if (a.y32) *(f32x4 *)(a.y32 + (long long)m * N + c) = v;
                o16[h] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                if (pp) {
                    const f32x4 p = v + *(const f32x4 *)(pp + c);
                    pe16[h] = make_uint2(pack2(p[0], p[1]), pack2(p[2], p[3]));
                }
            }
            if (a.y16) store_pair16(a.y16 + (long long)m * N + j * 16, fq, o16[0], o16[1]);
            if (pp) store_pair16(a.ype16 + (long long)m * N + j * 16, fq, pe16[0], pe16[1]);
            if (j % 4 == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- upscaling stage 1: A bf16 [P s s, K] (row = prompt, y, x) . W^T [4 C1, K] (column = (dy 2 + dx) C1 + c) ------------------
// out bf16 [P, 2s, 2s, C1] = GELU(LayerNorm2d_c(product + bias[c] + feat[2y + dy, 2x + dx, c]))
struct Up1Args {
    const uint16_t *A, *W;
    const float *bias, *feat, *gamma, *beta; float eps;
    int P, s;
    uint16_t *out;
    int lut;                    // 1: GELU through the LDS table (gemm_common.h: gelu_lut; round 5), 0: the packed polynomial (OVO_SAM_GELU_POLY)
};
template <int K, int C1, int NTHREADS>
__global__ void __launch_bounds__(NTHREADS) k_up1_ln(Up1Args a) {
    constexpr int N = 4 * C1, JPG = C1 / 16;                     // j-tiles per (dy, dx) group
    using S = Skinny<K, N>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *gs = (float *)(smem + S::W_BYTES), *bs = gs + C1, *cs = bs + C1;
    const float2 *lut = a.lut ? (const float2 *)(cs + C1) : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    if (lut) ovo_gemm_detail::gelu_lut_fill((float2 *)lut, tid, NTHREADS);
    S::load_w(smem, a.W, K, tid, NTHREADS);
    for (int i = tid; i < C1; i += NTHREADS) { gs[i] = a.gamma[i]; bs[i] = a.beta[i]; cs[i] = a.bias[i]; }
    __syncthreads();
    const int ss = a.s * a.s, M = a.P * ss, blocks = (M + 15) / 16, side = 2 * a.s;
    const float *gl = gs + fq * 4, *bl = bs + fq * 4, *cl = cs + fq * 4;
    for (int b = blockIdx.x * (NTHREADS / 64) + wave; b < blocks; b += gridDim.x * (NTHREADS / 64)) {
        const int m = b * 16 + fr, mc = m < M ? m : M - 1;
        bf16x8 af[S::KS];
        S::load_a(af, a.A, K, mc, fq);
        const int p = mc / ss, rem = mc - p * ss, y = rem / a.s, x = rem - y * a.s;
        f32x4 acc[S::NT];
#pragma unroll
        for (int j = 0; j < S::NT; ++j) {                        // bias + skip feature seed the accumulators
            const int g = j / JPG, c = (j % JPG) * 16 + fq * 4;
            const long long pix = (long long)(2 * y + (g >> 1)) * side + 2 * x + (g & 1);
            acc[j] = *(const f32x4 *)(cl + (j % JPG) * 16) + *(const f32x4 *)(a.feat + pix * C1 + c);
            if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        S::mma(acc, af, smem, fr, fq);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int jj = 0; jj < JPG; ++jj) { const f32x4 v = acc[g * JPG + jj]; sum += (v[0] + v[1]) + (v[2] + v[3]); }
            const float mean = quad_sum(sum) * (1.0f / C1);
            float sq = 0.f;
#pragma unroll
            for (int jj = 0; jj < JPG; ++jj) {
                f32x4 &v = acc[g * JPG + jj];
                v -= mean;
                sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            const float rstd = rsqrtf(quad_sum(sq) * (1.0f / C1) + a.eps);
            if (m >= M) continue;
            const long long pix = ((long long)p * side + 2 * y + (g >> 1)) * side + 2 * x + (g & 1);
            static_assert(JPG % 2 == 0, "column tiles are stored in pairs");
#pragma unroll
            for (int jj = 0; jj < JPG; jj += 2) {
                uint2 o16[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = acc[g * JPG + jj + h] * rstd * *(const f32x4 *)(gl + (jj + h) * 16) + *(const f32x4 *)(bl + (jj + h) * 16);
                    f32x2 lo, hi;
                    if (lut) { lo = f32x2{ovo_gemm_detail::gelu_lut(v[0], lut), ovo_gemm_detail::gelu_lut(v[1], lut)};
                               hi = f32x2{ovo_gemm_detail::gelu_lut(v[2], lut), ovo_gemm_detail::gelu_lut(v[3], lut)}; }
                    else { lo = gelu2(f32x2{v[0], v[1]}); hi = gelu2(f32x2{v[2], v[3]}); }
                    o16[h] = make_uint2(pack2(lo.x, lo.y), pack2(hi.x, hi.y));
                }
                store_pair16(a.out + pix * C1 + jj * 16, fq, o16[0], o16[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- upscaling stage 2 + hyper-network product: A bf16 [P s2 s2, K] . W^T [4 C2, K]  ->  masks f32 [P, n_out, 2 s2, 2 s2] -------
// masks[p, i, 2y + dy, 2x + dx] = sum_c hyper[p, first + i, c] * GELU(product[(dy, dx), c] + bias[c] + feat[2y + dy, 2x + dx, c])
struct Up2Args {
    const uint16_t *A, *W;
    const float *bias, *feat, *hyper;
    int n_mask, first, n_out, P, s2;
    float *out;
    int lut;                    // as Up1Args
};
template <int K, int C2>
__global__ void __launch_bounds__(512, 4) k_up2_masks(Up2Args a) {
    constexpr int N = 4 * C2, JPG = C2 / 16;
    using S = Skinny<K, N>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *cs = (float *)(smem + S::W_BYTES);
    const float2 *lut = a.lut ? (const float2 *)(cs + C2) : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    if (lut) ovo_gemm_detail::gelu_lut_fill((float2 *)lut, tid, 512);
    S::load_w(smem, a.W, K, tid, 512);
    for (int i = tid; i < C2; i += 512) cs[i] = a.bias[i];
    __syncthreads();
    const int ss = a.s2 * a.s2, M = a.P * ss, blocks = (M + 15) / 16, side = 2 * a.s2;
    const float *cl = cs + fq * 4;
    for (int b = blockIdx.x * 8 + wave; b < blocks; b += gridDim.x * 8) {
        const int m = b * 16 + fr, mc = m < M ? m : M - 1;
        bf16x8 af[S::KS];
        S::load_a(af, a.A, K, mc, fq);
        const int p = mc / ss, rem = mc - p * ss, y = rem / a.s2, x = rem - y * a.s2;
        f32x4 acc[S::NT];
#pragma unroll
        for (int j = 0; j < S::NT; ++j) {
            const int g = j / JPG, c = (j % JPG) * 16 + fq * 4;
            const long long pix = (long long)(2 * y + (g >> 1)) * side + 2 * x + (g & 1);
            acc[j] = *(const f32x4 *)(cl + (j % JPG) * 16) + *(const f32x4 *)(a.feat + pix * C2 + c);
        }
        S::mma(acc, af, smem, fr, fq);
        // part[g][i]: this lane's share (its 4 JPG channels) of output (dy, dx) = g, mask i
        float part[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[g][i] = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPG; ++jj) {
            const int c = jj * 16 + fq * 4;
            f32x4 h[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                h[i] = i < a.n_out ? *(const f32x4 *)(a.hyper + ((long long)p * a.n_mask + a.first + i) * C2 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = acc[g * JPG + jj];
                f32x2 lo, hi;
                if (lut) { lo = f32x2{ovo_gemm_detail::gelu_lut(v[0], lut), ovo_gemm_detail::gelu_lut(v[1], lut)};
                           hi = f32x2{ovo_gemm_detail::gelu_lut(v[2], lut), ovo_gemm_detail::gelu_lut(v[3], lut)}; }
                else { lo = gelu2(f32x2{v[0], v[1]}); hi = gelu2(f32x2{v[2], v[3]}); }
#pragma unroll
                for (int i = 0; i < 4; ++i) part[g][i] += (h[i][0] * lo.x + h[i][1] * lo.y) + (h[i][2] * hi.x + h[i][3] * hi.y);
            }
        }
        // reduce over the row's 4 lanes so that lane fq ends with the totals of (dy, dx) = fq: 8 + 4 exchanges instead of 32
        float half[2][4], mine[4];
        const bool up = (fq & 2) != 0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = up ? part[k][i] : part[2 + k][i], keep = up ? part[2 + k][i] : part[k][i];
                half[k][i] = keep + __shfl_xor(send, 32, 64);
            }
        const bool odd = (fq & 1) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = odd ? half[0][i] : half[1][i], keep = odd ? half[1][i] : half[0][i];
            mine[i] = keep + __shfl_xor(send, 16, 64);
        }
        if (m >= M) continue;
        const int Y = 2 * y + (fq >> 1), X = 2 * x + (fq & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < a.n_out) a.out[(((long long)p * a.n_out + i) * side + Y) * side + X] = mine[i];
    }
}


// ---- token -> image cross attention of the two-way transformer: T token queries x S image keys per prompt, head_dim 16 ----------
// The generic flash kernel pads the 8 query rows to a 64-row tile and head_dim 16 to 64 and ran at half the HBM rate of its K / V
// stream (2 MB per prompt).  Here one workgroup owns a prompt; lane = (key parity, head, token): a lane walks its wave's share of the
// keys with the 16-wide query in registers -- a wave instruction reads the 16 H-channel K (V) row of 64 / (H T) keys, each 32-byte
// head segment shared by the T token lanes -- keeping a running (max, sum, 16 outputs); the waves' partial states are merged through
// LDS.  K and V may be column blocks of one wider matrix (row stride kv_st): the fused K|V|Q projection is consumed in place.
struct T2iArgs {
    const uint16_t *q, *k, *v; long long kv_sb; int kv_st;
    uint16_t *o;
    int S, T, H; float scale;
};
__device__ __forceinline__ void unpack8(const uint4 r, f32x2 (&x)[4]) {
    x[0] = f32x2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    x[1] = f32x2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    x[2] = f32x2{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u)};
    x[3] = f32x2{__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_t2i_attention(T2iArgs a) {
    extern __shared__ __attribute__((aligned(16))) float part[];  // [NW][64][18]: max, sum, 16 outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HT = a.H * a.T, G = 64 / HT, ci = 16 * a.H;
    const int sub = lane / HT, ht = lane % HT, h = ht / a.T, t = ht % a.T;
    const long long p = blockIdx.x;
    f32x2 qv[8];
    {
        const uint16_t *qp = a.q + (p * a.T + t) * ci + h * 16;
        f32x2 lo[4], hi[4];
        unpack8(*(const uint4 *)qp, lo); unpack8(*(const uint4 *)(qp + 8), hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) { qv[i] = lo[i] * a.scale; qv[4 + i] = hi[i] * a.scale; }
    }
    const int chunk = (a.S + NW - 1) / NW, s0 = wave * chunk, s1 = min(a.S, s0 + chunk);
    const uint16_t *kp = a.k + p * a.kv_sb + h * 16, *vp = a.v + p * a.kv_sb + h * 16;
    float mx = -3.0e38f, den = 0.f;
    f32x2 out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = f32x2{0.f, 0.f};
    constexpr int U = 4;                                          // keys per softmax update (one rescale of the running state per U keys)
    for (int s = s0 + sub; s < s1; s += U * G) {
        uint4 kr[U][2], vr[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sk = min(s + u * G, a.S - 1);               // clamped: masked below
            const uint16_t *kk = kp + (long long)sk * a.kv_st, *vv = vp + (long long)sk * a.kv_st;
            kr[u][0] = *(const uint4 *)kk; kr[u][1] = *(const uint4 *)(kk + 8);
            vr[u][0] = *(const uint4 *)vv; vr[u][1] = *(const uint4 *)(vv + 8);
        }
        float sc[U], mloc = mx;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x2 x[4], y[4];
            unpack8(kr[u][0], x); unpack8(kr[u][1], y);
            f32x2 d = x[0] * qv[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) d = __builtin_elementwise_fma(x[i], qv[i], d);
#pragma unroll
            for (int i = 0; i < 4; ++i) d = __builtin_elementwise_fma(y[i], qv[4 + i], d);
            sc[u] = s + u * G < s1 ? d.x + d.y : -3.0e38f;
            mloc = fmaxf(mloc, sc[u]);
        }
        const float corr = __expf(mx - mloc);
        mx = mloc;
        den *= corr;
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] *= corr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pe = s + u * G < s1 ? __expf(sc[u] - mx) : 0.f;
            den += pe;
            f32x2 x[4], y[4];
            unpack8(vr[u][0], x); unpack8(vr[u][1], y);
#pragma unroll
            for (int i = 0; i < 4; ++i) { out[i] = __builtin_elementwise_fma(x[i], (f32x2)(pe), out[i]); out[4 + i] = __builtin_elementwise_fma(y[i], (f32x2)(pe), out[4 + i]); }
        }
    }
    float *mp = part + (wave * 64 + lane) * 18;
    mp[0] = mx; mp[1] = den;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mp[2 + 2 * i] = out[i].x; mp[3 + 2 * i] = out[i].y; }
    __syncthreads();
    if (tid < HT * 16) {                                          // thread = (head-token, output channel): merge NW x G partial states
        const int e = tid / 16, d = tid % 16;
        float M = -3.0e38f;
        for (int w = 0; w < NW; ++w)
            for (int g = 0; g < G; ++g) M = fmaxf(M, part[(w * 64 + g * HT + e) * 18]);
        float L = 0.f, O = 0.f;
        for (int w = 0; w < NW; ++w)
            for (int g = 0; g < G; ++g) {
                const float *pp = part + (w * 64 + g * HT + e) * 18;
                const float f = __expf(pp[0] - M);
                L += pp[1] * f; O += pp[2 + d] * f;
            }
        const int eh = e / a.T, et = e % a.T;
        const __bf16 r = (__bf16)(O / L);
        a.o[(p * a.T + et) * ci + eh * 16 + d] = *(const uint16_t *)&r;
    }
}

// ---- C bf16 [M, N] (row stride ldc) = A . W^T + bias + add[m % add_rows] ---------------------------------------------------------
// The K | V (| Q) projections over the per-prompt keys: at K = 256 and N <= 256 the tiled GEMMs spend their time in prologues and
// epilogues (2 TB/s); with the weights resident this is a plain stream of A in, C out.
struct LinArgs {
    const uint16_t *A, *W;
    const float *bias, *add; int add_rows, ld_add;
    uint16_t *C; int ldc;
    int M;
};
template <int K, int N, int NTHREADS>
__global__ void __launch_bounds__(NTHREADS, NTHREADS == 512 ? 4 : 1) k_skinny_linear(LinArgs a) {
    using S = Skinny<K, N>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *cs = (float *)(smem + S::W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    S::load_w(smem, a.W, K, tid, NTHREADS);
    for (int i = tid; i < N; i += NTHREADS) cs[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    const float *cl = cs + fq * 4;
    const int blocks = (a.M + 15) / 16;
    for (int b = blockIdx.x * (NTHREADS / 64) + wave; b < blocks; b += gridDim.x * (NTHREADS / 64)) {
        const int m = b * 16 + fr, mc = m < a.M ? m : a.M - 1;
        bf16x8 af[S::KS];
        S::load_a(af, a.A, K, mc, fq);
        f32x4 acc[S::NT];
        const float *ap = a.add ? a.add + (long long)(mc % a.add_rows) * a.ld_add + fq * 4 : nullptr;
#pragma unroll
        for (int j = 0; j < S::NT; ++j) {
            f32x4 v = *(const f32x4 *)(cl + j * 16);
            if (ap) v += *(const f32x4 *)(ap + j * 16);
            acc[j] = v;
            if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        S::mma(acc, af, smem, fr, fq);
        if (m >= a.M) continue;
        uint16_t *cp = a.C + (long long)m * a.ldc;
        static_assert(S::NT % 2 == 0, "column tiles are stored in pairs");
#pragma unroll
        for (int j = 0; j < S::NT; j += 2)
            store_pair16(cp + j * 16, fq, make_uint2(pack2(acc[j][0], acc[j][1]), pack2(acc[j][2], acc[j][3])),
                         make_uint2(pack2(acc[j + 1][0], acc[j + 1][1]), pack2(acc[j + 1][2], acc[j + 1][3])));
    }
}

// GELU of the two upscaling kernels: the table in LDS (default) or the packed polynomial (OVO_SAM_GELU_POLY=1; measurement)
int sam_gelu_lut() {
    static int lut = getenv("OVO_SAM_GELU_POLY") == nullptr;
    if (ovo_knobs_dynamic()) lut = getenv("OVO_SAM_GELU_POLY") == nullptr;
    return lut;
}

template <typename KernelT>
int set_lds(KernelT k, size_t lds, const char *who) {
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ovo_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return OVO_E_LAUNCH; }
    return OVO_OK;
}

}  // namespace

extern "C" int ovo_sam_proj_ln(const void *A, const void *W, const float *bias, const float *res, const void *res16, int64_t res_rows, const float *gamma,
                               const float *beta, float eps, const float *pe, int64_t pe_rows, float *y32, void *y16, void *ype16,
                               int64_t M, int N, int K, ovo_stream_t stream) {
    OVO_REQUIRE(M >= 0 && M < (1ll << 31) - 16, "bad row count");
    if (!((N == 256 && K == 128) || (N == 128 && K == 64))) return OVO_E_UNSUPPORTED;
    if (M == 0) return OVO_OK;
    OVO_REQUIRE(A && W && gamma && beta && (y32 || y16 || ype16), "null pointer");
    OVO_REQUIRE(!(res && res16) && (!(res || res16) || (res_rows > 0 && res_rows <= M)) && (!ype16 || (pe && pe_rows > 0)),
                "one residual at most; broadcast sources need their row counts");
    OVO_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)res | (uintptr_t)pe | (uintptr_t)y32 | (uintptr_t)y16 | (uintptr_t)ype16) % 16 == 0, "16-byte alignment");
    ProjLnArgs a;
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = bias; a.res = res; a.res16 = (const uint16_t *)res16; a.res_rows = (int)res_rows; a.gamma = gamma; a.beta = beta;
    a.eps = eps; a.pe = pe; a.pe_rows = (int)pe_rows; a.y32 = y32; a.y16 = (uint16_t *)y16; a.ype16 = (uint16_t *)ype16; a.M = (int)M;
    const int blocks = (int)((M + 15) / 16), grid = blocks < 512 * 8 ? (blocks + 7) / 8 : 512;
    hipStream_t st = (hipStream_t)stream;
#define GO(KK, NN)                                                                                           \
    {                                                                                                        \
        const size_t lds = (size_t)NN * KK * 2 + 3 * NN * sizeof(float);                                     \
        static bool done = false;                                                                            \
        if (!done) { if (int rc = set_lds(k_proj_ln<KK, NN>, lds, __func__)) return rc; done = true; }       \
        k_proj_ln<KK, NN><<<grid, 512, lds, st>>>(a);                                                        \
    }
    if (N == 256) GO(128, 256) else GO(64, 128)
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_up1_ln(const void *A, const void *W, const float *bias, const float *feat, const float *gamma, const float *beta,
                              float eps, int64_t P, int s, int C1, int K, void *out, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && s > 0 && P * s * s < (1ll << 31) - 16, "bad shape");
    if (!((C1 == 64 && K == 256) || (C1 == 32 && K == 128))) return OVO_E_UNSUPPORTED;
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(A && W && bias && feat && gamma && beta && out, "null pointer");
    OVO_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)feat | (uintptr_t)out) % 16 == 0, "16-byte alignment");
    Up1Args a;
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = bias; a.feat = feat; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.P = (int)P; a.s = s; a.out = (uint16_t *)out;
    a.lut = sam_gelu_lut();
    const int blocks = (int)((P * s * s + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
#define GO(KK, CC, NTH, WGS)                                                                                     \
    {                                                                                                            \
        const size_t lds = (size_t)4 * CC * KK * 2 + 3 * CC * sizeof(float) + ovo_gemm_detail::GELU_LUT_BYTES;   \
        static bool done = false;                                                                                \
        if (!done) { if (int rc = set_lds(k_up1_ln<KK, CC, NTH>, lds, __func__)) return rc; done = true; }       \
        const int wpb = NTH / 64, grid = blocks < WGS * wpb ? (blocks + wpb - 1) / wpb : WGS;                    \
        k_up1_ln<KK, CC, NTH><<<grid, NTH, lds, st>>>(a);                                                        \
    }
    if (C1 == 64) GO(256, 64, 1024, 256) else GO(128, 32, 512, 512)     // 128 KB of weights: one 16-wave workgroup per CU
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_up2_masks(const void *A, const void *W, const float *bias, const float *feat, const float *hyper, int n_mask, int first,
                                 int64_t P, int s2, int C2, int K, float *out, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && s2 > 0 && P * s2 * s2 < (1ll << 31) - 16, "bad shape");
    OVO_REQUIRE(n_mask > 0 && first >= 0 && first < n_mask && n_mask - first <= 4, "at most 4 mask tokens");
    if (!((C2 == 32 && K == 64) || (C2 == 16 && K == 32))) return OVO_E_UNSUPPORTED;
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(A && W && bias && feat && hyper && out, "null pointer");
    OVO_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)feat | (uintptr_t)hyper) % 16 == 0, "16-byte alignment");
    Up2Args a;
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = bias; a.feat = feat; a.hyper = hyper; a.n_mask = n_mask; a.first = first;
    a.n_out = n_mask - first; a.P = (int)P; a.s2 = s2; a.out = out;
    a.lut = sam_gelu_lut();
    const int blocks = (int)((P * s2 * s2 + 15) / 16), grid = blocks < 512 * 8 ? (blocks + 7) / 8 : 512;
    hipStream_t st = (hipStream_t)stream;
#define GO(KK, CC)                                                                                           \
    {                                                                                                        \
        const size_t lds = (size_t)4 * CC * KK * 2 + CC * sizeof(float) + ovo_gemm_detail::GELU_LUT_BYTES;   \
        static bool done = false;                                                                            \
        if (!done) { if (int rc = set_lds(k_up2_masks<KK, CC>, lds, __func__)) return rc; done = true; }     \
        k_up2_masks<KK, CC><<<grid, 512, lds, st>>>(a);                                                      \
    }
    if (C2 == 32) GO(64, 32) else GO(32, 16)
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_t2i_attention(const void *q, const void *k, const void *v, int64_t kv_batch_stride, int kv_token_stride, void *o, int64_t P,
                                     int S, int T, int H, float scale, ovo_stream_t stream) {
    OVO_REQUIRE(P >= 0 && S > 0 && T > 0 && H > 0, "bad shape");
    if (H * T > 64 || 64 % (H * T) != 0 || H * T * 16 > 1024) return OVO_E_UNSUPPORTED;
    if (P == 0) return OVO_OK;
    OVO_REQUIRE(q && k && v && o && kv_batch_stride % 8 == 0 && kv_token_stride % 8 == 0 && kv_token_stride >= 16 * H, "null / misaligned argument");
    OVO_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0, "16-byte alignment");
    T2iArgs a;
    a.q = (const uint16_t *)q; a.k = (const uint16_t *)k; a.v = (const uint16_t *)v; a.kv_sb = kv_batch_stride; a.kv_st = kv_token_stride;
    a.o = (uint16_t *)o; a.S = S; a.T = T; a.H = H; a.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    constexpr int NW = 16;
    const size_t lds = (size_t)NW * 64 * 18 * sizeof(float);
    static bool done = false;
    if (!done) { if (int rc = set_lds(k_t2i_attention<NW>, lds, __func__)) return rc; done = true; }
    k_t2i_attention<NW><<<(unsigned)P, 64 * NW, lds, st>>>(a);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_sam_linear(const void *A, const void *W, const float *bias, const float *add, int64_t add_rows, int ld_add, void *C, int ldc,
                              int64_t M, int N, int K, ovo_stream_t stream) {
    OVO_REQUIRE(M >= 0 && M < (1ll << 31) - 16, "bad row count");
    if (!((K == 256 && (N == 256 || N == 128)) || (K == 128 && (N == 128 || N == 64)))) return OVO_E_UNSUPPORTED;
    if (M == 0) return OVO_OK;
    OVO_REQUIRE(A && W && C && ldc >= N && ldc % 8 == 0, "null pointer / bad ldc (a multiple of 8 elements)");
    OVO_REQUIRE(!add || (add_rows > 0 && add_rows < (1ll << 31) && ld_add >= N && ld_add % 4 == 0), "periodic add needs its row count and stride");
    OVO_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)add | (uintptr_t)bias | (uintptr_t)C) % 16 == 0, "16-byte alignment");
    LinArgs a;
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = bias; a.add = add; a.add_rows = (int)add_rows; a.ld_add = ld_add;
    a.C = (uint16_t *)C; a.ldc = ldc; a.M = (int)M;
    const int blocks = (int)((M + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
#define GO(KK, NN, NTH, WGS)                                                                                          \
    {                                                                                                                 \
        const size_t lds = (size_t)NN * KK * 2 + NN * sizeof(float);                                                  \
        static bool done = false;                                                                                     \
        if (!done) { if (int rc = set_lds(k_skinny_linear<KK, NN, NTH>, lds, __func__)) return rc; done = true; }     \
        const int wpb = NTH / 64, grid = blocks < WGS * wpb ? (blocks + wpb - 1) / wpb : WGS;                         \
        k_skinny_linear<KK, NN, NTH><<<grid, NTH, lds, st>>>(a);                                                      \
    }
    if (K == 256 && N == 256) GO(256, 256, 1024, 256)       // 128 KB of weights: one 16-wave workgroup per CU
    else if (K == 256) GO(256, 128, 512, 512)
    else if (N == 128) GO(128, 128, 512, 512)
    else GO(128, 64, 512, 512)
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
