// winattn.hip -- the attention half of a Hiera stage-1 block up to the output projection, in ONE pass over the f32 residual stream (round 5):
//
//     att[window-major rows, :] = softmax(q k^T) v     with   q | k | v = LayerNorm(x) . Wqkv^T + b    per 8 x 8 window and head
//
// As three launches (LayerNorm inside the streaming QKV product -> bf16 q | k | v [rows, 336] -> window attention) the q | k | v tensor of a
// 12-frame group (528 MB) was written and read back through HBM: 200 + 186 us per block, each at its own HBM roofline.  Here a WAVE owns a
// window (64 tokens x 112 channels = 28 KB of x), the weights [336, 128] stay in LDS for the life of the workgroup, and nothing but x (in) and
// the attention output (out) touches HBM.  Everything between stays in registers because every product is set up so that the MFMA's C layout
// (lane = (column l16, row group g): 4 consecutive rows of one column) IS the operand layout of the next product (lane = (index l16, 8
// consecutive k of group g)) up to a permutation of the reduction index that both operands share:
//     K^T = Wk . X^T  (rows = head dim)   two head-dim tiles packed  ->  A operand of  S^T = K . Q^T      (k = head dim)
//     Q^T = Wq . X^T                       ''                         ->  B operand of  S^T
//     V   = X . Wv^T  (rows = keys)        two key tiles packed       ->  A operand of  O^T = V^T . P^T    (k = key)
//     S^T (rows = keys, columns = queries): softmax over the keys = in-lane over 16 values + two xor-shuffles; exp2 of the scores (the q rows
//          of Wqkv carry log2(e) / sqrt(head_dim), ovo_hiera_config_t.q_prescaled), packed  ->  B operand of  O^T
//     O^T (rows = head dim, column = query): 4 consecutive channels of one token per lane -> one 8-byte store.
// head_dim 56 is walked as 4 tiles of 16: the weights sit in LDS as (q | k | v, head) blocks of 64 rows whose last 8 (and their bias) are zeros.
// hiera_b+: the two stage-1 blocks (C = 112, 2 heads, 8 x 8 windows) and the stage-change block that follows them (112 -> 224 channels, 4 heads, queries
// pooled 2 x 2: k_win_attn112<true>, two passes of two heads).  From stage 2 on (C = 224) the weights (301 KB) do not fit LDS.
// Reference: sam2 MultiScaleBlock / MultiScaleAttention inside the image encoder, reached at mask_generator.py:113.
//
// Measured on the way (12 frames, stage 1; profiles/r05c_winattn_bench.txt has the final numbers):
//   * first form (unpadded 56-row blocks with selects, zero-initialised accumulators + bias adds, Q^T per query tile, scalar LayerNorm math): 250 us =
//     load phase 92 us (alone: x in at 3.8 TB/s) + product phase 160 us (alone) -- the two waves of a SIMD run in step and the phases do not overlap;
//   * 64-row zero-padded blocks (no selects), bias as the accumulators' initial value, fragments kept as whole uint4 (no v_mov in front of the MFMAs),
//     LayerNorm on explicit f32 pairs (v_pk_*, 340 v_mov gone), Q^T for all four query tiles per weight read: 214 us;
//   * quad reductions on v_permlane16/32_swap instead of ds_bpermute: 202 us;
//   * a start stagger of 10-40 us between the two waves of a SIMD: no effect (214-225 us);
//   * one wave per SIMD (512 registers) with the next window requested as soon as LayerNorm frees this one's 128 registers: 245 us -- the product
//     phase of a single wave per SIMD is a serial dependency chain (~20 us per window), two waves per SIMD hide more of it than the prefetch does.
#include <stdlib.h>

#include "gemm_common.h"

namespace {

using ovo_gemm_detail::bf16x8;
using ovo_gemm_detail::f32x4;
using ovo_gemm_detail::Mfma;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int C = 112, KP = 128, KS = 4, NH = 2, HD = 56, HP = 64, NROW = 3 * NH * HP, CPR = 16, WT = 64;
// LDS rows: (part q | k | v, head, head dim padded to 64) -- the 8 padding rows of every block (and their bias) are zeros, so head-dim tile 3 needs no select
constexpr int LDS_W = NROW * KP * 2, LDS_BYTES = LDS_W + NROW * 4 + 2 * KP * 4;

struct WinAttnArgs {
    const float *x; int B, H, W;
    const float *ln_g, *ln_b; float eps;
    const uint16_t *w; long long ldw; const float *bias;
    uint16_t *att; int ld_att;
    int n_win, nwh, nww;
    int part_rows, head0;          // rows between the q | k | v parts of Wqkv (= output width: 112, or 224 at the stage change); first head of this pass
};

__device__ __forceinline__ uint32_t pk2(float a, float b) {       // v_cvt_pk_bf16_f32 (RNE)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return *(const uint32_t *)&h;
}
__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint4 u = make_uint4(a, b, c, d);
    return *(const bf16x8 *)&u;
}
// reductions over the 4 lanes that share l16 (lane ^ 16, lane ^ 32) on the VALU: v_permlane16_swap / v_permlane32_swap of a value with itself leave the
// two partners' values side by side in every lane (a ds_bpermute round trip through LDS was ~120 cycles on the softmax's serial path, four per query tile)
__device__ __forceinline__ float quad_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// xor-1 / xor-8 partners inside a row of 16 lanes on the DPP path (quad_perm [1, 0, 3, 2]; row_ror 8)
__device__ __forceinline__ float pool_max4(float v) {
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xf, 0xf, true)));
    return fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xf, 0xf, true)));
}

// POOL: the stage-change block (sam2 MultiScaleBlock with q_stride 2): q is 2 x 2 max-pooled over the window's 8 x 8 tokens before the scores -- 16 queries
// per window attend to its 64 keys, att rows = window x 16 pooled positions.  Its projection is twice as wide (4 heads): two passes of two heads each.
template <bool POOL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_win_attn112(WinAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_bias = (float *)(smem + LDS_W), *s_g = s_bias + NROW, *s_b = s_g + KP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    for (int id = tid; id < NROW * CPR; id += 512) {              // weights: 16-byte chunk c of LDS row n at chunk (c ^ (n & 15)) of the row
        const int n = id >> 4, c = id & 15, blk = n >> 6, r = n & 63;             // blk = part * NH + head
        uint4 u = make_uint4(0u, 0u, 0u, 0u);
        if (r < HD) u = *(const uint4 *)(a.w + (long long)((blk >> 1) * a.part_rows + (a.head0 + (blk & 1)) * HD + r) * a.ldw + c * 8);
        *(uint4 *)(smem + (n * CPR + (c ^ (n & 15))) * 16) = u;
    }
    for (int n = tid; n < NROW; n += 512) {
        const int blk = n >> 6, r = n & 63;
        s_bias[n] = (a.bias && r < HD) ? a.bias[(blk >> 1) * a.part_rows + (a.head0 + (blk & 1)) * HD + r] : 0.f;
    }
    for (int i = tid; i < KP; i += 512) { s_g[i] = i < C ? a.ln_g[i] : 0.f; s_b[i] = i < C ? a.ln_b[i] : 0.f; }
    __syncthreads();
    const bool tail = g >= 2;                                     // K-step 3 of this lane group = channels 112..127: padding
    // a weight fragment: 16 bytes (8 channels, K-step ks, this lane's group) of LDS row `row`
    auto wfrag = [&](int row, int ks) -> bf16x8 { return *(const bf16x8 *)(smem + (row * CPR + ((ks * 4 + g) ^ (row & 15))) * 16); };
    for (int win = blockIdx.x * 8 + wave; win < a.n_win; win += gridDim.x * 8) {
        const int per = a.nwh * a.nww, b = win / per, wr = win - b * per, wy = wr / a.nww, wx = wr - wy * a.nww;
        const float *xw = a.x + (((long long)b * a.H + wy * 8) * a.W + wx * 8) * C;
        // ---- phase 1: the window's 64 rows, LayerNorm, bf16 fragments xf[token tile][K-step] (lane: token l16 of the tile, 8 channels of group g)
        bf16x8 xf[4][KS];
        {
            // K-step 3 of lane groups 2, 3 is the K padding (channels 112..127): those lanes re-read channels 96..111 (a valid address, no select) --
            // gamma = beta = 0 there turns whatever they hold into zeros, and `real` (0 or 1) keeps it out of the row statistics.
            // All arithmetic on f32 PAIRS that are adjacent in the loaded float4 (v_pk_*): no register shuffling in front of the packed ops.
            f32x4 v[4][KS][2];
            const int off3 = 96 + 8 * (g & 1);
            const float real = tail ? 0.f : 1.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float *xr = xw + ((long long)(2 * tt + (l16 >> 3)) * a.W + (l16 & 7)) * C;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float *q = xr + (ks < 3 ? 32 * ks + 8 * g : off3);
                    v[tt][ks][0] = *(const f32x4 *)q; v[tt][ks][1] = *(const f32x4 *)(q + 4);
                }
            }
            float rstd[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x2 s2 = (v[tt][0][0].lo + v[tt][0][0].hi) + (v[tt][0][1].lo + v[tt][0][1].hi);
#pragma unroll
                for (int ks = 1; ks < 3; ++ks) s2 += (v[tt][ks][0].lo + v[tt][ks][0].hi) + (v[tt][ks][1].lo + v[tt][ks][1].hi);
                s2 += ((v[tt][3][0].lo + v[tt][3][0].hi) + (v[tt][3][1].lo + v[tt][3][1].hi)) * real;
                const float mean = quad_sum(s2.x + s2.y) * (1.0f / C);
                f32x2 q2 = {0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    f32x2 part = {0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        v[tt][ks][h] -= mean;
                        part += v[tt][ks][h].lo * v[tt][ks][h].lo + v[tt][ks][h].hi * v[tt][ks][h].hi;
                    }
                    q2 += ks < 3 ? part : part * real;
                }
                rstd[tt] = rsqrtf(quad_sum(q2.x + q2.y) * (1.0f / C) + a.eps);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                     // (K-step outermost: one set of gamma / beta registers at a time)
                const int c0 = 32 * ks + 8 * g;                   // (gamma = beta = 0 at the padding channels: they come out as zeros)
                const f32x4 g0 = *(const f32x4 *)(s_g + c0), g1 = *(const f32x4 *)(s_g + c0 + 4);
                const f32x4 b0 = *(const f32x4 *)(s_b + c0), b1 = *(const f32x4 *)(s_b + c0 + 4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const f32x4 y0 = v[tt][ks][0] * rstd[tt] * g0 + b0, y1 = v[tt][ks][1] * rstd[tt] * g1 + b1;
                    xf[tt][ks] = frag(pk2(y0[0], y0[1]), pk2(y0[2], y0[3]), pk2(y1[0], y1[1]), pk2(y1[2], y1[3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- phase 2: per head
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            uint4 ka[4][2], qb[4][2], va[4][2];     // [token tile][K-step over head dim] A / B fragments of S^T; [head-dim tile][K-step over keys] A fragments of O^T
            const int rq = h * HP, rk = (NH + h) * HP, rvv = (2 * NH + h) * HP;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {   // K^T, then Q^T tiles (rows = head dim 16 ht + 4 g + j, column = token l16 of tile t); the bias seeds the accumulators
                    const int r0 = (which == 0 ? rk : rq) + 16 * ht;
                    const f32x4 bb = *(const f32x4 *)(s_bias + r0 + 4 * g);
                    f32x4 acc[4] = {bb, bb, bb, bb};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf16x8 w = wfrag(r0 + l16, ks);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = Mfma<bf16x8>::run(w, xf[t][ks], acc[t]);
                    }
                    if (POOL && which == 1) {
                        // token tile t = window rows 2 t, 2 t + 1: the pool cell (t, px) is lanes l16 in {2 px, 2 px + 1, 8 + 2 px, 9 + 2 px}; pooled query
                        // 4 t + px of the window goes to lane l16 = 4 t + px of the ONE query tile
                        f32x4 r = acc[0];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float m = __shfl(pool_max4(acc[t][j]), (lane & 48) | ((l16 & 3) << 1), 64);
                                if (t == 0 || (l16 >> 2) == t) r[j] = m;
                            }
                        acc[0] = r;
                    }
#pragma unroll
                    for (int t = 0; t < ((POOL && which == 1) ? 1 : 4); ++t) {
                        uint4 &d = which == 0 ? ka[t][ht >> 1] : qb[t][ht >> 1];
                        if (ht & 1) { d.z = pk2(acc[t][0], acc[t][1]); d.w = pk2(acc[t][2], acc[t][3]); }
                        else { d.x = pk2(acc[t][0], acc[t][1]); d.y = pk2(acc[t][2], acc[t][3]); }
                    }
                }
                {   // V tiles (rows = key 16 t + 4 g + j, column = head dim 16 ht + l16)
                    const float bv = s_bias[rvv + 16 * ht + l16];
                    const f32x4 bb = {bv, bv, bv, bv};
                    f32x4 acc[4] = {bb, bb, bb, bb};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf16x8 w = wfrag(rvv + 16 * ht + l16, ks);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = Mfma<bf16x8>::run(xf[t][ks], w, acc[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t & 1) { va[ht][t >> 1].z = pk2(acc[t][0], acc[t][1]); va[ht][t >> 1].w = pk2(acc[t][2], acc[t][3]); }
                        else { va[ht][t >> 1].x = pk2(acc[t][0], acc[t][1]); va[ht][t >> 1].y = pk2(acc[t][2], acc[t][3]); }
                    }
                }
            }
            // ---- one query tile at a time: scores, softmax over the keys, output
#pragma unroll
            for (int q = 0; q < (POOL ? 1 : 4); ++q) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) s[t] = Mfma<bf16x8>::run(*(const bf16x8 *)&ka[t][0], *(const bf16x8 *)&qb[q][0], zero);
#pragma unroll
                for (int t = 0; t < 4; ++t) s[t] = Mfma<bf16x8>::run(*(const bf16x8 *)&ka[t][1], *(const bf16x8 *)&qb[q][1], s[t]);
                float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
                for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
                mx = quad_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[t][j] = __builtin_amdgcn_exp2f(s[t][j] - mx);
                    sum += (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
                }
                const float inv = __builtin_amdgcn_rcpf(quad_sum(sum));
                const bf16x8 p0 = frag(pk2(s[0][0], s[0][1]), pk2(s[0][2], s[0][3]), pk2(s[1][0], s[1][1]), pk2(s[1][2], s[1][3]));
                const bf16x8 p1 = frag(pk2(s[2][0], s[2][1]), pk2(s[2][2], s[2][3]), pk2(s[3][0], s[3][1]), pk2(s[3][2], s[3][3]));
                uint16_t *dst = a.att + ((long long)win * (POOL ? 16 : WT) + 16 * q + l16) * a.ld_att + (a.head0 + h) * HD + 4 * g;
                f32x4 o[4];
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) o[ht] = Mfma<bf16x8>::run(*(const bf16x8 *)&va[ht][0], p0, zero);
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) o[ht] = Mfma<bf16x8>::run(*(const bf16x8 *)&va[ht][1], p1, o[ht]);
#pragma unroll
                for (int ht = 0; ht < 4; ++ht)
                    if (16 * ht + 4 * g < HD) *(uint2 *)(dst + 16 * ht) = make_uint2(pk2(o[ht][0] * inv, o[ht][1] * inv), pk2(o[ht][2] * inv, o[ht][3] * inv));
            }
        }
    }
}


// ---- C = 224 (hiera_b+ stage 2, blocks after the stage change): 4 heads of 56, 4 x 4 windows --------------------------------------------------------
// The same chain with a window = ONE 16-token tile: a wave takes two windows (32 tokens x 224 channels) at a time; scores are one 16 x 16 tile per window
// and head, and O^T = V^T . P^T has a reduction of 16 keys: v_mfma_f32_16x16x16_bf16, whose operand layout (4 consecutive k per lane) is the 16 x 16
// accumulator layout itself.  The weights of TWO heads fill LDS (336 rows x 448 bytes, row pitch 464: 16 consecutive rows start 13 x 16-byte chunks
// apart mod 16 -> conflict-free without a swizzle), so a block is two passes over x; blocks of 56 rows cannot be padded to 64 here (178 KB): head-dim
// tile 3 selects zeros for its rows / columns beyond 56.
namespace c224 {
constexpr int C2 = 224, KS2 = 7, ROWS = 3 * NH * HD, PITCH = 464, NW = 8;
constexpr int LDS2_W = ROWS * PITCH, LDS2_BYTES = LDS2_W + ROWS * 4 + 2 * C2 * 4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) k_win_attn224(WinAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_bias = (float *)(smem + LDS2_W), *s_g = s_bias + ROWS, *s_b = s_g + C2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    for (int id = tid; id < ROWS * 28; id += 64 * NW) {
        const int n = id / 28, c = id - 28 * n, blk = n / HD, r = n - blk * HD;            // blk = part * NH + head of the pass
        *(uint4 *)(smem + n * PITCH + c * 16) = *(const uint4 *)(a.w + (long long)((blk >> 1) * a.part_rows + (a.head0 + (blk & 1)) * HD + r) * a.ldw + c * 8);
    }
    for (int n = tid; n < ROWS; n += 64 * NW) {
        const int blk = n / HD, r = n - blk * HD;
        s_bias[n] = a.bias ? a.bias[(blk >> 1) * a.part_rows + (a.head0 + (blk & 1)) * HD + r] : 0.f;
    }
    for (int i = tid; i < C2; i += 64 * NW) { s_g[i] = a.ln_g[i]; s_b[i] = a.ln_b[i]; }
    __syncthreads();
    auto wfrag = [&](int row, int ks, bool valid) -> bf16x8 {
        uint4 u = *(const uint4 *)(smem + row * PITCH + (ks * 4 + g) * 16);
        if (!valid) u = make_uint4(0u, 0u, 0u, 0u);
        return *(const bf16x8 *)&u;
    };
    const int n_task = a.n_win >> 1;                             // (the launcher requires an even number of windows)
    for (int task = blockIdx.x * NW + wave; task < n_task; task += gridDim.x * NW) {
        // ---- phase 1: two windows' rows (lane: token l16 of window t, 8 channels of group g per K-step), LayerNorm, bf16 fragments
        bf16x8 xf[2][KS2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {                            // (one window at a time: 56 registers of f32 rows beside the fragments)
            f32x4 v[KS2][2];
            const int win = 2 * task + t, per = a.nwh * a.nww, b = win / per, wr = win - b * per, wy = wr / a.nww, wx = wr - wy * a.nww;
            const float *xr = a.x + (((long long)b * a.H + wy * 4 + (l16 >> 2)) * a.W + wx * 4 + (l16 & 3)) * C2 + 8 * g;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) { v[ks][0] = *(const f32x4 *)(xr + 32 * ks); v[ks][1] = *(const f32x4 *)(xr + 32 * ks + 4); }
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) s2 += (v[ks][0].lo + v[ks][0].hi) + (v[ks][1].lo + v[ks][1].hi);
            const float mean = quad_sum(s2.x + s2.y) * (1.0f / C2);
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[ks][h] -= mean;
                    q2 += v[ks][h].lo * v[ks][h].lo + v[ks][h].hi * v[ks][h].hi;
                }
            const float rstd = rsqrtf(quad_sum(q2.x + q2.y) * (1.0f / C2) + a.eps);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const int c0 = 32 * ks + 8 * g;
                const f32x4 g0 = *(const f32x4 *)(s_g + c0), g1 = *(const f32x4 *)(s_g + c0 + 4);
                const f32x4 b0 = *(const f32x4 *)(s_b + c0), b1 = *(const f32x4 *)(s_b + c0 + 4);
                const f32x4 y0 = v[ks][0] * rstd * g0 + b0, y1 = v[ks][1] * rstd * g1 + b1;
                xf[t][ks] = frag(pk2(y0[0], y0[1]), pk2(y0[2], y0[3]), pk2(y1[0], y1[1]), pk2(y1[2], y1[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase 2: per head of the pass
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            uint4 ka[2][2], qb[2][2];        // [window][K-step over head dim]: A / B fragments of S^T
            uint2 va[4][2];                  // [head-dim tile][window]: A fragment (16 keys) of O^T
            const int rq = h * HD, rk = (NH + h) * HD, rvv = (2 * NH + h) * HD;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const bool rv = ht < 3 || l16 < 8, cv = ht < 3 || g < 2;       // this lane's weight row / its 4 accumulator rows are real head dims
#pragma unroll
                for (int which = 0; which < 2; ++which) {        // K^T, then Q^T tiles (rows = head dim 16 ht + 4 g + j, column = token l16 of window t)
                    const int r0 = (which == 0 ? rk : rq) + 16 * ht;
                    f32x4 bb = *(const f32x4 *)(s_bias + r0 + 4 * g);
                    if (!cv) bb = (f32x4)(0.f);
                    f32x4 acc[2] = {bb, bb};
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) {
                        const bf16x8 w = wfrag(r0 + l16, ks, rv);
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t] = Mfma<bf16x8>::run(w, xf[t][ks], acc[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        uint4 &d = which == 0 ? ka[t][ht >> 1] : qb[t][ht >> 1];
                        if (ht & 1) { d.z = pk2(acc[t][0], acc[t][1]); d.w = pk2(acc[t][2], acc[t][3]); }
                        else { d.x = pk2(acc[t][0], acc[t][1]); d.y = pk2(acc[t][2], acc[t][3]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);            // (keeps the next tile's seven weight reads from being hoisted over this one: registers)
                }
                {   // V tiles (rows = key 4 g + j of window t, column = head dim 16 ht + l16)
                    const float bv = rv ? s_bias[rvv + 16 * ht + l16] : 0.f;
                    const f32x4 bb = {bv, bv, bv, bv};
                    f32x4 acc[2] = {bb, bb};
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) {
                        const bf16x8 w = wfrag(rvv + 16 * ht + l16, ks, rv);
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t] = Mfma<bf16x8>::run(xf[t][ks], w, acc[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) va[ht][t] = make_uint2(pk2(acc[t][0], acc[t][1]), pk2(acc[t][2], acc[t][3]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {                        // one window: 16 x 16 scores, softmax over its 16 keys, output
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 s = Mfma<bf16x8>::run(*(const bf16x8 *)&ka[t][0], *(const bf16x8 *)&qb[t][0], zero);
                s = Mfma<bf16x8>::run(*(const bf16x8 *)&ka[t][1], *(const bf16x8 *)&qb[t][1], s);
                const float mx = quad_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_exp2f(s[j] - mx);
                const float inv = __builtin_amdgcn_rcpf(quad_sum((s[0] + s[1]) + (s[2] + s[3])));
                const uint2 pp = make_uint2(pk2(s[0], s[1]), pk2(s[2], s[3]));
                uint16_t *dst = a.att + ((long long)(2 * task + t) * 16 + l16) * a.ld_att + (a.head0 + h) * HD + 4 * g;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) {
                    const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*(const bf16x4 *)&va[ht][t], *(const bf16x4 *)&pp, zero, 0, 0, 0);
                    if (ht < 3 || g < 2) *(uint2 *)(dst + 16 * ht) = make_uint2(pk2(o[0] * inv, o[1] * inv), pk2(o[2] * inv, o[3] * inv));
                }
            }
        }
    }
}
}  // namespace c224

}  // namespace

namespace ovo_gemm_detail {

int win_attn_launch(const float *x, int B, int H, int W, int ws, int d, int d_out, int heads, int pool, const float *ln_g, const float *ln_b, float eps,
                    const void *qkv_w, long long ldw, const float *qkv_b, void *att, int ld_att, hipStream_t s) {
    static const bool off_once = getenv("OVO_HIERA_NO_WINATTN") != nullptr;                // measurement / tests: the three-launch form
    if (ovo_knobs_dynamic() ? getenv("OVO_HIERA_NO_WINATTN") != nullptr : off_once) return OVO_E_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)qkv_w) & 15) != 0 || ((uintptr_t)att & 7) != 0 || ldw % 8 != 0 || ld_att < d_out || ld_att % 4 != 0) return OVO_E_UNSUPPORTED;
    if (d == c224::C2 && d_out == d && heads == 2 * NH && !pool && ws == 4 && H % 4 == 0 && W % 4 == 0 && ldw >= c224::C2) {     // stage 2: 4 x 4 windows
        const long long n_win = (long long)B * (H / 4) * (W / 4);
        if (n_win < 512 || n_win % 2 != 0 || n_win >= (1ll << 31) / 16) return OVO_E_UNSUPPORTED;
        static bool set2 = false;
        if (!set2) {
            hipError_t e = hipFuncSetAttribute((const void *)c224::k_win_attn224, hipFuncAttributeMaxDynamicSharedMemorySize, c224::LDS2_BYTES);
            if (e != hipSuccess) { ovo_set_error("win_attn_launch: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
            set2 = true;
        }
        WinAttnArgs a;
        a.x = x; a.B = B; a.H = H; a.W = W; a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps;
        a.w = (const uint16_t *)qkv_w; a.ldw = ldw; a.bias = qkv_b; a.att = (uint16_t *)att; a.ld_att = ld_att;
        a.n_win = (int)n_win; a.nwh = H / 4; a.nww = W / 4; a.part_rows = d_out;
        // (the event profiler books the pair of passes as one streaming-family launch: QKV product + scores + P . V flops; bytes = x in, attention out)
        const bool prof = ovo_prof_enabled();
        if (prof) {
            ovo_prof_begin(8, 2.0 * n_win * 16.0 * d * 3.0 * d_out + 4.0 * n_win * heads * 16.0 * 16.0 * HD, s);
            ovo_prof_shape((int)(n_win * 16), 3 * d_out, d); ovo_prof_flags(32 | 64); ovo_prof_bytes((double)n_win * 16.0 * (4.0 * d + 2.0 * d_out));
        }
        for (int h0 = 0; h0 < heads; h0 += NH) {
            a.head0 = h0;
            c224::k_win_attn224<<<256, 64 * c224::NW, c224::LDS2_BYTES, s>>>(a);
        }
        if (prof) ovo_prof_end(s);
        OVO_CHECK_LAUNCH();
        return OVO_OK;
    }
    const bool plain = !pool && d_out == C && heads == NH, change = pool && d_out == 2 * C && heads == 2 * NH;
    if (d != C || !(plain || change) || ws != 8 || H % 8 != 0 || W % 8 != 0 || ldw < KP) return OVO_E_UNSUPPORTED;
    const long long n_win = (long long)B * (H / 8) * (W / 8);
    if (n_win < 512 || n_win >= (1ll << 31) / WT) return OVO_E_UNSUPPORTED;                 // (short streams: the weight copy per workgroup would dominate)
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_win_attn112<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_win_attn112<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { ovo_set_error("win_attn_launch: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        set = true;
    }
    WinAttnArgs a;
    a.x = x; a.B = B; a.H = H; a.W = W; a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps;
    a.w = (const uint16_t *)qkv_w; a.ldw = ldw; a.bias = qkv_b; a.att = (uint16_t *)att; a.ld_att = ld_att;
    a.n_win = (int)n_win; a.nwh = H / 8; a.nww = W / 8; a.part_rows = d_out;
    const bool prof = ovo_prof_enabled();
    if (prof) {
        const double tq = pool ? 16.0 : 64.0;
        ovo_prof_begin(8, 2.0 * n_win * 64.0 * d * 3.0 * d_out + 4.0 * n_win * heads * tq * 64.0 * HD, s);
        ovo_prof_shape((int)(n_win * 64), 3 * d_out, d); ovo_prof_flags(32 | 64); ovo_prof_bytes((double)n_win * (64.0 * 4.0 * d + tq * 2.0 * d_out));
    }
    for (int h0 = 0; h0 < heads; h0 += NH) {                       // two heads per pass (their weights fill LDS)
        a.head0 = h0;
        if (pool) k_win_attn112<true><<<256, 512, LDS_BYTES, s>>>(a);
        else k_win_attn112<false><<<256, 512, LDS_BYTES, s>>>(a);
    }
    if (prof) ovo_prof_end(s);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // namespace ovo_gemm_detail

extern "C" int ovo_window_attention_f32(const float *x, int B, int H, int W, int window, int d, int d_out, int heads, int pool, const float *ln_g,
                                        const float *ln_b, float eps, const void *qkv_w, int64_t ldw, const float *qkv_b, void *att, int ld_att,
                                        ovo_stream_t stream) {
    OVO_REQUIRE(x && ln_g && ln_b && qkv_w && att && B > 0 && H > 0 && W > 0, "bad argument");
    return ovo_gemm_detail::win_attn_launch(x, B, H, W, window, d, d_out, heads, pool, ln_g, ln_b, eps, qkv_w, ldw, qkv_b, att, ld_att, (hipStream_t)stream);
}
