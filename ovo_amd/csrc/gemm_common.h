// gemm_common.h -- operand descriptor, MFMA wrappers and epilogue helpers shared by the MFMA GEMM kernels
// (gemm.hip: LDS-DMA ring kernels; gemm8p.hip: the 256-row ping-pong kernel).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace ovo_gemm_detail {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct GemmArgs {
    const char *A; long long lda;
    const char *W; long long ldw;
    const float *bias;
    void *C; long long ldc;
    const float *add; long long ld_add;
    int M, N, K;
    int out_dtype, act;
    float alpha;
    int nbn;
    int chunk, tiles;          // chunk > 0: XCD-chunked tile order (see launch())
    int add_rows;              // ovo_gemm_periodic: > 0 = `add` holds add_rows rows and product row m reads row m % add_rows
    int strip;                 // gemm8p: > 0 = tiles ordered in column strips of this many n-tiles (m fastest across a strip's rows)
    unsigned long long *best;  // ovo_gemm_argmax: packed (score, column) running maximum per row, or NULL
    int store, n_valid;        // with best: also store C?; columns >= n_valid (vocabulary padding) never win
    const float *rope_cos, *rope_sin;   // ovo_gemm_rope: rotary embedding of columns [0, rope_cols) in the epilogue, or NULL
    int rope_T, rope_hd, rope_cols, rope_t0;
    int win_per, win_ww, win_wh, win_nww, win_nwin, win_H, win_W;   // ovo_gemm_unwindow: win_per > 0 remaps C / add rows (see row_dest)
    // gemm_stream.hip, Hiera stages 1-2: A = LayerNorm (ln_mode 1) or plain cast (2) of the f32 rows of `ln_x` [*, ln_d], taken while the operand is
    // loaded -- product row m reads source row row_dest(m) (the win_* fields then describe the A side: window order -> spatial token, -1 = a
    // padding row of zeros) and C is written in product order; columns [ln_d, K) are zeros
    const float *ln_x, *ln_g, *ln_b;
    float ln_eps;
    int ln_d, ln_mode;
    uint16_t *qpool_out;       // with ln_mode, a window map and bf16 output: columns [0, qpool_cols) are NOT stored to C but 2 x 2 max-pooled over the
    int qpool_cols;            // window's tokens into qpool_out[pooled window-major row, qpool_cols] (Hiera's pooled queries, k_qpool)
    int pool_ww;               // with ln_mode and f32 output: > 0 = 2 x 2 max-pool of the window's tokens in the epilogue (window width), C rows = pooled spatial tokens
    int gelu_lut;              // gemm8p / gemm_stream: 1 = GELU (act 1) through the LDS table (default), 0 = the packed polynomial (OVO_GELU_POLY)
    int tail_wait;             // gemm8p: 1 = a wave waits for its epilogue stores before it ends (OVO_8P_TAILWAIT, measurement)
    int rope_lds;              // k_gemm8p: the rotary epilogue stages its table slice in LDS (OVO_8P_ROPE_LDS)
    int slab16;                // k_gemm8p: 2-byte outputs cross the epilogue's LDS slab already rounded (OVO_8P_NO_SLAB16: the f32 slab)
    int res_plain;             // k_gemm8p: the f32 + residual epilogue stores with the default cache policy instead of non-temporal (OVO_8P_RES_PLAIN, measurement)
    // k_gemm<128, 448> (gemm.hip, round 5): a workgroup owns whole rows (N = 448 = BN), so after the f32 result (+ residual) is stored the LayerNorm of the
    // rows that FOLLOWS the product in Hiera stage 3 (norm2 before the MLP) is taken from the accumulators: rln_out bf16 [C rows, rln_ld] = LN(C row)
    const float *rln_g, *rln_b; float rln_eps; uint16_t *rln_out; long long rln_ld;
    // LayerNorm FOLD (round 6, vit.hip; gemm8p only).  LN(x) . W^T + b = rstd (bf16(x) . W'^T - mean colsum(W')) + b' with W' = gamma . W and b' = b + W . beta:
    //   PRODUCER (the f32 += residual epilogue that writes x): xb_out = bf16(x row) beside the f32 row, and stat_out[part][row][2] = (sum, sum of squares) of
    //     the row's columns inside the 64-column wave tile `part` (N / 64 parts for either tile width; PART-major, stat_ld rows apart: a wave stores the 64 rows
    //     of a pass as one 512-byte run and the consumer's 256 threads read a part's 256 rows as four -- row-major [row][part] made every lane of both touch its
    //     own cache line: + 8 us per consumer launch at 16 K rows);
    //   CONSUMER (a 2-byte-output product whose A operand is xb): fold_stats / fold_parts = the producer's partials, fold_D = the row length they cover,
    //     fold_cs = column sums of W' (f32 [N]); `bias` = b'.  The epilogue forms mean / rstd per row once per tile (LDS) and applies the line above.
    uint16_t *xb_out; long long ld_xb;
    float *stat_out; long long stat_ld;
    const float *fold_stats; int fold_parts, fold_D; const float *fold_cs; float fold_eps;
    int dbg;                   // tools/ only (OVO_8P_DEBUG): 1 = leave before anything, 2 = leave after the prologue, 4 = no epilogue, 8 = epilogue without the 2-byte stores
    unsigned long long *stamps;   // tools/ only (OVO_8P_STAMPS = address of u64[tiles][4]): s_memrealtime at start / K-loop / epilogue / end
};

// ovo_gemm_unwindow: product row m is a token in window-major order (windows of wh x ww tiling an H x W grid that is padded up to
// whole windows); its C / add row is the token's spatial index (b*H + y)*W + x, or -1 for a padding position (row dropped).
__device__ __forceinline__ long long row_dest(const GemmArgs &g, int m) {
    if (g.win_per <= 0) return m;
    const int win = m / g.win_per, p = m - win * g.win_per;
    const int iy = p / g.win_ww, ix = p - iy * g.win_ww;
    const int b = win / g.win_nwin, wr = win - b * g.win_nwin;
    const int wy = wr / g.win_nww, wx = wr - wy * g.win_nww;
    const int y = wy * g.win_wh + iy, x = wx * g.win_ww + ix;
    return (y < g.win_H && x < g.win_W) ? ((long long)b * g.win_H + y) * g.win_W + x : -1;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <typename VT> struct Mfma;
template <> struct Mfma<bf16x8> {
    __device__ static f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static f32x16_t run32(bf16x8 a, bf16x8 b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<f16x8> {
    __device__ static f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static f32x16_t run32(f16x8 a, f16x8 b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <int BK> __device__ __forceinline__ int swz(int row) {
    return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

__device__ __forceinline__ void glds16(const void *src, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// GELU(x) = x/2 (1 + erf(x/sqrt2)) on PACKED f32 (v_pk_fma_f32: two values per VALU slot) with a polynomial erf and no
// transcendental op:  erf(z) ~ zc * P(s),  zc = clamp(z, +-3.5),  s = 2 zc^2 / 3.5^2 - 1,  P of degree 11 (Chebyshev-node
// weighted least squares, tools/ history in DESIGN.md).  |erf error| <= 1.9e-6, |GELU error| <= 8.6e-6 absolute for every
// x -- far below the bf16 rounding of the stored activation.  The library erff is ~50 branchy instructions and an
// exp/rcp form still pays two quarter-rate transcendentals per value; the GELU sits on the serial tail of every FC1
// tile (the epilogue does not overlap MFMA work), where it cost up to a quarter of the GEMM.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752f;
    const f32x2 zc = __builtin_elementwise_min(__builtin_elementwise_max(z, (f32x2)(-3.5f)), (f32x2)(3.5f));
    const f32x2 s = __builtin_elementwise_fma(zc * zc, (f32x2)(0.16326530612244897f), (f32x2)(-1.0f));
    f32x2 p = (f32x2)(-3.398861796e-03f);
    p = __builtin_elementwise_fma(p, s, (f32x2)(8.621919328e-03f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-8.698635955e-03f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(1.271555869e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-2.870869786e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(4.709060027e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-6.528488840e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(8.795614477e-02f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-1.145324569e-01f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(1.467849556e-01f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-2.007044758e-01f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(4.038725490e-01f));
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, p * zc, hx);
}

// GELU through a table in LDS (round 4).  The polynomial above is 19 packed instructions per PAIR of values -- and a packed f32 instruction
// issues at ~6.3 cycles with two waves per SIMD (tools/ubench.hip), i.e. ~60 cycles per value: the GELU of an FC1 epilogue cost 30-44 us of a
// 130-165 us launch (ViT / Hiera stage 3) and ~40 % of the streaming FC1s of Hiera stages 1-2.  Here Phi(x) = (1 + erf(x / sqrt 2)) / 2 is
// tabulated on [-6, 6) in steps of 1/64 as (value, difference to the next entry): 768 x 8 bytes, filled by the workgroup itself in its prologue
// (two erff per thread); a value costs clamp, fma, cvt, fract, shift-add, ds_read_b64, fma, mul = 7 plain VALU instructions + one LDS read.
// Linear interpolation error <= h^2 / 8 max|Phi''| = 7e-6; |GELU error| <= 9.1e-6 absolute over all x (the polynomial: 8.6e-6).
constexpr int GELU_LUT_N = 768;
constexpr int GELU_LUT_BYTES = GELU_LUT_N * 8;
// An 8-byte LDS store that the compiler's wait-count pass does not see.  A plain ds_write issued while LDS-DMA loads (buffer_load ... lds) are in flight gets
// `s_waitcnt vmcnt(0)` in front of it -- the pass cannot prove that it does not alias the DMA's destination -- which drains every stage a prologue has just
// prefetched (seen in k_gemm8p's ISA: the GELU table fill and the fold's (mean, rstd) rows, both behind the six prologue stages).  The caller orders the store
// itself: an `s_waitcnt lgkmcnt(0)` + barrier before the first read (the K-loop's own, for both users).
__device__ __forceinline__ void lds_store_b64_untracked(void *p, float a, float b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)p), "v"(v) : "memory");
}
template <bool UNTRACKED = false>
__device__ __forceinline__ void gelu_lut_fill(float2 *t, int tid, int nthreads) {
    for (int i = tid; i < GELU_LUT_N; i += nthreads) {
        const float x0 = (float)(i - GELU_LUT_N / 2) * (1.0f / 64.0f), x1 = (float)(i + 1 - GELU_LUT_N / 2) * (1.0f / 64.0f);
        const float p0 = 0.5f * (1.0f + erff(x0 * 0.70710678118654752f)), p1 = 0.5f * (1.0f + erff(x1 * 0.70710678118654752f));
        if constexpr (UNTRACKED) lds_store_b64_untracked(t + i, p0, p1 - p0);
        else t[i] = make_float2(p0, p1 - p0);
    }
}
__device__ __forceinline__ float gelu_lut(float x, const float2 *t) {
    const float u = fmaf(__builtin_amdgcn_fmed3f(x, -6.0f, 5.984375f), 64.0f, 384.0f);      // [0, 767]
    const int i = (int)u;
    const float2 e = t[i];
    return x * fmaf(__builtin_amdgcn_fractf(u), e.y, e.x);
}

__device__ __forceinline__ void act4(float *v, int act, const float2 *lut = nullptr) {
    if (act == 1 && lut) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_lut(v[r], lut);
    } else if (act == 1) {
        const f32x2 a = gelu2(f32x2{v[0], v[1]}), b = gelu2(f32x2{v[2], v[3]});
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else if (act == 2) {                                       // QuickGELU x * sigmoid(1.702 x)   (open_clip "-qg" cards)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
    } else if (act == 3) {                                       // ReLU (SAM2 decoder MLPs)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (act == 4) {                                       // sigmoid (SAM2 IoU head)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + __expf(-v[r]));
    } else if (act == 5) {                                       // GELU, tanh form (SigLIP towers): x * sigmoid(2 sqrt(2/pi) (x + 0.044715 x^3))
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u = 1.5957691216057308f * fmaf(0.044715f * v[r] * v[r], v[r], v[r]);
            v[r] = v[r] / (1.0f + __expf(-u));
        }
    }
}

// two f32 -> packed bf16 (round-to-nearest-even): v_cvt_pk_bf16_f32 on gfx950, one instruction per pair
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const bf16x2_t h = __builtin_convertvector(f32x2{a, b}, bf16x2_t);
    return *(const uint32_t *)&h;
}
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *(const uint32_t *)&h;
}

// One 4-column group of the epilogue (shared by the kernels): v = act(alpha * acc + bias) -> rotary embedding -> + residual ->
// running first-max argmax and/or the store of 4 consecutive columns of C row `mdst` (8 / 16 bytes).
// The arithmetic of one 4-column group of the epilogue: v = act(alpha * acc + bias) -> rotary embedding -> + residual.
// row of `add` that product row m (destination row md) reads
__device__ __forceinline__ long long add_row(const GemmArgs &g, int m, long long md) { return g.add_rows > 0 ? (long long)(m % g.add_rows) : md; }

// `tok` = m % rope_T and `nh` = n % rope_hd (only read with rope_cos set): callers that walk rows keep them incrementally -- an integer
// division by a run-time value is ~35 VALU instructions, once per 4 outputs it was a third of the QKV epilogue.
__device__ __forceinline__ void math4(const GemmArgs &g, int tok, int nh, int n, float (&v)[4], float4 bias, float4 addv, const float2 *lut = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
    v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
    if (g.act) act4(v, g.act, lut);
    if (g.rope_cos && n < g.rope_cols && tok >= g.rope_t0) {
        // rotary embedding of the (2i, 2i+1) pairs this lane holds: row = token m % T, column within the head n % hd
        const long long at = (long long)tok * g.rope_hd + nh;
        const float4 c = *(const float4 *)(g.rope_cos + at), sn = *(const float4 *)(g.rope_sin + at);
        const float y0 = v[0] * c.x - v[1] * sn.x, y1 = v[1] * c.y + v[0] * sn.y;
        const float y2 = v[2] * c.z - v[3] * sn.z, y3 = v[3] * c.w + v[2] * sn.w;
        v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
    }
    if (g.add) { v[0] += addv.x; v[1] += addv.y; v[2] += addv.z; v[3] += addv.w; }
}
__device__ __forceinline__ void math4(const GemmArgs &g, int m, int n, float (&v)[4], float4 bias, float4 addv) {
    math4(g, g.rope_cos ? m % g.rope_T : 0, g.rope_cos ? n % g.rope_hd : 0, n, v, bias, addv);
}

// One 4-column group of the epilogue straight from the accumulators (a lane owns 4 consecutive columns of one row): math4, then the
// running first-max argmax and/or the 8 / 16-byte store into C row `mdst`.
__device__ __forceinline__ void finish4(const GemmArgs &g, int m, long long mdst, int n, f32x4 a, float4 bias, float4 addv,
                                        float &row_best, int &row_arg) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    math4(g, m, n, v, bias, addv);
    if (g.best) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < g.n_valid && v[r] > row_best) { row_best = v[r]; row_arg = n + r; }   // ascending columns: ties keep the first
        if (!g.store) return;
    }
    if (g.out_dtype == 0) {
        *(float4 *)((float *)g.C + mdst * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 p;
        if (g.out_dtype == 2) { p.x = pack_bf16(v[0], v[1]); p.y = pack_bf16(v[2], v[3]); }
        else { p.x = pack_f16(v[0], v[1]); p.y = pack_f16(v[2], v[3]); }
        *(uint2 *)((uint16_t *)g.C + mdst * g.ldc + n) = p;
    }
}

// per-row close of the fused argmax: the row's columns of a wave tile sit in the 4 lanes that share lane & 15
__device__ __forceinline__ void finish_best(const GemmArgs &g, int m, int fq, float row_best, int row_arg) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float ob = __shfl_xor(row_best, o, 64);
        const int oa = __shfl_xor(row_arg, o, 64);
        if (ob > row_best || (ob == row_best && oa < row_arg)) { row_best = ob; row_arg = oa; }
    }
    if (fq == 0 && row_arg != 0x7fffffff) {
        uint32_t u = __float_as_uint(row_best);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        atomicMax(g.best + m, ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (uint32_t)row_arg));
    }
}

// variant of a product as the profiler and the tuned-tile table (gemm_tuned.h) see it: bit 0 f32 output, 1 residual add, 2-3 activation, 4 rotary epilogue,
// 5 window remap of the output / residual rows, 6 f32 A with LayerNorm / cast in the load, 7 fused argmax, 8 periodic residual
inline int gemm_flags(const GemmArgs &g) {
    return (g.out_dtype == 0 ? 1 : 0) | (g.add ? 2 : 0) | ((g.act & 3) << 2) | (g.rope_cos ? 16 : 0) | (g.win_per > 0 ? 32 : 0) | (g.ln_mode ? 64 : 0) |
           (g.best ? 128 : 0) | (g.add_rows > 0 ? 256 : 0) | (g.xb_out ? 512 : 0) | (g.fold_stats ? 1024 : 0);
}

// algorithmic HBM bytes of a product: every operand read once, the result written once (what PMC traffic is compared with, bench.py roofline)
inline double gemm_algorithmic_bytes(const GemmArgs &g) {
    const double mn = (double)g.M * g.N, a_bytes = g.ln_mode ? (double)g.M * g.ln_d * 4.0 : (double)g.M * g.K * 2.0;
    return a_bytes + (double)g.N * g.K * 2.0 + mn * (g.out_dtype == 0 ? 4.0 : 2.0) + (g.add ? (g.add_rows > 0 ? (double)g.add_rows * g.N * 4.0 : mn * 4.0) : 0.0) +
           (g.bias ? g.N * 4.0 : 0.0) +
           (g.xb_out ? mn * 2.0 + (double)g.M * (g.N / 64) * 8.0 : 0.0) +                 // LayerNorm fold, producer: the bf16 copy + the partial statistics
           (g.fold_stats ? (double)g.M * g.fold_parts * 8.0 + g.N * 4.0 : 0.0);           // consumer: the partials of its rows (once per row block of tiles at least) + row sums
}

// launch of the 256-row ping-pong kernels (gemm8p.hip); bn in {128, 256}; returns OVO_E_UNSUPPORTED when the shape does not fit
int gemm8p_launch(const GemmArgs &g, int bn, int in_dtype, hipStream_t s);

// The two halves of the LayerNorm fold (GemmArgs: xb_out ... fold_eps), as vit.hip calls them; both run on the ping-pong kernel or return OVO_E_UNSUPPORTED
// (nothing launched: the caller takes the LayerNorm + plain product path).
struct FoldOut { void *xb; long long ld_xb; float *stats; long long ld_stats; };        // producer: bf16 copy of the result rows, partial statistics [N / 64][ld_stats][2]
struct FoldIn { const float *stats; long long ld_stats; int parts; int D; const float *colsum; float eps; };
int gemm_fold_parts(int N);                                        // partial statistics per row a producer of width N writes: one per 64-column wave tile
bool gemm_fold_ok(int M, int N, int K);                            // the shape can run on the ping-pong kernel (either half of the fold)
int gemm_fold_producer(const ovo_gemm_t *p, const FoldOut &o, ovo_stream_t stream);
int gemm_fold_consumer(const ovo_gemm_t *p, const ovo_rope_t *rope, const FoldIn &f, ovo_stream_t stream);
int gemm_fold_rowstats(const float *x, long long ldx, int M, int D, void *xb, long long ld_xb, float *stats, ovo_stream_t stream);   // f32 rows -> bf16 copy + one partial per row
// the persistent 256 x 128 form (gemm8q.hip: one DMA ring across a workgroup's tiles, epilogue of tile j behind the K-loop of tile j + 1)
int gemm8q_launch(const GemmArgs &g, int in_dtype, hipStream_t s);
// the weights-resident streaming form for tall short-K products (gemm_stream.hip); OVO_E_UNSUPPORTED when the shape has no instantiation
int gemm_stream_launch(const GemmArgs &g, int in_dtype, hipStream_t s);
// the same kernel with the A operand taken from an f32 tensor through LayerNorm (mode 1) or a cast (mode 2) -- hiera.hip's way around
// k_ln_window / k_cast_pad for the layers the streaming form covers; OVO_E_UNSUPPORTED otherwise (nothing launched)
int gemm_f32a_stream(const ovo_gemm_t *p, const ovo_window_t *win, const float *x, int d, const float *gamma, const float *beta, float eps, int mode,
                     int pool2x2, ovo_stream_t stream, uint16_t *qpool_out = nullptr, int qpool_cols = 0);

// x f32 [rows, d] += fc2(GELU(fc1(LayerNorm(x)))) in ONE launch, the hidden row never leaving registers (mlp_stream.hip); OVO_E_UNSUPPORTED when the
// widths have no instantiation or rows < 16384 (nothing launched: the caller runs the two products)
int mlp_stream_launch(float *x, long long rows, int d, const float *ln_g, const float *ln_b, float eps, const void *w1, long long ldw1, const float *b1,
                      int hid, const void *w2, long long ldw2, const float *b2, hipStream_t s);

// ovo_gemm_unwindow (f32 C += residual, rows re-ordered from window order) that ALSO writes ln_out bf16 [C rows, ld_ln] = LayerNorm(C row; g, b, eps): the
// full-row tile of gemm.hip.  OVO_E_UNSUPPORTED (nothing launched) unless N = 448, K % 64 == 0, f32 output
int gemm_unwindow_rowln(const ovo_gemm_t *p, const ovo_window_t *win, const float *ln_g, const float *ln_b, float eps, void *ln_out, long long ld_ln,
                        ovo_stream_t stream);

// att bf16 [windows x 64 (16 with `pool`), ld_att] (window-major rows) = per-window, per-head softmax(q k^T) v of q | k | v = LayerNorm(x) . Wqkv^T + b, straight
// from the f32 token grid x [B, H, W, d] (winattn.hip; one launch per pair of heads); OVO_E_UNSUPPORTED (nothing launched) unless >= 512 windows and
// d = 112 with 8 x 8 windows and (d_out, heads, pool) = (112, 2, 0) or (224, 4, 1: queries 2 x 2 max-pooled inside the window), or d = d_out = 224 with
// 4 x 4 windows and 4 heads (k_win_attn224; the same contract as ovo_hip.h's ovo_win_attn)
int win_attn_launch(const float *x, int B, int H, int W, int ws, int d, int d_out, int heads, int pool, const float *ln_g, const float *ln_b, float eps,
                    const void *qkv_w, long long ldw, const float *qkv_b, void *att, int ld_att, hipStream_t s);

}  // namespace ovo_gemm_detail
