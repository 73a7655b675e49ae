// query.hip -- text-similarity query over instance descriptors or dense per-point accumulators
// (clip_utils.py:10-19, ovo.py:487-491).  Small-Q form: HBM-bound streaming of F with the text matrix in LDS.
// Large Q (BASELINE.json config 5, Q = 1000) is a GEMM and goes through gemm.hip.
//
// Each wave owns R = 4 consecutive rows per iteration: per 16-byte column chunk it issues 4 independent global
// loads (one per row, 1 KiB per wave-instruction, fully coalesced) and reuses the QC text-vector chunks it
// reads from LDS for all 4 rows, so LDS traffic and load latency are amortised 4x; the chunk loop is unrolled
// so ~16 loads per lane are in flight.  Epilogue per row: 1/cnt scale, SigLIP sigmoid, argmax + threshold.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include "common.h"

namespace {

template <int DT> struct Loader;
template <> struct Loader<0> {   // f32: 4 values per 16-byte load
    static constexpr int VEC = 4;
    __device__ static void load(const void *base, int64_t elem, float *v) {
        const float4 x = *(const float4 *)((const float *)base + elem);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    }
    __device__ static float one(const void *base, int64_t elem) { return ((const float *)base)[elem]; }
};
template <> struct Loader<1> {   // f16: 8 values per 16-byte load
    static constexpr int VEC = 8;
    __device__ static void load(const void *base, int64_t elem, float *v) {
        const uint4 x = *(const uint4 *)((const __half *)base + elem);
        const __half2 *h = (const __half2 *)&x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); v[2 * i] = f.x; v[2 * i + 1] = f.y; }
    }
    __device__ static float one(const void *base, int64_t elem) { return __half2float(((const __half *)base)[elem]); }
};
template <> struct Loader<2> {   // bf16
    static constexpr int VEC = 8;
    __device__ static void load(const void *base, int64_t elem, float *v) {
        const uint4 x = *(const uint4 *)((const uint16_t *)base + elem);
        const uint32_t *w = (const uint32_t *)&x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ static float one(const void *base, int64_t elem) { return __uint_as_float((uint32_t)((const uint16_t *)base)[elem] << 16); }
};

constexpr int R = 4;     // rows per wave iteration

template <int DT, int QC>
__global__ void __launch_bounds__(256) k_similarity(const void *__restrict__ F, int64_t n, int D, const float *__restrict__ T, int Q,
                                                    const int32_t *__restrict__ cnt, int siglip, float scale_exp, float bias,
                                                    float th, float *__restrict__ out_sim, long long *__restrict__ out_cls,
                                                    float *__restrict__ out_conf) {
    extern __shared__ __attribute__((aligned(16))) float sT[];                      // [QC][D]
    using L = Loader<DT>;
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int DV = D / L::VEC;                          // vector chunks per row
    const int64_t groups = (n + R - 1) / R;

    for (int q0 = 0; q0 < Q; q0 += QC) {
        const int qn = Q - q0 < QC ? Q - q0 : QC;
        __syncthreads();
        for (int i = threadIdx.x; i < QC * D; i += blockDim.x) sT[i] = i < qn * D ? T[(int64_t)q0 * D + i] : 0.f;
        __syncthreads();
        for (int64_t grp = wave0; grp < groups; grp += waves) {
            const int64_t r0 = grp * R;
            int64_t row[R];
#pragma unroll
            for (int j = 0; j < R; ++j) row[j] = r0 + j < n ? r0 + j : n - 1;      // clamp: tail rows recompute the last row
            float acc[R][QC];
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int q = 0; q < QC; ++q) acc[j][q] = 0.f;
#pragma unroll 2
            for (int c = lane; c < DV; c += 64) {
                float v[R][L::VEC];
#pragma unroll
                for (int j = 0; j < R; ++j) L::load(F, row[j] * D + (int64_t)c * L::VEC, v[j]);
#pragma unroll
                for (int q = 0; q < QC; ++q) {
                    float t[L::VEC];
#pragma unroll
                    for (int e = 0; e < L::VEC; e += 4) {
                        const float4 tt = *(const float4 *)(sT + q * D + c * L::VEC + e);
                        t[e] = tt.x; t[e + 1] = tt.y; t[e + 2] = tt.z; t[e + 3] = tt.w;
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j)
#pragma unroll
                        for (int e = 0; e < L::VEC; ++e) acc[j][q] = fmaf(v[j][e], t[e], acc[j][q]);
                }
            }
            for (int k = DV * L::VEC + lane; k < D; k += 64) {      // tail when D % VEC != 0
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float v = L::one(F, row[j] * D + k);
#pragma unroll
                    for (int q = 0; q < QC; ++q) acc[j][q] = fmaf(v, sT[q * D + k], acc[j][q]);
                }
            }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int64_t r = r0 + j;
                if (r >= n) break;                                                 // wave-uniform
                const float rs = cnt ? (cnt[r] > 0 ? 1.0f / (float)cnt[r] : 0.f) : 1.0f;
                float mine = 0.f;
#pragma unroll
                for (int q = 0; q < QC; ++q) {
                    float s = wave_sum(acc[j][q]) * rs;
                    if (siglip) s = 1.0f / (1.0f + __expf(-(s * scale_exp + bias)));
                    if (lane == q) mine = s;
                }
                if (out_sim && lane < qn) out_sim[r * Q + q0 + lane] = mine;
                if (out_cls) {
                    // first-max argmax over this chunk, merged with previous chunks through out_conf/out_cls
                    float best = lane < qn ? mine : -3.0e38f;
                    int arg = lane < qn ? q0 + lane : 0x7fffffff;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {                              // QC <= 16: lanes 0..15 hold the scores
                        const float ob = __shfl_xor(best, o, 64);
                        const int oa = __shfl_xor(arg, o, 64);
                        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
                    }
                    if (lane == 0) {
                        if (q0 > 0) {
                            const float pb = out_conf[r];
                            if (!(best > pb)) { best = pb; arg = (int)out_cls[r]; }
                        }
                        const bool last = q0 + QC >= Q;
                        if (last && best <= th) { best = 0.f; arg = -1; }
                        out_conf[r] = best;
                        out_cls[r] = arg;
                    }
                }
            }
        }
    }
}

template <int DT, int QC>
int launch(const void *F, int64_t n, int D, const float *T, int Q, const int32_t *cnt, int siglip, float se, float bias, float th,
           float *out_sim, long long *cls, float *conf, hipStream_t s) {
    const size_t lds = (size_t)QC * D * sizeof(float);
    if (lds > 160 * 1024) { ovo_set_error("ovo_similarity: D too large for the LDS text tile"); return OVO_E_ARG; }
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute((const void *)k_similarity<DT, QC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            ovo_set_error("ovo_similarity: hipFuncSetAttribute failed");
            return OVO_E_LAUNCH;
        }
        attr_done = true;
    }
    const int64_t groups = (n + R - 1) / R;
    const int per_cu = lds > 0 ? (int)((160 * 1024) / lds) : 4;          // resident blocks per CU by LDS
    int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    if ((int64_t)grid * 4 > groups) grid = (int)((groups + 3) / 4);
    if (grid < 1) grid = 1;
    k_similarity<DT, QC><<<grid, 256, lds, s>>>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf);
    return OVO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// MFMA form (D % 16 == 0): S^T[q][row] += T[q][k] F[row][k] on v_mfma_f32_16x16x4_f32 -- exact f32 (bitwise an fmaf
// chain), at the f32 vector rate, and NO cross-lane reduction: the k-sum happens inside the MFMA.  A wave owns 16
// rows; lane (row = lane&15, g = lane>>4) streams 32-byte pieces F[row][k0 + 8g .. +8) (4 lanes cover one 128-byte
// line per row) and pairs them with the same k of T from LDS (the k index of an MFMA is free as long as A and B
// agree).  Accumulator register r of lane (row, g) is S[q = 4g + r][row]: the argmax is 3 in-lane compares + two
// xor-shuffles.  HBM-bound: D*s bytes per row.
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DT> __device__ __forceinline__ void load8(const void *F, int64_t elem, float *v) {
    if (DT == 0) { Loader<0>::load(F, elem, v); Loader<0>::load(F, elem + 4, v + 4); }
    else Loader<DT>::load(F, elem, v);
}
template <int DT> __device__ __forceinline__ void load4(const void *F, int64_t elem, float *v) {
    if (DT == 0) Loader<0>::load(F, elem, v);
    else { for (int e = 0; e < 4; ++e) v[e] = Loader<DT>::one(F, elem + e); }
}

template <int DT>
__global__ void __launch_bounds__(256) k_similarity_mfma(const void *__restrict__ F, int64_t n, int D, const float *__restrict__ T, int Q,
                                                         const int32_t *__restrict__ cnt, int siglip, float scale_exp, float bias,
                                                         float th, float *__restrict__ out_sim, long long *__restrict__ out_cls,
                                                         float *__restrict__ out_conf, const int32_t *__restrict__ row_list = nullptr,
                                                         const int32_t *__restrict__ n_list = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sT[];                      // [min(Q, 16)][D]
    const int lane = threadIdx.x & 63, rr = lane & 15, g = lane >> 4;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // ovo_similarity_rows: only the rows named in row_list[0 .. *n_list) are evaluated (count read on the device: no host sync);
    // outputs stay indexed by the row itself, so a resident class / confidence map is patched in place
    if (row_list) { const int64_t m = *n_list; n = m < n ? m : n; }
    const int64_t groups = (n + 15) / 16;
    const int D32 = D & ~31;

    for (int q0 = 0; q0 < Q; q0 += 16) {
        const int qn = Q - q0 < 16 ? Q - q0 : 16;
        __syncthreads();
        for (int i = threadIdx.x; i < qn * D; i += blockDim.x) sT[i] = T[(int64_t)q0 * D + i];
        __syncthreads();
        // this lane's text row (A operand: i = q); lanes past the last query re-read it -- their output rows are never used
        const float *tq = sT + (rr < qn ? rr : qn - 1) * D;
        for (int64_t grp = wave0; grp < groups; grp += waves) {
            int64_t row = grp * 16 + rr;
            const bool live = row < n;
            if (!live) row = n - 1;
            if (row_list) row = row_list[row];
            const int64_t base = row * D;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k0 = 0; k0 < D32; k0 += 32) {
                float f[8];
                float4 t0, t1;
                if (DT == 0) {
                    // f32 rows: the two 16-byte loads of a lane are 64 B apart, so ONE load instruction covers a contiguous
                    // 64-byte half of each row's 128-byte line (4 lanes x 16 B) instead of four 16-byte pieces 32 B apart; the
                    // k index of an MFMA is free as long as both operands agree on it
                    load4<DT>(F, base + k0 + g * 4, f);
                    load4<DT>(F, base + k0 + 16 + g * 4, f + 4);
                    t0 = *(const float4 *)(tq + k0 + g * 4);
                    t1 = *(const float4 *)(tq + k0 + 16 + g * 4);
                } else {
                    load8<DT>(F, base + k0 + g * 8, f);
                    t0 = *(const float4 *)(tq + k0 + g * 8);
                    t1 = *(const float4 *)(tq + k0 + g * 8 + 4);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.x, f[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.y, f[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.z, f[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.w, f[3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.x, f[4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.y, f[5], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.z, f[6], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t1.w, f[7], acc, 0, 0, 0);
            }
            if (D32 < D) {                                                           // one 16-wide tail step
                float f[4];
                load4<DT>(F, base + D32 + g * 4, f);
                const float4 t0 = *(const float4 *)(tq + D32 + g * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.x, f[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.y, f[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.z, f[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t0.w, f[3], acc, 0, 0, 0);
            }
            // lane (row rr, g): acc[r] = S[q0 + 4g + r][row]
            const float rs = cnt ? (cnt[row] > 0 ? 1.0f / (float)cnt[row] : 0.f) : 1.0f;
            float best = -3.0e38f;
            int arg = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 4 * g + r;
                float s = acc[r] * rs;
                if (siglip) s = 1.0f / (1.0f + __expf(-(s * scale_exp + bias)));
                if (q < qn) {
                    if (out_sim && live) out_sim[row * Q + q0 + q] = s;
                    if (s > best) { best = s; arg = q0 + q; }
                }
            }
            if (out_cls) {
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oa = __shfl_xor(arg, o, 64);
                    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
                }
                if (g == 0 && live) {
                    if (q0 > 0) {
                        const float pb = out_conf[row];
                        if (!(best > pb)) { best = pb; arg = (int)out_cls[row]; }
                    }
                    const bool last = q0 + 16 >= Q;
                    if (last && best <= th) { best = 0.f; arg = -1; }
                    out_conf[row] = best;
                    out_cls[row] = arg;
                }
            }
        }
    }
}

template <int DT>
int launch_mfma(const void *F, int64_t n, int D, const float *T, int Q, const int32_t *cnt, int siglip, float se, float bias, float th,
                float *out_sim, long long *cls, float *conf, hipStream_t s) {
    const size_t lds = (size_t)(Q < 16 ? Q : 16) * D * sizeof(float);
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute((const void *)k_similarity_mfma<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            ovo_set_error("ovo_similarity: hipFuncSetAttribute failed");
            return OVO_E_LAUNCH;
        }
        attr_done = true;
    }
    const int64_t groups = (n + 15) / 16;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    if (const char *e = getenv("OVO_SIM_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }
    int64_t grid = 256 * per_cu;
    if (grid * 4 > groups) grid = (groups + 3) / 4;
    k_similarity_mfma<DT><<<(int)(grid < 1 ? 1 : grid), 256, lds, s>>>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf);
    return OVO_OK;
}

template <int DT>
int dispatch(const void *F, int64_t n, int D, const float *T, int Q, const int32_t *cnt, int siglip, float se, float bias, float th,
             float *out_sim, long long *cls, float *conf, hipStream_t s) {
    if (D % 16 == 0 && (size_t)16 * D * sizeof(float) <= 160 * 1024 && !getenv("OVO_SIM_VALU"))
        return launch_mfma<DT>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf, s);
    // smallest query chunk that covers Q in one pass (fewer wasted FMAs / LDS bytes), else 16-wide passes
    if (Q <= 4) return launch<DT, 4>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf, s);
    if (Q <= 8) return launch<DT, 8>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf, s);
    if (Q <= 12) return launch<DT, 12>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf, s);
    return launch<DT, 16>(F, n, D, T, Q, cnt, siglip, se, bias, th, out_sim, cls, conf, s);
}

// ---------------------------------------------------------------------------------------------------------
// Large vocabularies (Q in the hundreds .. thousands, BASELINE config 5): the score matrix comes from the f16/bf16
// MFMA GEMM (ovo_gemm, S = F . T^T) and this pass finishes it -- optional SigLIP epilogue in place, then
// max / first-argmax / threshold per row.  One wave per row, 16-byte loads; HBM-bound on 4*Q bytes per row.
__global__ void __launch_bounds__(256) k_row_argmax(float *__restrict__ S, int64_t n, int Q, int siglip, float scale_exp, float bias, float th,
                                                    long long *__restrict__ out_cls, float *__restrict__ out_conf) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < n; row += waves) {
        float *s = S + row * Q;
        float best = -3.0e38f;
        int arg = 0x7fffffff;
        for (int q = lane * 4; q < Q; q += 256) {                                    // Q % 4 == 0 (ovo_gemm's N constraint)
            float4 v = *(const float4 *)(s + q);
            if (siglip) {
                v.x = 1.0f / (1.0f + __expf(-(v.x * scale_exp + bias)));
                v.y = 1.0f / (1.0f + __expf(-(v.y * scale_exp + bias)));
                v.z = 1.0f / (1.0f + __expf(-(v.z * scale_exp + bias)));
                v.w = 1.0f / (1.0f + __expf(-(v.w * scale_exp + bias)));
                *(float4 *)(s + q) = v;
            }
            if (v.x > best) { best = v.x; arg = q; }
            if (v.y > best) { best = v.y; arg = q + 1; }
            if (v.z > best) { best = v.z; arg = q + 2; }
            if (v.w > best) { best = v.w; arg = q + 3; }
        }
        if (out_cls) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oa = __shfl_xor(arg, o, 64);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            if (lane == 0) {
                if (best <= th) { best = 0.f; arg = -1; }
                out_conf[row] = best;
                out_cls[row] = arg;
            }
        }
    }
}

}  // namespace

extern "C" int ovo_row_argmax(float *S, int64_t n, int Q, int siglip, float logit_scale, float logit_bias, float th,
                              int64_t *out_cls, float *out_conf, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && Q > 0 && Q % 4 == 0, "Q must be a positive multiple of 4");
    OVO_REQUIRE((out_cls == nullptr) == (out_conf == nullptr), "out_cls and out_conf go together");
    if (n == 0 || (!siglip && !out_cls)) return OVO_OK;
    OVO_REQUIRE(S && ((uintptr_t)S & 15) == 0, "S must be 16-byte aligned");
    int64_t grid = (n + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
    k_row_argmax<<<(int)grid, 256, 0, (hipStream_t)stream>>>(S, n, Q, siglip, expf(logit_scale), logit_bias, th, (long long *)out_cls, out_conf);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// The touched-rows form of the dense query: S is evaluated only for rows[0 .. *n_rows) (a device-side count, e.g. the list
// ovo_scatter_accum_touched emits) and out_cls / out_conf -- indexed by the row id, resident across keyframes -- are patched in place.
// A row's arithmetic is the one ovo_similarity performs, so the patched map is bit-identical to a full re-query.
extern "C" int ovo_similarity_rows(const void *F, int feat_dtype, const int32_t *rows, const int32_t *n_rows, int64_t max_rows, int D,
                                   const float *T, int Q, const int32_t *cnt, int siglip, float logit_scale, float logit_bias, float th,
                                   int64_t *out_cls, float *out_conf, ovo_stream_t stream) {
    OVO_REQUIRE(max_rows >= 0 && D > 0 && Q > 0, "bad shape");
    OVO_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "feat_dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    if (max_rows == 0) return OVO_OK;
    OVO_REQUIRE(F && T && rows && n_rows && out_cls && out_conf, "null pointer");
    OVO_REQUIRE(((uintptr_t)F & 15) == 0 && D % 16 == 0 && (size_t)16 * D * sizeof(float) <= 160 * 1024, "needs D % 16 == 0, D <= 2560 and 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    const float se = expf(logit_scale);
    const size_t lds = (size_t)(Q < 16 ? Q : 16) * D * sizeof(float);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = 256 * per_cu;
    const int64_t groups = (max_rows + 15) / 16;
    if (grid * 4 > groups) grid = (groups + 3) / 4;
    if (grid < 1) grid = 1;
#define OVO_ROWS_GO(DT)                                                                                                                      \
    do {                                                                                                                                     \
        if (lds > 64 * 1024) OVO_HIP(hipFuncSetAttribute((const void *)k_similarity_mfma<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_similarity_mfma<DT><<<(int)grid, 256, lds, s>>>(F, max_rows, D, T, Q, cnt, siglip, se, logit_bias, th, nullptr, (long long *)out_cls, \
                                                          out_conf, rows, n_rows);                                                           \
    } while (0)
    if (feat_dtype == 0) OVO_ROWS_GO(0);
    else if (feat_dtype == 1) OVO_ROWS_GO(1);
    else OVO_ROWS_GO(2);
#undef OVO_ROWS_GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

extern "C" int ovo_similarity(const void *F, int feat_dtype, int64_t n, int D, const float *T, int Q, const int32_t *cnt,
                              int siglip, float logit_scale, float logit_bias, float th, float *out_sim,
                              int64_t *out_cls, float *out_conf, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && D > 0 && Q > 0, "bad shape");
    OVO_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "feat_dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    OVO_REQUIRE((out_cls == nullptr) == (out_conf == nullptr), "out_cls and out_conf go together");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(F && T, "null pointer");
    OVO_REQUIRE(((uintptr_t)F & 15) == 0 && (D * (feat_dtype == 0 ? 4 : 2)) % 16 == 0 && D % 4 == 0, "F rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float se = expf(logit_scale);
    long long *cls = (long long *)out_cls;
    int rc;
    if (feat_dtype == 0) rc = dispatch<0>(F, n, D, T, Q, cnt, siglip, se, logit_bias, th, out_sim, cls, out_conf, s);
    else if (feat_dtype == 1) rc = dispatch<1>(F, n, D, T, Q, cnt, siglip, se, logit_bias, th, out_sim, cls, out_conf, s);
    else rc = dispatch<2>(F, n, D, T, Q, cnt, siglip, se, logit_bias, th, out_sim, cls, out_conf, s);
    if (rc != OVO_OK) return rc;
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
