// query.hip -- text-similarity query over instance descriptors or dense per-point accumulators
// (clip_utils.py:10-19, ovo.py:487-491).  Small-Q form: HBM-bound streaming of F, text matrix in LDS.
// Large Q (BASELINE.json config 5, Q = 1000) is a GEMM and goes through gemm.hip.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int QC = 16;   // queries per LDS chunk

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

template <int DT>
__global__ void __launch_bounds__(256) k_similarity(const void *__restrict__ F, int64_t n, int D, const float *__restrict__ T, int Q,
                                                    const int32_t *__restrict__ cnt, int siglip, float scale_exp, float bias,
                                                    float th, float *__restrict__ out_sim, long long *__restrict__ out_cls,
                                                    float *__restrict__ out_conf) {
    extern __shared__ float sT[];                      // [QC][D]
    using L = Loader<DT>;
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t rows_per_wave_iter = waves;
    const int DV = D / L::VEC;                          // vector chunks per row

    for (int q0 = 0; q0 < Q; q0 += QC) {
        const int qn = Q - q0 < QC ? Q - q0 : QC;
        __syncthreads();
        for (int i = threadIdx.x; i < QC * D; i += blockDim.x) sT[i] = i < qn * D ? T[(int64_t)q0 * D + i] : 0.f;
        __syncthreads();
        for (int64_t r = wave0; r < n; r += rows_per_wave_iter) {
            float acc[QC];
#pragma unroll
            for (int q = 0; q < QC; ++q) acc[q] = 0.f;
            for (int c = lane; c < DV; c += 64) {
                float v[L::VEC];
                L::load(F, r * D + (int64_t)c * L::VEC, v);
#pragma unroll
                for (int q = 0; q < QC; ++q) {
                    const float *t = sT + q * D + c * L::VEC;
#pragma unroll
                    for (int e = 0; e < L::VEC; ++e) acc[q] = fmaf(v[e], t[e], acc[q]);
                }
            }
            for (int k = DV * L::VEC + lane; k < D; k += 64) {      // tail when D % VEC != 0
                const float v = L::one(F, r * D + k);
#pragma unroll
                for (int q = 0; q < QC; ++q) acc[q] = fmaf(v, sT[q * D + k], acc[q]);
            }
            const float rs = cnt ? (cnt[r] > 0 ? 1.0f / (float)cnt[r] : 0.f) : 1.0f;
            float mine = 0.f;
#pragma unroll
            for (int q = 0; q < QC; ++q) {
                float s = wave_sum(acc[q]) * rs;
                if (siglip) s = 1.0f / (1.0f + __expf(-(s * scale_exp + bias)));
                if (lane == q) mine = s;
            }
            if (out_sim && lane < qn) out_sim[r * Q + q0 + lane] = mine;
            if (out_cls || out_conf) {
                // first-max argmax over this chunk, merged with previous chunks through out_conf/out_cls
                float best = lane < qn ? mine : -3.0e38f;
                int arg = lane < qn ? q0 + lane : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
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

}  // namespace

extern "C" int ovo_similarity(const void *F, int feat_dtype, int64_t n, int D, const float *T, int Q, const int32_t *cnt,
                              int siglip, float logit_scale, float logit_bias, float th, float *out_sim,
                              int64_t *out_cls, float *out_conf, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0 && D > 0 && Q > 0, "bad shape");
    OVO_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "feat_dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    OVO_REQUIRE((out_cls == nullptr) == (out_conf == nullptr), "out_cls and out_conf go together");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(F && T, "null pointer");
    OVO_REQUIRE(((uintptr_t)F & 15) == 0 && (D * (feat_dtype == 0 ? 4 : 2)) % 16 == 0, "F rows must be 16-byte aligned");
    const size_t lds = (size_t)QC * D * sizeof(float);
    OVO_REQUIRE(lds <= 160 * 1024, "D too large for the LDS text tile");
    hipStream_t s = (hipStream_t)stream;
    const int grid = ovo_grid((n + 3) / 4 * 256, 256, 256 * 4);
    const float se = expf(logit_scale);
    long long *cls = (long long *)out_cls;
#define LAUNCH(DT)                                                                                                  \
    do {                                                                                                            \
        if (lds > 64 * 1024) OVO_HIP(hipFuncSetAttribute((const void *)k_similarity<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_similarity<DT><<<grid, 256, lds, s>>>(F, n, D, T, Q, cnt, siglip, se, logit_bias, th, out_sim, cls, out_conf); \
    } while (0)
    if (feat_dtype == 0) LAUNCH(0);
    else if (feat_dtype == 1) LAUNCH(1);
    else LAUNCH(2);
#undef LAUNCH
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
