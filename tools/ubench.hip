// tools/ubench.hip -- instruction-cost probes behind the attention / epilogue designs (gfx950): issue cost per wave and throughput per SIMD of
// the VALU ops of an online softmax (v_exp_f32, v_fma_f32, packed f32, v_max3_f32, v_cvt_pk_bf16_f32, permlane swaps) alone and beside
// MFMAs, at 1 / 2 / 4 waves per SIMD.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP32(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) \
                 M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

enum { T_EXP, T_FMA, T_PKFMA, T_MAX3, T_CVT, T_PKMUL, T_ADD, T_PKADD, T_SWAP32, T_MFMA16, T_MFMA32, T_MFMA16_EXP16, T_MFMA16_EXP32, T_MFMA16_FMA32,
       T_MFMA16_MIX, T_MFMA16_SRCC, T_LDSB128, T_M32_F2, T_M32_F4, T_M32_F6, T_M32_F8, T_M32_MIX4, T_M32_MIX5, T_M32_EXP2, T_M32_EXP4, T_M32_LDS, T_M32_LDS2, T_M32_LDSTR, T_M32_LDS_MIX5, T_M32_LDS2_MIX5, T_N };
static const char *names[T_N] = {"v_exp_f32 x32", "v_fma_f32 x32", "v_pk_fma_f32 x16 (32 values)", "v_max3_f32 x32", "v_cvt_pk_bf16_f32 x16", "v_pk_mul_f32 x16",
                                 "v_add_f32 x32", "v_pk_add_f32 x16", "v_permlane32_swap x16", "mfma16x16x32 x16", "mfma32x32x16 x8", "mfma16 x16 + exp x16",
                                 "mfma16 x16 + exp x32", "mfma16 x16 + fma x32", "mfma16 x16 + (8 max3, 16 exp, 8 cvt)", "mfma16 x16, srcC != dst", "ds_read_b128 x16",
                                 "mfma32 x8 + 2 fma each", "mfma32 x8 + 4 fma each", "mfma32 x8 + 6 fma each", "mfma32 x8 + 8 fma each",
                                 "mfma32 x8 + (max3, 2 exp, cvt) each", "mfma32 x8 + (max3, 2 exp, cvt, fma) each", "mfma32 x8 + 2 exp each", "mfma32 x8 + 4 exp each",
                                 "mfma32 x8, A from ds_read_b128 each", "mfma32 x8, one ds_read_b128 per 2", "mfma32 x8, A from 2 ds_read_b64_tr each",
                                 "mfma32 x8, ds_read_b128 + mix5 each", "mfma32 x8, ds_read_b128 per 2 + mix5 each"};

template <int T>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int iters) {
    __shared__ float lds[16384];
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (threadIdx.x + i);
    f32x4 acc[16];
    f32x16 acc32[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc32[i] = (f32x16)(0.f);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * (threadIdx.x & 3)); }
    f32x4 negm = f32x4{-1.f, -1.f, -1.f, -1.f};
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (T == T_EXP) {
#define M(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            REP32(M)
#undef M
        } else if constexpr (T == T_FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
            REP32(M)
#undef M
        } else if constexpr (T == T_PKFMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(f32x2 *)&x[2 * i]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MAX3) {
#define M(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 31]), "v"(x[(i + 2) & 31]));
            REP32(M)
#undef M
        } else if constexpr (T == T_CVT) {
#define M(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i + 16]), "v"(x[((i + 1) & 15) + 16]));
            REP16(M)
#undef M
        } else if constexpr (T == T_PKMUL) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(f32x2 *)&x[2 * i]));
            REP16(M)
#undef M
        } else if constexpr (T == T_ADD) {
#define M(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
            REP32(M)
#undef M
        } else if constexpr (T == T_PKADD) {
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(f32x2 *)&x[2 * i]));
            REP16(M)
#undef M
        } else if constexpr (T == T_SWAP32) {
#define M(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 16]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MFMA16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        } else if constexpr (T == T_MFMA32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
        } else if constexpr (T == T_MFMA16_EXP16) {
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0); asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MFMA16_EXP32) {
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(x[i]), "+v"(x[i + 16]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MFMA16_FMA32) {
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(x[i]), "+v"(x[i + 16]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MFMA16_MIX) {     // the fast-path softmax of one (16 q x 64 keys) tile beside its 16 MFMAs
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0); asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); \
             if ((i & 1) == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[16 + (i >> 1)]) : "v"(x[i]), "v"(x[i + 1])); \
             else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[24 + (i >> 1)]) : "v"(x[i]), "v"(x[i - 1]));
            REP16(M)
#undef M
        } else if constexpr (T == T_MFMA16_SRCC) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, negm, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(acc[i]));
        } else if constexpr (T == T_LDSB128) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 v = *(const f32x4 *)&lds[((threadIdx.x & 63) * 4 + i * 256 + it * 4) & 16383 & ~3];
                acc[i] += v;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc32[i][0] + acc32[i][7];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
void run(float *out, long long *cyc, int n_per_iter) {
    const int iters = 2000;
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<T><<<256, threads>>>(out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<T><<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const int wps = threads / 256;
        const double cyc_wave = (double)h[0] / iters / n_per_iter;
        // counter ticks at 100 MHz on gfx9 (s_memrealtime) or the shader clock (s_memtime); print both views: wall-derived cycles at 2.4 GHz nominal
        const double wall_cyc_simd = ms * 1e-3 * 2.4e9 / iters / n_per_iter / wps;
        printf("%-40s %d wave/SIMD: counter/instr/wave %7.2f   wall@2.4GHz cyc/instr/SIMD %6.2f   (%.3f ms)\n", names[T], wps, cyc_wave, wall_cyc_simd, ms);
    }
}

template <int T>
__global__ void __launch_bounds__(512) k32(float *out, long long *cyc, int iters) {
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (threadIdx.x + i);
    f32x16 acc32[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc32[i] = (f32x16)(0.f);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * (threadIdx.x & 3)); }
    __shared__ __attribute__((aligned(16))) char lds32[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t *)lds32)[i] = 0x3c003c00u;
    __syncthreads();
    const char *lp = lds32 + (threadIdx.x & 63) * 16 + ((threadIdx.x >> 6) & 3) * 8192;      // conflict-free: 64 lanes x 16 bytes contiguous
    const char *lt = lds32 + (threadIdx.x & 63) * 8 + ((threadIdx.x >> 6) & 3) * 8192;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const uint32_t lpa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)lp;
    bf16x8 av[2] = {a, a};
#define LDA(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(av[((i) + 1) & 1]) : "v"(lpa), "n"(((i) & 7) * 1024));
#define MFL(i, j) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(j) & 1], b, acc32[i], 0, 0, 0);
#define LDT(i) { const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lt + ((i) & 7) * 1024)); \
                 const s16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lt + ((i) & 7) * 1024 + 512)); \
                 const uint2 l2 = *(const uint2 *)&lo, h2 = *(const uint2 *)&hh; uint4 raw = make_uint4(l2.x, l2.y, h2.x, h2.y); av[((i) + 1) & 1] = *(bf16x8 *)&raw; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define MF(i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
#define FMA(j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(j) & 31]));
#define EXP(j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(j) & 31]));
#define MX3(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[(j) & 31]) : "v"(x[((j) + 1) & 31]), "v"(x[((j) + 2) & 31]));
#define CVT(j) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[(j) & 31]) : "v"(x[((j) + 5) & 31]), "v"(x[((j) + 9) & 31]));
        if constexpr (T == T_MFMA32) {
#define M(i) MF(i)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_F2) {
#define M(i) MF(i) FMA(4 * i) FMA(4 * i + 1)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_F4) {
#define M(i) MF(i) FMA(4 * i) FMA(4 * i + 1) FMA(4 * i + 2) FMA(4 * i + 3)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_F6) {
#define M(i) MF(i) FMA(4 * i) FMA(4 * i + 1) FMA(4 * i + 2) FMA(4 * i + 3) FMA(4 * i + 16) FMA(4 * i + 17)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_F8) {
#define M(i) MF(i) FMA(4 * i) FMA(4 * i + 1) FMA(4 * i + 2) FMA(4 * i + 3) FMA(4 * i + 16) FMA(4 * i + 17) FMA(4 * i + 18) FMA(4 * i + 19)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_MIX4) {
#define M(i) MF(i) MX3(4 * i) EXP(4 * i + 1) EXP(4 * i + 2) CVT(4 * i + 3)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_MIX5) {
#define M(i) MF(i) MX3(4 * i) EXP(4 * i + 1) FMA(4 * i + 16) EXP(4 * i + 2) CVT(4 * i + 3)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_EXP2) {
#define M(i) MF(i) EXP(4 * i + 1) EXP(4 * i + 2)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_EXP4) {
#define M(i) MF(i) EXP(4 * i) EXP(4 * i + 1) EXP(4 * i + 2) EXP(4 * i + 3)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_LDS) {
#define M(i) LDA(i) MFL(i, i)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_LDS2) {
#define M(i) LDA(i) MFL(2 * i, i) MFL(2 * i + 1, i)
            M(0) M(1) M(2) M(3)
#undef M
        } else if constexpr (T == T_M32_LDSTR) {
#define M(i) LDT(i) MFL(i, i)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_LDS_MIX5) {
#define M(i) LDA(i) MFL(i, i) MX3(4 * i) EXP(4 * i + 1) FMA(4 * i + 16) EXP(4 * i + 2) CVT(4 * i + 3)
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
        } else if constexpr (T == T_M32_LDS2_MIX5) {
#define X5(i) MX3(4 * (i)) EXP(4 * (i) + 1) FMA(4 * (i) + 16) EXP(4 * (i) + 2) CVT(4 * (i) + 3)
#define M(i) LDA(i) MFL(2 * i, i) X5(2 * i) MFL(2 * i + 1, i) X5(2 * i + 1)
            M(0) M(1) M(2) M(3)
#undef M
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc32[i][0] + acc32[i][7];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
void run32(float *out, long long *cyc) {
    const int iters = 2000, n_per_iter = 8;
    for (int threads = 256; threads <= 512; threads += 256) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k32<T><<<256, threads>>>(out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k32<T><<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const int wps = threads / 256;
        printf("%-44s %d wave/SIMD: counter/mfma/wave %7.2f   wall@2.4GHz cyc/mfma/SIMD %6.2f   (%.3f ms)\n", names[T], wps, (double)h[0] / iters / n_per_iter,
               ms * 1e-3 * 2.4e9 / iters / n_per_iter / wps, ms);
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 16 * 8);
    run<T_EXP>(out, cyc, 32); run<T_FMA>(out, cyc, 32); run<T_PKFMA>(out, cyc, 16); run<T_MAX3>(out, cyc, 32); run<T_CVT>(out, cyc, 16);
    run<T_PKMUL>(out, cyc, 16); run<T_ADD>(out, cyc, 32); run<T_PKADD>(out, cyc, 16); run<T_SWAP32>(out, cyc, 16);
    run<T_MFMA16>(out, cyc, 16); run<T_MFMA16_EXP16>(out, cyc, 16); run<T_MFMA16_EXP32>(out, cyc, 16);
    run<T_MFMA16_FMA32>(out, cyc, 16); run<T_MFMA16_MIX>(out, cyc, 16); run<T_LDSB128>(out, cyc, 16);
    run32<T_MFMA32>(out, cyc); run32<T_M32_F2>(out, cyc); run32<T_M32_F4>(out, cyc); run32<T_M32_F6>(out, cyc); run32<T_M32_F8>(out, cyc);
    run32<T_M32_MIX4>(out, cyc); run32<T_M32_MIX5>(out, cyc); run32<T_M32_EXP2>(out, cyc); run32<T_M32_EXP4>(out, cyc);
    run32<T_M32_LDS>(out, cyc); run32<T_M32_LDS2>(out, cyc); run32<T_M32_LDSTR>(out, cyc); run32<T_M32_LDS_MIX5>(out, cyc); run32<T_M32_LDS2_MIX5>(out, cyc);
    return 0;
}
