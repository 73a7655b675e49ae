// gemm_stream.hip -- ovo_gemm for tall, short-K products (M >= 16 K rows, K <= 256): the SAM2 Hiera stage-1 / stage-2 layers at
// 1024^2 (64 K - 512 K tokens x 112 / 224 channels) and their like.  These are HBM streams -- 30-130 flops per byte moved -- and the
// tiled kernels (gemm.hip) ran them at ~2 TB/s: with 2-4 K-tiles a workgroup is all prologue and epilogue.  Here (skinny.h) a column
// group of the weights (<= 256 columns, <= 128 KB) is LDS-resident for the life of a workgroup, every wave streams its own 16-row blocks
// of A straight from global memory into MFMA fragments and finishes them in registers (bias, activation, residual, window -> spatial
// row mapping, cast), with no barrier after the weights are in.  Column groups of one product run side by side (blockIdx.x % groups),
// so the A rows they share meet in L2.
#include <stdlib.h>

#include <type_traits>

#include "skinny.h"

using namespace ovo_gemm_detail;

namespace {

template <int KS, int NT, typename VT, int NTHREADS>
__global__ void __launch_bounds__(NTHREADS) k_gemm_stream(GemmArgs g, int n_groups) {
    constexpr int K = KS * 32, NG = NT * 16;
    using S = ovo_skinny::Skinny<K, NG, VT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *cs = (float *)(smem + S::W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int group = blockIdx.x % n_groups, slot = blockIdx.x / n_groups, slots = gridDim.x / n_groups;
    const int n0 = group * NG;
    S::load_w(smem, (const uint16_t *)g.W + (long long)n0 * g.ldw, g.ldw, tid, NTHREADS);
    for (int i = tid; i < NG; i += NTHREADS) cs[i] = g.bias ? g.bias[n0 + i] : 0.f;
    __syncthreads();
    const float *cl = cs + fq * 4;
    const int blocks = (g.M + 15) / 16;
    constexpr int WPB = NTHREADS / 64;
    for (int b = slot * WPB + wave; b < blocks; b += slots * WPB) {
        const int m = b * 16 + fr, mc = m < g.M ? m : g.M - 1;
        VT af[KS];
        S::load_a(af, (const uint16_t *)g.A, g.lda, mc, fq);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        S::mma(acc, af, smem, fr, fq);
        const long long md = m < g.M ? row_dest(g, m) : -1;          // window-major product row -> spatial row (ovo_gemm_unwindow), -1 = padding
        if (md < 0) continue;
        const float *ap = g.add ? g.add + add_row(g, m, md) * g.ld_add + n0 + fq * 4 : nullptr;
        auto finish = [&](int j, float (&v)[4]) {
            const f32x4 bv = *(const f32x4 *)(cl + j * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[j][r] * g.alpha + bv[r];
            if (g.act) act4(v, g.act);
            if (ap) {
                const f32x4 r = *(const f32x4 *)(ap + j * 16);
                v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
            }
        };
        if (g.out_dtype == 0) {
            float *cp = (float *)g.C + md * g.ldc + n0 + fq * 4;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v[4];
                finish(j, v);
                *(float4 *)(cp + j * 16) = make_float4(v[0], v[1], v[2], v[3]);      // 4 lanes x 16 B = 64 contiguous bytes per row
                if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);                   // bounds the residual loads in flight (registers)
            }
        } else {
            // two-byte outputs: a lane holds 8 bytes of a column tile.  Tiles are finished in pairs and v_permlane16_swap trades the halves
            // between the lane pairs (fq, fq ^ 1), so that every lane stores 16 bytes and a store instruction writes 64 contiguous bytes per
            // row instead of 32: half the store instructions (the tiled kernels' direct epilogue was bound by exactly those).
            uint16_t *c16 = (uint16_t *)g.C + md * g.ldc + n0;
            const bool wide = g.ldc % 8 == 0;
#pragma unroll
            for (int j = 0; j + 1 < NT; j += 2) {
                float v[4], u[4];
                finish(j, v); finish(j + 1, u);
                uint32_t p0, p1, q0, q1;
                if (g.out_dtype == 2) { p0 = pack_bf16(v[0], v[1]); p1 = pack_bf16(v[2], v[3]); q0 = pack_bf16(u[0], u[1]); q1 = pack_bf16(u[2], u[3]); }
                else { p0 = pack_f16(v[0], v[1]); p1 = pack_f16(v[2], v[3]); q0 = pack_f16(u[0], u[1]); q1 = pack_f16(u[2], u[3]); }
                if (wide) {
                    ovo_skinny::store_pair16(c16 + j * 16, fq, make_uint2(p0, p1), make_uint2(q0, q1));
                } else {
                    *(uint2 *)(c16 + j * 16 + fq * 4) = make_uint2(p0, p1);
                    *(uint2 *)(c16 + (j + 1) * 16 + fq * 4) = make_uint2(q0, q1);
                }
                if (j % 4 == 2) __builtin_amdgcn_sched_barrier(0);
            }
            if (NT % 2) {
                float v[4];
                finish(NT - 1, v);
                *(uint2 *)(c16 + (NT - 1) * 16 + fq * 4) = g.out_dtype == 2 ? make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]))
                                                                             : make_uint2(pack_f16(v[0], v[1]), pack_f16(v[2], v[3]));
            }
        }
    }
}

template <int KS, int NT, typename VT>
int launch_stream(const GemmArgs &g, hipStream_t s) {
    constexpr int K = KS * 32, NG = NT * 16;
    constexpr size_t lds = (size_t)NG * K * 2 + NG * sizeof(float);
    // > half the LDS: one workgroup per CU -- 16 waves (128 VGPRs each), or 12 when the accumulators of a wide group need more; else two or more of 8 waves
    constexpr int NTHREADS = lds > 80 * 1024 ? (NT > 16 ? 768 : 1024) : 512;
    constexpr int PER_CU = lds > 80 * 1024 ? 1 : (lds > 52 * 1024 ? 2 : (lds > 39 * 1024 ? 3 : 4));
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm_stream<KS, NT, VT, NTHREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ovo_set_error("ovo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        attr_done = true;
    }
    const int n_groups = g.N / NG, blocks = (g.M + 15) / 16, wpb = NTHREADS / 64;
    int slots = (256 * PER_CU + n_groups - 1) / n_groups;             // workgroups = slots x groups ~ what the chip holds at once
    const int need = (blocks + wpb - 1) / wpb;
    if (slots > need) slots = need;
    if (slots < 1) slots = 1;
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(8, 2.0 * g.M * (double)g.N * g.K, s); ovo_prof_shape(g.M, g.N, g.K); }
    k_gemm_stream<KS, NT, VT, NTHREADS><<<slots * n_groups, NTHREADS, lds, s>>>(g, n_groups);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

}  // namespace

namespace ovo_gemm_detail {

// Returns OVO_E_UNSUPPORTED when the shape has no instantiation (the caller then takes a tiled kernel).
int gemm_stream_launch(const GemmArgs &g, int in_dtype, hipStream_t s) {
    if (g.best || g.rope_cos || in_dtype != 2) return OVO_E_UNSUPPORTED;
    if (g.M < 16384 || ((uintptr_t)g.C & 15) != 0 || g.ldc % 4 != 0) return OVO_E_UNSUPPORTED;
    // column groups: the widest of 256 / 224 / 112 / 64 / 32 that divides N (hiera_b+'s 112-multiples, powers of two); 288 / 144 for
    // hiera_l's stage 1 (K = 192: 144 channels padded)
    int ng = 0;
    for (int c : {256, 224, 112, 64, 32})
        if (g.N % c == 0) { ng = c; break; }
    if (g.K == 192 && g.N % 144 == 0 && g.N % 256 != 0) ng = g.N % 288 == 0 ? 288 : 144;
    if (g.K == 128 && g.N == 336) ng = 336;                          // Hiera stage-1 QKV: one group (86 KB of weights), A read once
    // measured (tools/gemm_bench.py, profiles/r02c_gemm_stream.txt): no gain over the tiled kernels with 6+ column groups (A re-read per group)
    // or for the narrow f32-residual product (524288, 112, 128), which both forms run at the HBM rate of its in-place C traffic
    if (g.N / (ng ? ng : 1) >= 6 || (g.K == 128 && g.N == 112 && g.out_dtype == 0)) return OVO_E_UNSUPPORTED;
#define GO(KK, NGG) if (g.K == KK && ng == NGG) return launch_stream<KK / 32, NGG / 16, bf16x8>(g, s);
    GO(128, 336) GO(128, 256) GO(128, 224) GO(128, 112) GO(128, 64)
    GO(192, 288) GO(192, 256) GO(192, 144) GO(192, 112)
    GO(256, 256) GO(256, 224) GO(256, 112) GO(256, 64) GO(256, 32)
#undef GO
    return OVO_E_UNSUPPORTED;
}

}  // namespace ovo_gemm_detail
