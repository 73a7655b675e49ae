// skinny.h -- the core of the weights-resident streaming products (samfuse.hip, gemm_stream.hip): A [M, K] two-byte rows in global
// memory, W [N, K] resident in LDS, one wave = 16 rows x N columns on v_mfma_f32_16x16x32_{bf16,f16}.  No LDS staging of A and no
// workgroup barrier after the weights are in: every wave streams its own row blocks, so 16 waves per CU hide the HBM latency and the
// kernel runs at the rate of its A-in / C-out traffic (3-4.4 TB/s measured) where a tiled GEMM with 2-8 K-tiles spends its time in
// prologues and epilogues (~2 TB/s).
#pragma once
#include "gemm_common.h"

namespace ovo_skinny {

using ovo_gemm_detail::bf16x8;
using ovo_gemm_detail::f32x4;
using ovo_gemm_detail::Mfma;

template <int K, int N, typename VT = bf16x8>
struct Skinny {
    static constexpr int CPR = K / 8, KS = K / 32, NT = N / 16, W_BYTES = N * K * 2;
    static_assert(K % 32 == 0 && N % 16 == 0 && CPR >= 4, "shape");
    // ds_read_b128 of a fragment: 16 lanes read the same 16-byte chunk of 16 consecutive rows; rows are K * 2 bytes apart, so the
    // chunk index is XORed with a row function that spreads those 16 reads over all 64 banks (applied when W is copied in, too).
    // For a chunk count that is not a power of two (K = 192) only the low 3 bits are XORed, which keeps the chunk inside its row.
    static __device__ __forceinline__ int swz(int n) {
        return (CPR & (CPR - 1)) != 0 ? (n & 7) : CPR >= 16 ? (n & 15) : CPR == 8 ? ((n >> 1) & 7) : ((n >> 2) & 3);
    }
    // W rows [n0, n0 + N) of a matrix with row stride ldw elements
    static __device__ __forceinline__ void load_w(char *lds, const uint16_t *W, long long ldw, int tid, int nthreads) {
        for (int id = tid; id < N * CPR; id += nthreads) {
            const int n = id / CPR, c = id % CPR;
            *(uint4 *)(lds + (n * CPR + (c ^ swz(n))) * 16) = *(const uint4 *)(W + (long long)n * ldw + c * 8);
        }
    }
    static __device__ __forceinline__ void load_a(VT (&a)[KS], const uint16_t *A, long long lda, long long row, int fq) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *(const VT *)(A + row * lda + (ks * 4 + fq) * 8);
    }
    // acc[j][r] += sum_k A[row fr][k] W[16 j + 4 fq + r][k]   (operands swapped: the accumulator holds 4 consecutive columns of one row)
    static __device__ __forceinline__ void mma(f32x4 (&acc)[NT], const VT (&a)[KS], const char *lds, int fr, int fq) {
        // groups of JG weight fragments (4 VGPRs each) are read, then multiplied; the scheduling fences keep the compiler from hoisting
        // every ds_read of the product ahead of the first MFMA (N / 16 x K / 32 fragments = 256+ VGPRs: it spilled the accumulators).
        // Four waves per SIMD cover a group's LDS latency.
        constexpr int JG = NT <= 8 ? NT : (NT % 8 == 0 ? 8 : NT % 7 == 0 ? 7 : NT % 6 == 0 ? 6 : NT % 5 == 0 ? 5 : 3);   // the largest divisor <= 8
        static_assert(NT % JG == 0, "column tiles per read group");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j0 = 0; j0 < NT; j0 += JG) {
                VT w[JG];
                // swz() depends on the row only through fr (16 j drops out), so a fragment address is (lane part for this ks) + j * constant
                const char *wp = lds + (fr * CPR + ((ks * 4 + fq) ^ swz(fr))) * 16;
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) w[jj] = *(const VT *)(wp + (j0 + jj) * (16 * CPR * 16));
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) acc[j0 + jj] = Mfma<VT>::run(w[jj], a[ks], acc[j0 + jj]);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
};

// Two adjacent column tiles' packed two-byte results (a lane holds 4 columns = 2 dwords of each).  v_permlane16_swap trades halves between
// the lane pairs (fq, fq ^ 1): lane (fr, fq) ends with 8 consecutive columns of tile j + (fq & 1), from column 8 (fq >> 1) -- one 16-byte
// store per lane and 64 contiguous bytes per row and instruction instead of two 8-byte stores of 32.  `tile_j` = address of column 0 of
// tile j in this lane's row (16-byte aligned); every lane of a row must call it (the exchange is within the row's 4 lanes).
__device__ __forceinline__ void store_pair16(uint16_t *tile_j, int fq, uint2 tj, uint2 tj1) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(tj.x, tj1.x, false, false), s1 = __builtin_amdgcn_permlane16_swap(tj.y, tj1.y, false, false);
    *(uint4 *)(tile_j + (fq & 1) * 16 + (fq >> 1) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);     // (a non-temporal store measured 2 % slower here)
}

}  // namespace ovo_skinny
