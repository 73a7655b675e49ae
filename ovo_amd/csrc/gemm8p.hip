// gemm8p.hip -- the large-tile MFMA GEMM of the batched encoder forwards: C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias) (+ add).
//
// Same contract and epilogue as k_gemm (gemm.hip); different main loop, built for M >= ~2000 rows (several keyframes' crops per
// forward) where the 128-row ring kernel sits on the LDS port (32x32 wave tiles: one LDS byte per 4 flop):
//
//   * 256 x BN x 64 tiles (BN = 256: 8 waves as 2 x 4, wave tile 128 x 64;  BN = 128: 4 x 2, wave tile 64 x 64), one workgroup per CU,
//     128 / 96 KB of LDS = two K-tile buffers, each split in four HALF-tiles: Ah0 / Ah1 hold sub-tile 0 / 1 (upper / lower half of the
//     rows) of EVERY wave's A rows, Bh0 / Bh1 likewise for the W rows.  A K-tile is consumed in four phases, one C quadrant each:
//         q0: read A.sub0 + B.sub0 -> acc[0][0]     q1: read B.sub1 -> acc[0][1]     q2: read A.sub1 -> acc[1][1]     q3: -> acc[1][0]
//     so a half-tile is dead two phases after it was read and is restaged (LDS-DMA, global_load_lds_dwordx4) for K-tile t+2 while K-tile t
//     is still being multiplied: q0 stages Bh1(t+1), q1 Ah1(t+1), q2 Ah0(t+2), q3 Bh0(t+2); every stage is read >= 5 phases later.
//   * the two wave groups {0..3} / {4..7} (one wave of each per SIMD) run ONE BARRIER APART: a phase is
//         [ds_read fragments | issue DMA | counted vmcnt] s_barrier [MFMA quadrant] s_barrier
//     and while one group multiplies, the other reads its next fragments -- the matrix pipe of a SIMD alternates between its two waves
//     and never waits for LDS (ping-pong).  Counted s_waitcnt vmcnt leaves the four newest stages in flight across the barriers.
//   * what bounds the K-loop (round 2 measurements): 256 x 128 tiles take 0.80 us per K-tile, 256 x 256 tiles 1.32 us -- in both cases
//     12-15 TB/s of L2 -> LDS DMA chip-wide (48 resp. 64 KB per workgroup and K-tile), i.e. the loop runs at the rate the tiles arrive,
//     with all the LDS there is already in flight.  A single-phase variant of the 256 x 128 kernel (three K-tile buffers, 16 ds_reads
//     then 32 MFMAs per K-tile, half the barriers) was written, verified bit-identical and measured: the same 0.80 us per K-tile
//     (FC1 66.1 vs 66.2 us, 8192^3 1258 vs 1238 TFLOP/s) -- phase overhead is not the bound; removed.  More flops per loaded byte
//     needs a larger tile than 256 x 256, which neither LDS (two buffers) nor the accumulator file allow.
//   * fragments: v_mfma_f32_16x16x32_{bf16,f16}, operands swapped as in k_gemm (a = W fragment, b = activation fragment: a lane owns 4
//     consecutive output columns); 128-byte LDS rows with the 16-byte-chunk XOR swizzle applied to the DMA source address and to the
//     ds_read_b128 address.
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

using namespace ovo_gemm_detail;

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {             // f(integral_constant<I>) ... f(integral_constant<N-1>): indices stay compile-time constants
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define OVO_8P_MFMA32_DEFAULT 0
#define OVO_8P_MERGED_DEFAULT 1
#define OVO_FENCE() asm volatile("" ::: "memory")
#define OVO_BARRIER()                      \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        OVO_FENCE();                       \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define OVO_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// MF = 16: v_mfma_f32_16x16x32 (acc[i][j] = one 16 x 16 tile);  MF = 32: v_mfma_f32_32x32x16 (round 5, VERDICT r4 item 1a: same LDS bytes per wave tile --
// every fragment is read once per K-tile either way -- half the VGPR operand reads per flop; STAGED epilogues only).  The epilogues see the accumulators as
// PIECES of 4 consecutive columns of one row: piece (i, j) of a lane = row RT i + lrow, columns CS j + lcol .. + 3 of the wave tile
//   MF 16: RT 16, CS 16, lrow = lane & 15, lcol = 4 (lane >> 4), value acc[i][j]
//   MF 32: RT 32, CS  8, lrow = lane & 31, lcol = 4 (lane >> 5), value acc32[i][j / 4][4 (j % 4) .. + 3]   (D[n][m] of W-fragment x activation-fragment:
//          lane holds m = lane % 32 and n = 8 (r / 4) + 4 (lane / 32) + r % 4 of the 32 x 32 tile)
// MERGED (round 6): a K-tile in TWO barrier intervals of two quadrants each (32 MFMAs per wave between barriers) instead of four of one -- see `body2`.
template <int BM, int BN, int WARPS_M, typename VT, bool STAGED, int MF = 16, bool MERGED = false>
__global__ void __launch_bounds__(512) k_gemm8p(GemmArgs g) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (its parse of lambdas that call LDS-DMA builtins drops the stub silently)
    constexpr int WARPS_N = 8 / WARPS_M;
    constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;            // wave tile
    constexpr int HM = WTM / 2, HN = WTN / 2;                        // one sub-tile ("half") of the wave tile
    constexpr int TMH = HM / 16, TNH = HN / 16;                      // 16x16 MFMA tiles per sub-tile
    constexpr int TMH32 = HM / 32, TNH32 = HN / 32;                  // 32x32 MFMA tiles per sub-tile (MF = 32)
    static_assert(MF == 16 || (MF == 32 && STAGED && TMH32 >= 1 && TNH32 >= 1), "MF = 32: staged epilogues, sub-tiles of >= 32 rows");
    constexpr int RT = MF, CS = MF == 16 ? 16 : 8, NI = WTM / RT, NJ = WTN / CS;      // pieces of a lane: NI x NJ (f32x4 each)
    constexpr int A_HALF = (BM / 2) * 128, B_HALF = (BN / 2) * 128;  // bytes of a half-tile: rows x 64 two-byte elements
    constexpr int BUF = 2 * A_HALF + 2 * B_HALF;
    constexpr int NA = A_HALF / (512 * 16), NB = B_HALF / (512 * 16);   // DMA pieces (1 KB = 8 rows per wave instruction) per thread
    static_assert(NA >= 1 && NB >= 1 && TMH >= 1 && TNH >= 1, "tile too small for 512 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    const int wr = wave / WARPS_N, wc = wave % WARPS_N, group = wave >> 2;
    int tile = blockIdx.x;
    if (g.chunk > 0) {                                   // XCD-chunked tile order (see k_gemm)
        tile = (blockIdx.x & 7) * g.chunk + (blockIdx.x >> 3);
        if (tile >= g.tiles) return;
    }
    int tm = tile / g.nbn, tn = tile % g.nbn;
    if (g.strip > 0) {                                   // column strips: an XCD's chunk of consecutive tiles is a compact block of the product
        const int nbm = (g.M + BM - 1) / BM, per = nbm * g.strip, sidx = tile / per, r = tile - sidx * per;
        const int ws = min(g.strip, g.nbn - sidx * g.strip);
        tm = r / ws;
        tn = sidx * g.strip + (r - tm * ws);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = MF == 16 ? fr : (lane & 31), lcol = 4 * (MF == 16 ? fq : (lane >> 5));
    // GELU table (gemm_common.h) behind the K-tile ring AND the epilogue slabs: filled now, first read after the K-loop's last barrier
    constexpr int SLAB32 = 8 * (BM / WARPS_M / 2) * ((BN / (8 / WARPS_M)) * 4 + 16), SLAB16 = 8 * (BM / WARPS_M) * ((BN / (8 / WARPS_M)) * 2 + 16), RING = 2 * (BM + BN) * 128;
    constexpr int LUT_OFF = STAGED ? (SLAB32 > RING ? (SLAB32 > SLAB16 ? SLAB32 : SLAB16) : (SLAB16 > RING ? SLAB16 : RING)) : RING;
    const float2 *lut = (STAGED && g.act == 1 && g.gelu_lut) ? (const float2 *)(smem + LUT_OFF) : nullptr;
    // LayerNorm fold, consumer side (gemm_common.h): (mean, rstd) of the tile's 256 rows, behind the table; filled in the prologue, read by the epilogue
    constexpr int ROPE_END = 8 * 32 * ((BN / (8 / WARPS_M)) * 4 + 16) + 2 * WARPS_M * 32 * (64 * 4 + 16);      // (rope_out's slabs + table slice: may reach past the table)
    constexpr int RS_OFF = LUT_OFF + GELU_LUT_BYTES > ROPE_END ? LUT_OFF + GELU_LUT_BYTES : ROPE_END;
    float2 *row_ms = (float2 *)(smem + RS_OFF);
    const bool fold = STAGED && g.fold_stats != nullptr;
    // (the table is FILLED after the prologue's six DMA stages are in flight -- two erff per thread under the first tiles' load latency, round 5)
#ifdef OVO_GEMM_DEBUG        // tools/ builds only (python -m ovo_amd.build --gemm-debug): early exits, per-phase time stamps, de-phased starts
    if (g.dbg & 1) return;
    auto stamp = [&](int k) { if (g.stamps && tid == 0) g.stamps[(long long)tile * 4 + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    if ((g.dbg >> 8) && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {      // OVO_8P_DELAY us: de-phase every other CU's tile sequence
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = (unsigned long long)(g.dbg >> 8) * 100;
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
#else
    auto stamp = [&](int) {};
#endif

    // ---- DMA sources: half h, piece (it * 8 + wave) = local rows [8 * piece, +8), lane -> (row, swizzled 16-byte chunk).
    // 32-bit byte offsets from the (wave-uniform) operand base: the launch checks that both operands span < 4 GB.
    uint32_t a_off[2][NA], b_off[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int r = (it * 8 + wave) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = m0 + (r / HM) * WTM + h * HM + (r % HM);
            gr = gr < g.M ? gr : g.M - 1;
            a_off[h][it] = (uint32_t)(((long long)gr * g.lda + c * 8) * 2);
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int r = (it * 8 + wave) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = n0 + (r / HN) * WTN + h * HN + (r % HN);
            gr = gr < g.N ? gr : g.N - 1;
            b_off[h][it] = (uint32_t)(((long long)gr * g.ldw + c * 8) * 2);
        }
    }
    // buffer_load_dwordx4 ... lds: resource = operand base (SGPRs), voffset = the lane's byte offset, soffset = the K-tile's byte offset.
    // A stage for a K-tile past the end goes through a zero-length resource: nothing is fetched (zeros land in a buffer nobody reads any
    // more), but the instruction still counts in vmcnt -- the loop body and its counted waits are the same for every K-tile.
    const int a_bytes = (int)((long long)g.M * g.lda * 2), b_bytes = (int)((long long)g.N * g.ldw * 2);
    auto stage_a = [&](int buf, int h, int kt, bool valid = true) {
        char *dst = smem + buf * BUF + h * A_HALF + wave * 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, valid ? a_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < NA; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + it * 8192), 16, a_off[h][it], kt * 128, 0, 0);
    };
    auto stage_b = [&](int buf, int h, int kt, bool valid = true) {
        char *dst = smem + buf * BUF + 2 * A_HALF + h * B_HALF + wave * 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, valid ? b_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < NB; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + it * 8192), 16, b_off[h][it], kt * 128, 0, 0);
    };

    // ---- fragment addresses: local row = (wave's first row in the half) + 16 i + fr, chunk (4 ks + fq) ^ swizzle(row)
    const int sw = (fr >> 1) & 7;
    const int off_a = (wr * HM + fr) * 128 + ((fq ^ sw) << 4);      // ks = 0; ks = 1 is the same address with bit 6 flipped
    const int off_b = (wc * HN + fr) * 128 + ((fq ^ sw) << 4);
    // MF = 32: local row = first row + 32 I + (lane & 31), chunk (2 s + (lane >> 5)) ^ swizzle(row): k-step s flips bits 5-6 of the address
    const int sw32 = ((lane & 31) >> 1) & 7;
    const int off32_a = (wr * HM + (lane & 31)) * 128 + (((lane >> 5) ^ sw32) << 4);
    const int off32_b = (wc * HN + (lane & 31)) * 128 + (((lane >> 5) ^ sw32) << 4);

    typedef __attribute__((ext_vector_type(16))) float f32x16;
    f32x4 acc[MF == 16 ? 2 * TMH : 1][MF == 16 ? 2 * TNH : 1];
    f32x16 acc32[MF == 32 ? 2 * TMH32 : 1][MF == 32 ? 2 * TNH32 : 1];
    if constexpr (MF == 16) {
#pragma unroll
        for (int i = 0; i < 2 * TMH; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 2 * TMH32; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TNH32; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    }
    // piece (i, j) of this lane (compile-time indices after unrolling)
    auto piece = [&](int i, int j) -> f32x4 {
        if constexpr (MF == 16) return acc[i][j];
        else { const f32x16 &t = acc32[i][j >> 2]; const int e = (j & 3) * 4; return f32x4{t[e], t[e + 1], t[e + 2], t[e + 3]}; }
    };
    constexpr int NKS = MF == 16 ? 2 : 4, FM = MF == 16 ? TMH : TMH32, FN = MF == 16 ? TNH : TNH32;     // k-steps per K-tile, fragments per sub-tile
    VT xa[FM][NKS], wb0[FN][NKS], wb1[FN][NKS];

    auto load_a = [&](const char *cur, int h) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                xa[i][ks] = MF == 16 ? *(const VT *)(cur + h * A_HALF + ((off_a ^ (ks << 6)) + i * 2048))
                                     : *(const VT *)(cur + h * A_HALF + ((off32_a ^ (ks << 5)) + i * 4096));
    };
    auto load_b = [&](const char *cur, int h, VT (&wb)[FN][NKS]) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                wb[j][ks] = MF == 16 ? *(const VT *)(cur + 2 * A_HALF + h * B_HALF + ((off_b ^ (ks << 6)) + j * 2048))
                                     : *(const VT *)(cur + 2 * A_HALF + h * B_HALF + ((off32_b ^ (ks << 5)) + j * 4096));
    };
    auto quadrant = [&](int ih, int jh, VT (&wb)[FN][NKS]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (MF == 16) acc[ih * TMH + i][jh * TNH + j] = Mfma<VT>::run(wb[j][ks], xa[i][ks], acc[ih * TMH + i][jh * TNH + j]);
                    else acc32[ih * TMH32 + i][jh * TNH32 + j] = Mfma<VT>::run32(wb[j][ks], xa[i][ks], acc32[ih * TMH32 + i][jh * TNH32 + j]);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    const int nt = g.K / 64;

    // fold: this thread's row's partial statistics, requested BEFORE the ring's first stages (older in the in-order vmcnt queue: the counted waits below
    // keep their meaning) and summed after the prologue's wait.  Unconditional loads (a load inside a select becomes guarded dword loads and full drains).
    float2 fpart[16];
    if (fold && tid < BM) {
        const int row = m0 + tid < g.M ? m0 + tid : g.M - 1;
        const float2 *sp = (const float2 *)g.fold_stats + row;                 // [part][row]: a wave instruction reads 64 consecutive rows of one part
#pragma unroll
        for (int q = 0; q < 16; ++q) fpart[q] = sp[(long long)(q < g.fold_parts ? q : g.fold_parts - 1) * g.stat_ld];
    }
    // ---- prologue: the six stages of "phases -6 .. -1": Ah0(0) Bh0(0) Bh1(0) Ah1(0) Ah0(1) Bh0(1)
    stage_a(0, 0, 0);
    stage_b(0, 0, 0);
    stage_b(0, 1, 0);
    stage_a(0, 1, 0);
    stage_a(1, 0, 1, nt > 1);
    stage_b(1, 0, 1, nt > 1);
    // the table's stores behind the six stages: untracked (gemm_common.h), or the compiler drains the stages in front of each of them
    if (lut) gelu_lut_fill<true>((float2 *)(smem + LUT_OFF), tid, 512);
    if constexpr (MERGED) OVO_VMCNT(2 * NA + NB);         // Ah0(0), Bh0(0), Bh1(0) landed (this wave's pieces): the first interval reads all three
    else OVO_VMCNT(2 * NA + 2 * NB);                      // Ah0(0), Bh0(0) landed (this wave's pieces)
    OVO_BARRIER();
    stamp(1);
    if (fold && tid < BM) {                               // partials in their fixed order -> mean, rstd (one-pass variance: the partials are sums and sums of squares)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float w = q < g.fold_parts ? 1.f : 0.f; s1 += w * fpart[q].x; s2 += w * fpart[q].y; }
        const float inv = 1.0f / (float)g.fold_D, mean = s1 * inv, var = fmaxf(s2 * inv - mean * mean, 0.f);
        lds_store_b64_untracked(row_ms + tid, mean, rsqrtf(var + g.fold_eps));      // (first read in the epilogue, behind the K-loop's waits and barriers)
    }
#ifdef OVO_GEMM_DEBUG
    if (g.dbg & 2) { OVO_VMCNT(0); return; }
#endif
    if (group == 1) OVO_BARRIER();                        // group 1 runs one barrier behind group 0 from here on
#ifdef OVO_GEMM_DEBUG
    if (g.dbg & 32) { load_b(smem, 1, wb1); }            // (ablation 32: wb1 holds something)
#endif

    // One K-tile = four phases.  The counted wait of a phase leaves exactly the stages of the last four phases in flight
    // (2 NA + 2 NB pieces), i.e. the stage issued four phases ago -- first read in the NEXT phase -- has landed.
    auto body = [&](auto PAR, int t) {
        constexpr int b = decltype(PAR)::value;
#ifdef OVO_GEMM_DEBUG        // ablations (tools/ builds; results are wrong): 16 = q2 multiplies the fragments q0 read, 32 = q1 too, 64 = no DMA traffic after the prologue
        const bool v1 = t + 1 < nt && !(g.dbg & 64), v2 = t + 2 < nt && !(g.dbg & 64);
#else
        const bool v1 = t + 1 < nt, v2 = t + 2 < nt;
#endif
        const char *cur = smem + b * BUF;
        // q0: A.sub0 x B.sub0
        load_a(cur, 0);
        load_b(cur, 0, wb0);
        stage_b(b ^ 1, 1, t + 1, v1);
        OVO_VMCNT(2 * NA + 2 * NB);
        OVO_BARRIER();
        quadrant(0, 0, wb0);
        OVO_BARRIER();
        // q1: A.sub0 x B.sub1
#ifdef OVO_GEMM_DEBUG
        if (!(g.dbg & 32))
#endif
        load_b(cur, 1, wb1);
        stage_a(b ^ 1, 1, t + 1, v1);
        OVO_VMCNT(2 * NA + 2 * NB);
        OVO_BARRIER();
        quadrant(0, 1, wb1);
        OVO_BARRIER();
        // q2: A.sub1 x B.sub1
#ifdef OVO_GEMM_DEBUG
        if (!(g.dbg & 16))
#endif
        load_a(cur, 1);
        stage_a(b, 0, t + 2, v2);
        OVO_VMCNT(2 * NA + 2 * NB);
        OVO_BARRIER();
        quadrant(1, 1, wb1);
        OVO_BARRIER();
        // q3: A.sub1 x B.sub0 (fragments kept from q0)
        stage_b(b, 0, t + 2, v2);
        OVO_VMCNT(2 * NA + 2 * NB);
        OVO_BARRIER();
        quadrant(1, 0, wb0);
        if (v1 || group == 0) OVO_BARRIER();              // group 1 skips its very last barrier: both groups execute 8 nt + 1
    };
    // MERGED: two intervals per K-tile.  What the ablations of the four-phase loop showed (profiles/r06_gemm_ablation.txt, 8192^3: 732 us): without the
    // fragment reads of q1 / q2 695, without any DMA after the prologue 673, without both 646 us = 1702 TFLOP/s -- the loop with NOTHING but q0's reads,
    // its MFMAs and its barriers still runs 29 % under the MFMA rate: ~17 cycles per MFMA issued and ~75 cycles per barrier interval of 16.  Here an
    // interval carries 32 MFMAs:
    //     A: read A.sub0, B.sub0, B.sub1 | stage Bh1(t+1), Ah1(t+1) | wait | barrier | q0, q1 | barrier
    //     B: read A.sub1                 | stage Ah0(t+2), Bh0(t+2) | wait | barrier | q2, q3 | barrier
    // Measured (profiles/r06_gemm_merged.txt, tools/gemm_bench.py, one session): 8192^3 740.7 -> 727.7 us, FC1 (16156, 4096, 1024) 132.8 -> 130.1, FC2 + f32
    // residual 114.8 -> 112.2, QKV + rotary 118.5 -> 115.5, K = 448 unchanged: +2 % where the K-loop dominates, bit-identical
    // (test_gemm_pingpong_merged_intervals_bit_identical); the default for the 256 x 256 tile (OVO_8P_MERGED=0: the four-phase loop).
    // A half-tile is restaged in the interval after the other group's last read of it; the load segment ends with lgkmcnt(0) so that those reads
    // HAVE returned before the barrier that lets the other group issue the DMA.  The wait of interval B leaves {Ah1(t+1), Ah0(t+2), Bh0(t+2)} in
    // flight (A.sub0 / B.sub0 / B.sub1 of K-tile t+1 landed: read two intervals later), that of interval A the four newest stages (Ah1(t) landed).
    auto body2 = [&](auto PAR, int t) {
        constexpr int b = decltype(PAR)::value;
        const bool v1 = t + 1 < nt, v2 = t + 2 < nt;
        const char *cur = smem + b * BUF;
        load_a(cur, 0);
        load_b(cur, 0, wb0);
        load_b(cur, 1, wb1);
        stage_b(b ^ 1, 1, t + 1, v1);
        stage_a(b ^ 1, 1, t + 1, v1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NA + 2 * NB) : "memory");
        OVO_BARRIER();
        quadrant(0, 0, wb0);
        quadrant(0, 1, wb1);
        OVO_BARRIER();
        load_a(cur, 1);
        stage_a(b, 0, t + 2, v2);
        stage_b(b, 0, t + 2, v2);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NA + NB) : "memory");
        OVO_BARRIER();
        quadrant(1, 1, wb1);
        quadrant(1, 0, wb0);
        if (v1 || group == 0) OVO_BARRIER();              // group 1 skips its very last barrier: both groups execute 4 nt + 1
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 1 < nt; t += 2) {                          // two K-tiles per trip (static buffer parity)
        if constexpr (MERGED) { body2(I0{}, t); body2(I1{}, t + 1); }
        else { body(I0{}, t); body(I1{}, t + 1); }
    }
    if (t < nt) { if constexpr (MERGED) body2(I0{}, t); else body(I0{}, t); }
    OVO_VMCNT(0);
    stamp(2);
#ifdef OVO_GEMM_DEBUG
    if (g.dbg & 4) { if (piece(0, 0)[0] == 12345.678f) *(float *)g.C = 1.f; return; }
#endif

    if constexpr (STAGED) {
        // ---- epilogue through LDS.  Stored straight from the accumulators a wave instruction writes 16 rows x 32-64 bytes; with 16-32
        // such groups per lane the store ISSUE (one request per row segment) cost more than the K-loop of a K = 1024 tile.  Instead each
        // wave transposes its tile through a private LDS slab (two passes of HM rows, rows padded by 16 bytes against bank conflicts)
        // and then reads rows back so that every store / residual load instruction covers whole 128-256-byte row segments.
        // (measured and dropped, round 4: 2-byte outputs WITHOUT any LDS round trip -- the lane's 4 columns of two neighbouring accumulator tiles packed and
        //  traded with v_permlane16_swap (8 consecutive columns per lane), then the two 32-column halves traded between lanes fr and fr ^ 8 (DPP row_ror:8) so
        //  that a 16-byte store per lane covers 8 rows x 128 bytes; no barrier, bit-identical.  With the generic math4 inlined 32 times it lost 15 % on the
        //  256 x 256 tile (instruction cache, see kind16 below); with the branch-free bodies it is 1-7 % AHEAD of the rounded-slab form in the micro-benchmark
        //  (FC1 + GELU 121.9 -> 117.3 us, Hiera stage-3 FC1 + GELU 116.1 -> 108.2) and 1-1.5 % BEHIND it in the bench (397.8 / 400.1 vs 404.0 / 403.9 frames/s):
        //  every wave's sixteen stores leave back to back with nothing between them, which the other stream's kernels pay for.  Debug bits 4 / 8 split the f32-slab
        //  epilogue's 7.9 us of a K = 1024 round into 5.0 us LDS + arithmetic and 2.9 us stores.)
        // epilogue kinds with a branch-free body in the accumulator layout (32 copies of the generic math4 -- every activation, rotary, residual -- were
        // 15 000 instructions of straight-line code: the 256 x 256 tile's epilogue ran out of the instruction cache and LOST 7 us): 0 plain, 1 table GELU.
        // (Rotary stays with the row-layout loop below: in this layout its 64 table loads per lane cover 16 rows x 64 bytes per instruction, and the ViT's
        // QKV measured 114.9 us -- 117.8 with all of a column tile's table rows fetched ahead of their arithmetic -- against 101-103.)
        // (2 = plain + the fused per-row argmax, round 6: the large-vocabulary query WITH its scores -- BASELINE configs[4] names "cosine scores within 1e-3
        // fp16" -- stored 8 bytes per lane straight from the accumulators (16 rows x 32 bytes per instruction: 4.08 ms for 1.25 M x 768 x 1000, a third of the
        // copy rate).  The row maximum is taken from the f32 values exactly as the unstaged form takes it; the scores then leave through the rounded slab.)
        const int kind16 = (g.out_dtype != 0 && !g.add && g.slab16) ? ((g.act == 0 && !g.rope_cos) ? (g.best ? 2 : 0) : ((g.act == 1 && lut && !g.rope_cos && !g.best) ? 1 : -1)) : -1;
        OVO_BARRIER();                                    // every wave's LDS-DMA has landed and every fragment read is done: LDS is free
        if (kind16 >= 0) {
            // 2-byte outputs without a residual: bias / activation / rotary are applied in the accumulator layout (a lane's 4 consecutive columns) and the
            // tile crosses LDS already ROUNDED -- half the slab bytes of the f32 form below, and the whole wave tile fits one pass (8 x 128 rows x 144 B).
            // Row stride 144 B = 36 dwords: the 16 rows x 8-byte pieces of a ds_write_b64 half-wave cover the 64 banks once (36 fr mod 64 = 16 distinct
            // multiples of 4); reads pair rows r and r + 8 in one 16-lane group (8 x 36 = 32 mod 64: the two rows' 128 bytes take opposite halves).
            constexpr int RB2 = WTN * 2 + 16, WROWS = 2 * HM;
            static_assert(WTN == 64, "read mapping below: 8 lanes x 16 bytes per row");
            char *slab = smem + wave * (WROWS * RB2);
            const int nw = n0 + wc * WTN;
            float4 bias_r[NJ], cs_r[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + j * CS + lcol;
                bias_r[j] = (g.bias && n < g.N) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                cs_r[j] = (fold && n < g.N) ? *(const float4 *)(g.fold_cs + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            auto write_all = [&](auto KIND_, auto FOLD_) {
                constexpr int KIND = decltype(KIND_)::value;
                constexpr bool FOLD = decltype(FOLD_)::value;
                static_for<0, NI>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    float row_mx = -3.0e38f;
                    int row_arg = 0x7fffffff;
                    float2 ms = make_float2(0.f, 1.f);
                    if constexpr (FOLD) ms = row_ms[wr * WTM + i * RT + lrow];                 // (mean, rstd) of this lane's row
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f32x4 pc = piece(i, j);
                        float v[4] = {pc[0], pc[1], pc[2], pc[3]};
                        if constexpr (FOLD) {                                             // LN(x) . W^T + b from bf16(x) . W'^T: rstd (acc - mean colsum) + b'
                            v[0] = ms.y * (v[0] - ms.x * cs_r[j].x); v[1] = ms.y * (v[1] - ms.x * cs_r[j].y);
                            v[2] = ms.y * (v[2] - ms.x * cs_r[j].z); v[3] = ms.y * (v[3] - ms.x * cs_r[j].w);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= g.alpha;                // (the operations of math4, in its order)
                        }
                        v[0] += bias_r[j].x; v[1] += bias_r[j].y; v[2] += bias_r[j].z; v[3] += bias_r[j].w;
                        if (KIND == 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = gelu_lut(v[r], lut);
                        }
                        if (KIND == 2) {                                                 // first maximum over ascending columns, as finish4's running pair
                            const int n = nw + j * CS + lcol;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n + r < g.n_valid && v[r] > row_mx) { row_mx = v[r]; row_arg = n + r; }
                        }
                        uint2 p;
                        if (g.out_dtype == 2) { p.x = pack_bf16(v[0], v[1]); p.y = pack_bf16(v[2], v[3]); }
                        else { p.x = pack_f16(v[0], v[1]); p.y = pack_f16(v[2], v[3]); }
                        *(uint2 *)(slab + (i * RT + lrow) * RB2 + (j * CS + lcol) * 2) = p;
                    }
                    if constexpr (KIND == 2 && MF == 16) {
                        const int m = m0 + wr * WTM + i * RT + lrow;
                        if (m < g.M) finish_best(g, m, fq, row_mx, row_arg);
                    }
                });
            };
            using NoFold = std::integral_constant<bool, false>;
            using Fold = std::integral_constant<bool, true>;
            if (kind16 == 0) { if (fold) write_all(std::integral_constant<int, 0>{}, Fold{}); else write_all(std::integral_constant<int, 0>{}, NoFold{}); }
            else if (kind16 == 2) write_all(std::integral_constant<int, 2>{}, NoFold{});
            else { if (fold) write_all(std::integral_constant<int, 1>{}, Fold{}); else write_all(std::integral_constant<int, 1>{}, NoFold{}); }
            OVO_FENCE();
            const int q = lane >> 3, c = (lane & 7) * 8, n = nw + c;
            const bool in0 = n < g.N, in1 = n + 4 < g.N;
#pragma unroll 4
            for (int it = 0; it < WROWS / 8; ++it) {
                const int r = 16 * (it >> 1) + 4 * (it & 1) + (q >> 1) + 8 * (q & 1), m = m0 + wr * WTM + r;
                const uint4 p = *(const uint4 *)(slab + r * RB2 + c * 2);
                const long long md = (m < g.M && in0) ? row_dest(g, m) : -1;
                if (md < 0) continue;
                uint16_t *dst = (uint16_t *)g.C + md * g.ldc + n;
#ifdef OVO_GEMM_DEBUG
                if ((g.dbg & 8) && p.x != 0x12345678u) continue;
#endif
                if (in1 && ((uintptr_t)dst & 15) == 0) __builtin_nontemporal_store(*(const __attribute__((ext_vector_type(4))) unsigned *)&p, (__attribute__((ext_vector_type(4))) unsigned *)dst);
                else {
                    *(uint2 *)dst = make_uint2(p.x, p.y);
                    if (in1) *(uint2 *)(dst + 4) = make_uint2(p.z, p.w);
                }
            }
            if (g.tail_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(3);
            return;
        }
        // The f32-slab form, instantiated per epilogue KIND so that the two forms the encoders run all day are branch-free and short (see kind16 above:
        // code size decides here): 3 = f32 output += f32 residual (out projection / FC2), 2 = 2-byte output with rotary (the ViT's QKV), -1 = anything
        // (math4 with every activation, either output type).
        constexpr int ROWB = WTN * 4 + 16;
        char *slab = smem + wave * (HM * ROWB);
        auto rows_out = [&](auto KIND_) {
        constexpr int KIND = decltype(KIND_)::value;
        auto mathk = [&](int tk, int nh, int n, float (&v)[4], float4 bias, float4 addv) __attribute__((always_inline)) {
            if constexpr (KIND < 0) math4(g, tk, nh, n, v, bias, addv, lut);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= g.alpha;            // (math4's operations, in its order)
                v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
                if constexpr (KIND == 2) {
                    if (n < g.rope_cols && tk >= g.rope_t0) {
                        const long long at = (long long)tk * g.rope_hd + nh;
                        const float4 c = *(const float4 *)(g.rope_cos + at), sn = *(const float4 *)(g.rope_sin + at);
                        const float y0 = v[0] * c.x - v[1] * sn.x, y1 = v[1] * c.y + v[0] * sn.y;
                        const float y2 = v[2] * c.z - v[3] * sn.z, y3 = v[3] * c.w + v[2] * sn.w;
                        v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
                    }
                }
                if constexpr (KIND == 3 || KIND == 4) { v[0] += addv.x; v[1] += addv.y; v[2] += addv.z; v[3] += addv.w; }
            }
        };
        // (measured and dropped, round 6: KIND 3's residual rows FETCHED AHEAD of the slab round trip -- two groups of 8 row instructions in flight, pass 1's
        //  issued under pass 0's rows; 252 VGPRs, no spill, bit-identical.  (16156, 1024, 1024) + f32 residual 52.3 vs 52.4 us, (16156, 1024, 4096) 110.5 vs
        //  108.6, bench 439.7 vs 442.4 frames/s (profiles/r06_gemm_addahead.txt): the +17 us of this epilogue is the 113 MB it moves, not its load latency.)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < NI / 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *(f32x4 *)(slab + (i * RT + lrow) * ROWB + (j * CS + lcol) * 4) = piece(pass * (NI / 2) + i, j);
            OVO_FENCE();
            const int mw = m0 + wr * WTM + pass * HM, nw = n0 + wc * WTN;
            if (KIND == 3 || KIND == 4 || (KIND < 0 && g.out_dtype == 0)) {          // f32 rows: 16 lanes x 16 bytes per row, 4 rows per instruction
                constexpr int LPR = WTN / 4, RPI = 64 / LPR;
                const int c = (lane % LPR) * 4, n = nw + c;
                const float4 bias = (g.bias && n < g.N) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                int tok = 0, nh = 0, tstep = 0;           // rotary position of the lane's row, advanced RPI rows per trip
                if (KIND < 0 && g.rope_cos) { tok = (mw + lane / LPR) % g.rope_T; nh = n % g.rope_hd; tstep = RPI % g.rope_T; }
                float keep1 = 0.f, keep2 = 0.f;           // KIND 4: (sum, sum of squares) of row mw + lane over this wave's 64 columns
#pragma unroll 4
                for (int it = 0; it < HM / RPI; ++it) {
                    const int r = it * RPI + lane / LPR, m = mw + r, tk = tok;
                    tok += tstep;
                    if (tok >= g.rope_T) tok -= g.rope_T;
                    const f32x4 a = *(const f32x4 *)(slab + r * ROWB + c * 4);
                    const long long md = (m < g.M && n < g.N) ? row_dest(g, m) : -1;
                    float s1 = 0.f, s2 = 0.f;
                    if (KIND != 4 && md < 0) continue;
                    if (KIND != 4 || md >= 0) {
                    const float4 addv = (KIND == 3 || KIND == 4 || g.add) ? *(const float4 *)(g.add + add_row(g, m, md) * g.ld_add + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    mathk(tk, nh, n, v, bias, addv);
                    { const f32x4 vv = {v[0], v[1], v[2], v[3]};
                      if (g.res_plain) *(f32x4 *)((float *)g.C + md * g.ldc + n) = vv;        // (OVO_8P_RES_PLAIN=1, measurement: the f32 stream kept in the caches for the LayerNorm that follows)
                      else __builtin_nontemporal_store(vv, (f32x4 *)((float *)g.C + md * g.ldc + n)); }
                    if constexpr (KIND == 4) {
                        // LayerNorm fold, producer side: the row's bf16 copy (the next product's A operand) and this wave's share of its (sum, sum of
                        // squares) -- the row's 64 columns sit in the 16 lanes of one DPP row: four row_shr adds, lane 15 of the row holds the total
                        static_assert(LPR == 16 && WTN == 64 && HM <= 64, "one DPP row per output row, one partial per 64-column wave tile, at most one row per lane and pass");
                        *(uint2 *)(g.xb_out + md * g.ld_xb + n) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                        s1 = (v[0] + v[1]) + (v[2] + v[3]); s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x111, 0xf, 0xf, true));
                        s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x111, 0xf, 0xf, true));
                        s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x112, 0xf, 0xf, true));
                        s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x112, 0xf, 0xf, true));
                        s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x114, 0xf, 0xf, true));
                        s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x114, 0xf, 0xf, true));
                        s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x118, 0xf, 0xf, true));
                        s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x118, 0xf, 0xf, true));
                    }
                    }
                    if constexpr (KIND == 4) {
                        // every lane active again: rows 4 it .. 4 it + 3 have their totals in lanes 15, 31, 47, 63 -- lanes 4 it .. 4 it + 3 fetch them (crossbar,
                        // no memory), so that after the HM / 4 trips lane L < HM holds row L of the pass and the statistics leave as ONE 256-512-byte store per wave and pass
                        const int src = ((lane & 3) * 16 + 15) * 4;
                        const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s1))), t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s2)));
                        if ((lane >> 2) == it) { keep1 = t1; keep2 = t2; }
                    }
                }
                if constexpr (KIND == 4) {
                    if (lane < HM && mw + lane < g.M && nw < g.N) *((float2 *)g.stat_out + (long long)(nw >> 6) * g.stat_ld + (mw + lane)) = make_float2(keep1, keep2);
                }
            } else {                                      // 2-byte rows: 8 lanes x 16 bytes per row, 8 rows per instruction
                constexpr int LPR = WTN / 8, RPI = 64 / LPR;
                const int c = (lane % LPR) * 8, n = nw + c;
                const bool in0 = n < g.N, in1 = n + 4 < g.N;
                const float4 bias0 = (g.bias && in0) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 bias1 = (g.bias && in1) ? *(const float4 *)(g.bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                int tok = 0, nh0 = 0, nh1 = 0, tstep = 0;
                if (KIND == 2 || (KIND < 0 && g.rope_cos)) { tok = (mw + lane / LPR) % g.rope_T; nh0 = n % g.rope_hd; nh1 = (n + 4) % g.rope_hd; tstep = RPI % g.rope_T; }
#pragma unroll 4
                for (int it = 0; it < HM / RPI; ++it) {
                    const int r = it * RPI + lane / LPR, m = mw + r, tk = tok;
                    tok += tstep;
                    if (tok >= g.rope_T) tok -= g.rope_T;
                    const f32x4 a0 = *(const f32x4 *)(slab + r * ROWB + c * 4), a1 = *(const f32x4 *)(slab + r * ROWB + c * 4 + 16);
                    const long long md = (m < g.M && in0) ? row_dest(g, m) : -1;
                    if (md < 0) continue;
                    float4 add0 = make_float4(0.f, 0.f, 0.f, 0.f), add1 = add0;
                    if (KIND < 0 && g.add) {
                        const float *ap = g.add + add_row(g, m, md) * g.ld_add + n;
                        add0 = *(const float4 *)ap;
                        if (in1) add1 = *(const float4 *)(ap + 4);
                    }
                    float v0[4] = {a0[0], a0[1], a0[2], a0[3]}, v1[4] = {a1[0], a1[1], a1[2], a1[3]};
                    mathk(tk, nh0, n, v0, bias0, add0);
                    mathk(tk, nh1, n + 4, v1, bias1, add1);
                    uint4 p;
                    if (g.out_dtype == 2) { p.x = pack_bf16(v0[0], v0[1]); p.y = pack_bf16(v0[2], v0[3]); p.z = pack_bf16(v1[0], v1[1]); p.w = pack_bf16(v1[2], v1[3]); }
                    else { p.x = pack_f16(v0[0], v0[1]); p.y = pack_f16(v0[2], v0[3]); p.z = pack_f16(v1[0], v1[1]); p.w = pack_f16(v1[2], v1[3]); }
                    uint16_t *dst = (uint16_t *)g.C + md * g.ldc + n;
                    // non-temporal: the tile is written once and read by a later kernel; keeping it out of the way of the operands the
                    // other workgroups are still streaming through L2 measured 4-10 % on the FC1 products (tools/gemm_bench.py)
#ifdef OVO_GEMM_DEBUG
                    if ((g.dbg & 8) && p.x != 0x12345678u) continue;          // tools/ builds only: the epilogue without its global stores
#endif
                    if (in1 && ((uintptr_t)dst & 15) == 0) __builtin_nontemporal_store(*(const __attribute__((ext_vector_type(4))) unsigned *)&p, (__attribute__((ext_vector_type(4))) unsigned *)dst);
                    else {
                        *(uint2 *)dst = make_uint2(p.x, p.y);
                        if (in1) *(uint2 *)(dst + 4) = make_uint2(p.z, p.w);
                    }
                }
            }
            OVO_FENCE();                                  // the slab is rewritten by the next pass: its reads above come first (same wave)
        }
        };
        // Rotary with the table rows of the workgroup's tokens staged in LDS (head_dim = the wave tile's 64 columns, so the four / two column waves of a
        // row group need the SAME [32 rows][64] slice of cos and sin): 32-row passes, per pass one cooperative load of the slice (4-8 float4 per thread)
        // instead of 4 table loads per lane and 8 rows from every wave: the ViT's QKV (13848, 3072, 1024) 99.2 -> 91.3 us, at 4 keyframes 37.6 -> 31.8
        // (plain: 81-83); same values, same arithmetic as math4: bit-identical.
        auto rope_out = [&](auto FOLD_) {
            constexpr bool FOLD = decltype(FOLD_)::value;
            constexpr int PR = 32, NPASS = WTM / PR, TRB = 64 * 4 + 16;
            constexpr int SLABS = 8 * PR * ROWB, TBL = WARPS_M * PR * TRB;      // table: cos block, then sin block
            char *sl = smem + wave * (PR * ROWB);
            char *tb = smem + SLABS;
            constexpr int NPIECE = 2 * WARPS_M * PR * 16 / 512;                  // float4 pieces of the slice per thread
            constexpr int LPR = WTN / 8, RPI = 64 / LPR;
            const int c = (lane % LPR) * 8, nw = n0 + wc * WTN, n = nw + c;
            const bool in0 = n < g.N, in1 = n + 4 < g.N, rot = n < g.rope_cols;
            const float4 bias0 = (g.bias && in0) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bias1 = (g.bias && in1) ? *(const float4 *)(g.bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float cs0[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FOLD) {
                const float4 q0 = in0 ? *(const float4 *)(g.fold_cs + n) : make_float4(0.f, 0.f, 0.f, 0.f), q1 = in1 ? *(const float4 *)(g.fold_cs + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                cs0[0] = q0.x; cs0[1] = q0.y; cs0[2] = q0.z; cs0[3] = q0.w; cs1[0] = q1.x; cs1[1] = q1.y; cs1[2] = q1.z; cs1[3] = q1.w;
            }
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                if (pass) OVO_BARRIER();                            // every wave is done with the previous slice
#pragma unroll
                for (int il = 0; il < PR / RT; ++il)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *(f32x4 *)(sl + (il * RT + lrow) * ROWB + (j * CS + lcol) * 4) = piece(pass * (PR / RT) + il, j);
#pragma unroll
                for (int k = 0; k < NPIECE; ++k) {
                    const int id = k * 512 + tid, c4 = id & 15, r = (id >> 4) % PR, wrg = ((id >> 4) / PR) % WARPS_M, tbl = (id >> 4) / (PR * WARPS_M);
                    const int tok = (m0 + wrg * WTM + pass * PR + r) % g.rope_T;
                    const float4 v = *(const float4 *)((tbl ? g.rope_sin : g.rope_cos) + (long long)tok * 64 + c4 * 4);
                    *(float4 *)(tb + tbl * TBL + (wrg * PR + r) * TRB + c4 * 16) = v;
                }
                OVO_BARRIER();
                const char *tc = tb + (wr * PR) * TRB + c * 4, *ts = tc + TBL;
#pragma unroll
                for (int it = 0; it < PR / RPI; ++it) {
                    const int r = it * RPI + lane / LPR, m = m0 + wr * WTM + pass * PR + r;
                    const f32x4 a0 = *(const f32x4 *)(sl + r * ROWB + c * 4), a1 = *(const f32x4 *)(sl + r * ROWB + c * 4 + 16);
                    const f32x4 c0 = *(const f32x4 *)(tc + r * TRB), c1 = *(const f32x4 *)(tc + r * TRB + 16);
                    const f32x4 s0 = *(const f32x4 *)(ts + r * TRB), s1 = *(const f32x4 *)(ts + r * TRB + 16);
                    const long long md = (m < g.M && in0) ? row_dest(g, m) : -1;
                    if (md < 0) continue;
                    float v0[4] = {a0[0], a0[1], a0[2], a0[3]}, v1[4] = {a1[0], a1[1], a1[2], a1[3]};
                    if constexpr (FOLD) {
                        const float2 ms = row_ms[wr * WTM + pass * PR + r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = ms.y * (v0[e] - ms.x * cs0[e]); v1[e] = ms.y * (v1[e] - ms.x * cs1[e]); }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] *= g.alpha; v1[e] *= g.alpha; }
                    }
                    v0[0] += bias0.x; v0[1] += bias0.y; v0[2] += bias0.z; v0[3] += bias0.w;
                    v1[0] += bias1.x; v1[1] += bias1.y; v1[2] += bias1.z; v1[3] += bias1.w;
                    if (rot && m % g.rope_T >= g.rope_t0) {
                        const float y0 = v0[0] * c0[0] - v0[1] * s0[0], y1 = v0[1] * c0[1] + v0[0] * s0[1];
                        const float y2 = v0[2] * c0[2] - v0[3] * s0[2], y3 = v0[3] * c0[3] + v0[2] * s0[3];
                        const float z0 = v1[0] * c1[0] - v1[1] * s1[0], z1 = v1[1] * c1[1] + v1[0] * s1[1];
                        const float z2 = v1[2] * c1[2] - v1[3] * s1[2], z3 = v1[3] * c1[3] + v1[2] * s1[3];
                        v0[0] = y0; v0[1] = y1; v0[2] = y2; v0[3] = y3; v1[0] = z0; v1[1] = z1; v1[2] = z2; v1[3] = z3;
                    }
                    uint4 p;
                    if (g.out_dtype == 2) { p.x = pack_bf16(v0[0], v0[1]); p.y = pack_bf16(v0[2], v0[3]); p.z = pack_bf16(v1[0], v1[1]); p.w = pack_bf16(v1[2], v1[3]); }
                    else { p.x = pack_f16(v0[0], v0[1]); p.y = pack_f16(v0[2], v0[3]); p.z = pack_f16(v1[0], v1[1]); p.w = pack_f16(v1[2], v1[3]); }
                    uint16_t *dst = (uint16_t *)g.C + md * g.ldc + n;
                    if (in1 && ((uintptr_t)dst & 15) == 0) __builtin_nontemporal_store(*(const __attribute__((ext_vector_type(4))) unsigned *)&p, (__attribute__((ext_vector_type(4))) unsigned *)dst);
                    else {
                        *(uint2 *)dst = make_uint2(p.x, p.y);
                        if (in1) *(uint2 *)(dst + 4) = make_uint2(p.z, p.w);
                    }
                }
                OVO_FENCE();
            }
        };
        if (g.out_dtype == 0 && g.add && g.act == 0 && !g.rope_cos) { if (g.xb_out) rows_out(std::integral_constant<int, 4>{}); else rows_out(std::integral_constant<int, 3>{}); }
        else if (g.out_dtype != 0 && !g.add && g.act == 0 && g.rope_cos && g.rope_hd == 64 && n0 % 64 == 0 && g.rope_lds) {
            if (fold) rope_out(std::integral_constant<bool, true>{}); else rope_out(std::integral_constant<bool, false>{});
        }
        else if (g.out_dtype != 0 && !g.add && g.act == 0 && g.rope_cos) rows_out(std::integral_constant<int, 2>{});
        else rows_out(std::integral_constant<int, -1>{});
    } else if constexpr (MF == 16) {
        // ---- epilogue straight from the accumulators (the fused-argmax form needs a row's columns in neighbouring lanes):
        // acc[i][j][r] = C[m = m0 + wr WTM + 16 i + fr][n = n0 + wc WTN + 16 j + 4 fq + r]
        float4 bias_r[2 * TNH];
#pragma unroll
        for (int j = 0; j < 2 * TNH; ++j) {
            const int n = n0 + wc * WTN + j * 16 + fq * 4;
            bias_r[j] = (g.bias && n < g.N) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // scores that only feed the argmax (the large-vocabulary query, BASELINE configs[4]: nothing stored, no residual, plain row order): a
        // row's maximum first (v_max3: half an instruction per score), then the first column that holds it (compare + select, last to first) --
        // 2.5 VALU instructions per score instead of the ~5 of the running (score, column) pair below; 10 M x 768 x 1000 texts 25.7 -> see DESIGN.md
        const bool argmax_only = g.best && !g.store && !g.add && g.win_per <= 0 && !g.rope_cos;
        if (argmax_only) {
            const bool plain = g.alpha == 1.0f && !g.bias && !g.act;
            const bool edge = n0 + wc * WTN + WTN > g.n_valid;         // this wave's columns reach past the vocabulary (wave-uniform)
            static_for<0, 2 * TMH>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                const int m = m0 + wr * WTM + i * 16 + fr;
                float v[2 * TNH][4];
#pragma unroll
                for (int j = 0; j < 2 * TNH; ++j) {
                    const int n = n0 + wc * WTN + j * 16 + fq * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r];
                    if (!plain) math4(g, 0, 0, n, v[j], bias_r[j], make_float4(0.f, 0.f, 0.f, 0.f));
                    if (edge) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[j][r] = n + r < g.n_valid ? v[j][r] : -3.0e38f;
                    }
                }
                float mx = fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[0][2], v[0][3]));
#pragma unroll
                for (int j = 1; j < 2 * TNH; ++j) mx = fmaxf(fmaxf(fmaxf(mx, v[j][0]), v[j][1]), fmaxf(v[j][2], v[j][3]));
                int arg = 0x7fffffff;
#pragma unroll
                for (int j = 2 * TNH - 1; j >= 0; --j)
#pragma unroll
                    for (int r = 3; r >= 0; --r) arg = v[j][r] == mx ? n0 + wc * WTN + j * 16 + fq * 4 + r : arg;
                if (!(mx > -3.0e38f)) arg = 0x7fffffff;                 // nothing valid here (as the running form: a score must exceed -3e38 to win)
                if (m < g.M) finish_best(g, m, fq, mx, arg);
            });
        } else
        static_for<0, 2 * TMH>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            const int m = m0 + wr * WTM + i * 16 + fr;
            const long long md = m < g.M ? row_dest(g, m) : -1;
            if (md >= 0) {
                float4 add_r[2 * TNH];
#pragma unroll
                for (int j = 0; j < 2 * TNH; ++j) {
                    const int n = n0 + wc * WTN + j * 16 + fq * 4;
                    add_r[j] = (g.add && n < g.N) ? *(const float4 *)(g.add + add_row(g, m, md) * g.ld_add + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float row_best = -3.0e38f;
                int row_arg = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 2 * TNH; ++j) {
                    const int n = n0 + wc * WTN + j * 16 + fq * 4;
                    if (n < g.N) finish4(g, m, md, n, acc[i][j], bias_r[j], add_r[j], row_best, row_arg);
                }
                if (g.best) finish_best(g, m, fq, row_best, row_arg);
            }
        });
    }
    if (g.tail_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
#endif
}

template <int BM, int BN, int WARPS_M, typename VT, bool STAGED, int MF = 16, bool MERGED = false>
int launch8p_(const GemmArgs &g0, hipStream_t s) {
    GemmArgs g = g0;
    g.dbg = 0; g.stamps = nullptr;
#ifdef OVO_GEMM_DEBUG        // the stamp buffer's ADDRESS comes from the environment: never in a production build
    g.dbg = (getenv("OVO_8P_DEBUG") ? atoi(getenv("OVO_8P_DEBUG")) : 0) | ((getenv("OVO_8P_DELAY") ? atoi(getenv("OVO_8P_DELAY")) : 0) << 8);
    g.stamps = getenv("OVO_8P_STAMPS") ? (unsigned long long *)strtoull(getenv("OVO_8P_STAMPS"), nullptr, 0) : nullptr;
#endif
    static bool no_chunk = getenv("OVO_GEMM_NO_CHUNK") != nullptr;              // tuning knobs: read once (see ovo_knobs_dynamic)
    static int strip_env = getenv("OVO_GEMM_STRIP") ? atoi(getenv("OVO_GEMM_STRIP")) : -1;
    static int tail_wait = getenv("OVO_8P_TAILWAIT") ? atoi(getenv("OVO_8P_TAILWAIT")) : 0;
    if (ovo_knobs_dynamic()) { no_chunk = getenv("OVO_GEMM_NO_CHUNK") != nullptr; strip_env = getenv("OVO_GEMM_STRIP") ? atoi(getenv("OVO_GEMM_STRIP")) : -1;
                               tail_wait = getenv("OVO_8P_TAILWAIT") ? atoi(getenv("OVO_8P_TAILWAIT")) : 0; }
    g.tail_wait = tail_wait;
    static int no_slab16 = getenv("OVO_8P_NO_SLAB16") != nullptr;
    if (ovo_knobs_dynamic()) no_slab16 = getenv("OVO_8P_NO_SLAB16") != nullptr;
    g.slab16 = !no_slab16;
    static int rope_lds_on = getenv("OVO_8P_ROPE_LDS") ? atoi(getenv("OVO_8P_ROPE_LDS")) : 1;      // (0: every wave loads its table rows itself)
    if (ovo_knobs_dynamic()) rope_lds_on = getenv("OVO_8P_ROPE_LDS") ? atoi(getenv("OVO_8P_ROPE_LDS")) : 1;
    g.rope_lds = rope_lds_on;
    static int gelu_poly = getenv("OVO_GELU_POLY") != nullptr;
    if (ovo_knobs_dynamic()) gelu_poly = getenv("OVO_GELU_POLY") != nullptr;
    g.gelu_lut = !gelu_poly;
    static int res_plain = getenv("OVO_8P_RES_PLAIN") ? atoi(getenv("OVO_8P_RES_PLAIN")) : 0;
    if (ovo_knobs_dynamic()) res_plain = getenv("OVO_8P_RES_PLAIN") ? atoi(getenv("OVO_8P_RES_PLAIN")) : 0;
    g.res_plain = res_plain;
    if (g.fold_stats || g.xb_out) {            // the LayerNorm fold lives in the staged epilogues the ViT's products take; anything else is refused, not approximated
        const bool cons_ok = !g.fold_stats || (STAGED && g.out_dtype != 0 && !g.add && !g.best && g.fold_cs && g.fold_parts >= 1 && g.fold_parts <= 16 && g.N % 64 == 0 && g.slab16 &&
                                               ((!g.rope_cos && (g.act == 0 || (g.act == 1 && g.gelu_lut))) || (g.rope_cos && g.act == 0 && g.rope_hd == 64 && g.rope_lds)));
        const bool prod_ok = !g.xb_out || (STAGED && g.out_dtype == 0 && g.add && g.act == 0 && !g.rope_cos && g.stat_out && g.N % 64 == 0 && BN / (8 / WARPS_M) == 64 && g.win_per <= 0);
        if (!cons_ok || !prod_ok || (g.fold_stats && g.xb_out)) { ovo_set_error("ovo_gemm: this product cannot take the folded LayerNorm"); return OVO_E_UNSUPPORTED; }
    }
    g.nbn = (g.N + BN - 1) / BN;
    const int nbm = (g.M + BM - 1) / BM;
    constexpr size_t ring = 2 * (size_t)(BM + BN) * 128;                                  // two K-tile buffers
    constexpr size_t slabs = 8 * (size_t)(BM / WARPS_M / 2) * ((BN / (8 / WARPS_M)) * 4 + 16);     // epilogue: 8 x HM rows x (4 WTN + 16) bytes
    constexpr size_t slabs16 = 8 * (size_t)(BM / WARPS_M) * ((BN / (8 / WARPS_M)) * 2 + 16);                   // 2-byte epilogue: 8 x WTM rows x (2 WTN + 16) bytes
    constexpr size_t lds = (STAGED ? (slabs > ring ? (slabs > slabs16 ? slabs : slabs16) : (slabs16 > ring ? slabs16 : ring)) : ring) + (STAGED ? GELU_LUT_BYTES : 0);
    constexpr size_t rope_lds = 8 * 32 * (size_t)((BN / (8 / WARPS_M)) * 4 + 16) + 2 * (size_t)WARPS_M * 32 * (64 * 4 + 16);      // rope_out: 32-row slabs + the table slice
    constexpr size_t lds_all = lds > rope_lds ? lds : rope_lds;
    constexpr size_t fold_lds = lds_all + 256 * 8;                                        // + (mean, rstd) of the tile's rows (the LayerNorm fold's consumer)
    static_assert(fold_lds <= 160 * 1024, "LDS");
    static bool attr_done = false;              // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm8p<BM, BN, WARPS_M, VT, STAGED, MF, MERGED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_lds);
        if (e != hipSuccess) { ovo_set_error("ovo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        attr_done = true;
    }
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(BN == 256 ? 3 : 0, 2.0 * g.M * (double)g.N * g.K, s); ovo_prof_shape(g.M, g.N, g.K); ovo_prof_flags(gemm_flags(g)); ovo_prof_bytes(gemm_algorithmic_bytes(g)); }     // kinds 3 / 0: 256x256 / 256x128
    g.tiles = nbm * g.nbn;
    g.chunk = (g.M > g.N || g.nbn % 8 != 0) && !no_chunk ? (g.tiles + 7) / 8 : 0;
    // tile order: measured to matter little (the K-loop is bound by the L2->LDS arrival rate, not by L2 misses); column strips of 8 n-tiles
    // gain ~5% on the widest products (N/BN >= 16: the per-XCD working set of a round drops under the 4 MB L2), nothing elsewhere
    g.strip = strip_env >= 0 ? strip_env : (g.nbn >= 16 ? 8 : 0);
    if (g.strip > 0) g.chunk = (g.tiles + 7) / 8;
    const int grid = g.chunk > 0 ? g.chunk * 8 : g.tiles;
    k_gemm8p<BM, BN, WARPS_M, VT, STAGED, MF, MERGED><<<grid, 512, (STAGED && g.fold_stats) ? fold_lds : ((STAGED && g.rope_cos) ? lds_all : lds), s>>>(g);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

template <int BM, int BN, int WARPS_M, typename VT>
int launch8p(const GemmArgs &g, hipStream_t s) {
    // OVO_8P_MFMA32 = 1: the 32 x 32 x 16 MFMA K-loop for the staged epilogues (bf16; measured against the 16 x 16 x 32 loop in profiles/r05*_gemm_mfma32.txt)
    static int mf32 = getenv("OVO_8P_MFMA32") ? atoi(getenv("OVO_8P_MFMA32")) : OVO_8P_MFMA32_DEFAULT;
    if (ovo_knobs_dynamic()) mf32 = getenv("OVO_8P_MFMA32") ? atoi(getenv("OVO_8P_MFMA32")) : OVO_8P_MFMA32_DEFAULT;
    // the fused argmax: straight from the accumulators -- unless the scores are stored too as plain 2-byte rows, which leave through the staged epilogue
    static int best_staged = getenv("OVO_8P_BEST_STAGED") ? atoi(getenv("OVO_8P_BEST_STAGED")) : 1;
    if (ovo_knobs_dynamic()) best_staged = getenv("OVO_8P_BEST_STAGED") ? atoi(getenv("OVO_8P_BEST_STAGED")) : 1;
    if (g.best && !(best_staged && g.store && g.out_dtype != 0 && !g.add && !g.act && !g.rope_cos && g.win_per <= 0 && BN / (8 / WARPS_M) == 64 && !getenv("OVO_8P_NO_SLAB16")))
        return launch8p_<BM, BN, WARPS_M, VT, false>(g, s);
    if (g.best) return launch8p_<BM, BN, WARPS_M, VT, true>(g, s);
    if constexpr (std::is_same<VT, bf16x8>::value) {
        if (mf32) return launch8p_<BM, BN, WARPS_M, VT, true, 32>(g, s);
    }
    // OVO_8P_MERGED: two barrier intervals per K-tile (the 256 x 256 tile's staged forms)
    static int merged = getenv("OVO_8P_MERGED") ? atoi(getenv("OVO_8P_MERGED")) : OVO_8P_MERGED_DEFAULT;
    if (ovo_knobs_dynamic()) merged = getenv("OVO_8P_MERGED") ? atoi(getenv("OVO_8P_MERGED")) : OVO_8P_MERGED_DEFAULT;
    if constexpr (BN == 256) {
        if (merged) return launch8p_<BM, BN, WARPS_M, VT, true, 16, true>(g, s);
    }
    return launch8p_<BM, BN, WARPS_M, VT, true>(g, s);
}


}  // namespace

namespace ovo_gemm_detail {

static int gemm8p_one(const GemmArgs &g, int bn, int in_dtype, hipStream_t s);

int gemm8p_launch(const GemmArgs &g, int bn, int in_dtype, hipStream_t s) {
    if (g.K % 64 != 0 || g.K < 64) return OVO_E_UNSUPPORTED;
    if ((long long)g.N * g.ldw * 2 >= (1ll << 32)) return OVO_E_UNSUPPORTED;                                              // 32-bit DMA offsets
    if ((long long)g.M * g.lda * 2 >= (1ll << 32)) {
        // an activation matrix past 4 GB (the 10 M-point query of BASELINE configs[4]: 15 GB of f16 features): row chunks of < 4 GB, one launch each.
        // Rows are independent; only the plain row mapping is chunked (no window map, no periodic residual, no operand-load LayerNorm)
        if (g.win_per > 0 || g.add_rows > 0 || g.ln_mode || g.rope_cos) return OVO_E_UNSUPPORTED;
        const long long rows = (((1ll << 32) - 1) / (g.lda * 2)) & ~255ll;
        if (rows < 256) return OVO_E_UNSUPPORTED;
        const size_t csz = g.out_dtype == 0 ? 4 : 2;
        for (long long m0 = 0; m0 < g.M; m0 += rows) {
            GemmArgs c = g;
            c.M = (int)((g.M - m0) < rows ? (g.M - m0) : rows);
            c.A = g.A + m0 * g.lda * 2;
            if (g.C) c.C = (char *)g.C + m0 * g.ldc * csz;
            if (g.add) c.add = g.add + m0 * g.ld_add;
            if (g.best) c.best = g.best + m0;
            const int rc = gemm8p_one(c, bn, in_dtype, s);
            if (rc != OVO_OK) return rc;
        }
        return OVO_OK;
    }
    return gemm8p_one(g, bn, in_dtype, s);
}

static int gemm8p_one(const GemmArgs &g, int bn, int in_dtype, hipStream_t s) {
    if (bn == 256) return in_dtype == 2 ? launch8p<256, 256, 2, bf16x8>(g, s) : launch8p<256, 256, 2, f16x8>(g, s);
    if (bn == 128) return in_dtype == 2 ? launch8p<256, 128, 4, bf16x8>(g, s) : launch8p<256, 128, 4, f16x8>(g, s);
    return OVO_E_UNSUPPORTED;
}

}  // namespace ovo_gemm_detail
