// gemm8q.hip -- the PERSISTENT form of the 256 x 128 ping-pong GEMM (gemm8p.hip) for two-byte outputs without a residual operand
// (the QKV and FC1 projections of the batched encoder forwards):  C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias)  (+ rotary embedding).
//
// What the one-tile-per-workgroup kernel loses (profiles/r02e_gemm8p_timeline.txt): with one workgroup per CU nothing overlaps a tile's
// prologue (ring fill from HBM / L2) and epilogue (every CU storing its 64-128 KB at once: ~3.3 TB/s chip-wide) -- 10 of 34 us per tile
// at K = 1024, a third of a workgroup's life at K = 448.  Here a workgroup walks through its tiles (b, b + G, ...) with
//   * ONE continuous LDS-DMA ring over all of them: the stages a K-tile issues for K-tiles t + 1 / t + 2 simply run on into the next
//     tile's first K-tiles (second set of source offsets), so there is one prologue per WORKGROUP, not per tile;
//   * TWO accumulator sets: when a tile's K-loop ends its 64 accumulator VGPRs move to `done` and the next tile starts at once; `done`
//     is finished (bias, activation, rotary embedding, bf16 pack) and stored in 8 steps spread over the next tile's first K-tiles --
//     16-byte stores of whole 64-byte row segments (v_permlane16_swap pairs two column tiles, as gemm_stream.hip) issued through a
//     buffer resource, so the writes trickle out at ~1.5 TB/s chip-wide behind the MFMAs instead of bursting at the end;
//   * every VMEM operation of the drain is a BUFFER operation with a fixed count per K-tile (bias loads and stores go through a
//     zero-length resource when there is nothing to do: nothing moves, vmcnt still counts), so the K-loop's counted s_waitcnt vmcnt --
//     which leave four ring stages in flight across the barriers -- stay compile-time constants (the drain's operations are added in).
// Same fragments, same k-order per output element, same epilogue arithmetic (math4) as gemm8p / gemm: bit-identical results
// (tests/test_gpu_encoder.py::test_gemm_persistent_kernel_vs_pingpong_kernel).
//
// MEASURED (round 3, profiles/r03_gemm8q.txt) -- and why ovo_gemm does NOT pick this kernel (OVO_GEMM_TILE=256x128p forces it):
//   FC1 + GELU (13848, 4096, 1024): 176 us vs 168 us for the one-tile 256 x 128 kernel and 146 us for the 256 x 256 one;
//   (49152, 1792, 448) + GELU: 188 vs 160 / 155;  8192^3: 1061 vs 849 / 711 us.
//   Ablation (OVO_8Q_DEBUG): without the epilogue ARITHMETIC the same launches take 141 / 117 us, without the stores 170 / 183: the
//   stores do hide behind the K-loop (6 us), the VALU work of the epilogue does not -- ~640 VALU instructions per lane and tile
//   (bias, polynomial GELU, pack) sit in the draining wave's load segment, the partner wave's 8 MFMAs (128 cycles) cover a fraction of
//   it, and every other wave waits at the next barrier.  Spread over all four phases of more K-tiles it might reach the "no arithmetic"
//   times -- which only TIE the 256 x 256 tile (its K-loop does 21 % more flops per LDS byte), so the experiment stops here.
#ifdef OVO_EXPERIMENTAL   // python -m ovo_amd.build --experimental: bit-identical to the one-tile kernel and measured slower
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

using namespace ovo_gemm_detail;

namespace {

#define OVO_FENCE() asm volatile("" ::: "memory")
#define OVO_BARRIER()                      \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        OVO_FENCE();                       \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define OVO_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int BM = 256, BN = 128, WARPS_M = 4, WARPS_N = 2;
constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;            // 64 x 64 wave tile
constexpr int HM = WTM / 2, HN = WTN / 2;                        // 32: one sub-tile ("half") of the wave tile
constexpr int TMH = HM / 16, TNH = HN / 16;                      // 2 x 2 MFMA tiles per sub-tile
constexpr int A_HALF = (BM / 2) * 128, B_HALF = (BN / 2) * 128;  // 16 KB, 8 KB
constexpr int BUF = 2 * A_HALF + 2 * B_HALF;                     // 48 KB per K-tile buffer, two of them
constexpr int NA = A_HALF / (512 * 16), NB = B_HALF / (512 * 16);   // 2, 1 DMA pieces per thread and half-tile
constexpr int PIECES = 2 * NA + 2 * NB;                          // per K-tile: the counted waits leave this many ring operations in flight
constexpr int STEPS = 8;                                         // drain steps per tile: (row block i, column-tile pair jp)

struct PArgs {
    GemmArgs g;
    int grid;          // resident workgroups: tile sequence of workgroup b = b, b + grid, ...
    unsigned c_bytes;  // extent of C in bytes (buffer resource)
};

// SPK = drain steps per K-tile (1: K >= 512; 2: K >= 320).  ROPE: the rotary embedding of the epilogue (PE's QKV projection; SPK 1 only),
// whose cos / sin rows are fetched like the bias -- ahead of their use, through counted buffer loads.
// A draining K-tile issues its buffer operations at three fixed points:
//   top (before q0)      L_TOP loads:  bias of the step's two column tiles [+ cos, sin of the first]
//   mid (after q1)       L_MID ops:    ROPE: cos, sin of the second tile;  SPK 2: the first step's store + the second step's bias
//   bottom (after q3)    1 store
template <typename VT, int SPK, bool ROPE>
__global__ void __launch_bounds__(512) k_gemm8q(PArgs p) {
    static_assert(!(ROPE && SPK == 2), "rope: one drain step per K-tile");
    constexpr int L_TOP = ROPE ? 4 : 2, L_MID = ROPE ? 2 : (SPK == 2 ? 3 : 0), S_BOT = 1;
#if __HIP_DEVICE_COMPILE__
    const GemmArgs &g = p.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WARPS_N, wc = wave % WARPS_N, group = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int nt = g.K / 64;

    // tile q of this workgroup (XCD-chunked order as k_gemm8p: XCD x walks a contiguous range of tiles)
    auto tile_of = [&](int q, int &m0, int &n0) __attribute__((always_inline)) -> bool {
        const int vb = blockIdx.x + q * p.grid;
        int tile = vb;
        if (g.chunk > 0) {
            if ((vb >> 3) >= g.chunk) return false;
            tile = (vb & 7) * g.chunk + (vb >> 3);
        }
        if (tile >= g.tiles) return false;
        m0 = (tile / g.nbn) * BM;
        n0 = (tile % g.nbn) * BN;
        return true;
    };
    int nq = 0;
    {
        int a, b;
        while (tile_of(nq, a, b)) ++nq;                   // (tiles of one workgroup are consecutive in q: both tests are monotonic)
    }
    if (nq == 0) return;

    // ---- DMA source offsets of a tile: half h, piece (it * 8 + wave) = local rows [8 piece, +8), lane -> (row, swizzled 16-byte chunk)
    struct Off { uint32_t a[2][NA], b[2][NB]; };
    auto offsets = [&](int m0, int n0, Off &o) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int it = 0; it < NA; ++it) {
                const int r = (it * 8 + wave) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
                int gr = m0 + (r / HM) * WTM + h * HM + (r % HM);
                gr = gr < g.M ? gr : g.M - 1;
                o.a[h][it] = (uint32_t)(((long long)gr * g.lda + c * 8) * 2);
            }
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int r = (it * 8 + wave) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
                int gr = n0 + (r / HN) * WTN + h * HN + (r % HN);
                gr = gr < g.N ? gr : g.N - 1;
                o.b[h][it] = (uint32_t)(((long long)gr * g.ldw + c * 8) * 2);
            }
        }
    };
    const int a_bytes = (int)((long long)g.M * g.lda * 2), b_bytes = (int)((long long)g.N * g.ldw * 2);
    // `use_nxt` (wave-uniform) picks the next tile's offsets: element-wise selects -- a reference to one of two structs would put both in scratch
    auto stage_a = [&](int buf, auto H_, int kt, const Off &c, const Off &n, bool use_nxt, bool valid) __attribute__((always_inline)) {
        constexpr int h = decltype(H_)::value;
        char *dst = smem + buf * BUF + h * A_HALF + wave * 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, valid ? a_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < NA; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + it * 8192), 16, use_nxt ? n.a[h][it] : c.a[h][it], kt * 128, 0, 0);
    };
    auto stage_b = [&](int buf, auto H_, int kt, const Off &c, const Off &n, bool use_nxt, bool valid) __attribute__((always_inline)) {
        constexpr int h = decltype(H_)::value;
        char *dst = smem + buf * BUF + 2 * A_HALF + h * B_HALF + wave * 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, valid ? b_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < NB; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + it * 8192), 16, use_nxt ? n.b[h][it] : c.b[h][it], kt * 128, 0, 0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    const int sw = (fr >> 1) & 7;
    const int off_a = (wr * HM + fr) * 128 + ((fq ^ sw) << 4);
    const int off_b = (wc * HN + fr) * 128 + ((fq ^ sw) << 4);

    f32x4 acc[2 * TMH][2 * TNH], done[2 * TMH][2 * TNH];
#pragma unroll
    for (int i = 0; i < 2 * TMH; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TNH; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; done[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    VT xa[TMH][2], wb0[TNH][2], wb1[TNH][2];

    auto load_a = [&](const char *cur, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TMH; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xa[i][ks] = *(const VT *)(cur + h * A_HALF + ((off_a ^ (ks << 6)) + i * 2048));
    };
    auto load_b = [&](const char *cur, int h, VT (&wb)[TNH][2]) {
#pragma unroll
        for (int j = 0; j < TNH; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wb[j][ks] = *(const VT *)(cur + 2 * A_HALF + h * B_HALF + ((off_b ^ (ks << 6)) + j * 2048));
    };
    auto quadrant = [&](int ih, int jh, VT (&wb)[TNH][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TMH; ++i)
#pragma unroll
                for (int j = 0; j < TNH; ++j)
                    acc[ih * TMH + i][jh * TNH + j] = Mfma<VT>::run(wb[j][ks], xa[i][ks], acc[ih * TMH + i][jh * TNH + j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- the tile being drained: destination of this lane's rows / columns
    //   done[i][j][r] = C[m = dm0 + wr WTM + 16 i + fr][n = dn0 + wc WTN + 16 j + 4 fq + r]
    int dm0 = 0, dn0 = 0;
    bool draining = false;
    u32x4 bias_v[2], rope_v[2];
    const unsigned bias_bytes = g.bias ? (unsigned)g.N * 4u : 0u;
    const unsigned rope_bytes = ROPE ? (unsigned)g.rope_T * (unsigned)g.rope_hd * 4u : 0u;
    auto col_of = [&](int step, int e) __attribute__((always_inline)) { return dn0 + wc * WTN + (2 * (step & 1) + e) * 16 + fq * 4; };
    auto row_of = [&](int step) __attribute__((always_inline)) { return dm0 + wr * WTM + ((step >> 1) & 3) * 16 + fr; };
    auto load_bias = [&](int step, bool active) __attribute__((always_inline)) {      // 2 loads: this lane's own 4 columns of the step's two column tiles
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, active ? bias_bytes : 0u, 0x00020000);
#pragma unroll
        for (int e = 0; e < 2; ++e) bias_v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)col_of(step, e) * 4u, 0, 0);   // columns >= N read zeros
    };
    auto load_rope = [&](int step, int e, bool active) __attribute__((always_inline)) {   // 2 loads: cos, sin at (token of the row, column within the head)
        const int n = col_of(step, e);
        const bool on = active && n < g.rope_cols;
        const unsigned at = on ? (unsigned)(((row_of(step) % g.rope_T) * g.rope_hd + n % g.rope_hd) * 4) : 0xfffffff0u;
        rope_v[0] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void *)g.rope_cos, 0, rope_bytes, 0x00020000), at, 0, 0);
        rope_v[1] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void *)g.rope_sin, 0, rope_bytes, 0x00020000), at, 0, 0);
    };
    // v = act(alpha * acc + bias) [-> rotary embedding] -> packed two-byte values: the arithmetic of math4 (gemm_common.h) without a residual
    auto finish_group = [&](int step, int e) __attribute__((always_inline)) -> uint2 {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        switch (step * 2 + e) {                         // register arrays want compile-time indices: 16 small cases, the arithmetic exists once
#define OVO_CASE(S) case S: t = done[(S) / 4][2 * (((S) / 2) % 2) + (S) % 2]; break;
            OVO_CASE(0) OVO_CASE(1) OVO_CASE(2) OVO_CASE(3) OVO_CASE(4) OVO_CASE(5) OVO_CASE(6) OVO_CASE(7)
            OVO_CASE(8) OVO_CASE(9) OVO_CASE(10) OVO_CASE(11) OVO_CASE(12) OVO_CASE(13) OVO_CASE(14) OVO_CASE(15)
#undef OVO_CASE
            default: break;
        }
        float v[4] = {t[0], t[1], t[2], t[3]};
#ifdef OVO_GEMM_DEBUG
        if (g.dbg & 1) return make_uint2(__float_as_uint(t[0]) ^ bias_v[e].x, __float_as_uint(t[1]));      // tools/ builds only: no epilogue arithmetic
#endif
        const u32x4 b = bias_v[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
        v[0] += __uint_as_float(b.x); v[1] += __uint_as_float(b.y); v[2] += __uint_as_float(b.z); v[3] += __uint_as_float(b.w);
        if (g.act) act4(v, g.act);
        if constexpr (ROPE) {
            const int n = col_of(step, e);
            if (n < g.rope_cols && row_of(step) % g.rope_T >= g.rope_t0) {      // (the class token's rows are left alone)
                const u32x4 c = rope_v[0], sn = rope_v[1];
                const float y0 = v[0] * __uint_as_float(c.x) - v[1] * __uint_as_float(sn.x), y1 = v[1] * __uint_as_float(c.y) + v[0] * __uint_as_float(sn.y);
                const float y2 = v[2] * __uint_as_float(c.z) - v[3] * __uint_as_float(sn.z), y3 = v[3] * __uint_as_float(c.w) + v[2] * __uint_as_float(sn.w);
                v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
            }
        }
        uint2 pk;
        if (g.out_dtype == 2) { pk.x = pack_bf16(v[0], v[1]); pk.y = pack_bf16(v[2], v[3]); }
        else { pk.x = pack_f16(v[0], v[1]); pk.y = pack_f16(v[2], v[3]); }
        return pk;
    };
    // the step's two packed groups -> lane (fr, fq) ends with 8 consecutive columns of tile 2 jp + (fq & 1), from column 8 (fq >> 1)
    // (skinny.h: store_pair16) -> ONE 16-byte buffer store (dropped through an out-of-range offset when there is nothing to write)
    auto store_step = [&](int step, uint2 pk0, uint2 pk1, bool active) __attribute__((always_inline)) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk0.x, pk1.x, false, false), s1 = __builtin_amdgcn_permlane16_swap(pk0.y, pk1.y, false, false);
        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
        const int m = row_of(step), n8 = dn0 + wc * WTN + (2 * (step & 1) + (fq & 1)) * 16 + (fq >> 1) * 8;
        bool ok = active && m < g.M && n8 < g.N;                             // N % 16 == 0: a column tile is all in or all out
#ifdef OVO_GEMM_DEBUG
        ok = ok && !(g.dbg & 2);                                             // tools/ builds only: no store leaves
#endif
        const unsigned off = ok ? (unsigned)(((long long)m * g.ldc + n8) * 2) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(out, __builtin_amdgcn_make_buffer_rsrc(g.C, 0, p.c_bytes, 0x00020000), off, 0, 0);
    };
    uint2 pk_first = make_uint2(0u, 0u);
    int dstep = STEPS;                                    // next step of the draining tile (STEPS = nothing left)

    // ---- tile hand-over: the finished accumulators become `done`, the destination of `done` is this tile's
    auto hand_over = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2 * TMH; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TNH; ++j) { done[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        dm0 = m0; dn0 = n0; dstep = 0; draining = true;
    };

    // ---- one K-tile of the continuous stream.  gk = global K-tile index (buffer parity = gk & 1); (cur, kt) = its tile's offsets and
    // K-tile; the stages for gk + 1 / gk + 2 belong to this tile or -- past its end -- to the next one (nxt, valid while has_next).
    int gk = 0;
    auto body = [&](auto DRAIN_, int kt, const Off &cur, const Off &nxt, bool has_next, bool last) __attribute__((always_inline)) {
        constexpr int DRAIN = decltype(DRAIN_)::value;
        // ring operations issued since the stage a phase waits for: the last four phases' pieces + this K-tile's drain operations so far;
        // DRAIN 1: the previous K-tile issued no drain operations (its stores would sit between the awaited stage and now)
        constexpr int W01 = PIECES + (DRAIN == 0 ? 0 : DRAIN == 1 ? L_TOP : L_TOP + L_MID + S_BOT);           // phases q0, q1
        constexpr int W23 = PIECES + (DRAIN == 0 ? 0 : DRAIN == 1 ? L_TOP + L_MID : L_TOP + L_MID + S_BOT);   // phases q2, q3
        const int b = gk & 1;
        const char *rd = smem + b * BUF;
        const bool in1 = kt + 1 < nt, in2 = kt + 2 < nt;
        const bool v1 = in1 || has_next, v2 = in2 || has_next;
        const int k1 = in1 ? kt + 1 : kt + 1 - nt, k2 = in2 ? kt + 2 : kt + 2 - nt;
        if constexpr (DRAIN != 0) {                        // top
            load_bias(dstep, dstep < STEPS);
            if constexpr (ROPE) load_rope(dstep, 0, dstep < STEPS);
        }
        // q0: A.sub0 x B.sub0
        load_a(rd, 0);
        load_b(rd, 0, wb0);
        stage_b(b ^ 1, H1{}, k1, cur, nxt, !in1, v1);
        OVO_VMCNT(W01);
        OVO_BARRIER();
        quadrant(0, 0, wb0);
        OVO_BARRIER();
        // q1: A.sub0 x B.sub1
        load_b(rd, 1, wb1);
        stage_a(b ^ 1, H1{}, k1, cur, nxt, !in1, v1);
        OVO_VMCNT(W01);
        OVO_BARRIER();
        quadrant(0, 1, wb1);
        OVO_BARRIER();
        if constexpr (DRAIN != 0 && ROPE) {                // mid: the first column tile is finished, the second tile's cos / sin are fetched
            pk_first = finish_group(dstep, 0);
            load_rope(dstep, 1, dstep < STEPS);
        }
        if constexpr (DRAIN != 0 && SPK == 2) {            // mid: the K-tile's first step leaves, the second step's bias is fetched
            const uint2 a0 = finish_group(dstep, 0), a1 = finish_group(dstep, 1);
            store_step(dstep, a0, a1, dstep < STEPS);
            dstep = dstep < STEPS ? dstep + 1 : dstep;
            load_bias(dstep, dstep < STEPS);
        }
        // q2: A.sub1 x B.sub1
        load_a(rd, 1);
        stage_a(b, H0{}, k2, cur, nxt, !in2, v2);
        OVO_VMCNT(W23);
        OVO_BARRIER();
        quadrant(1, 1, wb1);
        OVO_BARRIER();
        // q3: A.sub1 x B.sub0 (fragments kept from q0)
        stage_b(b, H0{}, k2, cur, nxt, !in2, v2);
        OVO_VMCNT(W23);
        OVO_BARRIER();
        quadrant(1, 0, wb0);
        if (!last || group == 0) OVO_BARRIER();           // group 1 skips its very last barrier: both groups execute the same number
        if constexpr (DRAIN != 0) {                        // bottom
            const uint2 a0 = ROPE ? pk_first : finish_group(dstep, 0), a1 = finish_group(dstep, 1);
            store_step(dstep, a0, a1, dstep < STEPS);
            dstep = dstep < STEPS ? dstep + 1 : dstep;
        }
        ++gk;
    };

    // ---- prologue (once per workgroup): Ah0(0) Bh0(0) Bh1(0) Ah1(0) Ah0(1) Bh0(1)
    Off cur, nxt;
    int m0 = 0, n0 = 0, m1 = 0, n1 = 0;
    tile_of(0, m0, n0);
    offsets(m0, n0, cur);
    bool has_next = nq > 1;
    if (has_next) { tile_of(1, m1, n1); offsets(m1, n1, nxt); } else nxt = cur;
    stage_a(0, H0{}, 0, cur, nxt, false, true);
    stage_b(0, H0{}, 0, cur, nxt, false, true);
    stage_b(0, H1{}, 0, cur, nxt, false, true);
    stage_a(0, H1{}, 0, cur, nxt, false, true);
    stage_a(1, H0{}, 1, cur, nxt, false, nt > 1);
    stage_b(1, H0{}, 1, cur, nxt, false, nt > 1);
    OVO_VMCNT(2 * NA + 2 * NB);
    OVO_BARRIER();
    if (group == 1) OVO_BARRIER();                        // group 1 runs one barrier behind group 0 from here on

    constexpr int DRAIN_KT = (STEPS + SPK - 1) / SPK;     // K-tiles a tile's drain takes
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;
    for (int q = 0; q < nq; ++q) {
        const bool last_tile = q + 1 == nq;
        int kt = 0;
        if (draining) {                                   // the previous tile's results leave during this tile's first K-tiles
            body(D1{}, kt, cur, nxt, has_next, last_tile && kt + 1 == nt);
            for (kt = 1; kt < DRAIN_KT; ++kt) body(D2{}, kt, cur, nxt, has_next, last_tile && kt + 1 == nt);
            draining = false;
        }
        for (; kt < nt; ++kt) body(D0{}, kt, cur, nxt, has_next, last_tile && kt + 1 == nt);
        hand_over(m0, n0);
        if (!last_tile) {
            cur = nxt; m0 = m1; n0 = n1;
            has_next = q + 2 < nq;
            if (has_next) { tile_of(q + 2, m1, n1); offsets(m1, n1, nxt); }
        }
    }
    // ---- the last tile's results: nothing left to hide them behind
    OVO_VMCNT(0);
    for (int st = 0; st < STEPS; ++st) {
        load_bias(st, true);
        if constexpr (ROPE) load_rope(st, 0, true);
        const uint2 a0 = finish_group(st, 0);
        if constexpr (ROPE) load_rope(st, 1, true);
        const uint2 a1 = finish_group(st, 1);
        store_step(st, a0, a1, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <typename VT, int SPK, bool ROPE>
int launch8q(const GemmArgs &g0, hipStream_t s) {
    PArgs p;
    p.g = g0;
    GemmArgs &g = p.g;
    g.dbg = 0; g.stamps = nullptr;
#ifdef OVO_GEMM_DEBUG
    static const int dbg = getenv("OVO_8Q_DEBUG") ? atoi(getenv("OVO_8Q_DEBUG")) : 0;
    g.dbg = dbg;
#endif
    g.nbn = (g.N + BN - 1) / BN;
    const int nbm = (g.M + BM - 1) / BM;
    constexpr size_t lds = 2 * (size_t)BUF;
    static bool attr_done = false;
    static int n_cu = 256;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm8q<VT, SPK, ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ovo_set_error("ovo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount & ~7;         // a multiple of 8: virtual block b + q * grid stays on XCD b % 8
        attr_done = true;
    }
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(0, 2.0 * g.M * (double)g.N * g.K, s); ovo_prof_shape(g.M, g.N, g.K); ovo_prof_flags(gemm_flags(g)); ovo_prof_bytes(gemm_algorithmic_bytes(g)); }     // kind 0: the 256 x 128 tile
    g.tiles = nbm * g.nbn;
    g.chunk = (g.tiles + 7) / 8;                          // XCD x walks tiles [x chunk, (x + 1) chunk)
    g.strip = 0;
    const int want = g.chunk * 8;
    p.grid = want < n_cu ? want : n_cu;
    p.c_bytes = (unsigned)((long long)g.M * g.ldc * 2);
    k_gemm8q<VT, SPK, ROPE><<<p.grid, 512, lds, s>>>(p);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

}  // namespace

namespace ovo_gemm_detail {

// Two-byte output, no residual / fused argmax / row remap, N % 16 == 0, K % 64 == 0 and >= 5 K-tiles, operands and C below 4 GB.
int gemm8q_launch(const GemmArgs &g, int in_dtype, hipStream_t s) {
    if (g.out_dtype == 0 || g.add || g.best || g.win_per > 0 || g.add_rows > 0) return OVO_E_UNSUPPORTED;
    if (g.K % 64 != 0 || g.K < 320 || g.N % 16 != 0) return OVO_E_UNSUPPORTED;
    if ((long long)g.M * g.lda * 2 >= (1ll << 31) || (long long)g.N * g.ldw * 2 >= (1ll << 31) || (long long)g.M * g.ldc * 2 >= (1ll << 32)) return OVO_E_UNSUPPORTED;
    if (((uintptr_t)g.C & 15) != 0 || g.ldc % 8 != 0) return OVO_E_UNSUPPORTED;
    if (in_dtype != 2) return OVO_E_UNSUPPORTED;                       // bf16 operands (the encoders); f16 products stay on gemm8p
    const bool spk2 = g.K < 512;
    if (g.rope_cos) return OVO_E_UNSUPPORTED;                          // (a rope variant exists in the template but is not instantiated: unverified)
    return spk2 ? launch8q<bf16x8, 2, false>(g, s) : launch8q<bf16x8, 1, false>(g, s);
}

}  // namespace ovo_gemm_detail

#else
#include "gemm_common.h"
namespace ovo_gemm_detail {
int gemm8q_launch(const GemmArgs &, int, hipStream_t) { return OVO_E_UNSUPPORTED; }      // not in a production build
}
#endif  // OVO_EXPERIMENTAL
