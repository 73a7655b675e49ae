// mlp_stream.hip -- the MLP half of a Hiera stage-1 / stage-2 block in ONE pass over the residual stream (round 5):
//
//     x[row, :] += W2 . GELU(W1 . LayerNorm(x[row, :]) + b1) + b2            x f32 [rows, D], W1 bf16 [HID, K1 >= D], W2 bf16 [D, HID]
//
// (sam2 hieradet.py MultiScaleBlock: `x = x + self.drop_path(self.mlp(self.norm2(x)))`, reached through SAM2AutomaticMaskGenerator.generate at
// /root/reference/ovo/entities/mask_generator.py:113.)  As two launches (gemm_stream.hip FC1 + the tiled FC2) the hidden activations -- 12 frames:
// 786 432 x 448 and 196 608 x 896 bf16 = 704 / 352 MB -- were written by FC1 and read back by FC2: 2.46 GB of HBM traffic per stage-1 block for
// 0.70 GB of residual stream in and out, 622 us per block at ~4 TB/s.  Here the hidden row never leaves the registers of the wave that made it:
//
//   * a wave owns RB blocks of 16 rows; their LayerNorm-ed bf16 A fragments (in-lane statistics, as gemm_stream.hip's f32-A loader) stay in registers;
//   * the hidden dimension is walked in chunks of HC = 64 units.  Per chunk the workgroup holds W1[chunk rows, :] and W2[:, chunk columns] in LDS
//     (double-buffered: the next chunk's 30 / 60 KB go into the other buffer by LDS-DMA under this chunk's products; ONE barrier per chunk), every wave multiplies its rows by the W1 chunk (v_mfma_f32_16x16x32_bf16, operands swapped: a lane's accumulator
//     holds 4 consecutive hidden units of one row), applies bias + table GELU, rounds to bf16 and feeds the result straight back as the B operand
//     of the FC2 partial product -- no LDS round trip, no cross-lane move: the W1 rows of a chunk are laid out in LDS in the ORDER the FC2
//     fragment wants them (LDS row 16 j + 4 fq + r holds hidden unit 32 (j / 2) + 8 fq + 4 (j % 2) + r, so that the two accumulator tiles 2 ks,
//     2 ks + 1 of lane (fr, fq) are exactly hidden units 32 ks + 8 fq + 0..7 of row fr = its B fragment of k-step ks);
//   * (launch shapes: mlp_stream_launch -- stage 1 as two 256-thread workgroups per CU, stage 2 as one of 512)
//   * a wave carries RB row blocks through every chunk (their FC2 accumulators live across the chunks): what a chunk costs beside its products -- one
//     barrier, the wait for its DMA -- is paid once per RB x 16 x 8 rows;
//   * FC2's accumulators (D / 16 tiles per row block) live across the chunks; the epilogue adds b2 and the residual (x re-read: L2 / Infinity Cache)
//     and stores f32 rows in place.
// Same roundings as the two-launch path (A, hidden in bf16; f32 accumulation in ascending k), same GELU table.
#include <stdlib.h>

#include <type_traits>

#include "skinny.h"

using namespace ovo_gemm_detail;

namespace {

struct MlpArgs {
    float *x; long long rows;
    const float *ln_g, *ln_b; float eps;
    const uint16_t *w1; long long ldw1; const float *b1;
    const uint16_t *w2; long long ldw2; const float *b2;
    int dbg;                       // OVO_MLP_DBG (diagnosis): 1 = a barrier after every chunk's products, 2 = wait for every DMA right after its issue
    // tools/ builds only (python -m ovo_amd.build --force --gemm-debug, tools/mlp_race.py): 4 / 8 = compare the chunk's weights IN LDS with their global
    // source right after the barrier / after the chunk's products (pieces brought in by ANOTHER wave), 16 = ~2000 idle cycles between the barrier and the
    // first fragment read, 32 = every wave reads its own pieces back before it enters the barrier, 64 = pad the workgroup's LDS so that only one fits a CU,
    // 128 = the next chunk's DMA is issued AFTER this chunk's products (nothing in flight under them)
    unsigned *dbg_out;             // [0] mismatching pieces, [1] of them equal to the chunk that was in the buffer before (c - 2), [2] workgroups with a
};                                 // non-zero LDS base, [3] records, then {blockIdx, chunk << 16 | piece, LDS_ALLOC register, when} per record

// K1 = padded input width (multiple of 32 >= D), D = model width, HID = hidden width, RB = 16-row blocks per wave, RI = row blocks that share one
// read of the weight fragments (RB / RI passes over a chunk's fragments: more rows per chunk amortise its barrier and DMA wait, registers bound RI)
template <int K1, int D, int HID, int RB, int RI, int NTHREADS, bool POLY, int HC>
__global__ void __launch_bounds__(NTHREADS, 2) k_mlp_stream(MlpArgs g, int n_slots) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (its parse of lambdas that call LDS-DMA builtins drops the stub silently)
    constexpr int NCH = HID / HC, KS1 = K1 / 32, KS2 = HC / 32, NT2 = D / 16;
    constexpr int CPR1 = K1 / 8, CPR2 = HC / 8;                                 // 16-byte chunks per LDS row of the two weight blocks
    constexpr int W1_BYTES = HC * K1 * 2, W2_BYTES = D * HC * 2, BUF = W1_BYTES + W2_BYTES;
    constexpr int P1 = HC * CPR1, P2 = D * CPR2, PIECES = P1 + P2, PPT = (PIECES + NTHREADS - 1) / NTHREADS;      // 16-byte pieces per chunk / per thread
    static_assert(HID % HC == 0 && D % 16 == 0 && K1 % 32 == 0 && K1 >= D && D % 8 == 0, "shape");
    using S1 = ovo_skinny::Skinny<K1, HC>;
    using S2 = ovo_skinny::Skinny<HC, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the two weight buffers (DMA destinations) ONLY
    // The tables live in STATIC LDS objects of their own: the compiler's wait-count pass orders every LDS read that may alias an in-flight LDS-DMA
    // behind an s_waitcnt vmcnt(0).  Fragment reads at constant offsets of the other buffer are provably disjoint; a table gather at a run-time index
    // into the same dynamic array was not -- the next chunk's DMA then had to land before the first GELU of this chunk (seen in the ISA).  Distinct
    // LDS variables carry distinct alias scopes.
    __shared__ __attribute__((aligned(16))) float b1s[HID];      // b1 in LDS-row order of each chunk (the permutation below)
    __shared__ __attribute__((aligned(16))) float b2s[D], lg[K1], lb[K1];
    __shared__ __attribute__((aligned(16))) float2 lut_s[POLY ? 1 : GELU_LUT_N];
    const float2 *lut = lut_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    constexpr int WPB = NTHREADS / 64;

    // hidden unit (within its chunk) held by LDS row q of the W1 block
    auto unit_of = [](int q) { const int j = q >> 4, n = q & 15; return 32 * (j >> 1) + 8 * (n >> 2) + 4 * (j & 1) + (n & 3); };
    if (!POLY) gelu_lut_fill(lut_s, tid, NTHREADS);
    for (int i = tid; i < HID; i += NTHREADS) b1s[i] = g.b1[(i / HC) * HC + unit_of(i % HC)];
    for (int i = tid; i < D; i += NTHREADS) b2s[i] = g.b2[i];
    for (int i = tid; i < K1; i += NTHREADS) { lg[i] = i < D ? g.ln_g[i] : 0.f; lb[i] = i < D ? g.ln_b[i] : 0.f; }

    // one chunk's weights, global -> LDS by DMA (global_load_lds_dwordx4: no staging registers; a wave instruction fills 64 consecutive 16-byte
    // slots).  Slot id < P1: W1 block, LDS row q = id / CPR1 (hidden unit chunk * HC + unit_of(q)), slot id % CPR1 holds source chunk slot ^ swz(q);
    // else W2 block, row n = (id - P1) / CPR2, columns [chunk * HC, + HC).  P1 and P2 are multiples of 64: a wave's 64 slots are all W1 or all W2.
    static_assert(P1 % 64 == 0 && P2 % 64 == 0, "a wave instruction must not straddle the two blocks");
    int src_off[PPT];                                   // element offset of this lane's piece at chunk 0 (its step per chunk is wave-uniform)
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int id = p * NTHREADS + tid;
        if (id < P1) {
            const int q = id / CPR1, c = (id % CPR1) ^ S1::swz(q);
            src_off[p] = (int)(unit_of(q) * g.ldw1 + c * 8);
        } else {
            const int n = (id - P1) / CPR2, c = ((id - P1) % CPR2) ^ S2::swz(n);
            src_off[p] = (int)(n * g.ldw2 + c * 8);
        }
    }
    // (the buffer index is a compile-time constant everywhere: with a run-time `(c & 1) * BUF` the compiler cannot tell the DMA's LDS destination
    //  from the other buffer's fragment reads and puts an s_waitcnt vmcnt(0) in front of the first ds_read after every DMA issue -- the next
    //  chunk's weights then land BEFORE this chunk's products start instead of under them)
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w1, 0, (int)((long long)HID * g.ldw1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w2, 0, (int)((long long)D * g.ldw2 * 2), 0x00020000);
    auto dma = [&](int chunk, auto BUF_) {
        char *base = smem + decltype(BUF_)::value * BUF;
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int id0 = p * NTHREADS + wave * 64;                // wave-uniform
            if (id0 >= PIECES) continue;
            // buffer_load_dwordx4 ... lds: resource = the weight matrix (SGPRs), voffset = the lane's byte offset, soffset = the chunk's
            if (id0 < P1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void *)(base + id0 * 16), 16, src_off[p] * 2, chunk * HC * (int)g.ldw1 * 2, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void *)(base + id0 * 16), 16, src_off[p] * 2, chunk * HC * 2, 0, 0);
        }
        if (g.dbg & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

#ifdef OVO_GEMM_DEBUG
    const unsigned lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);           // HW_REG_LDS_ALLOC: base [7:0], size [20:12] (granules)
    if (g.dbg_out && tid == 0 && (lds_alloc & 0xff)) atomicAdd(g.dbg_out + 2, 1u);
    // the 16-byte piece `id` of chunk `chunk` as the DMA fetches it (the set-up above, for any piece -- here: one another wave brought in)
    auto piece_src = [&](int id, int chunk) -> const uint4 * {
        if (id < P1) { const int q = id / CPR1, c = (id % CPR1) ^ S1::swz(q); return (const uint4 *)(g.w1 + ((long long)chunk * HC + unit_of(q)) * g.ldw1 + c * 8); }
        const int n = (id - P1) / CPR2, c = ((id - P1) % CPR2) ^ S2::swz(n);
        return (const uint4 *)(g.w2 + (long long)n * g.ldw2 + (long long)chunk * HC + c * 8);
    };
    auto verify = [&](int c, int par, unsigned when) {
        if (!g.dbg_out) return;
        for (int p = 0; p < PPT; ++p) {
            const int id = p * NTHREADS + ((tid + 64) % NTHREADS);                    // the next wave's piece
            if (id >= PIECES) continue;
            const uint4 have = *(const uint4 *)(smem + par * BUF + id * 16), want = *piece_src(id, c);
            if (have.x != want.x || have.y != want.y || have.z != want.z || have.w != want.w) {
                atomicAdd(g.dbg_out + 0, 1u);
                if (c >= 2) { const uint4 old = *piece_src(id, c - 2); if (have.x == old.x && have.y == old.y && have.z == old.z && have.w == old.w) atomicAdd(g.dbg_out + 1, 1u); }
                const unsigned at = atomicAdd(g.dbg_out + 3, 1u);
                if (at < 200) { unsigned *r = g.dbg_out + 8 + at * 4; r[0] = blockIdx.x; r[1] = ((unsigned)c << 16) | (unsigned)id; r[2] = lds_alloc; r[3] = when; }
            }
        }
    };
    const bool late_dma = (g.dbg & 128) != 0;
#else
    auto verify = [](int, int, unsigned) {};
    constexpr bool late_dma = false;
#endif
    const long long blocks = (g.rows + 15) / 16, groups = (blocks + WPB * RB - 1) / (WPB * RB);
    for (long long grp = blockIdx.x; grp < groups; grp += n_slots) {
        // (all waves of the workgroup run the same number of chunk iterations: the barriers below are workgroup-wide even for a wave without rows)
        __syncthreads();                                             // every wave is done with the previous group's buffers (first group: the tables above are written)
        dma(0, std::integral_constant<int, 0>{});
        // ---- this wave's rows: LayerNorm in the load (two-pass statistics over the 4 lanes (fr, 0..3) that hold a row), bf16 A fragments
        bf16x8 af[RB][KS1];
        long long row[RB];
#ifdef OVO_GEMM_DEBUG
        unsigned pre_h[RB][3];                                       // (diagnosis) hashes of the raw x values, of (mean, rstd) and of the gamma / beta values as read
#endif
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long b = (grp * WPB + wave) * RB + rb;
            const long long m = b * 16 + fr;
            row[rb] = (b < blocks && m < g.rows) ? m : -1;
            const float *xp = g.x + (row[rb] < 0 ? 0 : row[rb]) * D;
            float xv[KS1][8];
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int d0 = (ks * 4 + fq) * 8;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (row[rb] >= 0 && d0 < D) { lo = *(const float4 *)(xp + d0); hi = *(const float4 *)(xp + d0 + 4); }
                xv[ks][0] = lo.x; xv[ks][1] = lo.y; xv[ks][2] = lo.z; xv[ks][3] = lo.w;
                xv[ks][4] = hi.x; xv[ks][5] = hi.y; xv[ks][6] = hi.z; xv[ks][7] = hi.w;
                sum += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
            }
            sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)D;
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if ((ks * 4 + fq) * 8 < D) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float a0 = xv[ks][e] - mean, a1 = xv[ks][e + 1] - mean;
                        q += a0 * a0 + a1 * a1;
                    }
                }
            }
#ifdef OVO_GEMM_DEBUG
            const float q_loc_used = q;
            const unsigned long long exec_used = __builtin_amdgcn_read_exec();
#endif
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q / (float)D + g.eps);
#ifdef OVO_GEMM_DEBUG
            {
                unsigned hx = 0, ht = 0;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int d0 = (ks * 4 + fq) * 8;
                        hx = (hx * 16777619u) ^ (__float_as_uint(xv[ks][e]) + (unsigned)lane * 2654435761u);
                        ht = (ht * 16777619u) ^ (__float_as_uint(lg[d0 + e]) + 3u * __float_as_uint(lb[d0 + e]) + (unsigned)lane * 2654435761u);
                    }
                pre_h[rb][0] = hx; pre_h[rb][1] = __float_as_uint(mean) * 31u + __float_as_uint(rstd) + (unsigned)lane * 2654435761u; pre_h[rb][2] = ht;
                // (dbg & 512) the statistics AGAIN from the same registers, through the same instructions: does the wave reduction repeat?  Records
                // {workgroup, row block, lane, sum as used, sum again, local part as used, local part again} behind the other records
                if (g.dbg_out && (g.dbg & 1024) && lane < 16) {              // the statistics themselves, per row: [blocks][16][4] floats behind the hashes
                    const long long b = (grp * WPB + wave) * RB + rb;
                    if (b < blocks) {
                        float *so = (float *)(g.dbg_out + 8 + 800) + (long long)(5 + NCH) * blocks + (b * 16 + lane) * 4;
                        so[0] = sum; so[1] = q; so[2] = mean; so[3] = rstd;
                    }
                }
                if (g.dbg_out && (g.dbg & 512)) {
                    const int c_dummy = 0;
                    float loc = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        float v8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { v8[e] = xv[ks][e]; asm volatile("" : "+v"(v8[e])); }
                        loc += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
                    }
                    float s2 = loc;
                    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                    float mean2 = s2 / (float)D;
                    asm volatile("" : "+v"(mean2));
                    float q2 = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        if ((ks * 4 + fq) * 8 < D) {
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                float x0 = xv[ks][e], x1 = xv[ks][e + 1];
                                asm volatile("" : "+v"(x0), "+v"(x1));
                                const float a0 = x0 - mean2, a1 = x1 - mean2;
                                q2 += a0 * a0 + a1 * a1;
                            }
                        }
                    }
                    const float q2loc = q2;
                    q2 += __shfl_xor(q2, 16, 64); q2 += __shfl_xor(q2, 32, 64);
                    const float rstd2 = rsqrtf(q2 / (float)D + g.eps);
                    if (__float_as_uint(s2) != __float_as_uint(sum) || __float_as_uint(q2) != __float_as_uint(q) || __float_as_uint(rstd2) != __float_as_uint(rstd)) {
                        const unsigned at = atomicAdd(g.dbg_out + 5, 1u);
                        if (at < 40) {
                            unsigned *r = g.dbg_out + 8 + 800 - 8 * 40 + at * 8;          // (the last 40 x 8 words of the verify-record area)
                            r[0] = blockIdx.x; r[1] = (unsigned)((grp * WPB + wave) * RB + rb); r[2] = (unsigned)lane; r[3] = __float_as_uint(sum); r[4] = __float_as_uint(s2);
                            r[5] = __float_as_uint(q); r[6] = __float_as_uint(q2); r[7] = __float_as_uint(q2loc);
                            unsigned *r2 = g.dbg_out + 8 + 800 - 8 * 40 - 8 * 40 + at * 8;   // (a second record block below the first)
                            r2[0] = __float_as_uint(q_loc_used); r2[1] = __float_as_uint(mean); r2[2] = __float_as_uint(mean2); r2[3] = (unsigned)exec_used; r2[4] = (unsigned)(exec_used >> 32);
                            r2[5] = (unsigned)__builtin_amdgcn_read_exec(); r2[6] = (unsigned)(__builtin_amdgcn_read_exec() >> 32); r2[7] = (unsigned)c_dummy;
                        }
                    }
                }
            }
#endif
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int d0 = (ks * 4 + fq) * 8;
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float y0 = (xv[ks][e] - mean) * rstd * lg[d0 + e] + lb[d0 + e];
                    const float y1 = (xv[ks][e + 1] - mean) * rstd * lg[d0 + e + 1] + lb[d0 + e + 1];
                    pk[e >> 1] = (row[rb] >= 0 && d0 < D) ? pack_bf16(y0, y1) : 0u;
                }
                af[rb][ks] = *(const bf16x8 *)pk;
            }
        }
        f32x4 acc2[RB][NT2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NT2; ++j) acc2[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef OVO_GEMM_DEBUG
        // (diagnosis, dbg & 256) per row block: a hash of the LayerNorm-ed fragments, of every chunk's hidden fragments and of the FC2 accumulators, to be
        // compared with another variant's run on the host (tools/mlp_race.py): WHICH intermediate of a wrong row block is wrong first
        auto wave_xor = [&](unsigned h) { for (int o = 1; o < 64; o <<= 1) h ^= (unsigned)__shfl_xor((int)h, o, 64); return h; };
        auto hash8 = [&](const bf16x8 &v, unsigned h) { const uint4 u = *(const uint4 *)&v; return (h * 16777619u) ^ (u.x + 3u * u.y + 5u * u.z + 7u * u.w + (unsigned)lane * 2654435761u); };
        unsigned *hash_out = (g.dbg_out && (g.dbg & 256)) ? g.dbg_out + 8 + 800 : nullptr;
        constexpr int HW = 5 + NCH;                                  // words per row block: x, (mean, rstd), gamma / beta, af, chunk 0 .. NCH - 1, acc2
        if (hash_out) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                unsigned h = 0;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) h = hash8(af[rb][ks], h);
                h = wave_xor(h);
                const long long b = (grp * WPB + wave) * RB + rb;
                const unsigned h0 = wave_xor(pre_h[rb][0]), h1 = wave_xor(pre_h[rb][1]), h2 = wave_xor(pre_h[rb][2]);
                if (lane == 0 && b < blocks) { hash_out[b * HW] = h0; hash_out[b * HW + 1] = h1; hash_out[b * HW + 2] = h2; hash_out[b * HW + 3] = h; }
            }
        }
        unsigned hh[RB];
#endif

        auto chunk_body = [&](auto PAR_, int c) {
            constexpr int PAR = decltype(PAR_)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of chunk c have landed ...
#ifdef OVO_GEMM_DEBUG
            if (g.dbg & 32) {                                        // (diagnosis) ... and the wave has read its last one back
                const uint4 v = *(const uint4 *)(smem + PAR * BUF + (((PPT - 1) * NTHREADS + tid) < PIECES ? ((PPT - 1) * NTHREADS + tid) : tid) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(v.x) : "memory");
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                         // ... and everybody's: chunk c is in buffer PAR; every wave is done with buffer PAR ^ 1
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < NCH && !late_dma) dma(c + 1, std::integral_constant<int, PAR ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
#ifdef OVO_GEMM_DEBUG
            if (g.dbg & 16) { const unsigned long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < 2000ull) {} }
            if (g.dbg & 4) verify(c, PAR, 0u);
#endif
            const char *w1 = smem + PAR * BUF, *w2 = w1 + W1_BYTES;
            static_assert(RB % RI == 0, "row blocks per fragment pass");
#ifdef OVO_GEMM_DEBUG
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) hh[rb] = 0;
#endif
#pragma unroll
            for (int r0 = 0; r0 < RB; r0 += RI) {
                // one FC2 k-step (32 hidden units = two FC1 column tiles) at a time: FC1 tiles 2 kp, 2 kp + 1 over all of K1, bias + table GELU + bf16 --
                // the lane's two tiles ARE its FC2 fragment of k-step kp --, then that k-step of the FC2 partial.  (All four FC1 tiles of a chunk at
                // once held 16 more accumulators, 8 more weight fragments and 8 more bias registers live: 72 spilled registers at K1 = 256, RB = 2.)
                const float *bc = b1s + c * HC + fq * 4;
#pragma unroll
                for (int kp = 0; kp < KS2; ++kp) {
                    f32x4 acc1[RI][2];
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) acc1[ri][0] = acc1[ri][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const char *wp = w1 + (fr * CPR1 + ((ks * 4 + fq) ^ S1::swz(fr))) * 16 + (2 * kp) * (16 * CPR1 * 16);
                        const bf16x8 wa = *(const bf16x8 *)wp, wb = *(const bf16x8 *)(wp + 16 * CPR1 * 16);
#pragma unroll
                        for (int ri = 0; ri < RI; ++ri) {
                            acc1[ri][0] = Mfma<bf16x8>::run(wa, af[r0 + ri][ks], acc1[ri][0]);
                            acc1[ri][1] = Mfma<bf16x8>::run(wb, af[r0 + ri][ks], acc1[ri][1]);
                        }
                    }
                    bf16x8 hf[RI];
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) {
                        uint32_t pk[4];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 bv = *(const f32x4 *)(bc + (2 * kp + h) * 16);
                            if (POLY) {                              // packed polynomial erf (gemm_common.h: gelu2), no LDS access
                                const f32x2 a = gelu2(f32x2{acc1[ri][h][0] + bv[0], acc1[ri][h][1] + bv[1]});
                                const f32x2 b = gelu2(f32x2{acc1[ri][h][2] + bv[2], acc1[ri][h][3] + bv[3]});
                                pk[2 * h] = pack_bf16(a.x, a.y);
                                pk[2 * h + 1] = pack_bf16(b.x, b.y);
                            } else {
                                float v[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = gelu_lut(acc1[ri][h][r] + bv[r], lut);
                                pk[2 * h] = pack_bf16(v[0], v[1]);
                                pk[2 * h + 1] = pack_bf16(v[2], v[3]);
                            }
                        }
                        hf[ri] = *(const bf16x8 *)pk;
#ifdef OVO_GEMM_DEBUG
                        if (hash_out) hh[r0 + ri] = hash8(hf[ri], hh[r0 + ri]);
#endif
                    }
                    // FC2 partial, k-step kp: acc2[r0 + ri][j] += H . W2[:, chunk]^T
                    constexpr int JG = 4;                            // fragments per read group (the last group takes what is left)
#pragma unroll
                    for (int j0 = 0; j0 < NT2; j0 += JG) {
                        bf16x8 w[JG];
                        const char *wp = w2 + (fr * CPR2 + ((kp * 4 + fq) ^ S2::swz(fr))) * 16;
#pragma unroll
                        for (int jj = 0; jj < JG; ++jj)
                            if (j0 + jj < NT2) w[jj] = *(const bf16x8 *)(wp + (j0 + jj) * (16 * CPR2 * 16));
#pragma unroll
                        for (int jj = 0; jj < JG; ++jj)
                            if (j0 + jj < NT2) {
#pragma unroll
                                for (int ri = 0; ri < RI; ++ri) acc2[r0 + ri][j0 + jj] = Mfma<bf16x8>::run(w[jj], hf[ri], acc2[r0 + ri][j0 + jj]);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#ifdef OVO_GEMM_DEBUG
            if (g.dbg & 8) verify(c, PAR, 1u);
            if (hash_out) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const unsigned h = wave_xor(hh[rb]);
                    const long long b = (grp * WPB + wave) * RB + rb;
                    if (lane == 0 && b < blocks) hash_out[b * HW + 4 + c] = h;
                }
            }
#endif
            if (c + 1 < NCH && late_dma) dma(c + 1, std::integral_constant<int, PAR ^ 1>{});
            if (g.dbg & 1) __syncthreads();
        };
        {
            int c = 0;
            for (; c + 1 < NCH; c += 2) {                            // two chunks per trip: static buffer parity
                chunk_body(std::integral_constant<int, 0>{}, c);
                chunk_body(std::integral_constant<int, 1>{}, c + 1);
            }
            if (c < NCH) chunk_body(std::integral_constant<int, 0>{}, c);
        }
#ifdef OVO_GEMM_DEBUG
        if (hash_out) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                unsigned h = 0;
#pragma unroll
                for (int j = 0; j < NT2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) h = (h * 16777619u) ^ (__float_as_uint(acc2[rb][j][e]) + (unsigned)lane * 2654435761u);
                h = wave_xor(h);
                const long long b = (grp * WPB + wave) * RB + rb;
                if (lane == 0 && b < blocks) hash_out[b * HW + 4 + NCH] = h;
            }
        }
#endif
        // ---- epilogue: + b2 + residual, f32 rows in place (4 lanes x 16 B = 64 contiguous bytes per row and instruction)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (row[rb] < 0) continue;
            float *xp = g.x + row[rb] * D + fq * 4;
            const float *bc = b2s + fq * 4;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const f32x4 r = *(const f32x4 *)(xp + j * 16), bv = *(const f32x4 *)(bc + j * 16);
                *(float4 *)(xp + j * 16) = make_float4(acc2[rb][j][0] + bv[0] + r[0], acc2[rb][j][1] + bv[1] + r[1], acc2[rb][j][2] + bv[2] + r[2],
                                                       acc2[rb][j][3] + bv[3] + r[3]);
                if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#endif
}

template <int K1, int D, int HID, int RB, int RI, int NTHREADS, bool POLY, int HC>
int launch_mlp(const MlpArgs &g, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)(HC * K1 * 2 + D * HC * 2);                                   // dynamic: the weight buffers
    constexpr size_t lds_all = lds + (size_t)(HID + D + 2 * K1) * sizeof(float) + GELU_LUT_BYTES + 64;   // + the static tables
    static_assert(lds_all <= 160 * 1024, "LDS");
    // 8 waves per CU (2 per SIMD, up to 256 VGPRs each): ONE workgroup of 512 threads (the diagnosis variants: two of 256 when their LDS fits
    // twice).  Each workgroup walks row groups blockIdx.x, + slots, ...
    constexpr int PER_CU = (NTHREADS <= 256 && 2 * lds_all + 2048 <= 160 * 1024) ? 2 : 1;
    // The one-workgroup form is PINNED to one workgroup per CU by its LDS request: padded past half of the CU's 160 KB, a second workgroup
    // can never become resident beside it, whatever register count a later compiler lands at and whatever else runs on the other streams.
    constexpr size_t lds_pinned = (PER_CU == 1 && lds_all < 82 * 1024) ? lds + (82 * 1024 - lds_all) : lds;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_mlp_stream<K1, D, HID, RB, RI, NTHREADS, POLY, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pinned);
        if (e != hipSuccess) { ovo_set_error("ovo_mlp_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        attr_done = true;
    }
    const long long blocks = (g.rows + 15) / 16, groups = (blocks + (NTHREADS / 64) * RB - 1) / ((NTHREADS / 64) * RB);
    const int slots = (int)(groups < 256 * PER_CU ? groups : 256 * PER_CU);
    const bool prof = ovo_prof_enabled();
    // profiler kind 8 (the streaming GEMMs): flops of both products; algorithmic bytes = the stream in and out + the weights
    if (prof) { ovo_prof_begin(8, 2.0 * (double)g.rows * HID * (double)(K1 + D), s); ovo_prof_shape((int)g.rows, HID, K1); ovo_prof_flags(1 | 2 | 4 | 64);
                ovo_prof_bytes(8.0 * (double)g.rows * D + 2.0 * HID * (K1 + D)); }
    size_t lds_launch = lds_pinned;
#ifdef OVO_GEMM_DEBUG
    if ((g.dbg & 64) && lds_all < 84 * 1024) {                      // one workgroup per CU whatever its size: pad the dynamic LDS past half a CU's
        lds_launch = lds + (84 * 1024 - lds_all);
        (void)hipFuncSetAttribute((const void *)k_mlp_stream<K1, D, HID, RB, RI, NTHREADS, POLY, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
    }
#endif
    k_mlp_stream<K1, D, HID, RB, RI, NTHREADS, POLY, HC><<<slots, NTHREADS, lds_launch, s>>>(g, slots);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

}  // namespace

namespace ovo_gemm_detail {

// x f32 [rows, d] += fc2(GELU(fc1(LayerNorm(x)))) in one launch.  OVO_E_UNSUPPORTED (nothing launched) for shapes without an instantiation:
// the caller runs the two products.  Instantiations: Hiera hiera_b+ / hiera_s / hiera_t stage 1-2 widths (112, 224 | 96, 192) and hiera_l's 144, 288.
int mlp_stream_launch(float *x, long long rows, int d, const float *ln_g, const float *ln_b, float eps, const void *w1, long long ldw1, const float *b1,
                      int hid, const void *w2, long long ldw2, const float *b2, hipStream_t s) {
    auto read_off = [] { return getenv("OVO_NO_MLP_FUSE") != nullptr || getenv("OVO_GEMM_NO_STREAM") != nullptr || getenv("OVO_GEMM_TILE") != nullptr; };
    static int off = read_off(), gelu_poly = getenv("OVO_GELU_POLY") != nullptr;                   // (see ovo_knobs_dynamic)
    if (ovo_knobs_dynamic()) { off = read_off(); gelu_poly = getenv("OVO_GELU_POLY") != nullptr; }
    if (off || gelu_poly || rows < 16384 || hid != 4 * d || !x || !ln_g || !ln_b || !w1 || !b1 || !w2 || !b2) return OVO_E_UNSUPPORTED;
    if (ldw1 % 8 != 0 || ldw2 % 8 != 0 || (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)x) & 15) != 0) return OVO_E_UNSUPPORTED;
    MlpArgs g;
    g.x = x; g.rows = rows; g.ln_g = ln_g; g.ln_b = ln_b; g.eps = eps;
    static int dbg_env = getenv("OVO_MLP_DBG") ? atoi(getenv("OVO_MLP_DBG")) : 0;                  // diagnosis (tools/mlp_stress.py)
    if (ovo_knobs_dynamic()) dbg_env = getenv("OVO_MLP_DBG") ? atoi(getenv("OVO_MLP_DBG")) : 0;
    g.dbg = dbg_env;
    g.dbg_out = nullptr;
#ifdef OVO_GEMM_DEBUG        // the record buffer's ADDRESS comes from the environment: never in a production build
    g.dbg_out = getenv("OVO_MLP_DBG_OUT") ? (unsigned *)strtoull(getenv("OVO_MLP_DBG_OUT"), nullptr, 0) : nullptr;
#endif
    g.w1 = (const uint16_t *)w1; g.ldw1 = ldw1; g.b1 = b1; g.w2 = (const uint16_t *)w2; g.ldw2 = ldw2; g.b2 = b2;
    const int k1 = (int)ldw1;
    // GELU: the table in LDS (gemm_common.h: gelu_lut), as the two-launch path.  The packed polynomial (OVO_MLP_GELU_POLY=1) measured SLOWER here --
    // (786432, 112 -> 448): 408-444 us against 353-376 (profiles/r05a_mlp_stream_variants.txt): the kernel is bound by VALU issue beside the MFMAs,
    // and 9.5 packed instructions per value cost more of it than 7 plain ones + an LDS gather.  OVO_MLP_RB: variant number (measurement runs).
    static int rb_env = getenv("OVO_MLP_RB") ? atoi(getenv("OVO_MLP_RB")) : 0, lut_env = getenv("OVO_MLP_GELU_POLY") == nullptr;
    if (ovo_knobs_dynamic()) { rb_env = getenv("OVO_MLP_RB") ? atoi(getenv("OVO_MLP_RB")) : 0; lut_env = getenv("OVO_MLP_GELU_POLY") == nullptr; }
#define GO(KK, DD, RB, RI, NTH, HCC, CODE)                                                                                             \
    if (d == DD && k1 == KK && (rb_env == 0 || rb_env == CODE))                                                                        \
        return lut_env ? launch_mlp<KK, DD, 4 * DD, RB, RI, NTH, false, HCC>(g, s) : launch_mlp<KK, DD, 4 * DD, RB, RI, NTH, true, HCC>(g, s);
    // CODE (OVO_MLP_RB) = variant number of the measurement runs
    // Variant 1 = ONE 512-thread workgroup per CU, variants 2 / 3 = TWO 256-thread workgroups per CU (32-unit chunks at width 224 so that the LDS fits
    // twice; 2 or 4 / 2 row blocks per wave).  Measured (tools/mlp_bench.py, profiles/r06_mlp_variants.txt; 12 frames of hiera_b+; the two launches:
    // 600 / 410 us): stage 1 (112 -> 448) 402-405 us as variant 1, 332-343 as variant 2, 353-376 as variant 3; stage 2 (224 -> 896) 270-279 / 280-289 /
    // 282-301.  Stage 1 therefore runs as variant 2, stage 2 as variant 1.
    //
    // Round 5 kept variant 2 out of production: under tools/mlp_stress.py whole 16-row blocks came out wrong in 1 of 400 launches at 65 536 rows and in
    // every launch at 786 432 rows, "cause not found", with the suspicion on two workgroups' LDS-DMA rings sharing a CU.  Round 6 found it
    // (tools/mlp_race.py in a --gemm-debug build, tools/lds_dma_race.hip):
    //   * NOT the LDS-DMA protocol: every chunk's weights IN LDS, compared with their global source right after the wait + barrier and again after the
    //     chunk's products, were right in every failing launch (pieces another wave brought in; 0 wrong of ~10^9); the bare protocol with two
    //     workgroups really co-resident (LDS base != 0 read back from HW_REG_LDS_ALLOC) never read a stale or early piece; nor ds_bpermute beside it.
    //   * The first wrong intermediate of a wrong row block is always the LayerNorm's variance: the row SUM repeats bit for bit when recomputed from
    //     the same registers, the sum of squared deviations does not -- and only the partial of lanes 48 .. 63 differs (one (x - mean)^2 pair
    //     missing), in either the value used or the recomputed one.  That code is a chain of v_pk_add / v_pk_mul / v_pk_fma_f32 (the -O3 SLP
    //     vectoriser pairs the scalar f32 operations) around an EXEC-masked block (the last, partial 8-column chunk of a 112-wide row).
    //   * Built with -fno-slp-vectorize (no packed f32 in the loader) variants 1, 2, 3 are bit-stable: 0 of 2 260 stress launches, 0 of 60 at the
    //     size where every launch failed.  A bare `v_pk_add_f32 ; s_and_saveexec_b64` chain beside MFMA partner waves (tools/pk_exec_hazard.hip)
    //     does not reproduce it, so the exact trigger inside the packed sequence is not isolated; it needs a wave of ANOTHER workgroup on the SIMD
    //     (in its MFMA phase while this one normalises), which is why one-workgroup-per-CU launches never showed it.
    // ovo_amd/build.py therefore compiles the three files with an in-load LayerNorm (this one, gemm_stream.hip, winattn.hip) without SLP
    // vectorisation, `test_fused_mlp_two_workgroup_form_under_stress` repeats the launch that failed every time, and the one-workgroup forms stay
    // pinned to one workgroup per CU by their LDS request (launch_mlp).
    GO(128, 112, 2, 2, 256, 64, 2) GO(128, 112, 2, 2, 512, 64, 1) GO(128, 112, 4, 2, 256, 64, 3)
    GO(256, 224, 1, 1, 512, 64, 1) GO(256, 224, 1, 1, 256, 32, 2) GO(256, 224, 2, 1, 256, 32, 3)
    if (rb_env) return OVO_E_UNSUPPORTED;
    GO(128, 96, 2, 2, 512, 64, 1) GO(192, 192, 1, 1, 512, 64, 1)          // hiera_t / hiera_s
    GO(192, 144, 1, 1, 512, 64, 1) GO(320, 288, 1, 1, 512, 32, 1)          // hiera_l stages 1 and 2 (the reference's default trunk, ovo.yaml:35; 288 -> K 320: chunks of 32
                                                                           // hidden units so that two W1 / W2 chunk pairs fit the LDS beside the tables)
#undef GO
    return OVO_E_UNSUPPORTED;
}

}  // namespace ovo_gemm_detail

extern "C" int ovo_mlp_f32(float *x, int64_t rows, int d, const float *ln_g, const float *ln_b, float eps, const void *w1, int64_t ldw1, const float *b1,
                           int hidden, const void *w2, int64_t ldw2, const float *b2, ovo_stream_t stream) {
    OVO_REQUIRE(x && ln_g && ln_b && w1 && b1 && w2 && b2 && rows >= 0 && d > 0 && hidden > 0, "bad argument");
    if (rows == 0) return OVO_OK;
    const int rc = ovo_gemm_detail::mlp_stream_launch(x, rows, d, ln_g, ln_b, eps, w1, ldw1, b1, hidden, w2, ldw2, b2, (hipStream_t)stream);
    if (rc != OVO_OK) return rc;
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
