// gemm.hip -- C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias) (+ add), bf16/f16 inputs, fp32 accumulate on MFMA.
//
// The workhorse of the ViT / Hiera forward passes (QKV, projection, MLP, patch-embed, FPN 1x1 convs) and of
// the large-Q similarity query (BASELINE.json config 5).  Both operands are K-contiguous (activations
// [tokens, K], nn.Linear weights [out, K]), so A and W tiles are staged the same way.
//
// Structure (gfx950):
//   * block = NW waves as (NW/2) x 2: 4 waves for the 64x64 / 64x128 / 128x64 tiles (2 workgroups per CU), 8 waves for the
//     128x128 tile (1 workgroup per CU); BK in {32,64}; each wave owns (BM/(NW/2)) x (BN/2) as 16x16 MFMA tiles
//     (v_mfma_f32_16x16x32_{bf16,f16}).
//   * global -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round trip) into a ring of 3-4 LDS stages:
//     tiles t+1..t+NS-1 stay in flight across the (raw) barrier behind a COUNTED s_waitcnt vmcnt while tile t is
//     multiplied -- these GEMMs are 1-3 workgroup rounds long, so exposed load latency and L2->LDS traffic per flop of a
//     64-row tile, not MFMA issue, are the bound (tools/gemm_probe.hip; DESIGN.md section 3 has the measurements).
//   * tile order is XCD-aware (chunked) when the activation panels outweigh the weights; epilogue operands (bias, residual)
//     are fetched during the last k-tile; GELU runs on packed f32 with a polynomial erf; short-K launches ask only for the
//     LDS stages they use; ovo_gemm_argmax fuses a per-row first-max argmax (64-bit atomicMax) into the epilogue.
//   * the DMA writes LDS lane-linearly, so the bank-conflict swizzle (16-byte chunk index XOR a row
//     function) is applied to the per-lane SOURCE address and again when fragments are read (ds_read_b128).
//   * operands are swapped (a = W fragment, b = activation fragment): the accumulator holds C^T tiles, i.e.
//     each lane owns 4 consecutive output columns of one output row -> 8/16-byte epilogue stores.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "gemm_common.h"
#include "gemm_tuned.h"

using namespace ovo_gemm_detail;

namespace {


template <int BM, int BN, int BK, int NS, typename VT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) k_gemm(GemmArgs g) {
    constexpr int NT = 64 * NW;                       // threads: NW waves as (NW / 2) x 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPR = BK / 8;                       // 16-byte chunks per tile row
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / (NW / 2), WN = BN / 2, TM = WM / 16, TN = WN / 16, KS = BK / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    // Workgroups go round-robin over the 8 XCDs (XCD = blockIdx % 8), each with its own L2.  In chunked order XCD x walks
    // the contiguous tile range [x * chunk, (x + 1) * chunk) (m-major, n-minor): the n-tiles that share an A panel are
    // co-resident on ONE L2 and the panel leaves HBM once instead of once per XCD.
    int tile = blockIdx.x;
    if (g.chunk > 0) {
        tile = (blockIdx.x & 7) * g.chunk + (blockIdx.x >> 3);
        if (tile >= g.tiles) return;
    }
    const int m0 = (tile / g.nbn) * BM, n0 = (tile % g.nbn) * BN;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane source pointers of the DMA pieces (swizzled chunk of a clamped row), advanced by BK elements per tile
    constexpr int AI = A_BYTES / (16 * NT), BI = B_BYTES / (16 * NT);
    const char *a_src[AI], *b_src[BI];
#pragma unroll
    for (int it = 0; it < AI; ++it) {
        const int id = it * NT + tid, row = id / CPR, c = (id % CPR) ^ swz<BK>(row);
        int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
        a_src[it] = g.A + ((long long)gr * g.lda + c * 8) * 2;
    }
#pragma unroll
    for (int it = 0; it < BI; ++it) {
        const int id = it * NT + tid, row = id / CPR, c = (id % CPR) ^ swz<BK>(row);
        int gr = n0 + row; gr = gr < g.N ? gr : g.N - 1;
        b_src[it] = g.W + ((long long)gr * g.ldw + c * 8) * 2;
    }
    auto stage = [&](int buf, int kt) {
        char *sa = smem + buf * STAGE, *sb = sa + A_BYTES;
        const long long koff = (long long)kt * BK * 2;
#pragma unroll
        for (int it = 0; it < AI; ++it) glds16(a_src[it] + koff, sa + (it * NT + wave * 64) * 16);
#pragma unroll
        for (int it = 0; it < BI; ++it) glds16(b_src[it] + koff, sb + (it * NT + wave * 64) * 16);
    };
    auto frag = [&](const char *tile, int row, int chunk) -> VT {
        return *(const VT *)(tile + (row * CPR + (chunk ^ swz<BK>(row))) * 16);
    };
    auto compute = [&](int buf) {
        const char *sa = smem + buf * STAGE, *sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            VT xf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = frag(sa, wm0 + i * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = frag(sb, wn0 + j * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<VT>::run(wf[j], xf[i], acc[i][j]);
        }
    };

    // Epilogue operands (bias, residual) are fetched into registers during the LAST k-tile: issued after the main loop they
    // sat on the serial tail of every workgroup (2-4 us of a 15-25 us GEMM).
    float4 bias_r[TN], add_r[TM][TN];
    long long md[TM];                               // destination row of C / add for each of this lane's rows (-1: none)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 16 + fr;
        md[i] = m < g.M ? row_dest(g, m) : -1;
    }
    auto fetch_epilogue = [&]() {
        if constexpr (BN == 448) return;             // the full-row tile reads bias / residual in its epilogue (168 registers of prefetch do not fit beside 112 of accumulators)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 16 + fq * 4;
            bias_r[j] = (g.bias && n < g.N) ? *(const float4 *)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (g.add) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + i * 16 + fr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn0 + j * 16 + fq * 4;
                    if (md[i] >= 0 && n < g.N) add_r[i][j] = *(const float4 *)(g.add + add_row(g, m, md[i]) * g.ld_add + n);
                }
            }
        }
    };

    // NS-stage LDS ring: tiles t+1 .. t+NS-2 stay in flight (LDS-DMA) across the barrier while tile t is multiplied.
    // A tile's DMA pieces are ordered for the ds_reads only by the issuing waves' counted vmcnt followed by a barrier,
    // so: counted wait (leave the NEWER tiles' pieces outstanding) -> raw s_barrier -> restage the buffer tile t-1 used
    // (every wave has finished reading it once it passes this barrier) -> multiply tile t.
    const int nt = g.K / BK;
    constexpr int PIECES = AI + BI;                 // DMA instructions per thread per tile
    constexpr int AHEAD = NS - 2;                   // tiles allowed to stay in flight behind the one being waited for
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) stage(s, s);
    for (int t = 0; t < nt; ++t) {
        if (t + AHEAD < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifndef OVO_GEMM_PROBE_NO_LOAD
        if (t + NS - 1 < nt) stage((t + NS - 1) % NS, t + NS - 1);
#endif
        if (t == nt - 1) fetch_epilogue();          // behind the last DMA wait: their latency hides under the last multiply
#ifndef OVO_GEMM_PROBE_NO_MMA
        compute(t % NS);
#endif
    }

    // epilogue: acc[i][j][r] = C[m = m0+wm0+16i+fr][n = n0+wn0+16j+4fq+r].  One body per KIND, so that the forms the encoders launch hundreds of times per
    // frame are short and branch-free (every activation + rotary + argmax inlined into each of the TM x TN tiles is cold code fetched per workgroup):
    // 0 = 2-byte output, bias only; 1 = 2-byte output, GELU; 3 = f32 output += f32 residual; -1 = everything else
    auto epilogue = [&](auto KIND_) {
    constexpr int KIND = decltype(KIND_)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 16 + fr;
        if (m >= g.M || md[i] < 0) continue;
        float row_best = -3.0e38f;                               // fused first-max argmax of this lane's columns (g.best)
        int row_arg = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 16 + fq * 4;
            if (n >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * g.alpha;
            v[0] += bias_r[j].x; v[1] += bias_r[j].y; v[2] += bias_r[j].z; v[3] += bias_r[j].w;
            if constexpr (KIND == 1) {
                const f32x2 ga = gelu2(f32x2{v[0], v[1]}), gb = gelu2(f32x2{v[2], v[3]});
                v[0] = ga.x; v[1] = ga.y; v[2] = gb.x; v[3] = gb.y;
            } else if constexpr (KIND < 0) {
                if (g.act) act4(v, g.act);
            }
            if (KIND < 0 && g.rope_cos && n < g.rope_cols) {
                // rotary embedding of the (2i, 2i+1) pairs this lane holds: row = token m % T, column within the head n % hd
                const int t = m % g.rope_T;
                if (t >= g.rope_t0) {
                    const long long at = (long long)t * g.rope_hd + n % g.rope_hd;
                    const float4 c = *(const float4 *)(g.rope_cos + at), sn = *(const float4 *)(g.rope_sin + at);
                    const float y0 = v[0] * c.x - v[1] * sn.x, y1 = v[1] * c.y + v[0] * sn.y;
                    const float y2 = v[2] * c.z - v[3] * sn.z, y3 = v[3] * c.w + v[2] * sn.w;
                    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
                }
            }
            if (KIND == 3 || (KIND < 0 && g.add)) { v[0] += add_r[i][j].x; v[1] += add_r[i][j].y; v[2] += add_r[i][j].z; v[3] += add_r[i][j].w; }
            if (KIND < 0 && g.best) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.n_valid && v[r] > row_best) { row_best = v[r]; row_arg = n + r; }   // ascending columns: ties keep the first
                if (!g.store) continue;
            }
            if (KIND == 3 || (KIND < 0 && g.out_dtype == 0)) {
                *(float4 *)((float *)g.C + md[i] * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 p;
                if (g.out_dtype == 2) { p.x = pack_bf16(v[0], v[1]); p.y = pack_bf16(v[2], v[3]); }
                else { p.x = pack_f16(v[0], v[1]); p.y = pack_f16(v[2], v[3]); }
                *(uint2 *)((uint16_t *)g.C + md[i] * g.ldc + n) = p;
            }
        }
        if (KIND < 0 && g.best) {
            // the row's columns of this wave tile sit in the 4 lanes that share fr: two xor-shuffles, then ONE 64-bit atomicMax per
            // row and wave on (order-preserving float bits << 32 | ~column): larger score wins, equal scores keep the smaller column
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
    }
    };
    if constexpr (BN == 448) {
        // ---- full-row epilogue: C (f32) = acc + bias (+ residual), then LayerNorm of the row -> rln_out (bf16).  A row's 448 columns sit in two waves
        // (wave & 1: columns 0..223 / 224..447) x 4 lanes (fq) x 14 tiles: in-lane sums, two xor-shuffles, one exchange through LDS per statistic.
        bool ok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 16 + fr;
            ok[i] = m < g.M && md[i] >= 0;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn0 + j * 16 + fq * 4;
                f32x4 v = acc[i][j] * g.alpha;
                if (g.bias) v += *(const f32x4 *)(g.bias + n);
                if (g.add && ok[i]) v += *(const f32x4 *)(g.add + add_row(g, m, md[i]) * g.ld_add + n);
                acc[i][j] = v;
                if (ok[i]) *(f32x4 *)((float *)g.C + md[i] * g.ldc + n) = v;
            }
        }
        float *red = (float *)smem;                                // [BM][2]; the ring's stages are dead once every wave is past its last product
        float mean[TM], rstd[TM];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (fq == 0) red[(wm0 + i * 16 + fr) * 2 + (wave & 1)] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) mean[i] = (red[(wm0 + i * 16 + fr) * 2] + red[(wm0 + i * 16 + fr) * 2 + 1]) * (1.0f / 448.0f);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] -= mean[i];
                q += (acc[i][j][0] * acc[i][j][0] + acc[i][j][1] * acc[i][j][1]) + (acc[i][j][2] * acc[i][j][2] + acc[i][j][3] * acc[i][j][3]);
            }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (fq == 0) red[(wm0 + i * 16 + fr) * 2 + (wave & 1)] = q;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) rstd[i] = rsqrtf((red[(wm0 + i * 16 + fr) * 2] + red[(wm0 + i * 16 + fr) * 2 + 1]) * (1.0f / 448.0f) + g.rln_eps);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn0 + j * 16 + fq * 4;
            const f32x4 ga = *(const f32x4 *)(g.rln_g + n), be = *(const f32x4 *)(g.rln_b + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 y = acc[i][j] * rstd[i] * ga + be;
                if (ok[i]) *(uint2 *)(g.rln_out + md[i] * g.rln_ld + n) = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
            }
        }
        return;
    }
    const bool simple = !g.best && !g.rope_cos;
    if (simple && g.out_dtype != 0 && !g.add && g.act == 0) epilogue(std::integral_constant<int, 0>{});
    else if (simple && g.out_dtype != 0 && !g.add && g.act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (simple && g.out_dtype == 0 && g.add && g.act == 0) epilogue(std::integral_constant<int, 3>{});
    else epilogue(std::integral_constant<int, -1>{});
}

template <int BM, int BN> struct Stages { static constexpr int value = (BM == 128 && BN == 128) || (BM == 64 && BN == 64) ? 4 : 3; };

// environment knobs of the dispatcher (see ovo_knobs_dynamic)
struct GemmKnobs {
    bool no_chunk, w4, no_ns2, no_stream, no_8p, no_tuned, has_tile;
    char tile[16];
    void read() {
        no_chunk = getenv("OVO_GEMM_NO_CHUNK"); w4 = getenv("OVO_GEMM_W4"); no_ns2 = getenv("OVO_GEMM_NO_NS2");
        no_stream = getenv("OVO_GEMM_NO_STREAM"); no_8p = getenv("OVO_GEMM_NO_8P"); no_tuned = getenv("OVO_GEMM_NO_TUNED");
        const char *t = getenv("OVO_GEMM_TILE");
        has_tile = t != nullptr;
        snprintf(tile, sizeof(tile), "%s", t ? t : "");
    }
};
static const GemmKnobs &gemm_knobs() {
    static GemmKnobs k = [] { GemmKnobs x; x.read(); return x; }();
    if (ovo_knobs_dynamic()) k.read();
    return k;
}

template <int BM, int BN, int BK, typename VT, int NW = 4, int NS = Stages<BM, BN>::value>
int launch(const GemmArgs &g0, hipStream_t s) {
    GemmArgs g = g0;
    g.nbn = (g.N + BN - 1) / BN;
    const int nbm = (g.M + BM - 1) / BM;
    const size_t lds_max = (size_t)NS * (BM + BN) * BK * 2;
    // short-K GEMMs (the SAM2 stage-1/2 layers: K = 128..256) touch only their first K/BK ring stages: ask for just those,
    // so that more workgroups fit a CU and cover each other's load and store latency (these GEMMs are HBM streams)
    const int nt = g.K / BK;
    const size_t lds = (size_t)(nt < NS ? nt : NS) * (BM + BN) * BK * 2;
    static bool attr_done = false;              // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm<BM, BN, BK, NS, VT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) { ovo_set_error("ovo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        attr_done = true;
    }
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(4 + (BM == 128 ? 0 : 2) + (BN == 128 ? 0 : 1), 2.0 * g.M * (double)g.N * g.K, s); ovo_prof_shape(g.M, g.N, g.K); ovo_prof_flags(gemm_flags(g)); ovo_prof_bytes(gemm_algorithmic_bytes(g)); }   // kinds 4..7: 128x128, 128x64, 64x128, 64x64
    // Chunked order pays when the A panels outweigh the weights (M > N: per-XCD fills A/8 + W instead of A + W/8) or when
    // the n-tile count is not a multiple of 8 (round-robin then spreads every panel over every L2).
    g.tiles = nbm * g.nbn;
    g.chunk = (g.M > g.N || g.nbn % 8 != 0) && !gemm_knobs().no_chunk ? (g.tiles + 7) / 8 : 0;
    const int grid = g.chunk > 0 ? g.chunk * 8 : g.tiles;
    k_gemm<BM, BN, BK, NS, VT, NW><<<grid, 64 * NW, lds, s>>>(g);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

// The 256-row ping-pong kernels' tile width for a shape (256 or 128) and whether they beat the ring kernels there: the cost model of `dispatch` below
// (its comment block explains the constants); also the LayerNorm fold's choice, which runs on these kernels regardless of `wins`.
static int choose8p(int M, int N, int K, bool out_f32, bool add, bool *wins, bool tie256 = false) {
    auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    const double flop = 2.0 * M * (double)N * K, kt = K / 64;
    const double bytes = 2.0 * ((double)M * K + (double)N * K) + (double)M * N * (out_f32 ? 4.0 : 2.0) * (add ? 2.0 : 1.0);
    const double t_mem = bytes / 4.0e6;                                            // us at 4 TB/s
    // (the ring kernels' rate falls with K: 650 TFLOP/s from 16 K-tiles, ~560 at 7 (K = 448), ~400 at 4 (K = 256): profiles/r03c_gemm_variants.txt)
    const double t_ring = flop / (kt <= 4 ? 400.0e6 : (kt <= 8 ? 560.0e6 : 650.0e6)) + 4.0;
    const double r256 = (double)((blocks(256, 256) + 255) / 256), r128 = (double)((blocks(256, 128) + 255) / 256);
    // per-K-tile cost of the 256x128 form rises once every CU holds a tile and stays in its K-loop (measured: 0.62 us with <= 192 tiles
    // in flight, ~0.70 on a full chip with short K-loops whose phases interleave, 0.86 at K >= 3072: every CU in its K-loop at once)
    const double s128 = blocks(256, 128) <= 192 ? 0.62 : (kt >= 48 ? 0.86 : 0.70);
    // short K-loops over many rounds (Hiera stage 3: K = 448, 6-10 rounds): the rounds of different CUs drift apart and the 256 x 128 form's
    // shorter tiles overlap each other's prologues and epilogues (11 us per round measured instead of 14), the 256 x 256 form's do not (20.5):
    // (58800, 1344, 448) 126 us on the ring kernel -> 112; (196608, 1344, 256) 352 -> 300
    const double fix256 = (kt <= 8 && r256 >= 4) ? 10.0 : 8.0, fix128 = (kt <= 8 && r128 >= 4) ? 6.5 : 9.0;
    double t256 = r256 * (fix256 + 1.5 * kt), t128 = r128 * (fix128 + s128 * kt);
    t256 = t256 > t_mem ? t256 : t_mem; t128 = t128 > t_mem ? t128 : t_mem;
    const double t8 = t256 < t128 ? t256 : t128;
    if (wins) *wins = t8 < 0.93 * (t_ring > t_mem ? t_ring : t_mem);
    // (both under the memory floor: the model cannot tell them apart.  The fold's launches take the wide tile then -- its out-projection at 16 156 rows measured
    //  47.4 us on 256 x 256 against 53.6 on 256 x 128, tools/fold_bench.py with OVO_FOLD_TILE; the plain products keep the rule the tile table was measured against)
    return t256 < t128 || (tie256 && t256 == t128) ? 256 : 128;
}

template <typename VT>
int dispatch(const GemmArgs &g, hipStream_t s) {
    const GemmKnobs &kn = gemm_knobs();
    const bool k64 = g.K % 64 == 0;
    auto blocks = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
    // Tile choice (tools/gemm_bench.py, MI355X): these GEMMs are a handful of workgroup "rounds" long, so round
    // quantisation on 256 CUs decides.  cost = rounds x tile area at the workgroups a CU holds (2 for the 64/128-mixed
    // tiles, 1 for 128x128, whose round therefore counts half its area), x 1.4 for 64x64 (most L2 traffic per flop: it
    // only wins when it saves a round).  128x64 is listed first: on ties it beat 64x128 by 4-10% in the sweep.  128x128 (one
    // 8-wave workgroup per CU, 867 TFLOP/s at 8192^3, 10-18% ahead on the M = 1M decoder / 1k-text query GEMMs) enters
    // from 8 full rounds and wins ties.
    // The 8-wave 128x64 kernel needs 65 VGPRs, so a CU takes three of its workgroups once the ring is two stages deep (48 KB)
    // instead of three (72 KB): 768 resident workgroups instead of 512, each a little slower.  Taken when that saves
    // rounds: FC1 (1154, 4096, 1024) = 640 workgroups 24.0 -> 21.0 us, (1154, 8192, 1024) 37.2 -> 33.1; QKV (480 workgroups)
    // and the Hiera stage-3 GEMMs (800-900) are better off with the deeper ring (tools/gemm_bench.py).
    const long long b12864 = blocks(128, 64);
    const double r512 = (double)((b12864 + 511) / 512), r768 = 1.5 * (double)((b12864 + 767) / 768);
    const bool ns2_ok = k64 && !kn.w4 && !kn.no_ns2;
    bool ns2 = false;
    int bm = 128, bn = 64;
    {
        const int cand[3][2] = {{128, 64}, {64, 128}, {64, 64}};
        double best = -1.0;
        for (int i = 0; i < 3; ++i) {
            double rounds = (double)((blocks(cand[i][0], cand[i][1]) + 511) / 512);
            if (i == 0 && ns2_ok && r768 <= r512) rounds = r768;
            const double cost = rounds * cand[i][0] * cand[i][1] * (i == 2 ? 1.4 : 1.0);
            if (best < 0 || cost < best) { best = cost; bm = cand[i][0]; bn = cand[i][1]; }
        }
        if (blocks(128, 128) >= 2048 && (double)((blocks(128, 128) + 255) / 256) * 8192.0 <= best) { bm = 128; bn = 128; }
    }
    ns2 = ns2_ok && bm == 128 && bn == 64 && r768 <= r512;
    // The 256-row ping-pong kernels (gemm8p.hip) for the batched forwards.  One workgroup per CU, so their cost is
    //   rounds x (fixed + K-tiles x per-tile), fixed ~8-9 us (prologue from HBM, epilogue, nothing overlaps them), per 64-deep K-tile
    //   1.5 us (256x256) / 0.62-0.86 us (256x128), floored by the HBM time of the operands and the output;
    //   (a 256x128, BK = 32, 3-stage form of the ring kernel below with TWO workgroups per CU -- so that one tile's epilogue would overlap
    //   the other's K-loop -- measured 1.4-1.6x SLOWER than the ping-pong kernel: 16 MFMAs per barrier do not keep the K-loop fed)
    // the ring kernels above run ~650 TFLOP/s + 4 us at these sizes (tools/gemm_bench.py on MI355X, profiles/r02*_gemm_sweep.txt).
    // Tall short-K products are HBM streams: the weights-resident streaming kernel (gemm_stream.hip) runs them at 3+ TB/s, the tiled
    // kernels below at ~2 (tools/gemm_bench.py, profiles/r02c_gemm_stream.txt).
    const char *force_tile = kn.has_tile ? kn.tile : nullptr;
    if (!force_tile && !kn.no_tuned) {                               // measured choices first (gemm_tuned.h, tools/gemm_tune.py); family knobs still win
        const int f = gemm_flags(g);
        for (const TunedTile &e : kTunedTiles)
            if (e.M == g.M && e.N == g.N && e.K == g.K && e.flags == f && e.tile) {
                const bool is8p = e.tile[0] == '2', is_stream = e.tile[0] == 's';
                if (!(is8p && kn.no_8p) && !(is_stream && kn.no_stream)) force_tile = e.tile;
                break;
            }
    }
    if (g.M >= 16384 && g.K <= 256 && ((!force_tile && !kn.no_stream) || (force_tile && !strcmp(force_tile, "stream")))) {
        const int rc = gemm_stream_launch(g, std::is_same<VT, bf16x8>::value ? 2 : 1, s);
        if (rc != OVO_E_UNSUPPORTED) return rc;
    }
    if (k64 && g.M >= 2048 && g.N >= 256 && !force_tile && !kn.no_8p) {
        bool wins = false;
        const int bn8 = choose8p(g.M, g.N, g.K, g.out_dtype == 0, g.add != nullptr, &wins);
        if (wins) {
            const int rc = gemm8p_launch(g, bn8, std::is_same<VT, bf16x8>::value ? 2 : 1, s);
            if (rc != OVO_E_UNSUPPORTED) return rc;
        }
    }
    if (const char *force = force_tile) {                        // tuning knob (tools/gemm_bench.py): "128x128", "64x128", "256x256", ...
        int fm = 0, fn = 0;
        if (sscanf(force, "%dx%d", &fm, &fn) == 2 && (fm == 64 || fm == 128) && (fn == 64 || fn == 128)) { bm = fm; bn = fn; }
        if (!strcmp(force, "256x128p")) {                          // the persistent form (gemm8q.hip)
            const int rc = gemm8q_launch(g, std::is_same<VT, bf16x8>::value ? 2 : 1, s);
            if (rc != OVO_E_UNSUPPORTED) return rc;
        }
        if (fm == 256 && (fn == 256 || fn == 128) && k64) {      // (the tuned table keys on (M, N, K, flags) only: a launch the ping-pong kernel declines --
            const int rc = gemm8p_launch(g, fn, std::is_same<VT, bf16x8>::value ? 2 : 1, s);      // 32-bit DMA offsets -- falls through to the ring kernels)
            if (rc != OVO_E_UNSUPPORTED) return rc;
        }
    }
    if (bm == 128 && bn == 128) return k64 ? launch<128, 128, 64, VT, 8, 3>(g, s) : launch<128, 128, 32, VT, 8, 3>(g, s);
    // BK = 64: 8 waves per workgroup on every tile (two workgroups = 16 waves per CU): same LDS and L2 traffic as the 4-wave
    // form, twice the loads in flight and MFMA/LDS phases of different waves overlapping -- QKV 16.7 -> 14.9 us, FC1 25.1 ->
    // 21.7, (4096,1792,448) 18.6 -> 15.8 (tools/gemm_bench.py).  BK = 32 tiles are too small for 512 threads' 16-byte pieces.
    if (k64 && !kn.w4) {
        if (bm == 128 && bn == 64 && ns2) return launch<128, 64, 64, VT, 8, 2>(g, s);
        if (bm == 128 && bn == 64) return launch<128, 64, 64, VT, 8, 3>(g, s);
        if (bm == 64 && bn == 128) return launch<64, 128, 64, VT, 8, 3>(g, s);
        if (bm == 64 && bn == 64) return launch<64, 64, 64, VT, 8, 4>(g, s);
    }
#define GO(BM, BN)                                                         \
    if (bm == BM && bn == BN) return k64 ? launch<BM, BN, 64, VT>(g, s) : launch<BM, BN, 32, VT>(g, s);
    GO(64, 128) GO(128, 64) GO(64, 64)
#undef GO
    return OVO_E_UNSUPPORTED;
}

}  // namespace

struct RowLn { const float *g, *b; float eps; void *out; long long ld; };
static int gemm_entry(const ovo_gemm_t *p, unsigned long long *best, int store, int n_valid, ovo_stream_t stream, const ovo_rope_t *rope = nullptr,
                      const ovo_window_t *win = nullptr, long long add_rows = 0, const RowLn *rln = nullptr, const ovo_gemm_detail::FoldOut *fold_out = nullptr,
                      const ovo_gemm_detail::FoldIn *fold_in = nullptr) {
    OVO_REQUIRE(p, "null descriptor");
    OVO_REQUIRE(p->M >= 0 && p->N > 0 && p->K > 0, "bad shape");
    if (p->M == 0) return OVO_OK;
    OVO_REQUIRE(p->A && p->W && (p->C || (best && !store)), "null pointer");
    OVO_REQUIRE(p->K % 32 == 0, "K must be a multiple of 32 (pad activations and weights with zeros)");
    OVO_REQUIRE(p->N % 4 == 0, "N must be a multiple of 4");
    OVO_REQUIRE(p->in_dtype == 1 || p->in_dtype == 2, "in_dtype: 1 = f16, 2 = bf16");
    OVO_REQUIRE(p->out_dtype >= 0 && p->out_dtype <= 2, "out_dtype: 0 = f32, 1 = f16, 2 = bf16");
    OVO_REQUIRE(p->lda % 8 == 0 && p->ldw % 8 == 0 && (((uintptr_t)p->A | (uintptr_t)p->W) & 15) == 0, "A/W rows must be 16-byte aligned");
    OVO_REQUIRE(p->ldc % 4 == 0 && ((uintptr_t)p->C & 15) == 0, "C rows must be 16-byte aligned");
    OVO_REQUIRE(!p->add || (p->ld_add % 4 == 0 && ((uintptr_t)p->add & 15) == 0), "add rows must be 16-byte aligned");
    OVO_REQUIRE(!p->bias || ((uintptr_t)p->bias & 15) == 0, "bias must be 16-byte aligned");
    GemmArgs g;
    g.A = (const char *)p->A; g.lda = p->lda; g.W = (const char *)p->W; g.ldw = p->ldw; g.bias = p->bias;
    g.C = p->C; g.ldc = p->ldc; g.add = p->add; g.ld_add = p->ld_add;
    g.M = p->M; g.N = p->N; g.K = p->K; g.out_dtype = p->out_dtype; g.act = p->act; g.alpha = p->alpha; g.nbn = 0;
    g.best = best; g.store = store; g.n_valid = n_valid; g.add_rows = (int)add_rows; g.strip = 0;
    g.rope_cos = g.rope_sin = nullptr; g.rope_T = 1; g.rope_hd = 4; g.rope_cols = 0; g.rope_t0 = 0;
    g.ln_x = g.ln_g = g.ln_b = nullptr; g.ln_eps = 0.f; g.ln_d = 0; g.ln_mode = 0; g.pool_ww = 0; g.qpool_out = nullptr; g.qpool_cols = 0;
    g.rln_g = g.rln_b = nullptr; g.rln_eps = 0.f; g.rln_out = nullptr; g.rln_ld = 0;
    g.xb_out = nullptr; g.ld_xb = 0; g.stat_out = nullptr; g.stat_ld = 0; g.fold_stats = nullptr; g.fold_parts = 0; g.fold_D = 0; g.fold_cs = nullptr; g.fold_eps = 0.f;
    g.dbg = 0; g.slab16 = 0; g.rope_lds = 0; g.stamps = nullptr; g.win_per = 0; g.win_ww = g.win_wh = g.win_nww = g.win_nwin = 1; g.win_H = g.win_W = 0;
    if (win) {
        OVO_REQUIRE(win->B > 0 && win->H > 0 && win->W > 0 && win->wh > 0 && win->ww > 0, "bad window descriptor");
        const int nwh = (win->H + win->wh - 1) / win->wh, nww = (win->W + win->ww - 1) / win->ww;
        OVO_REQUIRE((long long)win->B * nwh * nww * win->wh * win->ww == p->M, "M must be B x windows x window size (padding rows included)");
        g.win_per = win->wh * win->ww; g.win_ww = win->ww; g.win_wh = win->wh; g.win_nww = nww; g.win_nwin = nwh * nww; g.win_H = win->H; g.win_W = win->W;
    }
    if (rope) {
        OVO_REQUIRE(rope->cos && rope->sin && rope->T > 0 && rope->hd > 0 && rope->hd % 4 == 0 && rope->cols % rope->hd == 0 && rope->cols <= p->N &&
                    rope->t0 >= 0 && (((uintptr_t)rope->cos | (uintptr_t)rope->sin) & 15) == 0, "bad rope descriptor");
        g.rope_cos = rope->cos; g.rope_sin = rope->sin; g.rope_T = rope->T; g.rope_hd = rope->hd; g.rope_cols = rope->cols; g.rope_t0 = rope->t0;
    }
    if (rln) {                                                   // full-row tile + LayerNorm of the result (Hiera stage 3's projection + norm2)
        if (p->N != 448 || p->K % 64 != 0 || p->out_dtype != 0 || p->in_dtype != 2 || p->act != 0 || p->alpha != 1.0f || add_rows != 0) return OVO_E_UNSUPPORTED;
        OVO_REQUIRE(rln->g && rln->b && rln->out && rln->ld >= 448 && rln->ld % 4 == 0 && ((uintptr_t)rln->out & 7) == 0 &&
                    (((uintptr_t)rln->g | (uintptr_t)rln->b) & 15) == 0, "bad row-LayerNorm operands");
        g.rln_g = rln->g; g.rln_b = rln->b; g.rln_eps = rln->eps; g.rln_out = (uint16_t *)rln->out; g.rln_ld = rln->ld;
        const int rc = launch<128, 448, 64, bf16x8, 8, 2>(g, (hipStream_t)stream);
        if (rc != OVO_OK) return rc;
        OVO_CHECK_LAUNCH();
        return OVO_OK;
    }
    if (fold_out || fold_in) {                                    // the LayerNorm fold: ping-pong kernel or nothing
        int bn = ovo_gemm_detail::gemm_fold_ok(p->M, p->N, p->K) ? choose8p(p->M, p->N, p->K, p->out_dtype == 0, p->add != nullptr, nullptr, true) : 0;
        if (bn && ovo_knobs_dynamic() && getenv("OVO_FOLD_TILE")) bn = atoi(getenv("OVO_FOLD_TILE")) == 128 ? 128 : 256;      // tools/fold_bench.py: either tile width on the fold's launches
        if (!bn || p->in_dtype != 2 || p->alpha != 1.0f || win || add_rows || best) return OVO_E_UNSUPPORTED;
        if (fold_out) {
            if (p->out_dtype != 0 || !p->add || p->act != 0 || rope) return OVO_E_UNSUPPORTED;
            OVO_REQUIRE(fold_out->xb && fold_out->stats && fold_out->ld_xb % 8 == 0 && ((uintptr_t)fold_out->xb & 15) == 0 && fold_out->ld_stats >= p->M &&
                        ((uintptr_t)fold_out->stats & 7) == 0, "bad fold outputs");
            g.xb_out = (uint16_t *)fold_out->xb; g.ld_xb = fold_out->ld_xb; g.stat_out = fold_out->stats; g.stat_ld = fold_out->ld_stats;
        } else {
            if (p->out_dtype == 0 || p->add || !(p->act == 0 || p->act == 1) || !p->bias) return OVO_E_UNSUPPORTED;
            OVO_REQUIRE(fold_in->stats && fold_in->colsum && fold_in->parts >= 1 && fold_in->parts <= 16 && fold_in->D > 0 && ((uintptr_t)fold_in->colsum & 15) == 0 &&
                        fold_in->ld_stats >= p->M && ((uintptr_t)fold_in->stats & 7) == 0, "bad fold inputs");
            g.fold_stats = fold_in->stats; g.stat_ld = fold_in->ld_stats; g.fold_parts = fold_in->parts; g.fold_D = fold_in->D; g.fold_cs = fold_in->colsum; g.fold_eps = fold_in->eps;
        }
        const int rc = gemm8p_launch(g, bn, 2, (hipStream_t)stream);
        if (rc != OVO_OK) return rc;
        OVO_CHECK_LAUNCH();
        return OVO_OK;
    }
    const int rc = p->in_dtype == 2 ? dispatch<bf16x8>(g, (hipStream_t)stream) : dispatch<f16x8>(g, (hipStream_t)stream);
    if (rc != OVO_OK) return rc;
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

// ---- the LayerNorm fold's entry points (gemm_common.h) ----
// The first LayerNorm of a forward has no producer product in front of it: one wave per row writes the bf16 copy and the row's (sum, sum of squares) as ONE
// partial (parts = 1) -- the same quantities, in the same layout, that the producer epilogue leaves for every later LayerNorm.
__global__ void __launch_bounds__(256) k_fold_rowstats(const float *__restrict__ x, long long ldx, int M, int D, uint16_t *__restrict__ xb, long long ld_xb, float2 *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 v = *(const float4 *)(x + row * ldx + c);
        *(uint2 *)(xb + row * ld_xb + c) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) stats[row] = make_float2(s1, s2);
}
int ovo_gemm_detail::gemm_fold_rowstats(const float *x, long long ldx, int M, int D, void *xb, long long ld_xb, float *stats, ovo_stream_t stream) {
    OVO_REQUIRE(x && xb && stats && M > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ld_xb % 4 == 0, "bad argument");
    k_fold_rowstats<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ldx, M, D, (uint16_t *)xb, ld_xb, (float2 *)stats);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
int ovo_gemm_detail::gemm_fold_parts(int N) { return N / 64; }      // one partial per 64-column wave tile of the producer (either tile width)
bool ovo_gemm_detail::gemm_fold_ok(int M, int N, int K) {
    // both halves run on the ping-pong kernel: batched forwards only (M >= 2048 rows, the dispatcher's own bound), whole 64-deep K-tiles, whole 64-column wave tiles
    return M >= 2048 && K % 64 == 0 && K >= 64 && N % 64 == 0 && N >= 256;
}
int ovo_gemm_detail::gemm_fold_producer(const ovo_gemm_t *p, const FoldOut &o, ovo_stream_t stream) { return gemm_entry(p, nullptr, 1, 0, stream, nullptr, nullptr, 0, nullptr, &o); }
int ovo_gemm_detail::gemm_fold_consumer(const ovo_gemm_t *p, const ovo_rope_t *rope, const FoldIn &f, ovo_stream_t stream) {
    return gemm_entry(p, nullptr, 1, 0, stream, rope, nullptr, 0, nullptr, nullptr, &f);
}

extern "C" int ovo_gemm_fold_out(const ovo_gemm_t *p, void *xb, int64_t ld_xb, float *stats, int64_t ld_stats, ovo_stream_t stream) {
    const ovo_gemm_detail::FoldOut o = {xb, (long long)ld_xb, stats, (long long)ld_stats};
    return ovo_gemm_detail::gemm_fold_producer(p, o, stream);
}
extern "C" int ovo_gemm_fold_stats(const float *x, int64_t ldx, int M, int D, void *xb, int64_t ld_xb, float *stats, ovo_stream_t stream) {
    return ovo_gemm_detail::gemm_fold_rowstats(x, ldx, M, D, xb, ld_xb, stats, stream);
}
extern "C" int ovo_gemm_fold_in(const ovo_gemm_t *p, const ovo_rope_t *rope, const float *stats, int64_t ld_stats, int parts, int D, const float *rowsum, float eps,
                                ovo_stream_t stream) {
    const ovo_gemm_detail::FoldIn f = {stats, (long long)ld_stats, parts, D, rowsum, eps};
    return ovo_gemm_detail::gemm_fold_consumer(p, rope, f, stream);
}

extern "C" int ovo_gemm(const ovo_gemm_t *p, ovo_stream_t stream) { return gemm_entry(p, nullptr, 1, 0, stream); }

// ovo_gemm with the rotary embedding of PE's attention (ovo_rope_qk) applied to the f32 accumulators of columns [0, cols) before
// the store: the QKV projection writes rotated q, k directly (one rounding instead of two, one launch less per block).
extern "C" int ovo_gemm_rope(const ovo_gemm_t *p, const ovo_rope_t *rope, ovo_stream_t stream) {
    OVO_REQUIRE(rope, "null rope descriptor");
    return gemm_entry(p, nullptr, 1, 0, stream, rope);
}

// ovo_gemm whose C / add rows are addressed in spatial order while the product rows arrive in window order (Hiera's attention
// output projection): the un-windowing + residual add pass of the block happens in the epilogue.
extern "C" int ovo_gemm_unwindow(const ovo_gemm_t *p, const ovo_window_t *win, ovo_stream_t stream) {
    OVO_REQUIRE(win, "null window descriptor");
    return gemm_entry(p, nullptr, 1, 0, stream, nullptr, win);
}

int ovo_gemm_detail::gemm_unwindow_rowln(const ovo_gemm_t *p, const ovo_window_t *win, const float *ln_g, const float *ln_b, float eps, void *ln_out,
                                         long long ld_ln, ovo_stream_t stream) {
    const RowLn r = {ln_g, ln_b, eps, ln_out, ld_ln};
    return gemm_entry(p, nullptr, 1, 0, stream, nullptr, win, 0, &r);
}
extern "C" int ovo_gemm_rowln(const ovo_gemm_t *p, const ovo_window_t *win, const float *ln_g, const float *ln_b, float eps, void *ln_out, int64_t ld_ln,
                              ovo_stream_t stream) {
    return ovo_gemm_detail::gemm_unwindow_rowln(p, win, ln_g, ln_b, eps, ln_out, ld_ln, stream);
}

// ovo_gemm whose `add` operand is periodic in the rows: product row m adds add[m % add_rows] (a per-pixel constant shared by every prompt of
// the SAM2 decoder: the projection of the positional code, so that k_proj(keys + pe) = keys . Wk^T + (pe . Wk^T)[pixel] needs no (keys + pe) tensor).
extern "C" int ovo_gemm_periodic(const ovo_gemm_t *p, int64_t add_rows, ovo_stream_t stream) {
    OVO_REQUIRE(p && p->add && add_rows > 0 && add_rows < (1ll << 31), "periodic add needs add and 0 < add_rows < 2^31");
    return gemm_entry(p, nullptr, 1, 0, stream, nullptr, nullptr, add_rows);
}

// C as ovo_gemm, plus a fused per-row first-max argmax over columns [0, n_valid): best u64 [M] must be ZERO on entry and holds, per
// row, (order-preserving bits of the best value << 32) | (0xffffffff - column); store_scores = 0 skips the C stores altogether
// (the 5 GB score matrix of a 1.25M x 1000 query is then never written).  Decode with ovo_decode_best.
extern "C" int ovo_gemm_argmax(const ovo_gemm_t *p, uint64_t *best, int store_scores, int n_valid, ovo_stream_t stream) {
    OVO_REQUIRE(best && n_valid > 0 && p && n_valid <= p->N, "best must be non-null, 0 < n_valid <= N");
    return gemm_entry(p, (unsigned long long *)best, store_scores != 0, n_valid, stream);
}

namespace {
__global__ void __launch_bounds__(256) k_decode_best(const unsigned long long *__restrict__ best, long long n, float th, long long *__restrict__ cls,
                                                     float *__restrict__ conf) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = best[i];
    uint32_t u = (uint32_t)(b >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float v = __uint_as_float(u);
    long long c = (long long)(0xffffffffu - (uint32_t)(b & 0xffffffffu));
    if (b == 0ull || v <= th) { v = 0.f; c = -1; }                // threshold semantics of ovo.py:487-491
    cls[i] = c;
    conf[i] = v;
}
}  // namespace

extern "C" int ovo_decode_best(const uint64_t *best, int64_t n, float th, int64_t *out_cls, float *out_conf, ovo_stream_t stream) {
    OVO_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return OVO_OK;
    OVO_REQUIRE(best && out_cls && out_conf, "null pointer");
    k_decode_best<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned long long *)best, n, th, (long long *)out_cls, out_conf);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
