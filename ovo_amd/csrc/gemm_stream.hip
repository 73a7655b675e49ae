// gemm_stream.hip -- ovo_gemm for tall, short-K products (M >= 16 K rows, K <= 256): the SAM2 Hiera stage-1 / stage-2 layers at
// 1024^2 (64 K - 512 K tokens x 112 / 224 channels) and their like.  These are HBM streams -- 30-130 flops per byte moved -- and the
// tiled kernels (gemm.hip) ran them at ~2 TB/s: with 2-4 K-tiles a workgroup is all prologue and epilogue.  Here (skinny.h) a column
// group of the weights (<= 256 columns, <= 128 KB) is LDS-resident for the life of a workgroup, every wave streams its own 16-row blocks
// of A straight from global memory into MFMA fragments and finishes them in registers (bias, activation, residual, window -> spatial
// row mapping, cast), with no barrier after the weights are in.  Column groups of one product run side by side ON ONE XCD (see the
// blockIdx mapping), so the A rows they share meet in that XCD's L2.
#include <stdlib.h>

#include <type_traits>

#include "skinny.h"

using namespace ovo_gemm_detail;

namespace {

// F32A: the A operand is the f32 residual stream itself -- LayerNorm (g.ln_mode 1) or a plain cast (2) of source row row_dest(m) is taken while
// the fragments are loaded (a row of K <= 256 values lives in the 4 lanes (fr, 0..3): in-lane sums + two xor shuffles give its statistics), so the
// normalised bf16 copy of the stream is never written or read: k_ln_window / k_cast_pad and their 6 bytes per element of HBM traffic disappear for
// the layers this kernel runs (Hiera stages 1-2, the FPN laterals of those stages, conv_s0 / conv_s1).
template <int KS, int NT, typename VT, int NTHREADS, bool F32A = false>
__global__ void __launch_bounds__(NTHREADS) k_gemm_stream(GemmArgs g, int n_groups, int slots) {
    constexpr int K = KS * 32, NG = NT * 16;
    using S = ovo_skinny::Skinny<K, NG, VT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *cs = (float *)(smem + S::W_BYTES);
    float *lg = cs + NG, *lb = lg + K;                               // F32A: LayerNorm weight / bias of the K (padded) input channels
    const float2 *lut = (g.act == 1 && g.gelu_lut) ? (const float2 *)(cs + NG + (F32A ? 2 * K : 0)) : nullptr;      // GELU table (gemm_common.h)
    if (lut) gelu_lut_fill((float2 *)lut, threadIdx.x, NTHREADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    // The column groups of one slot read the SAME rows of A: they must share an L2, i.e. sit on one XCD (workgroup i goes to XCD i % 8) and start
    // together.  With group = i % n_groups, slot = i / n_groups the groups of a slot landed on DIFFERENT XCDs and every one of them fetched the rows
    // itself (PMC, tools/pmc_shapes.py: (196608, 896, 256) fetched 404 MB for 100 MB of A -- four groups, four reads; these launches are HBM-bound)
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int group = within % n_groups, slot = (within / n_groups) * 8 + xcd;
    if (slot >= slots) return;
    const int n0 = group * NG;
    S::load_w(smem, (const uint16_t *)g.W + (long long)n0 * g.ldw, g.ldw, tid, NTHREADS);
    for (int i = tid; i < NG; i += NTHREADS) cs[i] = g.bias ? g.bias[n0 + i] : 0.f;
    if (F32A) {
        for (int i = tid; i < K; i += NTHREADS) {
            lg[i] = (g.ln_mode == 1 && i < g.ln_d) ? g.ln_g[i] : 0.f;
            lb[i] = (g.ln_mode == 1 && i < g.ln_d) ? g.ln_b[i] : 0.f;
        }
    }
    __syncthreads();
    const float *cl = cs + fq * 4;
    const int blocks = (g.M + 15) / 16;
    constexpr int WPB = NTHREADS / 64;
    // (measured and dropped, round 4: the next block's A rows fetched as soon as this block's rows are packed into fragments, to fly under its products
    //  and stores -- K = 256 products gained 4-7 % ((196608, 896, 256) 307 -> 294 us), K = 128 products lost 10-15 % to the wave of occupancy the extra
    //  registers cost ((786432, 448, 128) 326 -> 367); a full second register set one block ahead spilled 88 registers.  These launches are write-heavy
    //  streams -- 705 MB written for 352 MB read at (786432, 448, 128) -- at 3.2-3.5 TB/s of algorithmic bytes)
    for (int b = slot * WPB + wave; b < blocks; b += slots * WPB) {
        const int m = b * 16 + fr, mc = m < g.M ? m : g.M - 1;
        VT af[KS];
        if (F32A) {
            // this lane's 8 KS values of source row `src` (spatial token of product row mc; -1: a padding row of the window grid = zeros)
            const long long src = row_dest(g, mc);
            const float *xp = g.ln_x + (src < 0 ? 0 : src) * g.ln_d;
            float xv[KS][8];
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = (ks * 4 + fq) * 8;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (src >= 0 && d0 < g.ln_d) { lo = *(const float4 *)(xp + d0); hi = *(const float4 *)(xp + d0 + 4); }
                xv[ks][0] = lo.x; xv[ks][1] = lo.y; xv[ks][2] = lo.z; xv[ks][3] = lo.w;
                xv[ks][4] = hi.x; xv[ks][5] = hi.y; xv[ks][6] = hi.z; xv[ks][7] = hi.w;
                sum += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
            }
            float mean = 0.f, rstd = 1.f;
            if (g.ln_mode == 1) {                                     // two-pass statistics, as k_ln_window (columns >= ln_d hold zeros and are left out)
                sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
                mean = sum / (float)g.ln_d;
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if ((ks * 4 + fq) * 8 < g.ln_d) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const float a0 = xv[ks][e] - mean, a1 = xv[ks][e + 1] - mean;
                            q += a0 * a0 + a1 * a1;
                        }
                    }
                }
                q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
                rstd = rsqrtf(q / (float)g.ln_d + g.ln_eps);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = (ks * 4 + fq) * 8;
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    float y0 = xv[ks][e], y1 = xv[ks][e + 1];
                    if (g.ln_mode == 1) {
                        y0 = (y0 - mean) * rstd * lg[d0 + e] + lb[d0 + e];
                        y1 = (y1 - mean) * rstd * lg[d0 + e + 1] + lb[d0 + e + 1];
                    }
                    pk[e >> 1] = (src >= 0 && d0 < g.ln_d) ? pack_bf16(y0, y1) : 0u;
                }
                af[ks] = *(const VT *)pk;
            }
        } else {
            S::load_a(af, (const uint16_t *)g.A, g.lda, mc, fq);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        S::mma(acc, af, smem, fr, fq);
        const long long md = m < g.M ? (F32A ? (long long)m : row_dest(g, m)) : -1;   // window-major product row -> spatial row (ovo_gemm_unwindow), -1 = padding
        if (md < 0) continue;
        const float *ap = g.add ? g.add + add_row(g, m, md) * g.ld_add + n0 + fq * 4 : nullptr;
        auto finish = [&](int j, float (&v)[4]) {
            const f32x4 bv = *(const f32x4 *)(cl + j * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[j][r] * g.alpha + bv[r];
            // table GELU or nothing (the launch refuses other activations: the tiled kernels take them).  The generic act4 -- six activations, inlined into every
            // one of the NT column tiles of four epilogue forms -- made these kernels 26-40 K instructions long; the block loop ran out of the
            // instruction cache: (196608, 896, 256) 290 -> 246 us, (196608, 672, 256) 321 -> 271 with this body
            // (the table form only: with OVO_GELU_POLY=1 the launch declines GELU products as well -- the packed polynomial beside the table in every
            // tile was another 5 K instructions and 246 -> 271 us on the same product)
            if (g.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_lut(v[r], lut);
            }
            if (ap) {
                const f32x4 r = *(const f32x4 *)(ap + j * 16);
                v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
            }
        };
        if (F32A && g.pool_ww > 0) {
            // Hiera's stage-change skip path, maxpool2x2(proj(LN(x))): a 16-row block of window-major rows holds whole 2-row strips of a window
            // (16 % (2 ww) == 0), so the four tokens of a pooled position sit in the lanes fr, fr ^ 1, fr ^ ww, fr ^ (ww + 1) of one column group:
            // two xor shuffles take their maximum and the lane of the even / even token stores the pooled SPATIAL row -- the f32 projection of
            // every token (704 MB at 12 frames of stage 1) is never written.  (max is exact: the same bits as k_pool_unwindow's.)
            const long long src = row_dest(g, mc);                    // (b H + y) W + x of this lane's token; no padding rows here (checked at launch)
            const int x = (int)(src % g.win_W), y = (int)((src / g.win_W) % g.win_H), bb = (int)(src / ((long long)g.win_W * g.win_H));
            const long long dest = ((long long)bb * (g.win_H >> 1) + (y >> 1)) * (g.win_W >> 1) + (x >> 1);
            const bool writer = ((fr & 1) | (fr & g.pool_ww)) == 0;
            float *cp = (float *)g.C + dest * g.ldc + n0 + fq * 4;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v[4];
                finish(j, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaxf(v[r], __shfl_xor(v[r], 1, 64));
                    v[r] = fmaxf(v[r], __shfl_xor(v[r], g.pool_ww, 64));
                }
                if (writer) *(float4 *)(cp + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else if (F32A && g.qpool_cols > 0 && n0 < g.qpool_cols) {
            // Hiera's query pooling at a stage change (hieradet.py: q = do_pool(q)): the q columns of the QKV product are max-pooled 2 x 2 over the
            // window's tokens here (the four tokens of a pooled position sit in lanes fr, fr ^ 1, fr ^ ww, fr ^ (ww + 1), as in the skip path above)
            // and only the pooled rows are stored, in pooled window-major order -- k_qpool's read of every token's q and the store of it go away
            // (max commutes with the bf16 rounding: the same bits as pooling the stored values).
            const int win = mc / g.win_per, pw = mc - win * g.win_per, ly = pw / g.win_ww, lx = pw - ly * g.win_ww;
            const long long prow = (long long)win * (g.win_per >> 2) + (long long)(ly >> 1) * (g.win_ww >> 1) + (lx >> 1);
            const bool writer = ((fr & 1) | (fr & g.win_ww)) == 0;
            uint16_t *qp = g.qpool_out + prow * g.qpool_cols + n0 + fq * 4;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v[4];
                finish(j, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaxf(v[r], __shfl_xor(v[r], 1, 64));
                    v[r] = fmaxf(v[r], __shfl_xor(v[r], g.win_ww, 64));
                }
                if (writer) *(uint2 *)(qp + j * 16) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
            }
        } else if (g.out_dtype == 0) {
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

template <int KS, int NT, typename VT, bool F32A = false>
int launch_stream(const GemmArgs &g, hipStream_t s) {
    constexpr int K = KS * 32, NG = NT * 16;
    constexpr size_t lds = (size_t)NG * K * 2 + NG * sizeof(float) + (F32A ? 2 * K * sizeof(float) : 0) + GELU_LUT_BYTES;
    // > half the LDS: one workgroup per CU -- 16 waves (128 VGPRs each), or 12 when the accumulators of a wide group need more; else two or more of 8 waves
    // (the f32 A path holds a row's 8 KS floats next to the fragments: one step fewer waves, or the accumulators spill)
    constexpr int NTHREADS = lds > 80 * 1024 ? (F32A ? ((NT > 16 || KS >= 8) ? 512 : 768) : (NT > 16 ? 768 : 1024)) : 512;
    constexpr int PER_CU = lds > 80 * 1024 ? 1 : (lds > 52 * 1024 ? 2 : (lds > 39 * 1024 ? 3 : 4));
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm_stream<KS, NT, VT, NTHREADS, F32A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ovo_set_error("ovo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
        attr_done = true;
    }
    const int n_groups = g.N / NG, blocks = (g.M + 15) / 16, wpb = NTHREADS / 64;
    int slots = (256 * PER_CU + n_groups - 1) / n_groups;             // workgroups = slots x groups ~ what the chip holds at once
    const int need = (blocks + wpb - 1) / wpb;
    if (slots > need) slots = need;
    if (slots < 1) slots = 1;
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(8, 2.0 * g.M * (double)g.N * g.K, s); ovo_prof_shape(g.M, g.N, g.K); ovo_prof_flags(gemm_flags(g)); ovo_prof_bytes(gemm_algorithmic_bytes(g)); }
    GemmArgs gg = g;
    gg.gelu_lut = 1;
    k_gemm_stream<KS, NT, VT, NTHREADS, F32A><<<(slots + 7) / 8 * 8 * n_groups, NTHREADS, lds, s>>>(gg, n_groups, slots);
    if (prof) ovo_prof_end(s);
    return OVO_OK;
}

}  // namespace

namespace ovo_gemm_detail {

// Returns OVO_E_UNSUPPORTED when the shape has no instantiation (the caller then takes a tiled kernel).
int gemm_stream_launch(const GemmArgs &g, int in_dtype, hipStream_t s) {
    static int gelu_poly = getenv("OVO_GELU_POLY") != nullptr;
    if (ovo_knobs_dynamic()) gelu_poly = getenv("OVO_GELU_POLY") != nullptr;
    if (g.best || g.rope_cos || in_dtype != 2 || g.act > 1 || (g.act == 1 && gelu_poly)) return OVO_E_UNSUPPORTED;
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
    if (g.ln_mode) {                                                 // f32 A with LayerNorm / cast in the load: the shapes Hiera's stages 1-2 and FPN use
#define GO(KK, NGG) if (g.K == KK && ng == NGG) return launch_stream<KK / 32, NGG / 16, bf16x8, true>(g, s);
        GO(128, 336) GO(128, 256) GO(128, 224)
        GO(192, 288) GO(192, 256) GO(192, 144)
        GO(256, 256) GO(256, 224) GO(256, 64) GO(256, 32)
#undef GO
        return OVO_E_UNSUPPORTED;
    }
#define GO(KK, NGG) if (g.K == KK && ng == NGG) return launch_stream<KK / 32, NGG / 16, bf16x8>(g, s);
    GO(128, 336) GO(128, 256) GO(128, 224) GO(128, 112) GO(128, 64)
    GO(192, 288) GO(192, 256) GO(192, 144) GO(192, 112)
    GO(256, 256) GO(256, 224) GO(256, 112) GO(256, 64) GO(256, 32)
#undef GO
    return OVO_E_UNSUPPORTED;
}

// hiera.hip's entry: `p` as for ovo_gemm with A unused; A = LayerNorm (mode 1) / cast (mode 2) of x[source row, :d] (f32), source row of product
// row m = its spatial token when `win` describes a window partition (padding rows = zeros), m itself without.  OVO_E_UNSUPPORTED: no
// instantiation for the shape (the caller normalises / casts into a buffer and calls ovo_gemm).
static bool stream_off() {                              // OVO_GEMM_NO_STREAM / OVO_GEMM_TILE / OVO_NO_LN_FOLD (see ovo_knobs_dynamic)
    auto read = [] { return getenv("OVO_GEMM_NO_STREAM") || getenv("OVO_GEMM_TILE") || getenv("OVO_NO_LN_FOLD"); };
    static bool off = read();
    if (ovo_knobs_dynamic()) off = read();
    return off;
}

int gemm_f32a_stream(const ovo_gemm_t *p, const ovo_window_t *win, const float *x, int d, const float *gamma, const float *beta, float eps, int mode,
                     int pool2x2, ovo_stream_t stream, uint16_t *qpool_out, int qpool_cols) {
    if (!p || !x || p->in_dtype != 2 || p->M < 16384 || p->K > 256 || d <= 0 || d % 8 != 0 || d > p->K || (mode != 1 && mode != 2) ||
        (mode == 1 && (!gamma || !beta)) || ((uintptr_t)x & 15) != 0 || stream_off())
        return OVO_E_UNSUPPORTED;
    if (p->ldw % 8 != 0 || ((uintptr_t)p->W & 15) != 0 || (p->bias && ((uintptr_t)p->bias & 15) != 0) || p->add || p->N % 4 != 0 || p->K % 32 != 0 || p->act > 1)
        return OVO_E_UNSUPPORTED;
    GemmArgs g = {};
    g.A = nullptr; g.lda = 0; g.W = (const char *)p->W; g.ldw = p->ldw; g.bias = p->bias;
    g.C = p->C; g.ldc = p->ldc; g.add = nullptr; g.ld_add = 0;
    g.M = p->M; g.N = p->N; g.K = p->K; g.out_dtype = p->out_dtype; g.act = p->act; g.alpha = p->alpha;
    g.store = 1; g.rope_T = 1; g.rope_hd = 4;
    g.win_per = 0; g.win_ww = g.win_wh = g.win_nww = g.win_nwin = 1;
    if (win) {
        const int nwh = (win->H + win->wh - 1) / win->wh, nww = (win->W + win->ww - 1) / win->ww;
        if ((long long)win->B * nwh * nww * win->wh * win->ww != p->M) return OVO_E_UNSUPPORTED;
        g.win_per = win->wh * win->ww; g.win_ww = win->ww; g.win_wh = win->wh; g.win_nww = nww; g.win_nwin = nwh * nww; g.win_H = win->H; g.win_W = win->W;
    }
    g.ln_x = x; g.ln_g = gamma; g.ln_b = beta; g.ln_eps = eps; g.ln_d = d; g.ln_mode = mode;
    if (pool2x2) {      // whole even windows of width 2 / 4 / 8 tiling the grid exactly, f32 output, no activation between projection and pool
        if (!win || p->out_dtype != 0 || win->H % win->wh != 0 || win->W % win->ww != 0 || win->wh % 2 != 0 || (win->ww != 2 && win->ww != 4 && win->ww != 8) ||
            p->M % 16 != 0)
            return OVO_E_UNSUPPORTED;
        g.pool_ww = win->ww;
    }
    if (qpool_out) {    // pooled q: the same window conditions, bf16 output, the q columns a whole number of column groups, rows 8-byte aligned
        if (!win || p->out_dtype != 2 || win->H % win->wh != 0 || win->W % win->ww != 0 || win->wh % 2 != 0 || (win->ww != 2 && win->ww != 4 && win->ww != 8) ||
            p->M % 16 != 0 || qpool_cols <= 0 || qpool_cols % 4 != 0 || ((uintptr_t)qpool_out & 7) != 0 || pool2x2)
            return OVO_E_UNSUPPORTED;
        int ng = 0;                                                  // (the column-group width gemm_stream_launch will pick)
        for (int c : {256, 224, 112, 64, 32})
            if (p->N % c == 0) { ng = c; break; }
        if (p->K == 192 && p->N % 144 == 0 && p->N % 256 != 0) ng = p->N % 288 == 0 ? 288 : 144;
        if (p->K == 128 && p->N == 336) ng = 336;
        if (!ng || qpool_cols % ng != 0) return OVO_E_UNSUPPORTED;
        g.qpool_out = qpool_out; g.qpool_cols = qpool_cols;
    }
    const int rc = gemm_stream_launch(g, 2, (hipStream_t)stream);
    if (rc != OVO_OK) return rc;
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}

}  // namespace ovo_gemm_detail

extern "C" int ovo_gemm_f32a(const ovo_gemm_t *g, const ovo_window_t *win, const float *x, int d, const float *gamma, const float *beta, float eps,
                             int mode, int pool2x2, ovo_stream_t stream) {
    OVO_REQUIRE(g && x && g->W && g->C, "null pointer");
    return ovo_gemm_detail::gemm_f32a_stream(g, win, x, d, gamma, beta, eps, mode, pool2x2, stream);
}
