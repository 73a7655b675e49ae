// attention.hip -- fused softmax(Q K^T * scale) V for the ViT blocks and the Hiera (windowed / global) blocks.
//
// One workgroup = 4 waves; each wave owns QT q-tiles of 16 queries (QT = 1: 64 queries per workgroup, QT = 2: 128, used
// for long sequences so that every K / V^T fragment read from LDS feeds two MFMAs).  Keys/values are walked in tiles of
// 64: K tile row-major and V tile TRANSPOSED in LDS (padded rows, conflict-free fragment reads), online softmax kept in
// registers, the score matrix never materialised.
//   S^T tile  = mfma(a = K fragment [16 keys x 32 d], b = Q fragment [32 d x 16 queries])   -> lane owns one query
//   O^T tile  = mfma(a = V^T fragment [16 d x 32 keys], b = P fragment [32 keys x 16 queries])
// With this operand order a lane's accumulator registers all belong to the query (lane & 15), so the running
// max / sum / rescale are per-lane scalars plus two xor-shuffles, and the P fragment feeds the second MFMA
// straight from registers (the k index of an MFMA is free as long as a and b agree on it).
// Global loads of tile t+1 are issued into registers before tile t is multiplied and written to LDS after it
// (async-stage split), so HBM/L2 latency hides under the MFMAs even at one workgroup per CU.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct AttnArgs {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    long long q_sb, q_sh, q_st;   // element strides: batch, head, token (head_dim contiguous)
    long long k_sb, k_sh, k_st;
    long long v_sb, v_sh, v_st;
    long long o_sb, o_sh, o_st;
    int B, H, Tq, Tk, hd;
    float scale_log2e;
    int causal;                   // keys after the query are masked (text towers)
    int chunk, q_tiles;           // chunk > 0: 1-D grid in XCD-chunked order (see k_attention)
};

// float -> bf16 through the compiler's conversion (v_cvt_pk_bf16_f32 on gfx950: one instruction per pair, RNE)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return *(const uint32_t *)&h;
}

// xor-16 / xor-32 butterflies on VALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead of ds_bpermute round trips
// through the LDS crossbar: with one wave per SIMD nothing hides a ~100-cycle shuffle, and the online softmax has four per tile.
typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
__device__ __forceinline__ float max_xor16_32(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor16_32(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Online-softmax update for one query tile over one 64-key tile.  In: s = raw scores (lane: query lane & 15, keys 16 kt + 4 fq + r; masked
// entries -3e38).  Out: s = the probabilities exp2(score * scale_log2e - m_new) in fp32 (one FMA in front of v_exp_f32), m_run / l_run
// updated; returns the factor the caller's accumulators are rescaled by.  The running maximum is taken over the SCALED scores: a product
// is a canonical float, so the maxima compile to a v_max3_f32 chain -- on raw MFMA results every fmaxf operand was first quieted by a
// `v_max_f32 x, x, x` (96 extra VALU instructions per three q-tiles, a fifth of the softmax).  Products, FMAs and partial sums are packed
// (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two floats per lane and instruction).
template <bool TAIL>
__device__ __forceinline__ float softmax_tile(f32x4 (&s)[4], int nkt, float scale_log2e, float &m_run, float &l_run) {
    const f32x4 sc = f32x4{scale_log2e, scale_log2e, scale_log2e, scale_log2e};
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const f32x4 x = s[kt] * sc;
        mx = fmaxf(fmaxf(mx, x[0]), x[1]);                             // v_max3_f32
        mx = fmaxf(fmaxf(mx, x[2]), x[3]);
    }
    mx = max_xor16_32(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const f32x4 nm = f32x4{-m_new, -m_new, -m_new, -m_new};
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (!TAIL || kt < nkt) {
            s[kt] = __builtin_elementwise_fma(s[kt], sc, nm);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r]);     // v_exp_f32
            acc = acc + s[kt];
        } else {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};                     // sub-tile past the last key: p = +0 exactly, nothing exponentiated
        }
    }
    const float ps = sum_xor16_32((acc[0] + acc[1]) + (acc[2] + acc[3]));
    l_run = l_run * alpha + ps;
    m_run = m_new;
    return alpha;
}

#ifdef OVO_ATTN_TRACE
__device__ unsigned long long g_attn_trace[256];
#define ATTN_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0 && (i) < 256) g_attn_trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

template <int HD, int QT, bool TRIM = false>
__global__ void __launch_bounds__(256) k_attention(AttnArgs a) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int KROW = HD + 8;           // padded K-tile row (elements)
    constexpr int VB = HD / 16 + 1;        // V blocks per key group (+1 block of padding: bank spread)
    constexpr int CH = HD / 8;             // 16-byte chunks per head row
    constexpr int NLD = KT * CH / 256;     // 16-byte pieces per thread per tile (K and V each)
    constexpr int TILE_ELEMS = KT * KROW + (KT / 4) * VB * 64;
    constexpr bool DB = 2 * TILE_ELEMS * 2 <= 64 * 1024;      // double-buffered K/V tiles (one barrier per tile) when they fit
    __shared__ __attribute__((aligned(16))) uint16_t smem[(DB ? 2 : 1) * TILE_ELEMS];
    uint16_t *sK = smem, *sV = smem + KT * KROW;             // current tile (re-pointed per tile when DB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // Workgroups go round-robin over the 8 XCDs (XCD = linear id % 8), each with its own L2: XCD x takes the contiguous work
    // range [x * chunk, (x + 1) * chunk) in (batch*head major, q-tile minor) order, so the q-tiles that share one head's K / V
    // are co-resident on ONE L2 and K / V leave the fabric once instead of once per XCD.
    int qtile = blockIdx.x;
    int bh = blockIdx.y;
    if (a.chunk > 0) {
        const int work = (blockIdx.x & 7) * a.chunk + (blockIdx.x >> 3);
        if (work >= a.q_tiles * a.B * a.H) return;
        bh = work / a.q_tiles;
        qtile = work - bh * a.q_tiles;
    }
    const int b = bh / a.H, h = bh % a.H;
    const uint16_t *qp = a.q + b * a.q_sb + h * a.q_sh;
    const uint16_t *kp = a.k + b * a.k_sb + h * a.k_sh;
    const uint16_t *vp = a.v + b * a.v_sb + h * a.v_sh;

    // Q fragments: lane (query fr, group fq) holds d = ks*32 + fq*8 .. +8
    int q_row[QT];
    bf16x8 qf[QT][HD / 32];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        q_row[t] = (qtile * 4 + wave) * (16 * QT) + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            const int d0 = ks * 32 + fq * 8;
            if (q_row[t] < a.Tq && d0 < a.hd) raw = *(const uint4 *)(qp + (long long)q_row[t] * a.q_st + d0);
            qf[t][ks] = *(bf16x8 *)&raw;
        }
    }

    f32x4 oacc[QT][HD / 16];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -1.0e30f; l_run[t] = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 16; ++i) oacc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // two register sets: tiles t+1 and t+2 are in flight while tile t is multiplied (async-stage split, depth 2)
    uint4 k0r[NLD], v0r[NLD], k1r[NLD], v1r[NLD];
    // per-thread source offsets of the NLD pieces inside a tile (row, chunk); the tile base advances by a wave-uniform stride
    long long k_off[NLD], v_off[NLD];
    int f_row[NLD];
    bool f_ok[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int id = it * 256 + tid, row = id / CH, c = id % CH;
        f_row[it] = row; f_ok[it] = c * 8 < a.hd;
        k_off[it] = (long long)row * a.k_st + c * 8; v_off[it] = (long long)row * a.v_st + c * 8;
    }
    auto fetch = [&](uint4 (&kr)[NLD], uint4 (&vr)[NLD], int k0) {      // global -> registers, thread -> (key row, 16-byte chunk)
        const uint16_t *kt = kp + (long long)k0 * a.k_st, *vt = vp + (long long)k0 * a.v_st;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            kr[it] = make_uint4(0, 0, 0, 0);
            vr[it] = make_uint4(0, 0, 0, 0);
            if (f_ok[it] && k0 + f_row[it] < a.Tk) {
                kr[it] = *(const uint4 *)(kt + k_off[it]);
                vr[it] = *(const uint4 *)(vt + v_off[it]);
            }
        }
    };
    // registers -> LDS.  K row-major (padded rows).  V in [4 keys][16 d] blocks of 128 B (block (kg, dg) at (kg*VB + dg)*64
    // elements, VB = HD/16 + 1 keeps the four key groups of a fragment read on disjoint banks): written with plain 16-byte
    // stores, read back TRANSPOSED by ds_read_b64_tr_b16 -- no 2-byte scatter.
    auto commit = [&](const uint4 (&kr)[NLD], const uint4 (&vr)[NLD]) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int id = it * 256 + tid, row = id / CH, c = id % CH;
            *(uint4 *)(sK + row * KROW + c * 8) = kr[it];
            *(uint4 *)(sV + ((row >> 2) * VB + (c >> 1)) * 64 + (row & 3) * 16 + (c & 1) * 8) = vr[it];
        }
    };
    // TRIM (the instantiation for Tk <= 64: Hiera's 16-token windows and pooled 4 x 16 blocks, one ragged key tile): a wave whose
    // query rows all lie past Tq only helps to stage K / V, and 16-key sub-tiles past Tk are neither multiplied nor exponentiated --
    // results unchanged (their probabilities are +0), 126 -> 94 us on 8 frames' (16, 16) windows.  On multi-tile problems the same
    // tests cost more than they save (577 x 577: 69 -> 73 us), so the general instantiation carries none.
    // every instantiation: a wave with no query row below Tq skips the arithmetic (196-token windows: 3 of 16 waves; 73.8 -> 70.1 us)
    const bool wave_active = (qtile * 4 + wave) * (16 * QT) < a.Tq;
    auto compute = [&](int k0) {
        if (!wave_active) return;
        constexpr bool TAIL = TRIM;
        const int nkt = TAIL ? (a.Tk - k0 + 15) >> 4 : 4;           // 16-key sub-tiles of this tile that hold a key
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // ---- S^T = K Q^T : 4 tiles of 16 keys ----
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!TAIL || kt < nkt) {
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ++ks) {
                        const bf16x8 kf = *(const bf16x8 *)(sK + (kt * 16 + fr) * KROW + ks * 32 + fq * 8);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[kt], 0, 0, 0);
                    }
                }
            }
            ATTN_STAMP(100 + 4 * (k0 / KT));
            // ---- online softmax for query fr; this lane holds keys kt*16 + fq*4 + r ----
            if (k0 + KT > a.Tk) {                                   // wave-uniform: only the last tile has keys to mask
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + kt * 16 + fq * 4 + r >= a.Tk) s[kt][r] = -3.0e38f;
            }
            if (a.causal) {                                         // key 0 is never masked, so no row is ever all -inf in its first tile
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + kt * 16 + fq * 4 + r > q_row[t]) s[kt][r] = -3.0e38f;
            }
            const float alpha = softmax_tile<TAIL>(s, nkt, a.scale_log2e, m_run[t], l_run[t]);
            if (__any(alpha != 1.0f)) {                             // the running max settles after a few tiles
#pragma unroll
                for (int i = 0; i < HD / 16; ++i) {
                    oacc[t][i][0] *= alpha; oacc[t][i][1] *= alpha; oacc[t][i][2] *= alpha; oacc[t][i][3] *= alpha;
                }
            }
            ATTN_STAMP(101 + 4 * (k0 / KT));
            // ---- P fragments (bf16): element e of fragment kk <-> key (kk*2 + e/4)*16 + fq*4 + e%4 ----
            bf16x8 pf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint32_t tmp[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) tmp[e >> 1] = pack2(s[kk * 2 + (e >> 2)][e & 3], s[kk * 2 + (e >> 2)][(e & 3) + 1]);
                pf[kk] = *(bf16x8 *)tmp;
            }
            ATTN_STAMP(102 + 4 * (k0 / KT));
            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (TAIL && kk * 2 >= nkt) continue;            // both sub-tiles of this k-step hold no key: P = 0
                    // lane (d = dt*16 + fr, fq) needs V[keys (kk*2+h)*16 + fq*4 .. +4][d], h = 0,1: the transposing read of
                    // block (kg = (kk*2+h)*4 + fq, dg = dt) with every lane pointing at its own 8-byte slot of the block
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(sV + (((kk * 2) * 4 + fq) * VB + dt) * 64 + fr * 4));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(sV + (((kk * 2 + 1) * 4 + fq) * VB + dt) * 64 + fr * 4));
                    const uint2 l2 = *(const uint2 *)&lo, h2 = *(const uint2 *)&hi;
                    uint4 raw = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8 *)&raw, pf[kk], oacc[t][dt], 0, 0, 0);
                }
            }
        }
    };

    const int n_tiles = (a.Tk + KT - 1) / KT;
    ATTN_STAMP(0);
    fetch(k0r, v0r, 0);
    if (n_tiles > 1) fetch(k1r, v1r, KT);
    ATTN_STAMP(1);
    if (DB) {
        // two LDS buffers: tile t is multiplied out of buffer t & 1 while tile t+1 (fetched one iteration ago) is committed to
        // the other one and tile t+2 is fetched into the registers tile t just left -- ONE barrier per tile.
        commit(k0r, v0r);
        __syncthreads();
        for (int t = 0; t < n_tiles; t += 2) {
            ATTN_STAMP(2 + 4 * t);
            if (t + 2 < n_tiles) fetch(k0r, v0r, (t + 2) * KT);
            sK = smem; sV = smem + KT * KROW;
            ATTN_STAMP(4 + 4 * t);
            compute(t * KT);
            ATTN_STAMP(5 + 4 * t);
            if (t + 1 >= n_tiles) break;
            sK = smem + TILE_ELEMS; sV = sK + KT * KROW;
            commit(k1r, v1r);
            __syncthreads();
            ATTN_STAMP(6 + 4 * t);
            if (t + 3 < n_tiles) fetch(k1r, v1r, (t + 3) * KT);
            ATTN_STAMP(8 + 4 * t);
            compute((t + 1) * KT);
            ATTN_STAMP(9 + 4 * t);
            if (t + 2 < n_tiles) {
                sK = smem; sV = smem + KT * KROW;
                commit(k0r, v0r);
            }
            __syncthreads();
        }
    } else {
        for (int t = 0; t < n_tiles; t += 2) {
            __syncthreads();                                        // previous tile fully consumed
            commit(k0r, v0r);
            __syncthreads();
            if (t + 2 < n_tiles) fetch(k0r, v0r, (t + 2) * KT);
            compute(t * KT);
            if (t + 1 >= n_tiles) break;
            __syncthreads();
            commit(k1r, v1r);
            __syncthreads();
            if (t + 3 < n_tiles) fetch(k1r, v1r, (t + 3) * KT);
            compute((t + 1) * KT);
        }
    }
    ATTN_STAMP(250);
    // ---- store: lane holds O[q_row][dt*16 + fq*4 + r] ----
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (q_row[t] >= a.Tq) continue;
        const float inv = 1.0f / l_run[t];
        uint16_t *op = a.o + b * a.o_sb + h * a.o_sh + (long long)q_row[t] * a.o_st;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            const int d0 = dt * 16 + fq * 4;
            if (d0 < a.hd) {
                uint2 p;
                p.x = pack2(oacc[t][dt][0] * inv, oacc[t][dt][1] * inv);
                p.y = pack2(oacc[t][dt][2] * inv, oacc[t][dt][3] * inv);
                *(uint2 *)(op + d0) = p;
            }
        }
    }
}

// ---- tiny problems: Tq <= 64 and Tk <= 64 (Hiera's 8 x 8 / 4 x 4 windows and pooled blocks: tens of thousands of (window, head) pairs of
// 4-64 tokens).  The tiled kernel above gives each of them a whole 4-wave workgroup and a 64-key tile: 4-16 TFLOP/s, at half the HBM rate
// of their q / k / v / o stream.  Here ONE WAVE owns a (batch, head) pair: it stages its K / V rows in a private LDS slab sized for the
// (16-rounded) key count -- no workgroup barrier anywhere -- walks its 1-4 query tiles of 16, and the softmax is a single pass (one key
// tile: no running maximum).  Same MFMA operand scheme, same rounding points as k_attention (P in bf16, fp32 accumulation).
template <int HD>
__global__ void __launch_bounds__(256) k_attention_tiny(AttnArgs a, int krows) {
    constexpr int KROW = HD + 8, VB = HD / 16 + 1, CH = HD / 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t tsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const long long bh = (long long)blockIdx.x * 4 + wave;
    if (bh >= (long long)a.B * a.H) return;
    const int per_wave = krows * KROW + (krows / 4) * VB * 64;
    uint16_t *sK = tsm + wave * per_wave, *sV = sK + krows * KROW;
    const int b = (int)(bh / a.H), h = (int)(bh % a.H);
    const uint16_t *qp = a.q + b * a.q_sb + h * a.q_sh;
    const uint16_t *kp = a.k + b * a.k_sb + h * a.k_sh;
    const uint16_t *vp = a.v + b * a.v_sb + h * a.v_sh;
    // K row-major (padded rows), V in [4 keys][16 d] blocks read back transposed (as k_attention::commit)
    for (int id = lane; id < krows * CH; id += 64) {
        const int row = id / CH, c = id % CH;
        uint4 kr = make_uint4(0, 0, 0, 0), vr = make_uint4(0, 0, 0, 0);
        if (row < a.Tk && c * 8 < a.hd) {
            kr = *(const uint4 *)(kp + (long long)row * a.k_st + c * 8);
            vr = *(const uint4 *)(vp + (long long)row * a.v_st + c * 8);
        }
        *(uint4 *)(sK + row * KROW + c * 8) = kr;
        *(uint4 *)(sV + ((row >> 2) * VB + (c >> 1)) * 64 + (row & 3) * 16 + (c & 1) * 8) = vr;
    }
    __builtin_amdgcn_wave_barrier();
    const int nkt = krows >> 4;                                     // 16-key sub-tiles (1..4)
    for (int q0 = 0; q0 < a.Tq; q0 += 16) {
        const int q_row = q0 + fr;
        bf16x8 qf[HD / 32];
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            const int d0 = ks * 32 + fq * 8;
            if (q_row < a.Tq && d0 < a.hd) raw = *(const uint4 *)(qp + (long long)q_row * a.q_st + d0);
            qf[ks] = *(bf16x8 *)&raw;
        }
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
#pragma unroll
                for (int ks = 0; ks < HD / 32; ++ks) {
                    const bf16x8 kf = *(const bf16x8 *)(sK + (kt * 16 + fr) * KROW + ks * 32 + fq * 8);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kt * 16 + fq * 4 + r >= a.Tk || (a.causal && kt * 16 + fq * 4 + r > q_row)) s[kt][r] = -3.0e38f;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mx = fmaxf(fmaxf(mx, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        mx = max_xor16_32(mx);
        const float m = fmaxf(-1.0e30f, mx * a.scale_log2e);         // (k_attention's first tile: running max starts at -1e30)
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], a.scale_log2e, -m));
                    s[kt][r] = p;
                    ps += p;
                }
            } else s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ps = sum_xor16_32(ps);
        bf16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t tmp[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) tmp[e >> 1] = pack2(s[kk * 2 + (e >> 2)][e & 3], s[kk * 2 + (e >> 2)][(e & 3) + 1]);
            pf[kk] = *(bf16x8 *)tmp;
        }
        const float inv = 1.0f / ps;
        uint16_t *op = a.o + b * a.o_sb + h * a.o_sh + (long long)q_row * a.o_st;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk * 2 >= nkt) continue;
                const bool hi_ok = kk * 2 + 1 < nkt;                  // the slab holds only `krows` keys
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(sV + (((kk * 2) * 4 + fq) * VB + dt) * 64 + fr * 4));
                s16x4 hi = s16x4{0, 0, 0, 0};
                if (hi_ok) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(sV + (((kk * 2 + 1) * 4 + fq) * VB + dt) * 64 + fr * 4));
                const uint2 l2 = *(const uint2 *)&lo, h2 = *(const uint2 *)&hi;
                uint4 raw = make_uint4(l2.x, l2.y, h2.x, h2.y);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8 *)&raw, pf[kk], o, 0, 0, 0);
            }
            const int d0 = dt * 16 + fq * 4;
            if (q_row < a.Tq && d0 < a.hd) {
                uint2 pk;
                pk.x = pack2(o[0] * inv, o[1] * inv);
                pk.y = pack2(o[2] * inv, o[3] * inv);
                *(uint2 *)(op + d0) = pk;
            }
        }
    }
}


// ---- whole-head-resident form: Tk <= 592 keys of head_dim <= 64 (the ViT-L/14-336 blocks: 577 tokens; Hiera's 14 x 14 windows: 196).
// k_attention gives every 64 queries a workgroup that walks K / V in tiles -- fetch into registers, commit to LDS, one barrier per tile,
// one q-tile per wave (every K / V^T fragment read from LDS feeds ONE MFMA: the LDS pipe is as busy as the matrix pipe) -- and measured
// ~1600 cycles per (16 queries x 64 keys) on a SIMD whose MFMAs need 256.  Here K and V of one (batch, head) pair are copied into LDS
// ONCE (K: 128-byte rows, 16-byte chunk index XORed with (row >> 1) & 7; V: the [4 keys][16 d] blocks of k_attention with the block's
// d-group XORed with the key group & 3 -- no padding, 2 x 74 KB at 577 keys), one barrier, and then every wave walks all key tiles for
// its OWN 2-3 q-tiles out of read-only LDS with no further synchronisation: a fragment feeds 2-3 MFMAs, nothing is handed over between
// tiles, and the two waves of a SIMD cover each other's softmax with MFMAs.  A workgroup takes a contiguous range of a head's q-tiles
// (`splits` ranges per head: 577 queries = 37 q-tiles = 19 + 18, eight waves x (3, 3, 3, 2, 2, 2, 2, 2) / (3, 3, 2, ...)); the same
// arithmetic in the same order as k_attention -- bit-identical results (test_attention_resident_kernel_equals_tiled_kernel).
typedef __attribute__((address_space(3))) char lds_char;

// S^T of QT q-tiles against one 64-key tile (TAIL: the last, ragged one -- sub-tiles past the last key are not multiplied, keys past it masked)
template <int QT, bool TAIL>
__device__ __forceinline__ void resident_qk(const AttnArgs &a, const lds_char *sKt, int k0, const int (&kb)[2], const bf16x8 (&qf)[QT][2],
                                            f32x4 (&s)[QT][4], int fq) {
    const int nkt = TAIL ? (a.Tk - k0 + 15) >> 4 : 4;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!TAIL || kt < nkt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = *(const __attribute__((address_space(3))) bf16x8 *)(sKt + kt * (16 * 128) + kb[ks]);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[t][kt], 0, 0, 0);
            }
        }
    }
    if (TAIL) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + kt * 16 + fq * 4 + r >= a.Tk) s[t][kt][r] = -3.0e38f;
    }
}

// online softmax of those scores and O^T += V^T P^T
template <int QT, bool TAIL>
__device__ __forceinline__ void resident_pv(const AttnArgs &a, const lds_char *sVt, int k0, const int (&vb)[4], f32x4 (&s)[QT][4],
                                            f32x4 (&oacc)[QT][4], float (&m_run)[QT], float (&l_run)[QT]) {
    const int nkt = TAIL ? (a.Tk - k0 + 15) >> 4 : 4;
    bf16x8 pf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float alpha = softmax_tile<TAIL>(s[t], nkt, a.scale_log2e, m_run[t], l_run[t]);
        if (__any(alpha != 1.0f)) {                                 // the running max settles after a few tiles
#pragma unroll
            for (int i = 0; i < 4; ++i) oacc[t][i] = oacc[t][i] * alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t tmp[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) tmp[e >> 1] = pack2(s[t][kk * 2 + (e >> 2)][e & 3], s[t][kk * 2 + (e >> 2)][(e & 3) + 1]);
            pf[t][kk] = *(bf16x8 *)tmp;
        }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (TAIL && kk * 2 >= nkt) continue;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(sVt + (kk * 2) * (4 * 512) + vb[dt]));
            s16x4 hi = s16x4{0, 0, 0, 0};
            if (!TAIL || kk * 2 + 1 < nkt)
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(sVt + (kk * 2 + 1) * (4 * 512) + vb[dt]));
            const uint2 l2 = *(const uint2 *)&lo, h2 = *(const uint2 *)&hi;
            uint4 raw = make_uint4(l2.x, l2.y, h2.x, h2.y);
#pragma unroll
            for (int t = 0; t < QT; ++t) oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8 *)&raw, pf[t][kk], oacc[t][dt], 0, 0, 0);
        }
    }
}

// One wave, QT q-tiles from q-tile qt0, all key tiles out of the resident K / V.
template <int QT>
__device__ __forceinline__ void resident_wave(const AttnArgs &a, const lds_char *sK, const lds_char *sV, const uint16_t *qp, uint16_t *obase, int qt0, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    int q_row[QT];
    bf16x8 qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        q_row[t] = (qt0 + t) * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            const int d0 = ks * 32 + fq * 8;
            if (q_row[t] < a.Tq && d0 < a.hd) raw = *(const uint4 *)(qp + (long long)q_row[t] * a.q_st + d0);
            qf[t][ks] = *(bf16x8 *)&raw;
        }
    }
    f32x4 oacc[QT][4];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -1.0e30f; l_run[t] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // lane parts of the fragment addresses (bytes): K row (16 kt + fr) chunk (4 ks + fq) ^ (fr >> 1); V block (key group 4 x + fq, d-group dt ^ fq) + 8 fr
    int kb[2], vb[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kb[ks] = fr * 128 + (((ks * 4 + fq) ^ (fr >> 1)) << 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vb[dt] = (fq * 4 + (dt ^ fq)) * 128 + fr * 8;
    const int full = a.Tk >> 6;
    for (int tile = 0; tile < full; ++tile) {
        f32x4 s[QT][4];
        resident_qk<QT, false>(a, sK + tile * (64 * 128), tile * 64, kb, qf, s, fq);
        resident_pv<QT, false>(a, sV + tile * (16 * 512), tile * 64, vb, s, oacc, m_run, l_run);
    }
    if (a.Tk & 63) {
        f32x4 s[QT][4];
        resident_qk<QT, true>(a, sK + full * (64 * 128), full * 64, kb, qf, s, fq);
        resident_pv<QT, true>(a, sV + full * (16 * 512), full * 64, vb, s, oacc, m_run, l_run);
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (q_row[t] >= a.Tq) continue;
        const float inv = 1.0f / l_run[t];
        uint16_t *op = obase + (long long)q_row[t] * a.o_st;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int d0 = dt * 16 + fq * 4;
            if (d0 < a.hd) {
                uint2 p;
                p.x = pack2(oacc[t][dt][0] * inv, oacc[t][dt][1] * inv);
                p.y = pack2(oacc[t][dt][2] * inv, oacc[t][dt][3] * inv);
                *(uint2 *)(op + d0) = p;
            }
        }
    }
}

// a.q_tiles = q-tiles (of 16) per head, a.chunk = work items per XCD (work item = (head, split)); `splits` ranges of q-tiles per head
__global__ void __launch_bounds__(512) k_attention_resident(AttnArgs a, int splits, int krows) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    lds_char *sK = (lds_char *)rsm, *sV = sK + krows * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int work = (blockIdx.x & 7) * a.chunk + (blockIdx.x >> 3);          // items of one head next to each other on ONE XCD (its L2 serves K / V twice)
    if (work >= a.B * a.H * splits) return;
    const int bh = work / splits, split = work - bh * splits;
    const int b = bh / a.H, h = bh % a.H;
    const uint16_t *qp = a.q + b * a.q_sb + h * a.q_sh;
    const uint16_t *kp = a.k + b * a.k_sb + h * a.k_sh;
    const uint16_t *vp = a.v + b * a.v_sb + h * a.v_sh;
    // (measured alternatives, tools/attn_bench.py, 24 x 16 heads x 577^2: this copy through registers 81 us; the same image brought in by
    // LDS-DMA in tile order and consumed behind counted waits + one barrier per key tile 100 us -- waves in lock-step multiply and
    // exponentiate at the same time, the drift between them IS the overlap; DMA with one wait 85 us)
    const int pieces = krows * 8;
    for (int id0 = tid; id0 < pieces; id0 += 4 * blockDim.x) {                 // four 16-byte pieces of K and of V in flight per thread
        uint4 kr[4], vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = id0 + u * blockDim.x, row = id >> 3, c = id & 7;
            kr[u] = make_uint4(0, 0, 0, 0); vr[u] = make_uint4(0, 0, 0, 0);
            if (id < pieces && row < a.Tk && c * 8 < a.hd) {
                kr[u] = *(const uint4 *)(kp + (long long)row * a.k_st + c * 8);
                vr[u] = *(const uint4 *)(vp + (long long)row * a.v_st + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = id0 + u * blockDim.x, row = id >> 3, c = id & 7, kg = row >> 2;
            if (id < pieces) {
                *(uint4 *)(rsm + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kr[u];
                *(uint4 *)(rsm + krows * 128 + (kg * 4 + ((c >> 1) ^ (kg & 3))) * 128 + (row & 3) * 32 + (c & 1) * 16) = vr[u];
            }
        }
    }
    __syncthreads();
    // this workgroup's q-tiles [t0, t0 + nt), dealt to the waves as contiguous runs of 2-3 (the first `rem` waves take one more)
    const int per = a.q_tiles / splits, extra = a.q_tiles % splits;
    const int t0 = split * per + (split < extra ? split : extra), nt = per + (split < extra ? 1 : 0);
    const int base = nt / n_waves, rem = nt % n_waves;
    int mine = base + (wave < rem ? 1 : 0), first = t0 + wave * base + (wave < rem ? wave : rem);
    uint16_t *obase = a.o + b * a.o_sb + h * a.o_sh;
    while (mine > 0) {                                                          // (more than 3 only when a split holds more than 3 q-tiles per wave)
        if (mine >= 3) { resident_wave<3>(a, sK, sV, qp, obase, first, lane); first += 3; mine -= 3; }
        else if (mine == 2) { resident_wave<2>(a, sK, sV, qp, obase, first, lane); first += 2; mine -= 2; }
        else { resident_wave<1>(a, sK, sV, qp, obase, first, lane); first += 1; mine -= 1; }
    }
}

}  // namespace

extern "C" int ovo_attention(const ovo_attention_t *p, ovo_stream_t stream) {
    OVO_REQUIRE(p && p->q && p->k && p->v && p->o, "null pointer");
    OVO_REQUIRE(p->B > 0 && p->H > 0 && p->Tq > 0 && p->Tk > 0, "bad shape");
    OVO_REQUIRE(p->hd > 0 && p->hd <= 128 && p->hd % 8 == 0, "head_dim must be a multiple of 8, <= 128");
    const long long strides[12] = {p->q_sb, p->q_sh, p->q_st, p->k_sb, p->k_sh, p->k_st, p->v_sb, p->v_sh, p->v_st, p->o_sb, p->o_sh, p->o_st};
    for (int i = 0; i < 12; ++i) OVO_REQUIRE(strides[i] % 4 == 0, "strides must be multiples of 4 elements");
    OVO_REQUIRE(p->q_st % 8 == 0 && p->k_st % 8 == 0 && p->v_st % 8 == 0 && p->q_sh % 8 == 0 && p->k_sh % 8 == 0 && p->v_sh % 8 == 0 &&
                p->q_sb % 8 == 0 && p->k_sb % 8 == 0 && p->v_sb % 8 == 0, "q/k/v rows must be 16-byte aligned");
    OVO_REQUIRE(((((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v) & 15) == 0) && (((uintptr_t)p->o & 7) == 0), "misaligned base pointer");
    OVO_REQUIRE((long long)p->B * p->H * ((p->Tq + 63) / 64) < (1ll << 31) - 8, "too many workgroups");
    AttnArgs a;
    a.q = (const uint16_t *)p->q; a.k = (const uint16_t *)p->k; a.v = (const uint16_t *)p->v; a.o = (uint16_t *)p->o;
    a.q_sb = p->q_sb; a.q_sh = p->q_sh; a.q_st = p->q_st; a.k_sb = p->k_sb; a.k_sh = p->k_sh; a.k_st = p->k_st;
    a.v_sb = p->v_sb; a.v_sh = p->v_sh; a.v_st = p->v_st; a.o_sb = p->o_sb; a.o_sh = p->o_sh; a.o_st = p->o_st;
    a.B = p->B; a.H = p->H; a.Tq = p->Tq; a.Tk = p->Tk; a.hd = p->hd;
    a.scale_log2e = p->scale * 1.4426950408889634f;
    a.causal = p->causal;
    hipStream_t s = (hipStream_t)stream;
    // two q-tiles per wave once there are enough 128-query workgroups to fill the chip
    // (tools/attn_bench.py: 128-query workgroups win once there are >= 2 of them per CU; below that the 64-query form's
    // 3 waves/SIMD hide more latency: 4096 x 4096, 8 heads, hd 56: 100 us narrow vs 114 us wide)
    // -- and only for head_dim > 64: the 128-query form needs 198 VGPRs at head_dim 64 (2 waves per SIMD) against 128 (3-4) for the
    // 64-query form, and with several frames per launch there are always enough workgroups; tools/attn_bench.py, round 2:
    // 8 x 16 heads x 577^2 (four keyframes' ViT crops) 59.8 us wide vs 37.4 narrow, 4 x 8 x 4096^2 x 56 394 vs 292, while
    // 8 x 16 x 2048^2 x 128 stays 703 wide vs 849 narrow
    // -- and for Hiera's 14 x 14 windows (196 queries and keys): two 128-query workgroups per (window, head) stage K / V twice instead of
    // four times (64 + 64 + 64 + 4 queries): 12 frames' stage-3 windows 118 -> 104 us (tools/attn_bench.py, round 3)
    const bool wide = (getenv("OVO_ATTN_WIDE") != nullptr) ||
                      (!getenv("OVO_ATTN_NARROW") && ((p->hd > 64 && p->Tq >= 512 && (long long)((p->Tq + 127) / 128) * p->B * p->H >= 512) ||
                                                      (p->Tq > 128 && p->Tq <= 256 && p->Tk <= 256 && (long long)p->B * p->H >= 512) ||
                                                      // head_dim <= 64, long sequences: 12 frames' Hiera global blocks (96 x 4096^2 x 56) 730 us wide vs 772 narrow,
                                                      // four frames' (32 pairs: 1024 wide workgroups) 394 vs 292 -- wide from ten 128-query workgroups per CU
                                                      (p->hd <= 64 && p->Tq >= 2048 && (long long)((p->Tq + 127) / 128) * p->B * p->H >= 2560)));
    const int qpb = wide ? 128 : 64;
    dim3 grid((p->Tq + qpb - 1) / qpb, p->B * p->H);
    a.q_tiles = (int)grid.x; a.chunk = 0;
    // (also the form for more than 65535 batch x head rows -- windowed attention of several frames at once: the y dimension of a grid ends there)
    if ((grid.x > 1 && !getenv("OVO_ATTN_NO_CHUNK")) || grid.y > 65535) {   // several q-tiles share a head's K / V: keep them on one XCD's L2
        const long long total = (long long)grid.x * grid.y;
        a.chunk = (int)((total + 7) / 8);
        grid = dim3((unsigned)(a.chunk * 8), 1);
    }
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(1, 4.0 * p->B * p->H * (double)p->Tq * p->Tk * p->hd, s); ovo_prof_shape(p->B * p->H, p->Tq, p->Tk); }
    struct Done { bool on; hipStream_t s; ~Done() { if (on) ovo_prof_end(s); } } done{prof, s};
    // 65-592 keys of head_dim <= 64, not causal: K / V of a head resident in LDS (k_attention_resident)
    if (p->hd <= 64 && p->Tk > 64 && p->Tk <= 592 && !p->causal && !getenv("OVO_ATTN_NO_RESIDENT")) {
        const int krows = (p->Tk + 15) & ~15, nq = (p->Tq + 15) / 16;
        const size_t lds = (size_t)krows * 256;
        // one 8-wave workgroup per CU above half the LDS; below it 4-wave workgroups, so that one loads while another multiplies
        static const int force_threads = getenv("OVO_ATTN_RES_THREADS") ? atoi(getenv("OVO_ATTN_RES_THREADS")) : 0;    // tools/attn_bench.py
        static const int force_splits = getenv("OVO_ATTN_RES_SPLITS") ? atoi(getenv("OVO_ATTN_RES_SPLITS")) : 0;
        const int threads = force_threads ? force_threads : (lds > 80 * 1024 ? 512 : 256), waves = threads / 64;
        // at most 4 q-tiles per wave (a pass of 3 and one of 1: the 13 q-tiles of a 14 x 14 window on four waves, K / V staged once -- 84 us
        // against 94 as two workgroups of 7 + 6; tools/attn_variants.py) ...
        int splits = (nq + 4 * waves - 1) / (4 * waves);
        const long long heads = (long long)p->B * p->H;
        while (heads * splits < 768 && (splits + 1) * 2 * waves <= nq) ++splits;   // ... and 3+ workgroups per CU while every wave keeps 2 q-tiles
        if (force_splits) splits = force_splits;
        static bool attr_done = false;
        if (heads * splits < 192) goto tiled;                             // too few workgroups for the chip (one frame's two crops: 32 heads): the 64-query tiled form has 10x more
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void *)k_attention_resident, hipFuncAttributeMaxDynamicSharedMemorySize, 592 * 256);
            if (e != hipSuccess) { ovo_set_error("ovo_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return OVO_E_LAUNCH; }
            attr_done = true;
        }
        const long long items = heads * splits;
        a.q_tiles = nq; a.chunk = (int)((items + 7) / 8);
        k_attention_resident<<<(unsigned)(a.chunk * 8), threads, lds, s>>>(a, splits, krows);
        OVO_CHECK_LAUNCH();
        return OVO_OK;
    }
tiled:
    // tiny problems (one key tile, at most four query tiles): one wave per (batch, head) pair
    // (tools/attn_bench.py, 12 frames of hiera_b+: 16 x 16 windows 165 -> 79 us = 4.5 TB/s of q/k/v/o, pooled 4 x 16 blocks 321 -> 89 us;
    //  with 2-4 query tiles per pair -- 64 x 64, 49 x 49 -- the tiled kernel's four waves per pair are ahead: 180 vs 201 us, 33 vs 37)
    if (p->hd <= 64 && p->Tk <= 64 && p->Tq <= 16 && !getenv("OVO_ATTN_NO_TINY")) {
        const int krows = (p->Tk + 15) & ~15;
        const size_t lds = 4 * (size_t)(krows * (64 + 8) + (krows / 4) * (64 / 16 + 1) * 64) * 2;
        const long long nb = ((long long)p->B * p->H + 3) / 4;
        k_attention_tiny<64><<<(unsigned)nb, 256, lds, s>>>(a, krows);
        OVO_CHECK_LAUNCH();
        return OVO_OK;
    }
#define GO(HD)                                                       \
    do {                                                             \
        if (wide) k_attention<HD, 2><<<grid, 256, 0, s>>>(a);        \
        else k_attention<HD, 1><<<grid, 256, 0, s>>>(a);             \
    } while (0)
    // (the trimmed form for a LAST tile that is mostly padding -- 196 keys = 3 tiles + 4 keys -- measured no gain: 117 vs 118 us)
    if (p->hd <= 64 && p->Tk <= 64 && !wide) k_attention<64, 1, true><<<grid, 256, 0, s>>>(a);     // one ragged key tile: the trimmed form
    else if (p->hd <= 64) GO(64);
    else if (p->hd <= 96) GO(96);
    else GO(128);
#undef GO
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
