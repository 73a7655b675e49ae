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

// ---- online softmax, restated for the instruction mix of gfx950 (tools/ubench.hip: a lone wave issues one VALU instruction per ~8 cycles, two waves of
// a SIMD one per ~4; v_exp_f32 ~9; beside a stream of MFMAs every VALU instruction still costs 1-4 cycles of the SIMD).  Round 3's form spent ~80 VALU
// instructions per (16 queries x 64 keys) on a tile whose 16 MFMAs need 264 cycles.  Here:
//   * a score leaves the MFMA in log2 units: the encoders fold 1/sqrt(hd) * log2 e into the q rows of their QKV weights at load time (in fp32,
//     before the bf16 rounding; ovo_attention_t.scale == 0 -> scale_log2e == 1 here, no factor applied).  A caller that passes a scale instead gets
//     `prescale_q`: the bf16 Q fragments multiplied and ROUNDED AGAIN -- 2e-3 max output error at unit-variance q / k against 4e-7;
//   * the running reference maximum m of a query enters the QK^T product as its C operand (`negm` = four copies of -m, re-used by every first
//     k-step of the tile -- srcC and vdst of an MFMA are separate registers), so the MFMA delivers S' = S - m: no subtraction pass;
//   * FAST PATH (every tile after the first, unless some score of the wave outgrew the reference by more than THR): p = exp2(S') straight from the
//     accumulators -- per 16 scores of a lane 8 v_max3 + one compare/branch, 16 v_exp, 8 packed adds (row sum, kept PER LANE: the cross-lane sum
//     happens once, after the last tile), 8 v_cvt_pk.  No cross-lane operation at all;
//   * SLOW PATH (first tile; or a lane saw S' > THR): the classic update -- row maximum across the four lanes of a query, d = max(rowmax', 0)
//     (first tile: d = rowmax'), S' -= d, m += d, O and the row sum scaled by exp2(-d).  Probabilities stay <= 2^THR between rescales (bf16 /
//     f32 have the exponent range; the relative rounding of P does not depend on its scale).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
constexpr float ATTN_THR = 6.0f;

__device__ __forceinline__ float lane_max16(const f32x4 (&s)[4]) {           // 8 v_max3_f32 (MFMA results: no canonicalising v_max needed)
    float m = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
    m = fmaxf(fmaxf(m, s[0][3]), s[1][0]);
    m = fmaxf(fmaxf(m, s[1][1]), s[1][2]);
    m = fmaxf(fmaxf(m, s[1][3]), s[2][0]);
    m = fmaxf(fmaxf(m, s[2][1]), s[2][2]);
    m = fmaxf(fmaxf(m, s[2][3]), s[3][0]);
    m = fmaxf(fmaxf(m, s[3][1]), s[3][2]);
    return fmaxf(m, s[3][3]);
}

// Q fragment times scale * log2(e), rounded to bf16 again (callers that folded the factor into the projection pass 1 and skip this)
__device__ __forceinline__ bf16x8 prescale_q(bf16x8 q, float c) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const bf16x2 h = __builtin_convertvector(f32x2{(float)q[e] * c, (float)q[e + 1] * c}, bf16x2);
        r[e] = h[0]; r[e + 1] = h[1];
    }
    return r;
}

// slow path of one q-tile: s = S - m_old of this tile (masked entries -3e38), lm = the lane's maximum of them
template <bool FIRST, int NO>
__device__ __forceinline__ void softmax_rescale(f32x4 (&s)[4], float lm, f32x4 &negm, f32x4 &lsum, f32x4 (&oacc)[NO]) {
    const float rm = max_xor16_32(lm);
    const float d = FIRST ? rm : fmaxf(rm, 0.f);
    const f32x4 d4 = f32x4{d, d, d, d};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s[kt] = s[kt] - d4;
    negm = negm - d4;
    if (!FIRST) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        lsum = lsum * alpha;
#pragma unroll
        for (int i = 0; i < NO; ++i) oacc[i] = oacc[i] * alpha;
    }
}

// p = exp2(s) in place, row-sum partials, packed bf16 P fragments (element e of fragment kk <-> key (kk*2 + e/4)*16 + fq*4 + e%4)
template <bool TAIL>
__device__ __forceinline__ void softmax_exp(f32x4 (&s)[4], int nkt, f32x4 &lsum, bf16x8 (&pf)[2]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (!TAIL || kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r]);     // v_exp_f32
            lsum = lsum + s[kt];
        } else {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};                     // sub-tile past the last key: p = +0 exactly, nothing exponentiated
        }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        uint32_t tmp[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) tmp[e >> 1] = pack2(s[kk * 2 + (e >> 2)][e & 3], s[kk * 2 + (e >> 2)][(e & 3) + 1]);
        pf[kk] = *(bf16x8 *)tmp;
    }
}

__device__ __forceinline__ float row_total(const f32x4 &lsum) { return sum_xor16_32((lsum[0] + lsum[1]) + (lsum[2] + lsum[3])); }

#ifdef OVO_ATTN_TRACE
__device__ unsigned long long g_attn_trace[256];
#define ATTN_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0 && (i) < 256) g_attn_trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

template <int HD, int QT, bool TRIM = false>
__global__ void __launch_bounds__(256, (HD <= 64 && QT == 2) ? 2 : 1) k_attention(AttnArgs a) {
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
            if (a.scale_log2e != 1.0f) qf[t][ks] = prescale_q(qf[t][ks], a.scale_log2e);
        }
    }

    f32x4 oacc[QT][HD / 16];
    f32x4 negm[QT], lsum[QT];                 // -m (four copies: the C operand of the first k-step) and the lane's part of the row sum
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        negm[t] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            // ---- S^T - m = K Q^T - m : 4 tiles of 16 keys, the reference maximum as the C operand of the first k-step ----
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = negm[t];
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
            const float lm = lane_max16(s);
            if (k0 == 0) softmax_rescale<true>(s, lm, negm[t], lsum[t], oacc[t]);
            else if (__any(lm > ATTN_THR)) softmax_rescale<false>(s, lm, negm[t], lsum[t], oacc[t]);
            ATTN_STAMP(101 + 4 * (k0 / KT));
            bf16x8 pf[2];
            softmax_exp<TAIL>(s, nkt, lsum[t], pf);
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
        const float inv = 1.0f / row_total(lsum[t]);               // (cross-lane: before the divergent exit)
        if (q_row[t] >= a.Tq) continue;
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
            if (a.scale_log2e != 1.0f) qf[ks] = prescale_q(qf[ks], a.scale_log2e);
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
        // one key tile: k_attention's first-tile step (same arithmetic, same order)
        f32x4 negm = f32x4{0.f, 0.f, 0.f, 0.f}, lsum = f32x4{0.f, 0.f, 0.f, 0.f}, none[1];
        softmax_rescale<true>(s, lane_max16(s), negm, lsum, none);
        bf16x8 pf[2];
        softmax_exp<true>(s, nkt, lsum, pf);
        const float ps = row_total(lsum);
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


// ---- k_attention32: head_dim <= 64, any sequence length.  One wave = 32 queries on v_mfma_f32_32x32x16_bf16 (a 32-cycle instruction hides
// ~5 VALU issues; tools/ubench.hip), one workgroup = 4 waves = 128 queries, 2-3 workgroups per CU drifting against each other (MFMA phases of
// one beside the softmax of another).  K / V walk through LDS in 64-key tiles, double-buffered, ONE barrier per tile: tile t+1 is written after
// the barrier that opens tile t (its global loads were issued one tile earlier and re-issued for t+2 straight after the write: one register set).
//   S^T[32 keys x 32 q]  = mfma(a = K fragment [32 keys x 16 d], b = Q^T fragment [16 d x 32 q], c = -m)   lane: q = lane & 31, 16 keys per key block
//   O^T[32 d x 32 q]    += mfma(a = V^T fragment [32 d x 16 keys], b = P^T fragment [16 keys x 32 q])
// The k index of the second product is free as long as both operands agree on it: element e of k-step j of key block kb stands for key
// 32 kb + 16 j + 8 (e >> 2) + 4 hi + (e & 3) (hi = lane >> 5) -- exactly the keys of S accumulators 8 j .. 8 j + 7, so P goes from the
// exponentials into the B operand with a v_cvt_pk_bf16_f32 per pair and no cross-lane move; the V^T fragment is two transposing reads
// (ds_read_b64_tr_b16) of [4 keys][16 d] blocks.  Softmax: the fast / slow path scheme above (reference maximum as the C operand).
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool FIRST, bool ONES, int DB>
__device__ __forceinline__ void rescale32(f32x16 (&s)[2], float lm, f32x16 &negm, f32x4 &lsum, f32x16 (&o)[DB]) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lm), __float_as_uint(lm), false, false);
    const float rm = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const float d = FIRST ? rm : fmaxf(rm, 0.f);
    s[0] = s[0] - d; s[1] = s[1] - d;
    negm = negm - d;
    if (!FIRST) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        if (!ONES) lsum = lsum * alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = o[db] * alpha;
    }
}

__device__ __forceinline__ float lane_max32(const f32x16 &x, const f32x16 &y) {       // two independent v_max3 chains
    float m = fmaxf(fmaxf(x[0], x[1]), x[2]), n = fmaxf(fmaxf(y[0], y[1]), y[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) { m = fmaxf(fmaxf(m, x[i]), x[i + 1]); n = fmaxf(fmaxf(n, y[i]), y[i + 1]); }
    return fmaxf(fmaxf(m, x[15]), fmaxf(n, y[15]));
}

// (measured and dropped: a software-pipelined form -- tile t + 1's score products issued between tile t's exponentials, its lane maximum under tile t's
//  P V products, K one tile ahead of V through the same two buffers, two score sets alternating roles; bit-identical outputs, 231 VGPRs = 2 waves per
//  SIMD: 4096^2 x 56 457 -> 450 us, but 577^2 x 64 59.5 -> 71.8 and 196^2 windows 58.7 -> 70.0.  tools/ubench.hip says why: a 32 x 32 x 16 MFMA with its
//  A operand read from LDS plus this kernel's VALU mix (max3, 2 exp, cvt, fma per product) costs 50 cycles per SIMD at two waves and 45 with one LDS
//  read per two products, against 35 for the bare product -- the VALU issues extend the product's slot whichever wave they come from, so overlapping
//  them inside one wave buys what the three drifting waves per SIMD already get)
// ONES (head_dim <= 56, e.g. Hiera's 56): the zero padding of V up to 64 columns carries a column of ones at d = 56, so the row sum of the
// (bf16-rounded) probabilities comes out of the P V product as O^T row 56 -- no add per score, no separate accumulator to rescale.
// (measured and dropped: a 128-register build of the head_dim <= 56 form for 4 waves per SIMD -- 17 spilled registers inside the loop: 489 -> 577 us)
template <int NW, int KS, int DB, bool ONES>       // NW: waves per workgroup; KS: 16-wide k-steps of Q K^T (head_dim <= 16 KS); DB: 32-row blocks of O^T (head_dim <= 32 DB)
__global__ void __launch_bounds__(NW * 64, (NW == 4 && KS == 4) ? 3 : 2) k_attention32(AttnArgs a) {
    constexpr int NT = NW * 64;
    constexpr int KCH = 2 * KS, VCH = 4 * DB;                    // 16-byte chunks of a K row / a V row in LDS
    // K rows: 128 bytes with the chunk index XOR-swizzled (head_dim <= 64), else 32 KS + 16 bytes (the 16-byte pad spreads the 16 rows of a
    // ds_read_b128 lane group over all 64 banks: rows are -12 resp. +4 dwords apart mod 64); V: [key group][d-group][4 keys][16 d] blocks of 128 bytes
    constexpr int KRB = KS == 4 ? 128 : 32 * KS + 16, DGN = 2 * DB;
    constexpr int KT = 64, KBYTES = KT * KRB, TILE = KBYTES + KT * VCH * 16;
    constexpr int NLDK = (KT * KCH + NT - 1) / NT, NLDV = (KT * VCH + NT - 1) / NT;      // 16-byte pieces of K / of V per thread and tile
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int work = (blockIdx.x & 7) * a.chunk + (blockIdx.x >> 3);       // XCD-chunked: the q-blocks of one head next to each other on one L2
    if (work >= a.q_tiles * a.B * a.H) return;
    const int bh = work / a.q_tiles, qblk = work - bh * a.q_tiles;
    const int b = bh / a.H, h = bh % a.H;
    const uint16_t *qp = a.q + b * a.q_sb + h * a.q_sh;
    const uint16_t *kp = a.k + b * a.k_sb + h * a.k_sh;
    const uint16_t *vp = a.v + b * a.v_sb + h * a.v_sh;

    const int q_row = (qblk * NW + wave) * 32 + ql;
    bf16x8 qf[KS];                                               // B operand of k-step ks: d = 16 ks + 8 hi .. + 8
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 raw = make_uint4(0, 0, 0, 0);
        const int d0 = ks * 16 + hi * 8;
        if (q_row < a.Tq && d0 < a.hd) raw = *(const uint4 *)(qp + (long long)q_row * a.q_st + d0);
        qf[ks] = *(bf16x8 *)&raw;
        if (a.scale_log2e != 1.0f) qf[ks] = prescale_q(qf[ks], a.scale_log2e);
    }
    f32x16 o[DB], negm;
    f32x4 lsum = f32x4{0.f, 0.f, 0.f, 0.f};                      // the lane's part of the row sum, folded to four partial sums per tile (unused with ONES)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        negm[i] = 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db][i] = 0.f;
    }

    // staging through buffer loads: the lane's byte offset inside a tile in voffset, the tile's in soffset; a piece that must read zeros (chunk
    // past head_dim, row past the last key, piece index past the tile) gets an offset past the resource's extent -- the load returns 0 without a branch
    constexpr uint32_t OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)kp, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void *)vp, 0, 0x7ffffff0, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 kr[NLDK], vr[NLDV];
    int lds_k[NLDK], lds_v[NLDV], f_row[NLDK], f_vrow[NLDV];
    uint32_t k_off[NLDK], v_off[NLDV];
    bool f_vone[NLDV];
#pragma unroll
    for (int it = 0; it < NLDK; ++it) {                            // K piece id -> key row id / KCH, chunk id % KCH
        const int id = it * NT + tid, row = id / KCH, c = id - row * KCH;
        const bool live = id < KT * KCH;
        f_row[it] = row;
        k_off[it] = (live && c * 8 < a.hd) ? (uint32_t)((row * a.k_st + c * 8) * 2) : OOB;
        lds_k[it] = live ? row * KRB + ((KS == 4 ? (c ^ ((row >> 1) & 7)) : c) << 4) : -1;
    }
#pragma unroll
    for (int it = 0; it < NLDV; ++it) {
        // V: the 8 lanes of a ds_write_b128 group fill ONE 128-byte [4 keys][16 d] block (piece j of the block = key j >> 1, chunk 2 dg + (j & 1));
        // with lane -> (row, chunk) as for K the group's addresses alias mod 128 bytes: a 4-way bank conflict on every V write (PMC: 48 of 144 LDS cycles per tile)
        const int id = it * NT + tid, blk = id >> 3, j = id & 7, kg = blk / DGN, dg = blk - kg * DGN, vrow = kg * 4 + (j >> 1), vc = dg * 2 + (j & 1);
        const bool live = id < KT * VCH;
        f_vrow[it] = vrow;
        v_off[it] = (live && vc * 8 < a.hd) ? (uint32_t)((vrow * a.v_st + vc * 8) * 2) : OOB;
        lds_v[it] = live ? KBYTES + blk * 128 + j * 16 : -1;
        f_vone[it] = ONES && live && vc == VCH - 1;                // the piece holding the last 8 d columns (all padding with ONES)
    }
    const int k_tile_bytes = (int)(KT * a.k_st * 2), v_tile_bytes = (int)(KT * a.v_st * 2);
    auto fetch = [&](int t, bool ragged) {                         // tile t -> registers (ragged: the tile holds rows past the last key)
#pragma unroll
        for (int it = 0; it < NLDK; ++it) {
            uint32_t ko = k_off[it];
            if (ragged && t * KT + f_row[it] >= a.Tk) ko = OOB;
            kr[it] = __builtin_amdgcn_raw_buffer_load_b128(krs, ko, t * k_tile_bytes, 0);
        }
#pragma unroll
        for (int it = 0; it < NLDV; ++it) {
            uint32_t vo = v_off[it];
            if (ragged && t * KT + f_vrow[it] >= a.Tk) vo = OOB;
            vr[it] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vo, t * v_tile_bytes, 0);
        }
    };
    auto commit = [&](char *buf) {
#pragma unroll
        for (int it = 0; it < NLDK; ++it)
            if ((KT * KCH) % NT == 0 || lds_k[it] >= 0) *(u32x4 *)(buf + lds_k[it]) = kr[it];
#pragma unroll
        for (int it = 0; it < NLDV; ++it) {
            if (ONES && f_vone[it]) vr[it] = u32x4{0x3f80u, 0u, 0u, 0u};            // V[key][32 DB - 8] = 1.0 (bf16), the seven columns after it 0
            if ((KT * VCH) % NT == 0 || lds_v[it] >= 0) *(u32x4 *)(buf + lds_v[it]) = vr[it];
        }
    };
    // fragment addresses inside a tile buffer: K row (32 kb + ql), chunk 2 ks + hi (swizzled at head_dim <= 64);  V block (key group 8 kb + 4 j + hi (+2), d-group 2 db + (ql >> 4))
    int ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = ql * KRB + ((KS == 4 ? ((2 * ks + hi) ^ ((ql >> 1) & 7)) : (2 * ks + hi)) << 4);
    const int va = KBYTES + (hi * DGN + (ql >> 4)) * 128 + (ql & 15) * 8;
    const bool wave_active = (qblk * NW + wave) * 32 < a.Tq;     // (scalar: wave is)

    // one 64-key tile; FIRST: the tile opens the row (reference maximum := its maximum); mask (scalar): the tile may hold keys past Tk or
    // (causal) past the query.  ONE instantiation inside the loop: the accumulators keep their registers across iterations.
    // NKB = 1: only the tile's first 32-key block holds keys (the LAST tile of 577 = 9 x 64 + 1 or 196 = 3 x 64 + 4 keys: half the tile's
    // products and exponentials; instantiated outside the loop only)
    auto compute = [&](const char *buf, int k0, bool mask, auto FIRST_, auto NKB_) {
        constexpr bool FIRST = decltype(FIRST_)::value;
        constexpr int NKB = decltype(NKB_)::value;
        f32x16 s[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {                       // two independent accumulation chains, alternating
                const bf16x8 kf = *(const bf16x8 *)(buf + kb * (32 * KRB) + ka[ks]);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? negm : s[kb], 0, 0, 0);
            }
        // (measured and dropped: all eight K fragments and all sixteen V^T fragments fetched ahead of their products -- 32 + 32 more live
        //  registers, 2 waves per SIMD instead of 3: 4096^2 x 56 global blocks 526 -> 558 us; the third wave hides more than the prefetch)
        if (mask) {
            const int lim = (a.causal ? min(a.Tk - 1, q_row) : a.Tk - 1) - k0 - 4 * hi;      // last visible key of this lane's query, tile-relative
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[kb][r] = (kb * 32 + (r & 3) + 8 * (r >> 2) > lim) ? -3.0e38f : s[kb][r];
        }
        if (NKB == 1) s[1] = s[0];                                   // (maximum and rescale below walk both blocks)
        const float lm = lane_max32(s[0], s[1]);
        if (FIRST) rescale32<true, ONES, DB>(s, lm, negm, lsum, o);
        else if (__any(lm > ATTN_THR)) rescale32<false, ONES, DB>(s, lm, negm, lsum, o);
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
            if (!ONES) {                                             // 16 -> 8 -> 4 partial sums: as many adds as a 16-register accumulator, 12 registers fewer
                typedef __attribute__((ext_vector_type(8))) float f32x8;
                const f32x8 u = __builtin_shufflevector(s[kb], s[kb], 0, 1, 2, 3, 4, 5, 6, 7) + __builtin_shufflevector(s[kb], s[kb], 8, 9, 10, 11, 12, 13, 14, 15);
                lsum = lsum + (__builtin_shufflevector(u, u, 0, 1, 2, 3) + __builtin_shufflevector(u, u, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t tmp[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) tmp[e >> 1] = pack2(s[kb][8 * j + e], s[kb][8 * j + e + 1]);
                pf[kb][j] = *(bf16x8 *)tmp;
            }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const char *blk = buf + va + ((8 * kb + 4 * j) * DGN + 2 * db) * 128;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(blk));
                    const s16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(blk + 2 * DGN * 128));
                    const uint2 l2 = *(const uint2 *)&lo, h2 = *(const uint2 *)&hh;
                    uint4 raw = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&raw, pf[kb][j], o[db], 0, 0, 0);
                }
            }
        }
    };

    const int n_tiles = (a.Tk + KT - 1) / KT, n_full = a.Tk / KT;
    fetch(0, n_full == 0);
    commit(smem);
    if (n_tiles > 1) fetch(1, 1 >= n_full);
    __syncthreads();
    if (n_tiles > 1) {
        commit(smem + TILE);
        if (n_tiles > 2) fetch(2, 2 >= n_full);
    }
    using One = std::integral_constant<int, 1>;
    using Two = std::integral_constant<int, 2>;
    const int last = n_tiles - 1;
    const bool half_last = a.Tk - last * KT <= 32;               // the last tile's second key block is empty
    if (wave_active) {
        if (last == 0 && half_last) compute(smem, 0, true, std::true_type{}, One{});
        else compute(smem, 0, n_full == 0 || a.causal, std::true_type{}, Two{});
    }
    for (int t = 1; t < last; ++t) {                               // full tiles: ONE instantiation inside the loop
        __syncthreads();                                           // tile t is in buffer t & 1; every wave is done with tile t - 1 (buffer (t + 1) & 1)
        commit(smem + ((t + 1) & 1) * TILE);
        if (t + 2 < n_tiles) fetch(t + 2, t + 2 >= n_full);
        if (wave_active) compute(smem + (t & 1) * TILE, t * KT, a.causal != 0, std::false_type{}, Two{});
    }
    if (last >= 1) {
        __syncthreads();
        if (wave_active) {
            if (half_last) compute(smem + (last & 1) * TILE, last * KT, true, std::false_type{}, One{});
            else compute(smem + (last & 1) * TILE, last * KT, last >= n_full || a.causal, std::false_type{}, Two{});
        }
    }
    // ---- store: lane (q, hi) holds O^T rows d = 32 db + 8 g + 4 hi + (0..3) in o[db][4 g ..]; the two halves of a 16-byte row segment are
    // traded between lanes q and q + 32 (v_permlane32_swap) so that every lane stores 16 contiguous bytes: lanes < 32 at d = 32 db + 16 p,
    // lanes >= 32 at d = 32 db + 16 p + 8
    float l = 0.f;
    if (ONES) l = hi == 0 ? o[DB - 1][12] : 0.f;                  // O^T row 32 DB - 8 = 32 (DB - 1) + (12 & 3) + 8 (12 >> 2) + 4 hi at hi = 0
    else {
        l = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
    }
    {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float inv = 1.0f / l;
    uint16_t *op = a.o + b * a.o_sb + h * a.o_sh + (long long)q_row * a.o_st;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {                           // g = 2 pr (kept by lanes < 32), 2 pr + 1 (kept by lanes >= 32)
            uint32_t a0 = pack2(o[db][8 * pr + 0] * inv, o[db][8 * pr + 1] * inv), a1 = pack2(o[db][8 * pr + 2] * inv, o[db][8 * pr + 3] * inv);
            uint32_t b0 = pack2(o[db][8 * pr + 4] * inv, o[db][8 * pr + 5] * inv), b1 = pack2(o[db][8 * pr + 6] * inv, o[db][8 * pr + 7] * inv);
            // vdst = group 2 pr, src = group 2 pr + 1: lanes >= 32 of vdst trade with lanes < 32 of src
            const u32x2 x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            // lanes < 32: (a: own g = 2 pr, d + 0..3 | x[0] upper lanes' g = 2 pr, d + 4..7);  lanes >= 32: (lower lanes' g = 2 pr + 1 | own)
            const int d0 = db * 32 + pr * 16 + hi * 8;
            if (q_row < a.Tq && d0 < a.hd) *(uint4 *)(op + d0) = make_uint4(x0[0], x1[0], x0[1], x1[1]);
        }
    }
}


}  // namespace

namespace {
struct AttnKnobs {                                   // environment knobs of the dispatcher (see ovo_knobs_dynamic)
    bool wide, narrow, no_chunk, no_tiny;
    int force32;
    void read() {
        wide = getenv("OVO_ATTN_WIDE"); narrow = getenv("OVO_ATTN_NARROW"); no_chunk = getenv("OVO_ATTN_NO_CHUNK"); no_tiny = getenv("OVO_ATTN_NO_TINY");
        force32 = getenv("OVO_ATTN32") ? atoi(getenv("OVO_ATTN32")) : -1;
    }
};
const AttnKnobs &attn_knobs() {
    static AttnKnobs k = [] { AttnKnobs x; x.read(); return x; }();
    if (ovo_knobs_dynamic()) k.read();
    return k;
}
}  // namespace

extern "C" int ovo_attention(const ovo_attention_t *p, ovo_stream_t stream) {
    const AttnKnobs &kn = attn_knobs();
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
    a.scale_log2e = p->scale == 0.0f ? 1.0f : p->scale * 1.4426950408889634f;       // 0: the projection already carries scale * log2 e
    a.causal = p->causal;
    hipStream_t s = (hipStream_t)stream;
    // k_attention<HD, QT> (16 x 16 MFMA tiles, 64 or 128 queries per workgroup): only what k_attention32 below cannot address (misaligned output rows,
    // K / V slabs past 2 GB, OVO_ATTN32=0) -- every head_dim up to 128 goes to k_attention32 first.  History of the
    // 64- vs 128-query rule (rounds 2-3): DESIGN.md section 3.
    // (round 4: with the cheaper softmax the 64-query form also wins at head_dim 128 -- 8 x 16 x 2048^2 x 128: 512 us narrow vs 662 wide; the 128-query
    // form is left for OVO_ATTN_WIDE experiments; head_dim <= 64 goes to k_attention32 above unless OVO_ATTN32=0)
    const bool wide = kn.wide && !kn.narrow;
    const int qpb = wide ? 128 : 64;
    dim3 grid((p->Tq + qpb - 1) / qpb, p->B * p->H);
    a.q_tiles = (int)grid.x; a.chunk = 0;
    // (also the form for more than 65535 batch x head rows -- windowed attention of several frames at once: the y dimension of a grid ends there)
    if ((grid.x > 1 && !kn.no_chunk) || grid.y > 65535) {   // several q-tiles share a head's K / V: keep them on one XCD's L2
        const long long total = (long long)grid.x * grid.y;
        a.chunk = (int)((total + 7) / 8);
        grid = dim3((unsigned)(a.chunk * 8), 1);
    }
    const bool prof = ovo_prof_enabled();
    if (prof) { ovo_prof_begin(1, 4.0 * p->B * p->H * (double)p->Tq * p->Tk * p->hd, s); ovo_prof_shape(p->B * p->H, p->Tq, p->Tk);
                ovo_prof_bytes(2.0 * p->B * p->H * p->hd * (2.0 * p->Tq + 2.0 * p->Tk)); }
    struct Done { bool on; hipStream_t s; ~Done() { if (on) ovo_prof_end(s); } } done{prof, s};
    // More than 16 queries or 64 keys: 32 x 32 MFMA tiles, K / V streamed (k_attention32); 128 queries per workgroup, 64 when a (batch, head)
    // pair has no more.  OVO_ATTN32 = 0 / 1 overrides the shape rule (tools/attn_bench.py: 12 frames' shapes, a32 vs the 16 x 16 kernels with the
    // same softmax: 577^2 x 64 60.6 vs 66.8 us, 4096^2 x 56 466 vs 622, 196^2 x 56 windows 63 vs 76, 49 x 196 64 vs 79, 49^2 29 vs 34, 64^2 tie;
    // <= 16 queries x <= 64 keys stay with the one-wave-per-pair kernel: 222 / 79 / 96 us against 287 / 261 / 486)
    {
        const int force32 = kn.force32;
        const bool use32 = force32 >= 0 ? force32 != 0 : !(p->Tq <= 16 && p->Tk <= 64);
        // 16-byte output stores and 32-bit K / V byte offsets inside a (batch, head) slab
        const bool fits32 = ((uintptr_t)p->o & 15) == 0 && p->o_st % 8 == 0 && p->o_sh % 8 == 0 && p->o_sb % 8 == 0 &&
                            ((long long)p->Tk + 64) * p->k_st * 2 < (1ll << 31) && ((long long)p->Tk + 64) * p->v_st * 2 < (1ll << 31);
        if (use32 && fits32) {
            // head_dim <= 64: KS 4 / DB 2; <= 80: 5 / 3 (SigLIP's 72, hiera_l's 72, ViT-H's 80); <= 96: 6 / 3; <= 128: 8 / 4.  The ones column needs 8 free d
            // columns at the end of the last 32-row block.  64-query workgroups only in the head_dim <= 64 form (its staging fits 128 threads' registers)
            const int qpw = (p->Tq <= 64 && p->hd <= 64) ? 64 : 128;
            const long long qb = (p->Tq + qpw - 1) / qpw, total = qb * p->B * p->H;
            a.q_tiles = (int)qb; a.chunk = (int)((total + 7) / 8);
            const unsigned g = (unsigned)(a.chunk * 8);
            if (qpw == 64) {
                if (p->hd <= 56) k_attention32<2, 4, 2, true><<<g, 128, 0, s>>>(a);
                else k_attention32<2, 4, 2, false><<<g, 128, 0, s>>>(a);
            } else if (p->hd <= 56) k_attention32<4, 4, 2, true><<<g, 256, 0, s>>>(a);
            else if (p->hd <= 64) k_attention32<4, 4, 2, false><<<g, 256, 0, s>>>(a);
            else if (p->hd <= 80) k_attention32<4, 5, 3, true><<<g, 256, 0, s>>>(a);
            else if (p->hd <= 88) k_attention32<4, 6, 3, true><<<g, 256, 0, s>>>(a);
            else if (p->hd <= 96) k_attention32<4, 6, 3, false><<<g, 256, 0, s>>>(a);
            else if (p->hd <= 120) k_attention32<4, 8, 4, true><<<g, 256, 0, s>>>(a);
            else k_attention32<4, 8, 4, false><<<g, 256, 0, s>>>(a);
            OVO_CHECK_LAUNCH();
            return OVO_OK;
        }
    }
    // tiny problems (one key tile, at most four query tiles): one wave per (batch, head) pair
    // (tools/attn_bench.py, 12 frames of hiera_b+: 16 x 16 windows 165 -> 79 us = 4.5 TB/s of q/k/v/o, pooled 4 x 16 blocks 321 -> 89 us;
    //  with 2-4 query tiles per pair -- 64 x 64, 49 x 49 -- the tiled kernel's four waves per pair are ahead: 180 vs 201 us, 33 vs 37)
    if (p->hd <= 64 && p->Tk <= 64 && p->Tq <= 16 && !kn.no_tiny) {
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
