// attention.hip -- fused softmax(Q K^T * scale) V for the ViT blocks and the Hiera (windowed / global) blocks.
//
// One workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 queries.  Keys/values are
// walked in tiles of 64: K tile row-major and V tile TRANSPOSED in LDS (padded rows, conflict-free fragment
// reads), online softmax kept in registers, never materialising the score matrix.
//   S^T tile  = mfma(a = K fragment [16 keys x 32 d], b = Q fragment [32 d x 16 queries])   -> lane owns one query
//   O^T tile  = mfma(a = V^T fragment [16 d x 32 keys], b = P fragment [32 keys x 16 queries])
// With this operand order a lane's accumulator registers all belong to the query (lane & 15), so the running
// max / sum / rescale are per-lane scalars plus two xor-shuffles, and the P fragment feeds the second MFMA
// straight from registers (the k index of an MFMA is free as long as a and b agree on it).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnArgs {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    long long q_sb, q_sh, q_st;   // element strides: batch, head, token (head_dim contiguous)
    long long k_sb, k_sh, k_st;
    long long v_sb, v_sh, v_st;
    long long o_sb, o_sh, o_st;
    int B, H, Tq, Tk, hd;
    float scale_log2e;
};

__device__ __forceinline__ uint16_t f2bf(float x) {
    const uint32_t u = __float_as_uint(x);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int HD>
__global__ void __launch_bounds__(256) k_attention(AttnArgs a) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int KROW = HD + 8;           // padded K-tile row (elements)
    constexpr int VROW = KT + 8;           // padded V^T-tile row (elements)
    constexpr int CH = HD / 8;             // 16-byte chunks per head row
    __shared__ __attribute__((aligned(16))) uint16_t sK[KT * KROW];
    __shared__ __attribute__((aligned(16))) uint16_t sV[HD * VROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int q_row = blockIdx.x * 64 + wave * 16 + fr;       // this lane's query
    const uint16_t *qp = a.q + b * a.q_sb + h * a.q_sh;
    const uint16_t *kp = a.k + b * a.k_sb + h * a.k_sh;
    const uint16_t *vp = a.v + b * a.v_sb + h * a.v_sh;

    // Q fragments: lane (query fr, group fq) holds d = ks*32 + fq*8 .. +8
    bf16x8 qf[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
        uint4 raw = make_uint4(0, 0, 0, 0);
        const int d0 = ks * 32 + fq * 8;
        if (q_row < a.Tq && d0 < a.hd) raw = *(const uint4 *)(qp + (long long)q_row * a.q_st + d0);
        qf[ks] = *(bf16x8 *)&raw;
    }

    f32x4 oacc[HD / 16];
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;

    for (int k0 = 0; k0 < a.Tk; k0 += KT) {
        __syncthreads();                                        // previous tile fully consumed
        // ---- stage K (row-major) ----
#pragma unroll
        for (int it = 0; it < KT * CH / 256; ++it) {
            const int id = it * 256 + tid, row = id / CH, c = id % CH;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (k0 + row < a.Tk && c * 8 < a.hd) raw = *(const uint4 *)(kp + (long long)(k0 + row) * a.k_st + c * 8);
            *(uint4 *)(sK + row * KROW + c * 8) = raw;
        }
        // ---- stage V transposed: consecutive lanes take consecutive keys -> contiguous LDS writes ----
#pragma unroll
        for (int it = 0; it < KT * CH / 256; ++it) {
            const int id = it * 256 + tid, row = id % KT, c = id / KT;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (k0 + row < a.Tk && c * 8 < a.hd) raw = *(const uint4 *)(vp + (long long)(k0 + row) * a.v_st + c * 8);
            const uint16_t *e = (const uint16_t *)&raw;
#pragma unroll
            for (int j = 0; j < 8; ++j) sV[(c * 8 + j) * VROW + row] = e[j];
        }
        __syncthreads();

        // ---- S^T = K Q^T : 4 tiles of 16 keys ----
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                const bf16x8 kf = *(const bf16x8 *)(sK + (kt * 16 + fr) * KROW + ks * 32 + fq * 8);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // ---- online softmax for query fr; this lane holds keys kt*16 + fq*4 + r ----
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + fq * 4 + r;
                const float v = key < a.Tk ? s[kt][r] * a.scale_log2e : -1.0e30f;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[kt][r] - m_new);
                s[kt][r] = p;
                ps += p;
            }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < HD / 16; ++i) {
            oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha;
        }
        // ---- P fragments (bf16): element e of fragment kk <-> key (kk*2 + e/4)*16 + fq*4 + e%4 ----
        bf16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint16_t t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = f2bf(s[kk * 2 + (e >> 2)][e & 3]);
            pf[kk] = *(bf16x8 *)t;
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint16_t *row = sV + (dt * 16 + fr) * VROW;
                uint2 lo = *(const uint2 *)(row + (kk * 2) * 16 + fq * 4);
                uint2 hi = *(const uint2 *)(row + (kk * 2 + 1) * 16 + fq * 4);
                uint4 raw = make_uint4(lo.x, lo.y, hi.x, hi.y);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8 *)&raw, pf[kk], oacc[dt], 0, 0, 0);
            }
        }
    }
    // ---- store: lane holds O[q_row][dt*16 + fq*4 + r] ----
    if (q_row < a.Tq) {
        const float inv = 1.0f / l_run;
        uint16_t *op = a.o + b * a.o_sb + h * a.o_sh + (long long)q_row * a.o_st;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            const int d0 = dt * 16 + fq * 4;
            if (d0 < a.hd) {
                uint2 p;
                p.x = (uint32_t)f2bf(oacc[dt][0] * inv) | ((uint32_t)f2bf(oacc[dt][1] * inv) << 16);
                p.y = (uint32_t)f2bf(oacc[dt][2] * inv) | ((uint32_t)f2bf(oacc[dt][3] * inv) << 16);
                *(uint2 *)(op + d0) = p;
            }
        }
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
    OVO_REQUIRE((long long)p->B * p->H <= 65535, "B*H exceeds the grid y limit");
    AttnArgs a;
    a.q = (const uint16_t *)p->q; a.k = (const uint16_t *)p->k; a.v = (const uint16_t *)p->v; a.o = (uint16_t *)p->o;
    a.q_sb = p->q_sb; a.q_sh = p->q_sh; a.q_st = p->q_st; a.k_sb = p->k_sb; a.k_sh = p->k_sh; a.k_st = p->k_st;
    a.v_sb = p->v_sb; a.v_sh = p->v_sh; a.v_st = p->v_st; a.o_sb = p->o_sb; a.o_sh = p->o_sh; a.o_st = p->o_st;
    a.B = p->B; a.H = p->H; a.Tq = p->Tq; a.Tk = p->Tk; a.hd = p->hd;
    a.scale_log2e = p->scale * 1.4426950408889634f;
    dim3 grid((p->Tq + 63) / 64, p->B * p->H);
    hipStream_t s = (hipStream_t)stream;
    const bool prof = ovo_prof_enabled();
    if (prof) ovo_prof_begin(1, 4.0 * p->B * p->H * (double)p->Tq * p->Tk * p->hd, s);
    struct Done { bool on; hipStream_t s; ~Done() { if (on) ovo_prof_end(s); } } done{prof, s};
    if (p->hd <= 64) k_attention<64><<<grid, 256, 0, s>>>(a);
    else if (p->hd <= 96) k_attention<96><<<grid, 256, 0, s>>>(a);
    else k_attention<128><<<grid, 256, 0, s>>>(a);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
