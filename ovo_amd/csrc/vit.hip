// vit.hip -- ViT forward driver: the whole layer loop of an open_clip / PE vision transformer in one C call.
//
// Launch sequence per block (residual stream x kept in fp32, GEMM operands bf16, fp32 accumulation):
//   LN1 -> QKV GEMM(+bias, RoPE of q and k in the epilogue) -> fused attention -> out-proj GEMM(+bias, += x)
//   LN2 -> FC1 GEMM(+bias, GELU) -> FC2 GEMM(+bias, += x)
// With folded weights (ovo_vit_layer_t.qkv_wf ..., ABI v12) and a batched forward the two LayerNorm launches disappear: the out-projection / FC2 epilogue that
// writes x also writes bf16(x) and per-row partial (sum, sum of squares); the QKV / FC1 product multiplies bf16(x) by W' = gamma . W and its epilogue applies
// rstd (acc - mean colsum(W')) + b' (gemm_common.h "LayerNorm FOLD"): 5 launches per block and 2 x 99 MB less traffic per block at 16 K tokens.
// 7 launches per block, no host synchronisation, nothing allocated, every launch on the caller's stream (the whole forward is ONE C call:
// ~200 launches in ~1 ms of host time, which the pipeline hides by batching 12 keyframes' crops per forward; nothing here captures graphs).
#include "common.h"
#include "gemm_common.h"

namespace {

struct Ws {
    uint16_t *col;     // [B*P, kpad]       im2col patches
    float *patch;      // [B*P, width]      patch-embed GEMM output
    float *x;          // [M, width]        residual stream
    uint16_t *h;       // [M, width]        LayerNorm output
    uint16_t *qkv;     // [M, 3*width]
    uint16_t *att;     // [M, width]
    uint16_t *u;       // [M, mlp]
    float *stats;      // [16, M, 2]        LayerNorm fold: partial (sum, sum of squares) of the rows of x, part-major
    size_t bytes;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

Ws carve(const ovo_vit_config_t &c, int B, void *base) {
    const size_t G = c.image_size / c.patch, P = G * G, T = P + c.n_prefix, M = (size_t)B * T;
    Ws w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t n) { char *r = p ? p + off : nullptr; off += align256(n); return r; };
    w.col = (uint16_t *)take((size_t)B * P * c.kpad * 2);
    w.patch = (float *)take((size_t)B * P * c.width * 4);
    w.x = (float *)take(M * c.width * 4);
    w.h = (uint16_t *)take(M * c.width * 2);
    w.qkv = (uint16_t *)take(M * 3 * c.width * 2);
    w.att = (uint16_t *)take(M * c.width * 2);
    w.u = (uint16_t *)take(M * c.mlp_dim * 2);
    w.stats = (float *)take(M * 16 * 2 * 4);
    w.bytes = off;
    return w;
}

int gemm(const void *A, long long lda, const void *W, long long ldw, const float *bias, void *C, long long ldc, int out_dtype,
         const float *add, long long ld_add, int M, int N, int K, int act, ovo_stream_t s) {
    ovo_gemm_t g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.add = add; g.ld_add = ld_add;
    g.M = M; g.N = N; g.K = K; g.in_dtype = 2; g.out_dtype = out_dtype; g.act = act; g.alpha = 1.0f;
    return ovo_gemm(&g, s);
}

}  // namespace

#define TRY(call)                  \
    do {                           \
        const int rc__ = (call);   \
        if (rc__ != OVO_OK) return rc__; \
    } while (0)

extern "C" {

size_t ovo_vit_workspace_bytes(const ovo_vit_config_t *cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve(*cfg, B, nullptr).bytes;
}

int ovo_vit_forward(const ovo_vit_config_t *cfg, const ovo_vit_weights_t *w, const float *images, int B, float *out,
                    void *ws, size_t ws_bytes, ovo_stream_t stream) {
    OVO_REQUIRE(cfg && w && images && out && ws && B > 0, "null argument");
    const ovo_vit_config_t &c = *cfg;
    OVO_REQUIRE(c.patch > 0 && c.image_size >= c.patch && c.width % c.heads == 0 && c.layers >= 0, "bad config");   // a trailing partial patch is dropped, as the stride-p conv does
    OVO_REQUIRE(c.width % 32 == 0 && c.mlp_dim % 32 == 0 && c.kpad % 32 == 0 && c.kpad >= 3 * c.patch * c.patch, "dims must be multiples of 32");
    OVO_REQUIRE(w->patch_w && w->layers && w->ln_post_g && w->ln_post_b, "missing weights");
    OVO_REQUIRE(c.n_prefix == 0 || w->prefix, "class embedding missing");
    OVO_REQUIRE(!c.pre_ln || (w->ln_pre_g && w->ln_pre_b), "ln_pre weights missing");
    OVO_REQUIRE(!c.use_rope || (w->rope_cos && w->rope_sin), "rope tables missing");
    OVO_REQUIRE(c.pool == 0 || c.pool == 2 || (c.pool == 1 && w->proj_w && c.n_prefix == 1 && c.out_dim % 4 == 0), "pool=1 needs a class token and proj");
    OVO_REQUIRE(c.pool != 2 || (w->map_q && w->map_kv_w && w->map_kv_b && w->map_proj_w && w->map_proj_b && w->map_ln_g && w->map_ln_b &&
                                w->map_fc1_w && w->map_fc1_b && w->map_fc2_w && w->map_fc2_b && c.out_dim == c.width), "pool=2 needs the map_* weights");
    const int G = c.image_size / c.patch, P = G * G, T = P + c.n_prefix, M = B * T, hd = c.width / c.heads;
    OVO_REQUIRE(hd % 8 == 0 && hd <= 128, "head_dim must be a multiple of 8, <= 128");
    Ws k = carve(c, B, ws);
    OVO_REQUIRE(ws_bytes >= k.bytes, "workspace too small");
    const int D = c.width;

    // patch embedding: im2col -> GEMM -> (+class token, +pos, ln_pre) -> x
    TRY(ovo_im2col(images, B, 3, c.image_size, c.image_size, c.patch, c.patch, 0, k.col, c.kpad, stream));
    TRY(gemm(k.col, c.kpad, w->patch_w, c.kpad, w->patch_b, k.patch, D, 0, nullptr, 0, B * P, D, c.kpad, 0, stream));
    TRY(ovo_vit_embed(k.patch, w->prefix, c.n_prefix, w->pos, B, P, D, c.pre_ln ? w->ln_pre_g : nullptr,
                      c.pre_ln ? w->ln_pre_b : nullptr, c.ln_eps, k.x, stream));

    const float scale = c.q_prescaled ? 0.0f : 1.0f / sqrtf((float)hd);      // 0: the q rows of qkv_w / qkv_b (and map_q) carry log2 e / sqrt(hd)
    // The LayerNorm fold: every layer carries folded weights, all four products of a block sit on the ping-pong kernel at this batch, and a row's partial
    // statistics fit the 16 slots (width <= 1024).  OVO_VIT_LNFOLD=0 keeps the LayerNorm kernels.
    namespace gd = ovo_gemm_detail;
    static int fold_env = getenv("OVO_VIT_LNFOLD") ? atoi(getenv("OVO_VIT_LNFOLD")) : 1;
    if (ovo_knobs_dynamic()) fold_env = getenv("OVO_VIT_LNFOLD") ? atoi(getenv("OVO_VIT_LNFOLD")) : 1;
    const int parts = gd::gemm_fold_parts(D);
    bool fold = fold_env && c.layers > 0 && parts <= 16 && gd::gemm_fold_ok(M, D, D) && gd::gemm_fold_ok(M, D, c.mlp_dim) && gd::gemm_fold_ok(M, 3 * D, D) &&
                gd::gemm_fold_ok(M, c.mlp_dim, D) && (c.act == 0 || c.act == 1) && (!c.use_rope || hd == 64);
    for (int l = 0; l < c.layers && fold; ++l) {
        const ovo_vit_layer_t &L = w->layers[l];
        fold = L.qkv_wf && L.qkv_bf && L.qkv_cs && L.fc1_wf && L.fc1_bf && L.fc1_cs;
    }
    if (fold) {
        TRY(gd::gemm_fold_rowstats(k.x, D, M, D, k.h, D, k.stats, stream));      // layer 0's LayerNorm 1: one partial per row
        int have = 1;                                                            // partials per row in k.stats
        for (int l = 0; l < c.layers; ++l) {
            const ovo_vit_layer_t &L = w->layers[l];
            ovo_gemm_t g;
            g.A = k.h; g.lda = D; g.W = L.qkv_wf; g.ldw = D; g.bias = L.qkv_bf; g.C = k.qkv; g.ldc = 3 * D; g.add = nullptr; g.ld_add = 0;
            g.M = M; g.N = 3 * D; g.K = D; g.in_dtype = 2; g.out_dtype = 2; g.act = 0; g.alpha = 1.0f;
            const ovo_rope_t r = {w->rope_cos, w->rope_sin, T, hd, 2 * D, c.n_prefix};
            const gd::FoldIn f1 = {k.stats, M, have, D, L.qkv_cs, c.ln_eps};
            TRY(gd::gemm_fold_consumer(&g, c.use_rope ? &r : nullptr, f1, stream));
            ovo_attention_t a = {};
            a.q = k.qkv; a.k = k.qkv + D; a.v = k.qkv + 2 * D; a.o = k.att;
            a.q_sb = a.k_sb = a.v_sb = (int64_t)T * 3 * D; a.q_sh = a.k_sh = a.v_sh = hd; a.q_st = a.k_st = a.v_st = 3 * D;
            a.o_sb = (int64_t)T * D; a.o_sh = hd; a.o_st = D;
            a.B = B; a.H = c.heads; a.Tq = T; a.Tk = T; a.hd = hd; a.scale = scale;
            TRY(ovo_attention(&a, stream));
            const gd::FoldOut fo = {k.h, D, k.stats, M};
            g.A = k.att; g.lda = D; g.W = L.out_w; g.ldw = D; g.bias = L.out_b; g.C = k.x; g.ldc = D; g.add = k.x; g.ld_add = D;
            g.M = M; g.N = D; g.K = D; g.out_dtype = 0; g.act = 0;
            TRY(gd::gemm_fold_producer(&g, fo, stream));
            have = parts;
            g.A = k.h; g.lda = D; g.W = L.fc1_wf; g.ldw = D; g.bias = L.fc1_bf; g.C = k.u; g.ldc = c.mlp_dim; g.add = nullptr; g.ld_add = 0;
            g.M = M; g.N = c.mlp_dim; g.K = D; g.out_dtype = 2; g.act = c.act;
            const gd::FoldIn f2 = {k.stats, M, have, D, L.fc1_cs, c.ln_eps};
            TRY(gd::gemm_fold_consumer(&g, nullptr, f2, stream));
            if (l + 1 < c.layers) {
                g.A = k.u; g.lda = c.mlp_dim; g.W = L.fc2_w; g.ldw = c.mlp_dim; g.bias = L.fc2_b; g.C = k.x; g.ldc = D; g.add = k.x; g.ld_add = D;
                g.M = M; g.N = D; g.K = c.mlp_dim; g.out_dtype = 0; g.act = 0;
                TRY(gd::gemm_fold_producer(&g, fo, stream));
            } else {                                                             // ln_post reads x itself
                TRY(gemm(k.u, c.mlp_dim, L.fc2_w, c.mlp_dim, L.fc2_b, k.x, D, 0, k.x, D, M, D, c.mlp_dim, 0, stream));
            }
        }
    }
    for (int l = 0; l < c.layers && !fold; ++l) {
        const ovo_vit_layer_t &L = w->layers[l];
        TRY(ovo_layernorm(k.x, D, M, D, L.ln1_g, L.ln1_b, c.ln_eps, k.h, D, 2, stream));
        if (c.use_rope) {                                                // q, k rotated in the projection's epilogue
            ovo_gemm_t g;
            g.A = k.h; g.lda = D; g.W = L.qkv_w; g.ldw = D; g.bias = L.qkv_b; g.C = k.qkv; g.ldc = 3 * D; g.add = nullptr; g.ld_add = 0;
            g.M = M; g.N = 3 * D; g.K = D; g.in_dtype = 2; g.out_dtype = 2; g.act = 0; g.alpha = 1.0f;
            const ovo_rope_t r = {w->rope_cos, w->rope_sin, T, hd, 2 * D, c.n_prefix};
            TRY(ovo_gemm_rope(&g, &r, stream));
        } else {
            TRY(gemm(k.h, D, L.qkv_w, D, L.qkv_b, k.qkv, 3 * D, 2, nullptr, 0, M, 3 * D, D, 0, stream));
        }
        ovo_attention_t a = {};
        a.q = k.qkv; a.k = k.qkv + D; a.v = k.qkv + 2 * D; a.o = k.att;
        a.q_sb = a.k_sb = a.v_sb = (int64_t)T * 3 * D; a.q_sh = a.k_sh = a.v_sh = hd; a.q_st = a.k_st = a.v_st = 3 * D;
        a.o_sb = (int64_t)T * D; a.o_sh = hd; a.o_st = D;
        a.B = B; a.H = c.heads; a.Tq = T; a.Tk = T; a.hd = hd; a.scale = scale;
        TRY(ovo_attention(&a, stream));
        TRY(gemm(k.att, D, L.out_w, D, L.out_b, k.x, D, 0, k.x, D, M, D, D, 0, stream));
        TRY(ovo_layernorm(k.x, D, M, D, L.ln2_g, L.ln2_b, c.ln_eps, k.h, D, 2, stream));
        TRY(gemm(k.h, D, L.fc1_w, D, L.fc1_b, k.u, c.mlp_dim, 2, nullptr, 0, M, c.mlp_dim, D, c.act, stream));
        TRY(gemm(k.u, c.mlp_dim, L.fc2_w, c.mlp_dim, L.fc2_b, k.x, D, 0, k.x, D, M, D, c.mlp_dim, 0, stream));
    }

    if (c.pool == 0) {
        TRY(ovo_layernorm(k.x, D, M, D, w->ln_post_g, w->ln_post_b, c.ln_eps, out, D, 0, stream));
    } else if (c.pool == 2) {
        // SigLIP "map" head: one learned query per image attends over ln_post(tokens); y = att @ Wo; out = y + mlp(LN(y)).
        // The query projection does not depend on the input and arrives precomputed (map_q).
        float *y = k.patch;                                               // [B, D] (the patch buffer is free by now)
        TRY(ovo_layernorm(k.x, D, M, D, w->ln_post_g, w->ln_post_b, c.ln_eps, k.h, D, 2, stream));
        TRY(gemm(k.h, D, w->map_kv_w, D, w->map_kv_b, k.qkv, 2 * D, 2, nullptr, 0, M, 2 * D, D, 0, stream));
        ovo_attention_t a = {};
        a.q = w->map_q; a.k = k.qkv; a.v = k.qkv + D; a.o = k.att;
        a.q_sb = 0; a.q_sh = hd; a.q_st = D;
        a.k_sb = a.v_sb = (int64_t)T * 2 * D; a.k_sh = a.v_sh = hd; a.k_st = a.v_st = 2 * D;
        a.o_sb = D; a.o_sh = hd; a.o_st = D;
        a.B = B; a.H = c.heads; a.Tq = 1; a.Tk = T; a.hd = hd; a.scale = scale;
        TRY(ovo_attention(&a, stream));
        TRY(gemm(k.att, D, w->map_proj_w, D, w->map_proj_b, y, D, 0, nullptr, 0, B, D, D, 0, stream));
        TRY(ovo_layernorm(y, D, B, D, w->map_ln_g, w->map_ln_b, c.ln_eps, k.h, D, 2, stream));
        TRY(gemm(k.h, D, w->map_fc1_w, D, w->map_fc1_b, k.u, c.mlp_dim, 2, nullptr, 0, B, c.mlp_dim, D, c.act, stream));
        TRY(gemm(k.u, c.mlp_dim, w->map_fc2_w, c.mlp_dim, w->map_fc2_b, out, D, 0, y, D, B, D, c.mlp_dim, 0, stream));
    } else {
        // ln_post on the class token of each image (row stride T*D), then @ proj
        TRY(ovo_layernorm(k.x, (int64_t)T * D, B, D, w->ln_post_g, w->ln_post_b, c.ln_eps, k.h, D, 2, stream));
        TRY(gemm(k.h, D, w->proj_w, D, nullptr, out, c.out_dim, 0, nullptr, 0, B, c.out_dim, D, 0, stream));
    }
    return OVO_OK;
}

}  // extern "C"
