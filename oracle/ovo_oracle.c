/*
 * ovo_oracle.c -- CPU restatement of the reference's geometry hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (ovo_amd/) never does.  Every function states the reference lines it follows
 * (paths relative to the reference checkout).
 *
 * Floating-point contract.  The reference runs torch-CPU `einsum("mn,bn->bm")` / `mm` with a
 * contraction length of 3 or 4.  For >= ~1000 rows that is bit-identical to a left-to-right fused
 * multiply-add chain
 *        acc = m0*p0;  acc = fmaf(m1,p1,acc);  acc = fmaf(m2,p2,acc);  [acc = fmaf(m3,p3,acc)]
 * (SURVEY.md section 7 "hard parts"; re-checked against tests/golden/geometry_*.npz, which were
 * produced by the reference itself).  Divisions are IEEE, rounding to pixels is half-to-even
 * (torch.round), float->int32 conversion of non-finite / out-of-range values follows x86 cvttss2si
 * (INT32_MIN).  Build with -ffp-contract=off so nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float dot3(const float *m, float x, float y, float z) {
    float acc = m[0] * x;
    acc = fmaf(m[1], y, acc);
    acc = fmaf(m[2], z, acc);
    return acc;
}

static inline float dot4(const float *m, float x, float y, float z, float w) {
    float acc = m[0] * x;
    acc = fmaf(m[1], y, acc);
    acc = fmaf(m[2], z, acc);
    acc = fmaf(m[3], w, acc);
    return acc;
}

static inline int32_t f2i_x86(float v) {
    /* tensor.int() on x86: truncation, INT32_MIN when unrepresentable */
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int32_t)v;
}

/* geometry_utils.py:205-276  compute_frustum_aabb / points_inside_aabb_mask /
 * points_inside_frustum_mask / compute_frustum_point_ids.
 * aabb = {minx,miny,minz,maxx,maxy,maxz}; planes = 6 rows of (a,b,c,d); a point is kept when it is
 * inside the closed AABB and plane . (x,y,z,1) <= 0 for all six planes.  Output: ascending indices. */
int64_t orc_frustum_ids(const float *pts, int64_t n, const float *aabb, const float *planes, int64_t *out_idx) {
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (!(x >= aabb[0] && x <= aabb[3] && y >= aabb[1] && y <= aabb[4] && z >= aabb[2] && z <= aabb[5])) continue;
        int ok = 1;
        for (int p = 0; p < 6 && ok; ++p) ok = dot4(planes + 4 * p, x, y, z, 1.0f) <= 0.0f;
        if (ok) out_idx[c++] = i;
    }
    return c;
}

/* geometry_utils.py:26-43 project_3d_points on homogeneous points that get the w2c transform first.
 * stride = 3 (w := 1) or 4 floats per point. */
static inline void project_one(const float *p, int stride, const float *w2c, const float *K,
                               float *zc, int32_t *u, int32_t *v) {
    float x = p[0], y = p[1], z = p[2], w = stride == 4 ? p[3] : 1.0f;
    float lx = dot4(w2c, x, y, z, w), ly = dot4(w2c + 4, x, y, z, w);
    float lz = dot4(w2c + 8, x, y, z, w), lw = dot4(w2c + 12, x, y, z, w);
    *zc = lz;                                   /* depth test uses the un-normalised z (:76) */
    float cx = lx / lw, cy = ly / lw, cz = lz / lw;
    float pu = dot3(K, cx, cy, cz), pv = dot3(K + 3, cx, cy, cz), pw = dot3(K + 6, cx, cy, cz);
    *u = f2i_x86(rintf(pu / pw));
    *v = f2i_x86(rintf(pv / pw));
}

void orc_project(const float *pts, int64_t n, int stride, const float *w2c, const float *K, int32_t *out_uv) {
    for (int64_t i = 0; i < n; ++i) {
        float zc;
        project_one(pts + (int64_t)stride * i, stride, w2c, K, &zc, out_uv + 2 * i, out_uv + 2 * i + 1);
    }
}

/* geometry_utils.py:46-89 match_3d_points_to_2d_pixels.  Keeps a point when its pixel is inside the
 * image, |z - depth[v,u]| < th and depth[v,u] != 0.  Outputs (index, (u,v)) in ascending index order. */
int64_t orc_match(const float *pts, int64_t n, int stride, const float *w2c, const float *K,
                  const float *depth, int h, int w, float th, int64_t *out_idx, int32_t *out_uv) {
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        float zc; int32_t u, v;
        project_one(pts + (int64_t)stride * i, stride, w2c, K, &zc, &u, &v);
        if (!(u < w && v < h && u >= 0 && v >= 0)) continue;
        float d = depth[(int64_t)v * w + u];
        if (!(fabsf(zc - d) < th) || d == 0.0f) continue;
        out_idx[c] = i; out_uv[2 * c] = u; out_uv[2 * c + 1] = v; ++c;
    }
    return c;
}

/* vanilla_mapper.py:46-85 VanillaMapper.map, the part after the "explained pixel" scatter:
 *  valid = depth>0 with explained pixels cleared (:55,:61); 3x3 stride-1 erosion of `valid` when the
 *  map is not empty (:62, pooling defined :27-29: ~maxpool(~mask), zero... -inf padding => border
 *  pixels only see in-image neighbours); [::ds, ::ds] subsample (:67-68); unproject
 *  ((x-cx)*d)/fx (:74-76); c2w transform (:79); rows emitted in row-major pixel order.
 *  explained may be NULL (first frame).  Returns number of points appended. */
int64_t orc_backproject(const float *depth, const uint8_t *rgb, const uint8_t *explained, int h, int w,
                        int erode, int ds, const float *K, const float *c2w,
                        float *out_xyz, uint8_t *out_rgb) {
    uint8_t *valid = (uint8_t *)malloc((size_t)h * w);
    for (int i = 0; i < h * w; ++i) valid[i] = depth[i] > 0.0f && !(explained && explained[i]);
    int64_t c = 0;
    for (int y = 0; y < h; y += ds) for (int x = 0; x < w; x += ds) {
        int ok = valid[y * w + x];
        if (ok && erode) {
            for (int dy = -1; dy <= 1 && ok; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                if (!valid[yy * w + xx]) { ok = 0; break; }
            }
        }
        if (!ok) continue;
        float d = depth[y * w + x];
        float x3 = (((float)x - K[2]) * d) / K[0];
        float y3 = (((float)y - K[5]) * d) / K[4];
        out_xyz[3 * c + 0] = dot4(c2w, x3, y3, d, 1.0f);
        out_xyz[3 * c + 1] = dot4(c2w + 4, x3, y3, d, 1.0f);
        out_xyz[3 * c + 2] = dot4(c2w + 8, x3, y3, d, 1.0f);
        if (rgb) memcpy(out_rgb + 3 * c, rgb + 3 * ((int64_t)y * w + x), 3);
        ++c;
    }
    free(valid);
    return c;
}

/* clip_utils.py:10-19: S = F . T^T (row-major F[n,d], T[q,d]); siglip: sigmoid(S*exp(scale)+bias).
 * Plain fp32 accumulation in index order -- a float oracle, compared with a tolerance. */
void orc_similarity(const float *F, int64_t n, const float *T, int q, int d, int siglip,
                    float logit_scale, float logit_bias, float *out) {
    float es = expf(logit_scale);
    for (int64_t i = 0; i < n; ++i) for (int j = 0; j < q; ++j) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) acc += (double)F[i * d + k] * (double)T[(int64_t)j * d + k];
        float s = (float)acc;
        if (siglip) s = 1.0f / (1.0f + expf(-(s * es + logit_bias)));
        out[i * q + j] = s;
    }
}

/* segment_utils.py:218-230: pairwise intersection counts of bit-packed masks (words of 64 bit). */
void orc_mask_intersections(const uint64_t *bits, int n, int64_t words, int32_t *inter) {
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {
        int64_t c = 0;
        for (int64_t k = 0; k < words; ++k) c += __builtin_popcountll(bits[i * words + k] & bits[j * words + k]);
        inter[i * n + j] = inter[j * n + i] = (int32_t)c;
    }
}
