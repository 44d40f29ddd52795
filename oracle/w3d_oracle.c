/*
 * w3d_oracle.c — CPU restatement of the Gaussian rasterizer behind Wheat-3DGS's
 * gaussian_renderer.render() / flashsplat_render().
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (wheat-3dgs_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in third-party submodules that are
 * NOT vendored in /root/reference (empty directories):
 *     ashawkey/diff-gaussian-rasterization @ 8829d14f814fccdaf840b7b0f3021a616583c0a1
 *     florinshen/flashsplat-rasterization  @ 189c483ffa33dd6d5661343ce496df0c6eb80a0c
 *     bkerbl/simple-knn (unpinned)                          [reference README.md:21, .gitmodules:1-9]
 * so what follows restates their *published* algorithm (3D Gaussian Splatting, Kerbl et al.
 * 2023, tile rasterizer with the depth/alpha outputs of the ashawkey fork, and the FlashSplat
 * contribution scatter) and is anchored on the reference's own call sites:
 *     gaussian_renderer/__init__.py:22-106   render()            (argument marshalling)
 *     gaussian_renderer/__init__.py:109-218  flashsplat_render() (8 outputs, used_mask subset)
 *     scene/gaussian_model.py:27-31,131-132  get_covariance      (cov3D must match; tests pin it)
 *     utils/sh_utils.py:57-112               eval_sh             (SH->RGB must match; tests pin it)
 *     scene/cameras.py:56-59, utils/graphics_utils.py:38-71      (transposed matrices)
 *     scene/gaussian_model.py:461-463        add_densification_stats (reads means2D.grad[:, :2])
 *     scene/gaussian_model.py:148            distCUDA2 (mean of 3 nearest squared distances)
 * The pieces of the reference that ARE importable (eval_sh, get_covariance, camera matrices)
 * pin the corresponding stages through tests/golden/ fixtures; the rasterizer constants
 * (0.2 near plane, 1.3 FoV clamp, 0.3 px^2 dilation, ceil(3 sigma), 1/255, 0.99, 1e-4) follow
 * SURVEY.md Appendix A and cannot be checked against the absent CUDA sources.
 *
 * All arithmetic is fp32 in the order written here; compile with -ffp-contract=off so that
 * the integer results derived from it (radii, tile rectangles, per-tile order) are
 * reproducible bit for bit.  Per-Gaussian gradient sums are accumulated in double.
 *
 * Matrix convention (reference scene/cameras.py:56-58): viewmatrix / projmatrix are the
 * TRANSPOSED matrices, i.e. flat index [4*c + r] holds maths element (r, c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16

/* Rounding-sensitivity probe (tests only).  Mode 0 is the restated algorithm: expf(power), per-Gaussian gradient sums in
 * double.  Bit 0 set: exp2f(power * log2 e) instead of expf — the same function with other roundings.  Bit 1 set: the
 * per-Gaussian sums are accumulated in fp32 in whatever order the threads arrive, which is what a GPU implementation with
 * float atomics (the reference's CUDA rasterizer included) does.  Bit 2 set: FMA-contracted exponent (w3do_power).
 * Bit 3 set: the suffix recurrence of the backward walk in its other algebraic form (w3do_lerp).
 * Every variant is a valid fp32 evaluation of the same
 * formulas: the spread between a probe run and the mode-0 run is the uncertainty ANY fp32 implementation has against this
 * oracle — (pixel, Gaussian) pairs whose alpha sits on the 1/255 threshold and pixels whose transmittance sits on 1e-4
 * flip, and ill-conditioned per-Gaussian sums move. */
static int g_exp_mode = 0;
void w3do_set_exp_mode(int mode) { g_exp_mode = mode; }
/* Conditioning of the densification statistic (tests only): when set, w3do_backward adds |term| of every summand of
 * dL/dmean2D.x / .y into abs_out[2g], abs_out[2g+1] (doubles, zeroed by the caller) — with every DIFFERENCE inside a
 * summand replaced by the sum of its operands' magnitudes: (colour - colour behind) becomes |colour| + |colour behind|, the
 * two products G dx conic.x and G dy conic.y are taken separately (which also covers an implementation that sums the
 * moments sum(m dx), sum(m dy) first and combines them with the conic afterwards).  That is the classical running error
 * bound: rounding an operand by 2^-24 moves the summand by at most 2^-24 times this magnitude.  sum|terms| / |sum terms| is the condition number of
 * that Gaussian's sum: an fp32 implementation that rounds each term (relative 2^-24) can be off by ~2^-24 * sum|terms|
 * however it orders the additions — which exceeds 1e-4 of the result where terms cancel. */
static double *g_abs_out = NULL;
void w3do_set_abs_sums(double *abs_out) { g_abs_out = abs_out; }
static inline float w3do_exp(float x) { return (g_exp_mode & 1) ? exp2f(x * 1.4426950408889634f) : expf(x); }
/* the exponent of a (pixel, Gaussian) pair.  Probe bit 2: the same expression with the multiply-adds contracted into FMAs,
 * as a GPU compiler does by default (nvcc -fmad=true for the reference's CUDA build; this file itself is compiled with
 * -ffp-contract=off) */
static inline float w3do_power(const float *co, float dx, float dy) {
    if (g_exp_mode & 4) return fmaf(-0.5f, fmaf(co[2] * dy, dy, co[0] * dx * dx), -(co[1] * dx * dy));
    return -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
}
/* the backward walk's recurrence for what lies behind an entry: a * c + (1 - a) * acc.  Probe bit 3: the algebraically equal
 * acc + a * (c - acc) — other roundings of the same value (the gradient takes differences c - acc, which amplifies them where
 * a Gaussian's colour is close to what lies behind it) */
static inline float w3do_lerp(float acc, float c, float a) {
    if (g_exp_mode & 8) return acc + a * (c - acc);
    return a * c + (1.f - a) * acc;
}
/* one term of a per-Gaussian gradient sum: double accumulator, or (probe bit 1) an fp32 one */
static inline void acc_add(double *a, float *af, float x) {
    if (af) {
#pragma omp atomic
        *af += x;
    } else {
#pragma omp atomic
        *a += (double)x;
    }
}

typedef struct W3DOView {
    int H, W;
    float tanfovx, tanfovy;
    float scale_modifier;
    int sh_degree; /* active degree 0..3 */
    int sh_coeffs; /* coefficients stored per Gaussian (16 for max degree 3) */
    float bg[3];
    float view[16];
    float proj[16];
    float campos[3];
} W3DOView;

typedef struct W3DOState {
    int P, H, W, gx, gy;
    long R;
    float *depth, *xy, *conic_op, *rgb, *cov3D;
    unsigned char *clamped;
    int *radii, *rect;
    uint32_t *ranges;     /* 2*T */
    uint32_t *point_list; /* R */
    float *final_T;
    uint32_t *n_contrib;
} W3DOState;

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* ---------------------------------------------------------------- small helpers */
static void xform4x3(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static void quat_to_R(const float *q, float R[3][3]) {
    /* same formula as reference utils/general_utils.py:78-99 but WITHOUT renormalising:
       the caller already passes unit quaternions (scene/gaussian_model.py:105-107). */
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}
/* Sigma = (R S)(R S)^T, stored [xx,xy,xz,yy,yz,zz] — must equal get_covariance
   (reference scene/gaussian_model.py:27-31). */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *q, float *c) {
    float R[3][3], L[3][3];
    quat_to_R(q, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * s[j];
    c[0] = L[0][0] * L[0][0] + L[0][1] * L[0][1] + L[0][2] * L[0][2];
    c[1] = L[0][0] * L[1][0] + L[0][1] * L[1][1] + L[0][2] * L[1][2];
    c[2] = L[0][0] * L[2][0] + L[0][1] * L[2][1] + L[0][2] * L[2][2];
    c[3] = L[1][0] * L[1][0] + L[1][1] * L[1][1] + L[1][2] * L[1][2];
    c[4] = L[1][0] * L[2][0] + L[1][1] * L[2][1] + L[1][2] * L[2][2];
    c[5] = L[2][0] * L[2][0] + L[2][1] * L[2][1] + L[2][2] * L[2][2];
}
/* T = J W (2x3), the EWA projection Jacobian times the view rotation; also reports whether
   the FoV clamp (1.3 tan) was active on x / y. t is the view-space mean. */
static void ewa_T(const W3DOView *v, const float *t_in, float T[2][3], float *tx_c, float *ty_c,
                  int *clx, int *cly) {
    float fx = (float)v->W / (2.f * v->tanfovx), fy = (float)v->H / (2.f * v->tanfovy);
    float limx = 1.3f * v->tanfovx, limy = 1.3f * v->tanfovy;
    float tz = t_in[2];
    float txtz = t_in[0] / tz, tytz = t_in[1] / tz;
    *clx = (txtz < -limx || txtz > limx);
    *cly = (tytz < -limy || tytz > limy);
    float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    *tx_c = tx;
    *ty_c = ty;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    const float *V = v->view;
    /* W(r,c) = V[4c + r] */
    for (int c = 0; c < 3; c++) {
        T[0][c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
        T[1][c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
    }
}
static void sym6_to_mat(const float *c, float S[3][3]) {
    S[0][0] = c[0]; S[0][1] = c[1]; S[0][2] = c[2];
    S[1][0] = c[1]; S[1][1] = c[3]; S[1][2] = c[4];
    S[2][0] = c[2]; S[2][1] = c[4]; S[2][2] = c[5];
}
static void cov2d_from_T(float T[2][3], const float *cov3D, float *a, float *b, float *c) {
    float S[3][3], TS[2][3];
    sym6_to_mat(cov3D, S);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) TS[i][j] = T[i][0] * S[0][j] + T[i][1] * S[1][j] + T[i][2] * S[2][j];
    *a = TS[0][0] * T[0][0] + TS[0][1] * T[0][1] + TS[0][2] * T[0][2] + 0.3f;
    *b = TS[0][0] * T[1][0] + TS[0][1] * T[1][1] + TS[0][2] * T[1][2];
    *c = TS[1][0] * T[1][0] + TS[1][1] * T[1][1] + TS[1][2] * T[1][2] + 0.3f;
}
static void sh_to_rgb(int deg, int M, const float *sh /* M x 3 */, const float *pos, const float *campos,
                      float *rgb, unsigned char *clamped) {
    float d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / len, y = d[1] / len, z = d[2] / len;
    (void)M;
    for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k)*3 + ch]
        float r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SH(12) +
                        SH_C3[4] * x * (4.f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[ch] = (r < 0.f);
        rgb[ch] = r < 0.f ? 0.f : r;
    }
}

typedef struct {
    uint32_t key; /* depth bits */
    uint32_t g;
} KV;
static int kv_cmp(const void *a, const void *b) {
    const KV *x = (const KV *)a, *y = (const KV *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->g != y->g) return x->g < y->g ? -1 : 1; /* stable sort of index-ordered emission */
    return 0;
}

void w3do_free(void *h) {
    W3DOState *s = (W3DOState *)h;
    if (!s) return;
    free(s->depth); free(s->xy); free(s->conic_op); free(s->rgb); free(s->cov3D); free(s->clamped);
    free(s->radii); free(s->rect); free(s->ranges); free(s->point_list); free(s->final_T); free(s->n_contrib);
    free(s);
}

/* ---------------------------------------------------------------- forward
 * Restates SURVEY.md Appendix A.1-A.3 (+A.6 when gt_mask/used_count are given).
 * Returns an opaque state handle for w3do_backward / the getters; free with w3do_free. */
void *w3do_forward(const W3DOView *v, int P, const float *means3D, const float *shs, const float *colors_precomp,
                   const float *opacities, const float *scales, const float *rotations,
                   const float *cov3D_precomp, float *out_color, float *out_depth, float *out_alpha,
                   int *radii_out, int nthreads,
                   /* FlashSplat extras, all nullable */
                   const float *gt_mask, int num_obj, float *used_count, int *contrib_num, float *proj_xy,
                   float *gs_depth) {
    const int H = v->H, W = v->W;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int T = gx * gy;
    W3DOState *s = (W3DOState *)calloc(1, sizeof(W3DOState));
    s->P = P; s->H = H; s->W = W; s->gx = gx; s->gy = gy;
    s->depth = (float *)calloc((size_t)P + 1, sizeof(float));
    s->xy = (float *)calloc((size_t)P * 2 + 1, sizeof(float));
    s->conic_op = (float *)calloc((size_t)P * 4 + 1, sizeof(float));
    s->rgb = (float *)calloc((size_t)P * 3 + 1, sizeof(float));
    s->cov3D = (float *)calloc((size_t)P * 6 + 1, sizeof(float));
    s->clamped = (unsigned char *)calloc((size_t)P * 3 + 1, 1);
    s->radii = (int *)calloc((size_t)P + 1, sizeof(int));
    s->rect = (int *)calloc((size_t)P * 4 + 1, sizeof(int));
    s->ranges = (uint32_t *)calloc((size_t)T * 2 + 1, sizeof(uint32_t));
    s->final_T = (float *)calloc((size_t)H * W + 1, sizeof(float));
    s->n_contrib = (uint32_t *)calloc((size_t)H * W + 1, sizeof(uint32_t));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif

    /* ---- A.1 preprocess */
#pragma omp parallel for schedule(static)
    for (int g = 0; g < P; g++) {
        s->radii[g] = 0;
        if (proj_xy) { proj_xy[2 * g] = 0.f; proj_xy[2 * g + 1] = 0.f; }
        if (gs_depth) gs_depth[g] = 0.f;
        const float *p = means3D + 3 * (size_t)g;
        float pv[3];
        xform4x3(v->view, p, pv);
        if (pv[2] <= 0.2f) continue; /* near cull */
        float ph[4];
        xform4x4(v->proj, p, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
        float *c3 = s->cov3D + 6 * (size_t)g;
        if (cov3D_precomp) memcpy(c3, cov3D_precomp + 6 * (size_t)g, 6 * sizeof(float));
        else cov3d_from_scale_rot(scales + 3 * (size_t)g, v->scale_modifier, rotations + 4 * (size_t)g, c3);
        float Tm[2][3], txc, tyc; int clx, cly;
        ewa_T(v, pv, Tm, &txc, &tyc, &clx, &cly);
        float a, b, c;
        cov2d_from_T(Tm, c3, &a, &b, &c);
        float det = a * c - b * b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {c * det_inv, -b * det_inv, a * det_inv};
        float mid = 0.5f * (a + c);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float px = ((pp[0] + 1.0f) * (float)W - 1.0f) * 0.5f;
        float py = ((pp[1] + 1.0f) * (float)H - 1.0f) * 0.5f;
        int r = (int)my_radius;
        int minx = (int)((px - (float)r) / (float)TILE), miny = (int)((py - (float)r) / (float)TILE);
        int maxx = (int)((px + (float)r + (float)(TILE - 1)) / (float)TILE);
        int maxy = (int)((py + (float)r + (float)(TILE - 1)) / (float)TILE);
        minx = minx < 0 ? 0 : (minx > gx ? gx : minx);
        miny = miny < 0 ? 0 : (miny > gy ? gy : miny);
        maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
        maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (colors_precomp) {
            for (int ch = 0; ch < 3; ch++) s->rgb[3 * (size_t)g + ch] = colors_precomp[3 * (size_t)g + ch];
        } else {
            sh_to_rgb(v->sh_degree, v->sh_coeffs, shs + (size_t)g * v->sh_coeffs * 3, p, v->campos,
                      s->rgb + 3 * (size_t)g, s->clamped + 3 * (size_t)g);
        }
        s->depth[g] = pv[2];
        s->radii[g] = r;
        s->xy[2 * (size_t)g] = px; s->xy[2 * (size_t)g + 1] = py;
        s->conic_op[4 * (size_t)g + 0] = conic[0];
        s->conic_op[4 * (size_t)g + 1] = conic[1];
        s->conic_op[4 * (size_t)g + 2] = conic[2];
        s->conic_op[4 * (size_t)g + 3] = opacities[g];
        s->rect[4 * (size_t)g + 0] = minx; s->rect[4 * (size_t)g + 1] = miny;
        s->rect[4 * (size_t)g + 2] = maxx; s->rect[4 * (size_t)g + 3] = maxy;
        if (proj_xy) { proj_xy[2 * g] = px; proj_xy[2 * g + 1] = py; }
        if (gs_depth) gs_depth[g] = pv[2];
    }
    memcpy(radii_out, s->radii, (size_t)P * sizeof(int));

    /* ---- A.2 binning: per-tile lists ordered by (depth bits, Gaussian index) */
    uint32_t *count = (uint32_t *)calloc((size_t)T + 1, sizeof(uint32_t));
    for (int g = 0; g < P; g++) {
        if (s->radii[g] <= 0) continue;
        const int *rc = s->rect + 4 * (size_t)g;
        for (int ty = rc[1]; ty < rc[3]; ty++)
            for (int tx = rc[0]; tx < rc[2]; tx++) count[ty * gx + tx]++;
    }
    long R = 0;
    for (int t = 0; t < T; t++) { s->ranges[2 * t] = (uint32_t)R; R += count[t]; s->ranges[2 * t + 1] = (uint32_t)R; }
    s->R = R;
    KV *kv = (KV *)malloc(((size_t)R + 1) * sizeof(KV));
    memset(count, 0, (size_t)T * sizeof(uint32_t));
    for (int g = 0; g < P; g++) {
        if (s->radii[g] <= 0) continue;
        const int *rc = s->rect + 4 * (size_t)g;
        uint32_t key; memcpy(&key, &s->depth[g], 4);
        for (int ty = rc[1]; ty < rc[3]; ty++)
            for (int tx = rc[0]; tx < rc[2]; tx++) {
                int t = ty * gx + tx;
                KV *e = kv + s->ranges[2 * t] + count[t]++;
                e->key = key; e->g = (uint32_t)g;
            }
    }
    free(count);
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < T; t++) {
        uint32_t b = s->ranges[2 * t], e = s->ranges[2 * t + 1];
        if (e - b > 1) qsort(kv + b, e - b, sizeof(KV), kv_cmp);
    }
    s->point_list = (uint32_t *)malloc(((size_t)R + 1) * sizeof(uint32_t));
    for (long i = 0; i < R; i++) s->point_list[i] = kv[i].g;
    free(kv);

    /* ---- A.3 blend forward (+A.6 contribution scatter) */
    if (used_count) memset(used_count, 0, (size_t)(num_obj + 1) * P * sizeof(float));
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < T; t++) {
        int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        uint32_t b = s->ranges[2 * t], e = s->ranges[2 * t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                float Tr = 1.0f, C[3] = {0, 0, 0}, D = 0.f, A = 0.f;
                uint32_t contributor = 0, last = 0;
                int ncontrib_accepted = 0;
                float pxf = (float)px, pyf = (float)py;
                int label = 0;
                if (gt_mask) label = (int)gt_mask[(size_t)py * W + px];
                for (uint32_t i = b; i < e; i++) {
                    contributor++;
                    uint32_t g = s->point_list[i];
                    float dx = s->xy[2 * (size_t)g] - pxf, dy = s->xy[2 * (size_t)g + 1] - pyf;
                    const float *co = s->conic_op + 4 * (size_t)g;
                    float power = w3do_power(co, dx, dy);
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, co[3] * w3do_exp(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = Tr * (1.f - alpha);
                    if (test_T < 0.0001f) break; /* this entry is NOT applied */
                    float w = alpha * Tr;
                    for (int ch = 0; ch < 3; ch++) C[ch] += s->rgb[3 * (size_t)g + ch] * w;
                    D += s->depth[g] * w;
                    A += w;
                    if (used_count && gt_mask && label >= 0 && label <= num_obj) {
#pragma omp atomic
                        used_count[(size_t)label * P + g] += w;
                    }
                    ncontrib_accepted++;
                    Tr = test_T;
                    last = contributor;
                }
                size_t pix = (size_t)py * W + px;
                s->final_T[pix] = Tr;
                s->n_contrib[pix] = last;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + Tr * v->bg[ch];
                out_depth[pix] = D;
                out_alpha[pix] = A;
                if (contrib_num) contrib_num[pix] = ncontrib_accepted;
            }
    }
    return s;
}

/* ---------------------------------------------------------------- attribution of differences (tests only)
 * The blend is threshold-laden: a (pixel, Gaussian) pair is skipped when power > 0 or alpha < 1/255, and a pixel stops at
 * the first entry that would take its transmittance below 1e-4.  A pair that sits ON one of these thresholds can fall on
 * either side in another fp32 evaluation of the same formulas; when it does, that pixel's contributor set changes and with it
 * the weights of every later entry (T is multiplied by 1 - alpha >= ... of the flipped entry) and the colour composited
 * behind every earlier one — so every Gaussian blended at that pixel moves.
 *
 * w3do_fragile_pixels: out[pix] = 1 where the oracle's own walk of the pixel meets a pair with
 *     |ln(alpha * 255)| <= tol,  or  |test_T / 1e-4 - 1| <= eps + tol_T,  or  |power| <= tol (the power > 0 skip),
 * where tol = eps + 2e-6 * (0.5 (|A| dx^2 + |C| dy^2) + |B dx dy|): "on the threshold" is measured in units of what TWO fp32
 * evaluations of the exponent can differ by — a few ulps of the sum of its terms' magnitudes, which for a needle-shaped Gaussian
 * (conic condition 1e4 ... 1e7) hundreds of pixels from its centre is 0.01 ... 1, not 1e-7 — and tol_T accumulates what that
 * does to the transmittance (sum of alpha / (1 - alpha) times the pair's exponent noise).  For ordinary Gaussians tol = eps.
 * w3do_mark_contributors: flags[g] = 1 for every Gaussian that is BLENDED (applied) at a pixel with pixel_flags != 0. */
void w3do_fragile_pixels(void *h, float eps, unsigned char *out) {
    W3DOState *s = (W3DOState *)h;
    const int H = s->H, W = s->W, gx = s->gx, T = s->gx * s->gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < T; t++) {
        int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        uint32_t b = s->ranges[2 * t], e = s->ranges[2 * t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                float Tr = 1.0f, pxf = (float)px, pyf = (float)py, tol_T = 0.0f;
                unsigned char frag = 0;
                for (uint32_t i = b; i < e; i++) {
                    uint32_t g = s->point_list[i];
                    float dx = s->xy[2 * (size_t)g] - pxf, dy = s->xy[2 * (size_t)g + 1] - pyf;
                    const float *co = s->conic_op + 4 * (size_t)g;
                    float power = w3do_power(co, dx, dy);
                    const float noise = 2e-6f * (0.5f * (fabsf(co[0]) * dx * dx + fabsf(co[2]) * dy * dy) + fabsf(co[1] * dx * dy));
                    const float tol = eps + noise;
                    if (fabsf(power) <= tol && co[3] >= 1.0f / 255.0f) frag = 1;
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, co[3] * w3do_exp(power));
                    if (alpha > 0.0f && fabsf(logf(alpha * 255.0f)) <= tol) frag = 1;
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = Tr * (1.f - alpha);
                    tol_T += noise * alpha / (1.f - alpha);
                    if (fabsf(test_T * 10000.0f - 1.0f) <= eps + tol_T) frag = 1;
                    if (test_T < 0.0001f) break;
                    Tr = test_T;
                }
                out[(size_t)py * W + px] = frag;
            }
    }
}

void w3do_mark_contributors(void *h, const unsigned char *pixel_flags, unsigned char *flags) {
    W3DOState *s = (W3DOState *)h;
    const int H = s->H, W = s->W, gx = s->gx, T = s->gx * s->gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < T; t++) {
        int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        uint32_t b = s->ranges[2 * t];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                size_t pix = (size_t)py * W + px;
                if (!pixel_flags[pix]) continue;
                float pxf = (float)px, pyf = (float)py;
                /* everything up to the last contributor that passes the skips is blended (the walk the backward makes);
                 * entries behind it are marked too when they pass the skips: a termination flip would blend them */
                uint32_t e = s->ranges[2 * t + 1];
                for (uint32_t i = b; i < e; i++) {
                    uint32_t g = s->point_list[i];
                    float dx = s->xy[2 * (size_t)g] - pxf, dy = s->xy[2 * (size_t)g + 1] - pyf;
                    const float *co = s->conic_op + 4 * (size_t)g;
                    float power = w3do_power(co, dx, dy);
                    /* (same exponent-noise window as w3do_fragile_pixels) */
                    const float noise = 2e-6f * (0.5f * (fabsf(co[0]) * dx * dx + fabsf(co[2]) * dy * dy) + fabsf(co[1] * dx * dy));
                    if (power > 1e-3f + noise) continue;
                    float alpha = fminf(0.99f, co[3] * w3do_exp(fminf(power, 0.0f)));
                    if (alpha * w3do_exp(fminf(noise, 80.0f)) < 0.99f / 255.0f) continue;
                    flags[g] = 1;           /* (benign race: every writer stores 1) */
                    if (i - b >= s->n_contrib[pix] + 64u) break;   /* far behind the stop: cannot be reached by a flip */
                }
            }
    }
}

long w3do_num_rendered(void *h) { return ((W3DOState *)h)->R; }
void w3do_get_binning(void *h, uint32_t *ranges, uint32_t *point_list) {
    W3DOState *s = (W3DOState *)h;
    memcpy(ranges, s->ranges, (size_t)s->gx * s->gy * 2 * sizeof(uint32_t));
    memcpy(point_list, s->point_list, (size_t)s->R * sizeof(uint32_t));
}
void w3do_get_geom(void *h, float *depth, float *xy, float *conic_op, float *rgb, float *cov3D, int *rect,
                   unsigned char *clamped) {
    W3DOState *s = (W3DOState *)h;
    size_t P = (size_t)s->P;
    if (depth) memcpy(depth, s->depth, P * 4);
    if (xy) memcpy(xy, s->xy, P * 8);
    if (conic_op) memcpy(conic_op, s->conic_op, P * 16);
    if (rgb) memcpy(rgb, s->rgb, P * 12);
    if (cov3D) memcpy(cov3D, s->cov3D, P * 24);
    if (rect) memcpy(rect, s->rect, P * 16);
    if (clamped) memcpy(clamped, s->clamped, P * 3);
}
void w3do_get_pixel_state(void *h, float *final_T, uint32_t *n_contrib) {
    W3DOState *s = (W3DOState *)h;
    size_t n = (size_t)s->H * s->W;
    if (final_T) memcpy(final_T, s->final_T, n * 4);
    if (n_contrib) memcpy(n_contrib, s->n_contrib, n * 4);
}

/* ---------------------------------------------------------------- backward
 * Restates SURVEY.md Appendix A.4-A.5.  dL_ddepth / dL_dalpha may be NULL (Wheat-3DGS's loss
 * uses colour only, train_vanilla_3dgs.py:74-80).  Output arrays are overwritten.
 * dL_dmeans2D is (P,3) with the third column 0 and the first two scaled by W/2, H/2 —
 * the quantity add_densification_stats reads (scene/gaussian_model.py:462). */
void w3do_backward(void *h, const W3DOView *v, const float *means3D, const float *shs,
                   const float *colors_precomp, const float *opacities, const float *scales,
                   const float *rotations, const float *cov3D_precomp, const float *dL_dcolor,
                   const float *dL_ddepth, const float *dL_dalpha_px, float *dL_dmeans3D, float *dL_dmeans2D,
                   float *dL_dcolors, float *dL_dshs, float *dL_dopacity, float *dL_dscales,
                   float *dL_drots, float *dL_dcov3D, int nthreads) {
    W3DOState *s = (W3DOState *)h;
    const int P = s->P, H = s->H, W = s->W, gx = s->gx, gy = s->gy, T = gx * gy;
    (void)opacities; (void)cov3D_precomp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    /* double accumulators: [mean2D.x, mean2D.y, conic.x, conic.y(half), conic.z, opacity, r, g, b, depth] */
    enum { NA = 10 };
    double *acc = (double *)calloc((size_t)P * NA + 1, sizeof(double));
    float *accf = (g_exp_mode & 2) ? (float *)calloc((size_t)P * NA + 1, sizeof(float)) : NULL;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    /* ---- A.4 blend backward: reverse walk per pixel */
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < T; t++) {
        int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        uint32_t b = s->ranges[2 * t], e = s->ranges[2 * t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                size_t pix = (size_t)py * W + px;
                const float T_final = s->final_T[pix];
                float Tr = T_final;
                const uint32_t last_contributor = s->n_contrib[pix];
                float dLdp[3] = {dL_dcolor[pix], dL_dcolor[(size_t)H * W + pix], dL_dcolor[2 * (size_t)H * W + pix]};
                float dLdd = dL_ddepth ? dL_ddepth[pix] : 0.f;
                float dLda = dL_dalpha_px ? dL_dalpha_px[pix] : 0.f;
                float accum_rec[3] = {0, 0, 0}, accum_d = 0.f, accum_a = 0.f;
                float last_alpha = 0.f, last_color[3] = {0, 0, 0}, last_depth = 0.f;
                float pxf = (float)px, pyf = (float)py;
                float bg_dot = v->bg[0] * dLdp[0] + v->bg[1] * dLdp[1] + v->bg[2] * dLdp[2];
                for (uint32_t k = last_contributor; k-- > 0;) {
                    uint32_t g = s->point_list[b + k];
                    (void)e;
                    float dx = s->xy[2 * (size_t)g] - pxf, dy = s->xy[2 * (size_t)g + 1] - pyf;
                    const float *co = s->conic_op + 4 * (size_t)g;
                    float power = w3do_power(co, dx, dy);
                    if (power > 0.0f) continue;
                    float G = w3do_exp(power);
                    float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    Tr = Tr / (1.f - alpha);
                    float dch = alpha * Tr;
                    float dL_dalpha = 0.f;
                    /* running magnitude of dL_dalpha with every difference replaced by the sum of its operands' magnitudes
                     * (tests only, g_abs_out): what the rounding of the operands can move the term by */
                    double mag_dalpha = 0.0;
                    double *a = acc + (size_t)g * NA;
                    float *af = accf ? accf + (size_t)g * NA : NULL;
#define ACC(i, x) acc_add(a + (i), af ? af + (i) : NULL, (x))
                    for (int ch = 0; ch < 3; ch++) {
                        float c = s->rgb[3 * (size_t)g + ch];
                        accum_rec[ch] = w3do_lerp(accum_rec[ch], last_color[ch], last_alpha);
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dLdp[ch];
                        mag_dalpha += (fabs((double)c) + fabs((double)accum_rec[ch])) * fabs((double)dLdp[ch]);
                        ACC(6 + ch, dch * dLdp[ch]);
                    }
                    {
                        float cd = s->depth[g];
                        accum_d = w3do_lerp(accum_d, last_depth, last_alpha);
                        last_depth = cd;
                        dL_dalpha += (cd - accum_d) * dLdd;
                        ACC(9, dch * dLdd);
                        accum_a = w3do_lerp(accum_a, 1.0f, last_alpha);
                        dL_dalpha += (1.0f - accum_a) * dLda;
                        mag_dalpha += (fabs((double)cd) + fabs((double)accum_d)) * fabs((double)dLdd) + (1.0 + fabs((double)accum_a)) * fabs((double)dLda);
                    }
                    dL_dalpha *= Tr;
                    mag_dalpha *= Tr;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    mag_dalpha += fabs((double)(T_final / (1.f - alpha)) * bg_dot);
                    float dL_dG = co[3] * dL_dalpha;
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    float dG_ddely = -gdy * co[2] - gdx * co[1];
                    ACC(0, dL_dG * dG_ddelx * ddelx_dx);
                    ACC(1, dL_dG * dG_ddely * ddely_dy);
                    if (g_abs_out) {
                        /* ... and the exponent: power is a difference of products of size |conic| d^2, so its rounding error —
                         * a RELATIVE error of G, hence of the whole summand — is 2^-24 times their magnitude, not times
                         * |power| (thin, rotated footprints: large terms that cancel) */
                        const double pmag = 0.5 * (fabs((double)co[0]) * dx * dx + fabs((double)co[2]) * dy * dy) + fabs((double)co[1] * dx * dy);
                        const double mG = (double)co[3] * mag_dalpha * (1.0 + pmag);
#pragma omp atomic
                        g_abs_out[2 * (size_t)g] += (fabs((double)gdx * co[0]) + fabs((double)gdy * co[1])) * mG * ddelx_dx;
#pragma omp atomic
                        g_abs_out[2 * (size_t)g + 1] += (fabs((double)gdy * co[2]) + fabs((double)gdx * co[1])) * mG * ddely_dy;
                    }
                    ACC(2, -0.5f * gdx * dx * dL_dG);
                    ACC(3, -0.5f * gdx * dy * dL_dG);
                    ACC(4, -0.5f * gdy * dy * dL_dG);
                    ACC(5, G * dL_dalpha);
                }
            }
    }

#undef ACC
    if (accf) {
        for (size_t i = 0; i < (size_t)P * NA; i++) acc[i] = (double)accf[i];
        free(accf);
    }

    /* ---- A.5 preprocess backward */
#pragma omp parallel for schedule(static)
    for (int g = 0; g < P; g++) {
        float *gm3 = dL_dmeans3D + 3 * (size_t)g;
        gm3[0] = gm3[1] = gm3[2] = 0.f;
        float *gm2 = dL_dmeans2D + 3 * (size_t)g;
        gm2[0] = gm2[1] = gm2[2] = 0.f;
        dL_dopacity[g] = 0.f;
        if (dL_dcolors) for (int k = 0; k < 3; k++) dL_dcolors[3 * (size_t)g + k] = 0.f;
        if (dL_dshs) memset(dL_dshs + (size_t)g * v->sh_coeffs * 3, 0, (size_t)v->sh_coeffs * 3 * sizeof(float));
        if (dL_dscales) for (int k = 0; k < 3; k++) dL_dscales[3 * (size_t)g + k] = 0.f;
        if (dL_drots) for (int k = 0; k < 4; k++) dL_drots[4 * (size_t)g + k] = 0.f;
        if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)g + k] = 0.f;
        if (s->radii[g] <= 0) continue;
        const double *a = acc + (size_t)g * NA;
        const float *p = means3D + 3 * (size_t)g;
        float dmean2D[2] = {(float)a[0], (float)a[1]};
        float dconic[3] = {(float)a[2], (float)a[3], (float)a[4]};
        float dcol[3] = {(float)a[6], (float)a[7], (float)a[8]};
        float ddepth = (float)a[9];
        gm2[0] = dmean2D[0]; gm2[1] = dmean2D[1];
        dL_dopacity[g] = (float)a[5];

        /* (i) conic -> cov2D -> cov3D and the view-space mean (through J) */
        float pv[3];
        xform4x3(v->view, p, pv);
        float Tm[2][3], txc, tyc; int clx, cly;
        ewa_T(v, pv, Tm, &txc, &tyc, &clx, &cly);
        const float *c3 = s->cov3D + 6 * (size_t)g;
        float ca, cb, cc;
        cov2d_from_T(Tm, c3, &ca, &cb, &cc);
        float denom = ca * cc - cb * cb;
        float denom2inv = 1.0f / (denom * denom + 0.0000001f);
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float dcov[6] = {0, 0, 0, 0, 0, 0};
        float dmean[3] = {0, 0, 0};
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * dconic[0] + 2.f * cb * cc * dconic[1] + (denom - ca * cc) * dconic[2]);
            dL_dc = denom2inv * (-ca * ca * dconic[2] + 2.f * ca * cb * dconic[1] + (denom - ca * cc) * dconic[0]);
            dL_db = denom2inv * 2.f * (cb * cc * dconic[0] - (denom + 2.f * cb * cb) * dconic[1] + ca * cb * dconic[2]);
            /* dL/dSigma3 = T^T [[da, db/2],[db/2, dc]] T ; off-diagonals doubled in the 6-vector */
            dcov[0] = Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc;
            dcov[3] = Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc;
            dcov[5] = Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc;
            dcov[1] = 2.f * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db +
                      2.f * Tm[1][0] * Tm[1][1] * dL_dc;
            dcov[2] = 2.f * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db +
                      2.f * Tm[1][0] * Tm[1][2] * dL_dc;
            dcov[4] = 2.f * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db +
                      2.f * Tm[1][1] * Tm[1][2] * dL_dc;
        }
        /* dL/dT = 2 G T Sigma, G = [[da, db/2],[db/2, dc]] */
        float S[3][3];
        sym6_to_mat(c3, S);
        float TS[2][3];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) TS[i][j] = Tm[i][0] * S[0][j] + Tm[i][1] * S[1][j] + Tm[i][2] * S[2][j];
        float dT[2][3];
        for (int j = 0; j < 3; j++) {
            dT[0][j] = 2.f * TS[0][j] * dL_da + TS[1][j] * dL_db;
            dT[1][j] = 2.f * TS[1][j] * dL_dc + TS[0][j] * dL_db;
        }
        const float *V = v->view;
        /* T = J W  =>  dL/dJ(i,k) = sum_j dT(i,j) W(k,j), W(k,j) = V[4j+k] */
        float dJ00 = dT[0][0] * V[0] + dT[0][1] * V[4] + dT[0][2] * V[8];
        float dJ02 = dT[0][0] * V[2] + dT[0][1] * V[6] + dT[0][2] * V[10];
        float dJ11 = dT[1][0] * V[1] + dT[1][1] * V[5] + dT[1][2] * V[9];
        float dJ12 = dT[1][0] * V[2] + dT[1][1] * V[6] + dT[1][2] * V[10];
        float fx = (float)W / (2.f * v->tanfovx), fy = (float)H / (2.f * v->tanfovy);
        float tz = 1.f / pv[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = (clx ? 0.f : 1.f) * -fx * tz2 * dJ02;
        float dty = (cly ? 0.f : 1.f) * -fy * tz2 * dJ12;
        float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * txc) * tz3 * dJ02 + (2.f * fy * tyc) * tz3 * dJ12;
        dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;

        /* (ii) 2-D mean -> 3-D mean through the perspective divide */
        const float *M = v->proj;
        float mh[4];
        xform4x4(M, p, mh);
        float mw = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        dmean[0] += (M[0] * mw - M[3] * mul1) * dmean2D[0] + (M[1] * mw - M[3] * mul2) * dmean2D[1];
        dmean[1] += (M[4] * mw - M[7] * mul1) * dmean2D[0] + (M[5] * mw - M[7] * mul2) * dmean2D[1];
        dmean[2] += (M[8] * mw - M[11] * mul1) * dmean2D[0] + (M[9] * mw - M[11] * mul2) * dmean2D[1];

        /* (iii) depth output -> 3-D mean through the z row of the view matrix */
        dmean[0] += V[2] * ddepth; dmean[1] += V[6] * ddepth; dmean[2] += V[10] * ddepth;

        /* (iv) colour: either straight to colors_precomp or through the SH basis */
        if (colors_precomp) {
            if (dL_dcolors) for (int k = 0; k < 3; k++) dL_dcolors[3 * (size_t)g + k] = dcol[k];
        } else if (shs) {
            const int Mc = v->sh_coeffs, deg = v->sh_degree;
            const float *sh = shs + (size_t)g * Mc * 3;
            float *dsh = dL_dshs + (size_t)g * Mc * 3;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = s->clamped[3 * (size_t)g + ch] ? 0.f : dcol[ch];
            float d0[3] = {p[0] - v->campos[0], p[1] - v->campos[1], p[2] - v->campos[2]};
            float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            float x = d0[0] / len, y = d0[1] / len, z = d0[2] / len;
            float ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k)*3 + ch]
#define DSH(k) dsh[(k)*3 + ch]
                float gch = dRGB[ch];
                float ddx = 0, ddy = 0, ddz = 0;
                DSH(0) = SH_C0 * gch;
                if (deg > 0) {
                    DSH(1) = -SH_C1 * y * gch; DSH(2) = SH_C1 * z * gch; DSH(3) = -SH_C1 * x * gch;
                    ddx = -SH_C1 * SH(3); ddy = -SH_C1 * SH(1); ddz = SH_C1 * SH(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = SH_C2[0] * xy * gch; DSH(5) = SH_C2[1] * yz * gch;
                        DSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * gch;
                        DSH(7) = SH_C2[3] * xz * gch; DSH(8) = SH_C2[4] * (xx - yy) * gch;
                        ddx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                        ddy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                        ddz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (deg > 2) {
                            DSH(9) = SH_C3[0] * y * (3.f * xx - yy) * gch; DSH(10) = SH_C3[1] * xy * z * gch;
                            DSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * gch;
                            DSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gch;
                            DSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * gch;
                            DSH(14) = SH_C3[5] * z * (xx - yy) * gch; DSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * gch;
                            ddx += SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                                   SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy);
                            ddy += SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                   SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                   SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy;
                            ddz += SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                   SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                   SH_C3[5] * SH(14) * (xx - yy);
                        }
                    }
                }
#undef SH
#undef DSH
                ddir[0] += ddx * gch; ddir[1] += ddy * gch; ddir[2] += ddz * gch;
            }
            /* d normalize(v)/dv applied to ddir */
            float sum2 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
            float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - d0[0] * d0[0]) * ddir[0] - d0[1] * d0[0] * ddir[1] - d0[2] * d0[0] * ddir[2]) * inv32;
            dmean[1] += (-d0[0] * d0[1] * ddir[0] + (sum2 - d0[1] * d0[1]) * ddir[1] - d0[2] * d0[1] * ddir[2]) * inv32;
            dmean[2] += (-d0[0] * d0[2] * ddir[0] - d0[1] * d0[2] * ddir[1] + (sum2 - d0[2] * d0[2]) * ddir[2]) * inv32;
        }
        gm3[0] = dmean[0]; gm3[1] = dmean[1]; gm3[2] = dmean[2];

        /* (v) cov3D -> scale, rotation (gradient w.r.t. the quaternion AS GIVEN) */
        if (cov3D_precomp) {
            if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)g + k] = dcov[k];
        } else if (scales) {
            if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)g + k] = dcov[k];
            const float *q = rotations + 4 * (size_t)g;
            float R[3][3], L[3][3];
            quat_to_R(q, R);
            float mod = v->scale_modifier;
            float sc[3] = {mod * scales[3 * (size_t)g], mod * scales[3 * (size_t)g + 1], mod * scales[3 * (size_t)g + 2]};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * sc[j];
            float Gs[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                              {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                              {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dLm[3][3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    dLm[i][j] = 2.f * (Gs[i][0] * L[0][j] + Gs[i][1] * L[1][j] + Gs[i][2] * L[2][j]);
            for (int j = 0; j < 3; j++)
                dL_dscales[3 * (size_t)g + j] = mod * (dLm[0][j] * R[0][j] + dLm[1][j] * R[1][j] + dLm[2][j] * R[2][j]);
            float GR[3][3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) GR[i][j] = dLm[i][j] * sc[j];
            float r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * (size_t)g + 0] = 2.f * (-z * GR[0][1] + y * GR[0][2] + z * GR[1][0] - x * GR[1][2] - y * GR[2][0] + x * GR[2][1]);
            dL_drots[4 * (size_t)g + 1] = 2.f * (y * GR[0][1] + z * GR[0][2] + y * GR[1][0] - 2.f * x * GR[1][1] - r * GR[1][2] + z * GR[2][0] + r * GR[2][1] - 2.f * x * GR[2][2]);
            dL_drots[4 * (size_t)g + 2] = 2.f * (-2.f * y * GR[0][0] + x * GR[0][1] + r * GR[0][2] + x * GR[1][0] + z * GR[1][2] - r * GR[2][0] + z * GR[2][1] - 2.f * y * GR[2][2]);
            dL_drots[4 * (size_t)g + 3] = 2.f * (-2.f * z * GR[0][0] - r * GR[0][1] + x * GR[0][2] + r * GR[1][0] - 2.f * z * GR[1][1] + y * GR[1][2] + x * GR[2][0] + y * GR[2][1]);
        }
    }
    free(acc);
    (void)gy;
}

/* ---------------------------------------------------------------- distCUDA2 (A.7)
 * mean of the squared distances to the 3 nearest OTHER points; brute force. */
void w3do_knn_dist2(int N, const float *pts, float *out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        float best[3] = {INFINITY, INFINITY, INFINITY};
        const float *p = pts + 3 * (size_t)i;
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const float *q = pts + 3 * (size_t)j;
            float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            float d = dx * dx + dy * dy + dz * dz;
            if (d < best[2]) {
                best[2] = d;
                if (best[2] < best[1]) { float tmp = best[1]; best[1] = best[2]; best[2] = tmp; }
                if (best[1] < best[0]) { float tmp = best[0]; best[0] = best[1]; best[1] = tmp; }
            }
        }
        int n = N - 1 < 3 ? N - 1 : 3;
        float sum = 0.f;
        for (int k = 0; k < n; k++) sum += best[k];
        out[i] = n > 0 ? sum / 3.0f : 0.f;
    }
}
