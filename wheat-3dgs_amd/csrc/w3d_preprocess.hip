// w3d_preprocess.hip — per-Gaussian stages of the rasterizer, forward and backward.
//
// Forward (replaces the preprocess stage reached through
// reference gaussian_renderer/__init__.py:89-97): near-plane cull, projection, 3-D covariance
// from scale/quaternion (or precomputed), EWA 2-D covariance, conic, radius, tile rectangle,
// SH -> RGB.  One thread per Gaussian, reads coalesced across the wave; writes the packed
// per-Gaussian records of the state buffer and the depth key / id pair the sort consumes.
//
// Backward (Appendix A.5): 2-D gradient record -> means3D, SH | colours, scale + quaternion | cov3D,
// opacity, and the screen-space gradient used by densification.
//
// The geometric part of the forward is compiled with FP contraction OFF and written in one
// fixed operation order so that every integer derived from it (radii, tile rectangles, depth
// keys and therefore the per-tile order) is reproducible bit for bit against the CPU oracle.
#include "w3d_common.h"

namespace {

__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

struct Cam {
    float V[16], M[16], campos[3];
};

__device__ __forceinline__ void load_cam(const w3d_view &v, Cam &c) {
#pragma unroll
    for (int i = 0; i < 16; i++) { c.V[i] = v.viewmatrix[i]; c.M[i] = v.projmatrix[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) c.campos[i] = v.campos[i];
}

// ---- exact-order geometry (contraction off) -------------------------------------------------
struct Geo {
    float depth, px, py;
    float cov3D[6];
    float Tm[2][3];
    float a, b, c;       // 2-D covariance incl. the 0.3 dilation
    float txc, tyc;      // clamped view-space x,y
    bool clx, cly;
};

__device__ __forceinline__ void quat_to_R(const float *q, float R[3][3]) {
#pragma clang fp contract(off)
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

__device__ __forceinline__ void cov3d_from_scale_rot(const float *scale, float mod, const float *q, float *c) {
#pragma clang fp contract(off)
    float R[3][3], L[3][3];
    quat_to_R(q, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * s[j];
    c[0] = L[0][0] * L[0][0] + L[0][1] * L[0][1] + L[0][2] * L[0][2];
    c[1] = L[0][0] * L[1][0] + L[0][1] * L[1][1] + L[0][2] * L[1][2];
    c[2] = L[0][0] * L[2][0] + L[0][1] * L[2][1] + L[0][2] * L[2][2];
    c[3] = L[1][0] * L[1][0] + L[1][1] * L[1][1] + L[1][2] * L[1][2];
    c[4] = L[1][0] * L[2][0] + L[1][1] * L[2][1] + L[1][2] * L[2][2];
    c[5] = L[2][0] * L[2][0] + L[2][1] * L[2][1] + L[2][2] * L[2][2];
}

// T = J W and the dilated 2-D covariance.  pv = view-space mean.
__device__ __forceinline__ void ewa(const w3d_view &v, const Cam &cam, const float *pv, const float *cov3D, Geo &g) {
#pragma clang fp contract(off)
    float fx = (float)v.image_width / (2.f * v.tanfovx), fy = (float)v.image_height / (2.f * v.tanfovy);
    float limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
    float tz = pv[2];
    float txtz = pv[0] / tz, tytz = pv[1] / tz;
    g.clx = (txtz < -limx || txtz > limx);
    g.cly = (tytz < -limy || tytz > limy);
    float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    g.txc = tx; g.tyc = ty;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    const float *V = cam.V;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        g.Tm[0][c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
        g.Tm[1][c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
    }
    float S[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    float TS[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) TS[i][j] = g.Tm[i][0] * S[0][j] + g.Tm[i][1] * S[1][j] + g.Tm[i][2] * S[2][j];
    g.a = TS[0][0] * g.Tm[0][0] + TS[0][1] * g.Tm[0][1] + TS[0][2] * g.Tm[0][2] + 0.3f;
    g.b = TS[0][0] * g.Tm[1][0] + TS[0][1] * g.Tm[1][1] + TS[0][2] * g.Tm[1][2];
    g.c = TS[1][0] * g.Tm[1][0] + TS[1][1] * g.Tm[1][1] + TS[1][2] * g.Tm[1][2] + 0.3f;
}

__device__ __forceinline__ void xform4x3(const float *m, const float *p, float *o) {
#pragma clang fp contract(off)
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform4x4(const float *m, const float *p, float *o) {
#pragma clang fp contract(off)
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// 4-byte-aligned 16-byte vector (gfx950 handles unaligned dwordx4 global accesses)
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
#define W3D_SH_CHUNKS 720   // 16-B chunks in 64 f_rest rows of 45 floats
#ifndef W3D_SH_ROUNDS
#define W3D_SH_ROUNDS 2     // preprocess forward: the 64 rows of a wave staged through LDS in this many rounds (2: 32 rows at a time)
#endif
#ifndef W3D_PRE_OCC
#define W3D_PRE_OCC 4       // ... and the waves per SIMD it is compiled for.  One round needs 46 KB of LDS per workgroup AND 129 VGPRs
                            // (twelve 16-B chunks in flight per lane): 3 waves per SIMD either way — round 5 halved only the LDS and
                            // measured nothing.  Two rounds: 23 KB, 101 VGPRs, 4 waves: 120.8 -> 112.5 us (same-box A/B,
                            // profiles/r06/ab_preprocess_fwd.txt; 5 waves spill 50 registers: 117 us, 6 waves 242: 183 us)
#endif

// Loads the active SH coefficients of one Gaussian into c[3k + ch].
//   interleaved layout: sh -> this Gaussian's (M,3) block (reference get_features layout);
//   split layout (raw path): dc -> (1,3) block, rest -> ((M-1),3) block, as GaussianModel stores them.
__device__ __forceinline__ void load_sh(int deg, const float *__restrict__ sh, const float *__restrict__ dc,
                                        const float *__restrict__ rest, float *c) {
    const int ncoef = (deg + 1) * (deg + 1);
    if (sh) {
        const int nvec = (ncoef * 3 + 3) / 4;
        if ((reinterpret_cast<uintptr_t>(sh) & 15) == 0) {
            const float4 *sh4 = reinterpret_cast<const float4 *>(sh);
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (i < nvec) { float4 t = sh4[i]; c[4 * i] = t.x; c[4 * i + 1] = t.y; c[4 * i + 2] = t.z; c[4 * i + 3] = t.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 48; i++)
                if (i < ncoef * 3) c[i] = sh[i];
        }
    } else {
        c[0] = dc[0]; c[1] = dc[1]; c[2] = dc[2];
        const int nrest = (ncoef - 1) * 3;            // 0, 9, 24 or 45 floats
        const F4U *r4 = reinterpret_cast<const F4U *>(rest);
#pragma unroll
        for (int i = 0; i < 11; i++)
            if (4 * i + 3 < nrest) { F4U t = r4[i]; c[3 + 4 * i] = t.x; c[4 + 4 * i] = t.y; c[5 + 4 * i] = t.z; c[6 + 4 * i] = t.w; }
#pragma unroll
        for (int i = 0; i < 45; i++)
            if (i >= (nrest & ~3) && i < nrest) c[3 + i] = rest[i];
    }
}

// SH -> RGB (+0.5, clamp at 0) from preloaded coefficients c[3k + ch].
__device__ __forceinline__ void sh_to_rgb(int deg, const float *c, const float *pos, const float *campos,
                                          float *rgb, uint32_t &clamped) {
#pragma clang fp contract(off)
    float d0 = pos[0] - campos[0], d1 = pos[1] - campos[1], d2 = pos[2] - campos[2];
    float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    float x = d0 / len, y = d1 / len, z = d2 / len;
    float r[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
#define SH(k) c[(k)*3 + ch]
        float res = SH_C0 * SH(0);
        if (deg > 0) {
            res = res - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.f * zz - xx - yy) * SH(6) +
                      SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                          SH_C3[2] * y * (4.f * zz - xx - yy) * SH(11) +
                          SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SH(12) +
                          SH_C3[4] * x * (4.f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                          SH_C3[6] * x * (xx - 3.f * yy) * SH(15);
                }
            }
        }
#undef SH
        res += 0.5f;
        r[ch] = res;
    }
    clamped = (r[0] < 0.f ? 1u : 0u) | (r[1] < 0.f ? 2u : 0u) | (r[2] < 0.f ? 4u : 0u);
    rgb[0] = fmaxf(r[0], 0.f); rgb[1] = fmaxf(r[1], 0.f); rgb[2] = fmaxf(r[2], 0.f);
}

// ---------------------------------------------------------------------------------------------
// Exact footprint culling (tile_cull): which tiles of the rect can reach alpha >= 1/255 for this Gaussian at
// ANY pixel centre?  alpha = o*exp(-q(d)) with q(d) = 0.5*(A dx^2 + C dy^2) + B dx dy, so the reachable region is the
// ellipse q <= tau = ln(255 o).  It is rasterised at tile granularity row by row: within the strip of pixel rows of
// one tile row (offsets w in [w0, w1] from the centre) the ellipse spans u in [l(w), r(w)] with
//     r(w), l(w) = (-B w +- sqrt(2 A tau - det w^2)) / A,       det = A C - B^2,
// r concave / l convex in w, so the strip's extent is taken at the strip ends or at the stationary points
// w* = -+ B sqrt(2 tau / (C det)) where r, l reach the ellipse's own extent +- sqrt(2 tau C / det).  The ellipse cut by a
// strip is convex, hence the tiles it meets in a row are exactly the ones whose pixel columns meet [l, r]: one
// contiguous run of mask bits per row, no per-tile test (the per-lane loop over up to 64 tiles used to dominate
// the kernel: a wave runs as long as its largest rect).  tau carries a +1e-3 margin plus the fp32 evaluation noise of the blend's own
// exponent over the rect (w3d_q_noise, w3d_common.h) and the intervals a +0.01 px margin, the determinant is taken to ~1 ulp
// (w3d_conic_det), so a tile is only dropped when the Gaussian provably contributes nothing there:
// every output is unchanged, R and R_walk shrink.  Bit k of the result = k-th tile of the rect, row-major.
// LIST GRID (w3d_view.list_share): with lsx / lsy > 0 the mask lives on the grid of list cells (2^lsx x 2^lsy tiles each): the rect
// is the cells the tile rect touches, and a cell's bit is the OR of its tiles' bits — every tile row still contributes its own
// run [t_lo, t_hi] of tiles (so nothing is lost to the coarser grid), shifted down to cells.  Bit k = k-th cell of the cell rect.
__device__ __forceinline__ uint64_t footprint_tile_mask(float mx, float my, float A, float B, float C, float tau,
                                                        int minx, int miny, int maxx, int maxy, int lsx = 0, int lsy = 0) {
    if (tau < 0.f) return 0ull;                       // o <= 1/255: alpha >= 1/255 is unreachable anywhere
    const float det = w3d_conic_det(A, B, C);
    if (!(A > 0.f && C > 0.f && det > 0.f)) return ~0ull;
    // (hardware rcp / sqrt, ~1 ulp: this mask only has to be conservative, and the 0.01-px / 1e-3 margins dwarf that)
    const float T2 = 2.f * tau, idet = __builtin_amdgcn_rcpf(det), iA = __builtin_amdgcn_rcpf(A);
    const float pad = 0.01f;
    const float wmax = __builtin_amdgcn_sqrtf(T2 * A * idet) + pad;    // the ellipse's half height
    const float xext = __builtin_amdgcn_sqrtf(T2 * C * idet) + pad;    // ... and half width
    const float wstar = B * __builtin_amdgcn_sqrtf(T2 * idet * __builtin_amdgcn_rcpf(C));     // l is extremal at +wstar, r at -wstar
    const int cminx = minx >> lsx, cminy = miny >> lsy;
    const int rw = ((maxx + (1 << lsx) - 1) >> lsx) - cminx;
    uint64_t m = 0ull;
    for (int ty = miny; ty < maxy; ty++) {
        const float w0 = (float)(ty * W3D_TILE) - my, w1 = w0 + (float)(W3D_TILE - 1);
        const float a = fmaxf(w0, -wmax), b = fminf(w1, wmax);
        if (a > b) continue;                          // the strip misses the ellipse
        const float sa = __builtin_amdgcn_sqrtf(fmaxf(T2 * A - det * a * a, 0.f)), sb = __builtin_amdgcn_sqrtf(fmaxf(T2 * A - det * b * b, 0.f));
        const float ra = (-B * a + sa) * iA, rb = (-B * b + sb) * iA;
        const float la = (-B * a - sa) * iA, lb = (-B * b - sb) * iA;
        const float xr = ((-wstar >= a && -wstar <= b) ? xext : fmaxf(ra, rb) + pad) + mx;
        const float xl = ((wstar >= a && wstar <= b) ? -xext : fminf(la, lb) - pad) + mx;
        // tile tx holds the pixel columns [16 tx, 16 tx + 15]
        const int t_lo = max(minx, (int)ceilf((xl - (float)(W3D_TILE - 1)) * (1.0f / W3D_TILE)));
        const int t_hi = min(maxx - 1, (int)floorf(xr * (1.0f / W3D_TILE)));
        if (t_lo > t_hi) continue;
        const int c_lo = t_lo >> lsx, c_hi = t_hi >> lsx;
        const int cnt = c_hi - c_lo + 1, start = ((ty >> lsy) - cminy) * rw + (c_lo - cminx);
        m |= (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << start;
    }
    return m;
}

// Activations of the raw (pre-activation) parameter path — what GaussianModel's getters apply
// (reference scene/gaussian_model.py:33-41,101-121): torch.exp, torch.sigmoid, F.normalize ON THE DEVICE.  The fused path must hand
// the rasterizer the very bits those torch kernels produce (radii, tile ranges and lists are integer work and compared
// bit for bit with the activated API fed torch's activations, tests/test_gpu_raw_bitexact.py):
//   exp      torch's kernel calls ::exp(float) = this expf (ROCm device library, no fast-math on either side);
//   sigmoid  1 / (1 + exp(-x)) with an IEEE division (aten UnarySpecialOpsKernel.cu);
//   normalize  x / max(sqrt(sum of squares), 1e-12) — one IEEE division PER COMPONENT (not one reciprocal), the sum of squares in
//            the order torch's reduction of a contiguous row of four takes on this device: correctly rounded squares added
//            pairwise, (s0 + s1) + (s2 + s3) — found empirically (profiles/activation_probe.py, profiles/r06/activation_probe.json:
//            0 of 4 M rows over 60 binades differ from F.normalize with this order and a per-component division; left-to-right
//            order 469 k, an fma chain 669 k, one reciprocal 2.5 M).
#ifndef W3D_NORM_ORDER
#define W3D_NORM_ORDER 1       // 1: (s0+s1)+(s2+s3)   0: ((s0+s1)+s2)+s3   2: fma chain  (0, 2: probe builds only)
#endif
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void act_normalize(const float *r, float *q, float &inv_norm) {
#pragma clang fp contract(off)
    const float s0 = r[0] * r[0], s1 = r[1] * r[1], s2 = r[2] * r[2], s3 = r[3] * r[3];
#if W3D_NORM_ORDER == 0
    const float ss = ((s0 + s1) + s2) + s3;
#elif W3D_NORM_ORDER == 1
    const float ss = (s0 + s1) + (s2 + s3);
#else
    const float ss = __builtin_fmaf(r[3], r[3], __builtin_fmaf(r[2], r[2], __builtin_fmaf(r[1], r[1], s0)));
    (void)s1; (void)s2; (void)s3;
#endif
    const float d = fmaxf(sqrtf(ss), 1e-12f);
    inv_norm = 1.0f / d;              // (the backward's chain rule through the normalisation; not on the integer path)
    q[0] = r[0] / d; q[1] = r[1] / d; q[2] = r[2] / d; q[3] = r[3] / d;
}

// RAW: `shs` is f_dc (P,1,3), `f_rest` is (P,M-1,3), opacities are logits, scales are log-scales,
// rotations are un-normalised quaternions; the activations are applied here.
template <bool RAW>
__global__ void __launch_bounds__(256, W3D_PRE_OCC)
preprocess_fwd_kernel(w3d_view v, int P, int gx, int gy, int lsx, int lsy, const float *__restrict__ means3D,
                      const float *__restrict__ shs, const float *__restrict__ f_rest, const float *__restrict__ colors_precomp,
                      const float *__restrict__ opacities, const float *__restrict__ scales,
                      const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                      int32_t *__restrict__ radii, float4 *__restrict__ grec, ushort4 *__restrict__ rect,
                      uint8_t *__restrict__ clamped_out,
                      uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ counters,
                      uint4 *__restrict__ tile_mask, const uint8_t *__restrict__ used_mask, uint4 *__restrict__ minmax) {
#pragma clang fp contract(off)
    // f_rest rows of one wave's 64 Gaussians (64 x 180 B, contiguous in memory) on their way to the lanes
    __shared__ float4 s_sh[RAW ? 4 : 1][RAW ? W3D_SH_CHUNKS / W3D_SH_ROUNDS : 1];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < P;
    Cam cam;
    load_cam(v, cam);
    int radius = 0;
    uint32_t key = W3D_INVALID_KEY;
    bool vis = false;
    float p[3] = {0.f, 0.f, 0.f}, px = 0.f, py = 0.f, conx = 0.f, cony = 0.f, conz = 0.f, op_in = 0.f, depth = 0.f;
    float dc[3] = {0.f, 0.f, 0.f};
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    // the wave-cooperative SH load needs whole 16-B chunks of whole degree-3 rows
    const bool coop = RAW && !colors_precomp && v.sh_degree == 3 && v.sh_coeffs == 16 &&
                      (reinterpret_cast<uintptr_t>(f_rest) & 15) == 0;
    float s_in[3] = {0.f, 0.f, 0.f};
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    // subset render (flashsplat_render(used_mask=...), reference gaussian_renderer/__init__.py:151-156): a Gaussian the
    // caller's byte mask leaves out is culled before any of its parameters is requested
    const bool used = valid && (!used_mask || used_mask[g] != 0);
    if (used) {
        p[0] = means3D[3 * (size_t)g]; p[1] = means3D[3 * (size_t)g + 1]; p[2] = means3D[3 * (size_t)g + 2];
        // scale, rotation and opacity are requested together with the position (one memory latency instead of three
        // dependent ones; 32 B per Gaussian that the culled ones would not have needed)
        if (!cov3D_precomp) {
            s_in[0] = scales[3 * (size_t)g]; s_in[1] = scales[3 * (size_t)g + 1]; s_in[2] = scales[3 * (size_t)g + 2];
            q_in = reinterpret_cast<const float4 *>(rotations)[g];
        }
        op_in = opacities[g];
        if (coop) { dc[0] = shs[3 * (size_t)g]; dc[1] = shs[3 * (size_t)g + 1]; dc[2] = shs[3 * (size_t)g + 2]; }
    }
    do {
        if (!used) break;
        float pv[3];
        xform4x3(cam.V, p, pv);
        if (!(pv[2] > W3D_NEAR)) break;   // near cull (also rejects NaN depth)
        float ph[4];
        xform4x4(cam.M, p, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float ppx = ph[0] * pw, ppy = ph[1] * pw;
        float c3[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)g + i];
        } else {
            float s[3] = {s_in[0], s_in[1], s_in[2]};
            const float4 q4 = q_in;
            float q[4] = {q4.x, q4.y, q4.z, q4.w};
            if (RAW) {
                s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
                float inv_n;
                const float r[4] = {q[0], q[1], q[2], q[3]};
                act_normalize(r, q, inv_n);
            }
            cov3d_from_scale_rot(s, v.scale_modifier, q, c3);
        }
        Geo geo;
        ewa(v, cam, pv, c3, geo);
        float det = geo.a * geo.c - geo.b * geo.b;
        if (det == 0.0f) break;
        float det_inv = 1.f / det;
        conx = geo.c * det_inv; cony = -geo.b * det_inv; conz = geo.a * det_inv;
        float mid = 0.5f * (geo.a + geo.c);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        px = ((ppx + 1.0f) * (float)v.image_width - 1.0f) * 0.5f;
        py = ((ppy + 1.0f) * (float)v.image_height - 1.0f) * 0.5f;
        // radius can exceed int range for degenerate inputs: clamp before converting
        int r = (int)fminf(my_radius, 1.0e9f);
        float rf = (float)r;
        float fminx = (px - rf) / (float)W3D_TILE, fminy = (py - rf) / (float)W3D_TILE;
        float fmaxx = (px + rf + (float)(W3D_TILE - 1)) / (float)W3D_TILE;
        float fmaxy = (py + rf + (float)(W3D_TILE - 1)) / (float)W3D_TILE;
        // clamp in float first (same result as int clamp for in-range values, safe for huge ones)
        minx = (int)fminf(fmaxf(fminx, -1.0f), (float)gx + 1.0f); miny = (int)fminf(fmaxf(fminy, -1.0f), (float)gy + 1.0f);
        maxx = (int)fminf(fmaxf(fmaxx, -1.0f), (float)gx + 1.0f); maxy = (int)fminf(fmaxf(fmaxy, -1.0f), (float)gy + 1.0f);
        minx = min(gx, max(0, minx)); miny = min(gy, max(0, miny));
        maxx = min(gx, max(0, maxx)); maxy = min(gy, max(0, maxy));
        if ((maxx - minx) * (maxy - miny) == 0) break;
        radius = r;
        depth = pv[2];
        vis = true;
    } while (0);

    float c[48];
    if (RAW && coop) {
        // Every lane of the wave fetches 16-B chunks of the wave's contiguous 64-row span (fully coalesced: 1 KB per
        // instruction instead of 64 rows 180 B apart), skipping chunks that only culled Gaussians own, and the rows
        // are picked up from LDS (pitch 45 words: conflict-free).  Nobody leaves the wave before this point.
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const size_t g0 = (size_t)blockIdx.x * blockDim.x + (size_t)wv * 64;
        const uint64_t vm = __ballot(vis);
        if (vm) {
            const int rows = (int)min((size_t)64, (size_t)P - g0);          // > 0 whenever a lane is visible
            const int nfl = rows * 45, nch = nfl >> 2;
            const float4 *src = reinterpret_cast<const float4 *>(f_rest + g0 * 45);
#if W3D_SH_ROUNDS == 1
            float4 t[W3D_SH_CHUNKS / 64 + 1];
#pragma unroll
            for (int i = 0; i < W3D_SH_CHUNKS / 64 + 1; i++) {
                const int ch = lane + 64 * i;
                const int r0 = (4 * ch) / 45, r1 = (4 * ch + 3) / 45;
                const bool need = ch < nch && (((vm >> r0) | (vm >> min(r1, 63))) & 1ull);
                t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (need) t[i] = src[ch];
            }
#pragma unroll
            for (int i = 0; i < W3D_SH_CHUNKS / 64 + 1; i++) {
                const int ch = lane + 64 * i;
                if (ch < W3D_SH_CHUNKS) s_sh[wv][ch] = t[i];
            }
            // (a ragged last wave: the 1-3 floats behind its last whole chunk)
            if ((nfl & 3) && lane < (nfl & 3)) reinterpret_cast<float *>(s_sh[wv])[4 * nch + lane] = f_rest[g0 * 45 + 4 * nch + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (vis) {
                const float *row = reinterpret_cast<const float *>(s_sh[wv]) + 45 * lane;
                c[0] = dc[0]; c[1] = dc[1]; c[2] = dc[2];
#pragma unroll
                for (int i = 0; i < 45; i++) c[3 + i] = row[i];
            }
#else
            // two rounds of 32 rows (row 32 starts on a chunk boundary: 32 * 45 floats = 360 chunks): half the LDS, half the
            // chunks in flight per lane; the lanes of the round's half of the wave pick up their rows
            constexpr int HC = W3D_SH_CHUNKS / 2;                  // 360 chunks per round
            if (vis) { c[0] = dc[0]; c[1] = dc[1]; c[2] = dc[2]; }
#pragma unroll
            for (int rnd = 0; rnd < 2; rnd++) {
                if (((vm >> (32 * rnd)) & 0xFFFFFFFFull) == 0ull) continue;     // nobody of this half is visible
                float4 t[HC / 64 + 1];
#pragma unroll
                for (int i = 0; i < HC / 64 + 1; i++) {
                    const int lc = lane + 64 * i, ch = HC * rnd + lc;
                    const int r0 = (4 * ch) / 45, r1 = (4 * ch + 3) / 45;
                    const bool need = lc < HC && ch < nch && (((vm >> r0) | (vm >> min(r1, 63))) & 1ull);
                    t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (need) t[i] = src[ch];
                }
#pragma unroll
                for (int i = 0; i < HC / 64 + 1; i++) {
                    const int lc = lane + 64 * i;
                    if (lc < HC) s_sh[wv][lc] = t[i];
                }
                // (a ragged last wave: the 1-3 floats behind its last whole chunk)
                if ((nfl & 3) && lane < (nfl & 3) && nch >= HC * rnd && nch < HC * (rnd + 1))
                    reinterpret_cast<float *>(s_sh[wv])[4 * (nch - HC * rnd) + lane] = f_rest[g0 * 45 + 4 * nch + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (vis && (lane >> 5) == rnd) {
                    const float *row = reinterpret_cast<const float *>(s_sh[wv]) + 45 * (lane & 31);
#pragma unroll
                    for (int i = 0; i < 45; i++) c[3 + i] = row[i];
                }
                __builtin_amdgcn_wave_barrier();
            }
#endif
        }
    }
    if (vis) {
        float rgb[3];
        uint32_t cl = 0;
        if (colors_precomp) {
            rgb[0] = colors_precomp[3 * (size_t)g]; rgb[1] = colors_precomp[3 * (size_t)g + 1]; rgb[2] = colors_precomp[3 * (size_t)g + 2];
        } else {
            if (RAW) { if (!coop) load_sh(v.sh_degree, nullptr, shs + 3 * (size_t)g, f_rest + (size_t)g * (v.sh_coeffs - 1) * 3, c); }
            else load_sh(v.sh_degree, shs + (size_t)g * v.sh_coeffs * 3, nullptr, nullptr, c);
            sh_to_rgb(v.sh_degree, c, p, cam.campos, rgb, cl);
        }
        key = __float_as_uint(depth);
        const float opac = RAW ? act_sigmoid(op_in) : op_in;
        {
            // the Gaussian's 64-B record, in the layout the blend kernels stage (W3DLayout::o_grec)
            float4 *r = grec + 4 * (size_t)g;
            r[0] = make_float4(px, py, __uint_as_float((uint32_t)minx | ((uint32_t)miny << 16)),
                               __uint_as_float((uint32_t)maxx | ((uint32_t)maxy << 16)));
            r[1] = make_float4(conx, cony, conz, opac);
            r[3] = make_float4(-0.5f * W3D_LOG2E * conx, -W3D_LOG2E * cony, -0.5f * W3D_LOG2E * conz, opac);
        }
        if (v.tile_cull) {
            // which tiles of the rect can this Gaussian reach at all?
            // o <= 1/255 can never reach alpha >= 1/255: tau < 0 drops every tile
            float tau = (opac > 0.f) ? (__logf(255.0f * opac) + 1e-3f) : -1.0f;
            if (tau >= 0.f) {
                // ... widened by what the blend's fp32 exponent can be off by anywhere in the published rect (its pixel columns
                // [16 minx, 16 maxx - 1], rows likewise): needles only — w3d_q_noise, w3d_common.h
                const float Dx = fmaxf(fabsf(px - (float)(W3D_TILE * minx)), fabsf((float)(W3D_TILE * maxx - 1) - px));
                const float Dy = fmaxf(fabsf(py - (float)(W3D_TILE * miny)), fabsf((float)(W3D_TILE * maxy - 1) - py));
                tau += w3d_q_noise(conx, cony, conz, Dx, Dy);
            }
            // (a) the rect itself: the published rule takes the bounding SQUARE of the 3-sigma circle of the major axis; the
            // pixels that can reach alpha >= 1/255 lie inside the ellipse q <= tau, whose own axis-aligned extent is
            // +- sqrt(2 tau C / det) x +- sqrt(2 tau A / det) — for the stretched, faint Gaussians a trained scene is full of a
            // fraction of the square (round 4: 45 % of the densified scene's list entries came from rects of more than 64 tiles,
            // which carry no mask).  Same margins as the mask: +1e-3 on tau, +0.01 px on the extents.
            int cminx = minx, cminy = miny, cmaxx = maxx, cmaxy = maxy;
            const float det = w3d_conic_det(conx, cony, conz);
            if (tau < 0.f) {
                cminx = cminy = cmaxx = cmaxy = 0;
            } else if (conx > 0.f && conz > 0.f && det > 0.f) {
                const float T2 = 2.f * tau, idet = __builtin_amdgcn_rcpf(det);
                const float xext = __builtin_amdgcn_sqrtf(T2 * conz * idet) * 1.0001f + 0.01f;
                const float yext = __builtin_amdgcn_sqrtf(T2 * conx * idet) * 1.0001f + 0.01f;
                // tile t holds the pixel columns (rows) [16 t, 16 t + 15]
                // (clamped as floats before the conversion: an extent beyond the int range — a degenerate conic — or a NaN
                //  leaves the published rect's bound in place instead of wrapping around)
                cminx = max(minx, (int)fmaxf(ceilf((px - xext - (float)(W3D_TILE - 1)) * (1.0f / W3D_TILE)), -1.f));
                cmaxx = min(maxx, (int)fminf(floorf((px + xext) * (1.0f / W3D_TILE)), 65535.f) + 1);
                cminy = max(miny, (int)fmaxf(ceilf((py - yext - (float)(W3D_TILE - 1)) * (1.0f / W3D_TILE)), -1.f));
                cmaxy = min(maxy, (int)fminf(floorf((py + yext) * (1.0f / W3D_TILE)), 65535.f) + 1);
                if (cminx >= cmaxx || cminy >= cmaxy) cminx = cminy = cmaxx = cmaxy = 0;
            }
            // (b) within it, tile by tile — for rects of up to 64 tiles (larger ones are left whole); on the LIST grid when the
            // view shares lists between neighbouring tiles (lsx, lsy: cells of 2^lsx x 2^lsy tiles — the record then holds the
            // rect of cells and one bit per cell, which is what the binning stage bins)
            const int lminx = cminx >> lsx, lminy = cminy >> lsy;
            const int lmaxx = (cmaxx + (1 << lsx) - 1) >> lsx, lmaxy = (cmaxy + (1 << lsy) - 1) >> lsy;
            const int rw = lmaxx - lminx, rn = rw * (lmaxy - lminy);
            uint64_t m = ~0ull;
            if (rn <= 64) m = footprint_tile_mask(px, py, conx, cony, conz, tau, cminx, cminy, cmaxx, cmaxy, lsx, lsy);
            tile_mask[g] = make_uint4((uint32_t)lminx | ((uint32_t)lminy << 16), (uint32_t)lmaxx | ((uint32_t)lmaxy << 16), (uint32_t)m,
                                      (uint32_t)(m >> 32));
        }
        grec[4 * (size_t)g + 2] = make_float4(rgb[0], rgb[1], rgb[2], depth);
        rect[g] = make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy);
        clamped_out[g] = (uint8_t)cl;
    }
    else if (valid) {
        // culled Gaussians write zeros into the densely packed arrays: whole 64-B lines leave the wave (stores with holes where
        // the culled lanes sit cost more than the 40 % extra bytes: 0.16 -> 0.14 ms)
        // (the 64-B record is NOT written: every lane owns a whole line of the record array, so skipping the culled ones leaves no
        //  holes inside a line — 40 % of the array's bytes stay unwritten — and nothing reads the record of a Gaussian that is in no
        //  list: the blend gathers through the lists, flash_extras and the deterministic gather test the rect first)
        if (v.tile_cull) tile_mask[g] = make_uint4(0, 0, 0, 0);
        clamped_out[g] = 0;
    }
    {
        // the interval of this workgroup's visible depth keys, for the depth sort's bucket grid (w3d_binning.hip depth_grid): a culled
        // Gaussian's key, 0xFFFFFFFF, is neutral for the minimum.  Two of the keys (the first lanes of waves 1 and 3, culled or
        // not) go along as the grid's population SAMPLE: one 16-B store per workgroup, nothing to initialise
        __shared__ uint32_t s_mm[3][4];
        uint32_t kmn = key, kmx = vis ? key : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, off, 64));
            kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = kmn; s_mm[1][threadIdx.x >> 6] = kmx; s_mm[2][threadIdx.x >> 6] = key; }
        __syncthreads();
        if (threadIdx.x == 0)
            minmax[blockIdx.x] = make_uint4(min(min(s_mm[0][0], s_mm[0][1]), min(s_mm[0][2], s_mm[0][3])),
                                            max(max(s_mm[1][0], s_mm[1][1]), max(s_mm[1][2], s_mm[1][3])), s_mm[2][1], s_mm[2][3]);
    }
    if (!valid) return;
    radii[g] = radius;
    keys[g] = key;
    // (no id array is written: the depth sort's first pass — the only reader of the unsorted order — takes the value of key i
    //  to be i, w3d_binning.hip radix_scatter_kernel `drop_invalid`; s_vals0 only serves as a ping-pong buffer of later passes)
    (void)vals;
    // an all-zero rect marks a culled Gaussian for the binning walk and for the backward pass; written
    // here for EVERY Gaussian so that no stale bytes of a recycled state buffer can ever be read
    if (radius == 0) rect[g] = make_ushort4(0, 0, 0, 0);
    // the visible count is NOT accumulated here (31k same-address atomics cost more than the whole
    // kernel): the depth sort's last pass yields it for free (w3d_binning.hip).
    (void)counters;
}

// proj_xy / gs_depth outputs of the FlashSplat variant (zeros for culled Gaussians)
__global__ void flash_extras_kernel(int P, const uint32_t *__restrict__ keys_unused, const float4 *__restrict__ grec,
                                    const ushort4 *__restrict__ rect,
                                    const int32_t *__restrict__ radii, float *__restrict__ proj_xy,
                                    float *__restrict__ gs_depth) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const ushort4 rc = rect[g];
    const bool vis = ((int)rc.z - (int)rc.x) * ((int)rc.w - (int)rc.y) > 0;
    (void)radii;
    if (proj_xy) {
        const float4 p = vis ? grec[4 * (size_t)g] : make_float4(0.f, 0.f, 0.f, 0.f);
        proj_xy[2 * (size_t)g] = p.x; proj_xy[2 * (size_t)g + 1] = p.y;
    }
    if (gs_depth) gs_depth[g] = vis ? grec[4 * (size_t)g + 2].w : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Backward of the per-Gaussian stages (Appendix A.5).  visible <=> the forward wrote a record,
// flagged by rect having a non-empty area (rect is zero-initialised per call for culled ones).
// Extra pointers of the raw (pre-activation) path; all NULL otherwise.
struct RawBwd {
    const float *f_rest;       // (P, M-1, 3)
    const float *opacity_logit;  // (P,)
    float *dL_df_rest;         // (P, M-1, 3)
    float *gnorm_out;          // (P,) ||dL/dmean2D.xy|| (0 for culled), nullable
    const int32_t *radii;      // (P,) for the fused statistics, nullable
    float *accum, *denom, *max_radii;   // densification statistics updated in place, nullable
    // fused optimizer (ADAM instantiation): blocks in the order xyz, f_dc, f_rest, opacity, scaling, rotation
    float *pw[6], *m[6], *v[6];
    float step_size[6];                 // lr / bias_correction1 per block
    uint32_t skip_mask;                 // bit i: leave block i untouched
    float b1, b2, eps;
    float inv_sqrt_bc2[6];              // per block
    const uint32_t *counters;           // forward counters: [1] = list length needed, [3] = list capacity used
    float *dcolor_out;                  // MODE 2: (P,3) clamp-masked dL/dRGB per Gaussian (zeros for culled)
    // MODE 3: the non-zero 64-B gradient rows themselves (w3d_exchange.hip's row format), appended to rows_out
    float4 *rows_out;
    uint32_t rows_cap;
    uint32_t *rows_count;               // device counter, zeroed by the launcher
    float norm_scale;                   // row word 1 = ||dL/dmean2D|| * norm_scale (0: no norm wanted)
};

// Adam on the `rows` x DIM contiguous floats a workgroup owns in parameter block b, gradients taken from an LDS stage
// (row stride rstride, column offset coff).  16-B nontemporal accesses when the span is aligned (U groups of 3 loads in
// flight per thread), dword accesses otherwise and for the tail.
template <int DIM, int U>
__device__ __forceinline__ void coop_adam(const RawBwd &raw, int b, size_t g0, int rows, const float *stage, int rstride, int coff) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    float *pb = raw.pw[b] + g0 * DIM, *mb = raw.m[b] + g0 * DIM, *vb = raw.v[b] + g0 * DIM;
    const float step = raw.step_size[b];
    const int n = rows * DIM;
    const bool vec = (((uintptr_t)pb | (uintptr_t)mb | (uintptr_t)vb) & 15) == 0;
    const int nv = vec ? n / 4 : 0;
    for (int q0 = threadIdx.x; q0 < nv; q0 += 256 * U) {
        f4 pp[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + u * 256;
            if (q < nv) {
                pp[u] = __builtin_nontemporal_load(reinterpret_cast<f4 *>(pb) + q);
                mm[u] = __builtin_nontemporal_load(reinterpret_cast<f4 *>(mb) + q);
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<f4 *>(vb) + q);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + u * 256;
            if (q < nv) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int e = 4 * q + c;
                    float a = pp[u][c], cm = mm[u][c], cv = vv[u][c];
                    w3d_adam1(a, stage[(e / DIM) * rstride + coff + e % DIM], cm, cv, step, raw.b1, raw.b2, raw.eps, raw.inv_sqrt_bc2[b]);
                    pp[u][c] = a; mm[u][c] = cm; vv[u][c] = cv;
                }
                __builtin_nontemporal_store(pp[u], reinterpret_cast<f4 *>(pb) + q);
                __builtin_nontemporal_store(mm[u], reinterpret_cast<f4 *>(mb) + q);
                __builtin_nontemporal_store(vv[u], reinterpret_cast<f4 *>(vb) + q);
            }
        }
    }
    for (int e = 4 * nv + threadIdx.x; e < n; e += 256) {
        float pp = pb[e], mm = mb[e], vv = vb[e];
        w3d_adam1(pp, stage[(e / DIM) * rstride + coff + e % DIM], mm, vv, step, raw.b1, raw.b2, raw.eps, raw.inv_sqrt_bc2[b]);
        pb[e] = pp; mb[e] = mm; vb[e] = vv;
    }
}

// RAW: shs/dL_dshs are the f_dc blocks, scales/rotations/opacity are pre-activation and the
// gradients are chained through exp / normalize / sigmoid before being written.
// FAST16 (16 SH coefficients, the only case Wheat-3DGS uses): the SH gradient rows of the workgroup's
// 256 Gaussians are staged in LDS (row stride 49 floats: conflict-free) and written out as one
// contiguous, fully coalesced span instead of 45-48 dword stores per lane at a 180/192-B stride.
#define W3D_SHROW 49
// ADAM (with RAW and FAST16): instead of writing the gradients, apply the optimizer update to the parameter blocks and
// their moments in place (w3d_backward_raw_adam) — unless the forward overflowed its list buffer, in which case
// nothing is touched and the host repeats the view.
// MODE 2 (with RAW, view-parallel exchange): the SH gradient of a view is the outer product of the SH basis at the view
// direction with the clamp-masked dL/dRGB, so only those 3 floats per Gaussian are written (dcolor_out) — the ranks
// exchange them instead of 48 floats and rebuild the sum over views in sh_adam_lowrank_kernel.  The gradients of the
// other blocks are written as in MODE 0.  No LDS stage, no workgroup barrier.
// MODE 3 (with RAW, the sparse form of that exchange): the same 14 floats are not written densely and packed by a second kernel
// (pack_rows_kernel: 120 MB written and read again at 2 M Gaussians) — every Gaussian whose 14 gradient floats and norm are not
// all zero gets its 64-B row {index, norm * scale, dRGB[3], d xyz[3], d opacity, d scaling[3], d rotation[4]} appended to
// rows_out right here: ballot + wave totals in LDS + ONE counter atomic per workgroup.  Row order is unspecified (as the pack
// kernel's was: the replicated optimizer finds a Gaussian's row through the index, and adds views in view order).
template <bool HAS_SH, bool HAS_SCALE_ROT, bool RAW, bool FAST16, int MODE = 0>
__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(w3d_view v, int P, const float *means3D, const float *shs, const float *scales, const float *rotations,
                      const float *__restrict__ cov3D_precomp, RawBwd raw,     // (no __restrict__ on the parameter inputs:
                      // the fused-Adam flavour updates the same memory through raw.pw[] later in the kernel; the workgroup
                      // barrier before the update is the only ordering it relies on)
                      const ushort4 *__restrict__ rect, const uint8_t *__restrict__ clamped,
                      float *grad2d /* read; written back as zeros under records_kept_clean */,
                      float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dcolors,
                      float *__restrict__ dL_dshs, float *__restrict__ dL_dopacity, float *__restrict__ dL_dscales,
                      float *__restrict__ dL_drots, float *__restrict__ dL_dcov3D) {
    constexpr bool ADAM = MODE == 1;
    constexpr bool LOWRANK = MODE == 2 || MODE == 3;            // no SH gradient rows: dL/dRGB stands for them
    constexpr bool STAGED = HAS_SH && FAST16 && !LOWRANK;       // SH gradient rows go through the LDS stage
    __shared__ uint32_t s_rowtot[MODE == 3 ? 5 : 1];
    __shared__ float sh_stage[STAGED ? 256 * W3D_SHROW : 1];
    const uint32_t blk = blockIdx.x;
    const int gid = blk * blockDim.x + threadIdx.x;
    const bool active = gid < P;
    if (!STAGED && MODE != 3 && !active) return;      // (MODE 3: every thread reaches the workgroup barrier of the row compaction)
    float dRGB_out[3] = {0.f, 0.f, 0.f};
    const int g = active ? gid : P - 1;       // FAST16: idle lanes of the last block still reach the barrier
    const int Mc = v.sh_coeffs;
    // (uniform over the grid) the forward's lists fitted their buffer, so this backward is final
    const bool adam_ok = ADAM && raw.counters[1] <= raw.counters[3];
    float inv_qnorm = 1.f;
    float q_act[4] = {1.f, 0.f, 0.f, 0.f}, s_act[3] = {0.f, 0.f, 0.f};
    const ushort4 rc = rect[g];
    // (`active`: the idle lanes of the last workgroup alias Gaussian P - 1 only to stay in the barriers — they must not act
    //  on it: under records_kept_clean a lane that zeroed P - 1's record before its owner had read it cost that Gaussian its
    //  gradient, once in ~40 launches of a 5 000-Gaussian scene — tests/test_gpu_fused.py::test_last_gaussian_of_a_ragged_workgroup)
    const bool vis = active && ((int)rc.z - (int)rc.x) * ((int)rc.w - (int)rc.y) > 0;
    float dmean[3] = {0.f, 0.f, 0.f};
    float dm2[2] = {0.f, 0.f};
    float dop = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dcol[3] = {0.f, 0.f, 0.f};
    // coefficient k, channel c of the SH gradient: interleaved (P,M,3) or split dc | rest blocks
    float *dsh = HAS_SH ? (RAW ? dL_dshs + 3 * (size_t)g : dL_dshs + (size_t)g * Mc * 3) : nullptr;
    float *dsh_rest = (HAS_SH && RAW) ? raw.dL_df_rest + (size_t)g * (Mc - 1) * 3 : nullptr;
#define DSH(k, c) (*(STAGED ? (sh_stage + threadIdx.x * W3D_SHROW + 3 * (k) + (c)) \
                            : ((RAW && (k) > 0) ? (dsh_rest + 3 * ((k)-1) + (c)) : (dsh + 3 * (k) + (c)))))
    if (!vis) {
        if (HAS_SH && !LOWRANK) {
            if (FAST16) {
#pragma unroll
                for (int i = 0; i < 48; i++) sh_stage[threadIdx.x * W3D_SHROW + i] = 0.f;
            } else if (RAW) {
                dsh[0] = dsh[1] = dsh[2] = 0.f;
                for (int i = 0; i < (Mc - 1) * 3; i++) dsh_rest[i] = 0.f;
            } else {
                for (int i = 0; i < Mc * 3; i++) dsh[i] = 0.f;
            }
        }
    } else {
        Cam cam;
        load_cam(v, cam);
        float4 *rec = reinterpret_cast<float4 *>(grad2d + (size_t)g * W3D_G2D_STRIDE);
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        if (v.records_kept_clean && !v.deterministic) {
            // this lane is the record's only reader: hand the buffer back zeroed (w3d_view.records_kept_clean) instead of
            // running a zeroing pass in front of every blend backward (the blend adds to floats 0..9 only)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            rec[0] = z; rec[1] = z; rec[2] = z;
        }
        dm2[0] = r0.x; dm2[1] = r0.y;
        const float dconic[3] = {r0.z, r0.w, r1.x};
        dop = r1.y;
        dcol[0] = r1.z; dcol[1] = r1.w; dcol[2] = r2.x;
        const float ddepth = r2.y;
        float p[3] = {means3D[3 * (size_t)g], means3D[3 * (size_t)g + 1], means3D[3 * (size_t)g + 2]};
        float pv[3];
        xform4x3(cam.V, p, pv);
        float c3[6];
        float q[4] = {1.f, 0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f};
        if (HAS_SCALE_ROT) {
            s[0] = scales[3 * (size_t)g]; s[1] = scales[3 * (size_t)g + 1]; s[2] = scales[3 * (size_t)g + 2];
            const float4 q4 = reinterpret_cast<const float4 *>(rotations)[g];
            q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
            if (RAW) {
                s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
                const float r4[4] = {q[0], q[1], q[2], q[3]};
                act_normalize(r4, q, inv_qnorm);
                s_act[0] = s[0]; s_act[1] = s[1]; s_act[2] = s[2];
                q_act[0] = q[0]; q_act[1] = q[1]; q_act[2] = q[2]; q_act[3] = q[3];
            }
            cov3d_from_scale_rot(s, v.scale_modifier, q, c3);
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)g + i];
        }
        Geo geo;
        ewa(v, cam, pv, c3, geo);
        // (i) conic -> cov2D -> cov3D, view-space mean
        const float ca = geo.a, cb = geo.b, cc = geo.c;
        const float denom = ca * cc - cb * cb;
        const float denom2inv = 1.0f / (denom * denom + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float(*Tm)[3] = geo.Tm;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * dconic[0] + 2.f * cb * cc * dconic[1] + (denom - ca * cc) * dconic[2]);
            dL_dc = denom2inv * (-ca * ca * dconic[2] + 2.f * ca * cb * dconic[1] + (denom - ca * cc) * dconic[0]);
            dL_db = denom2inv * 2.f * (cb * cc * dconic[0] - (denom + 2.f * cb * cb) * dconic[1] + ca * cb * dconic[2]);
            dcov[0] = Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc;
            dcov[3] = Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc;
            dcov[5] = Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc;
            dcov[1] = 2.f * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db + 2.f * Tm[1][0] * Tm[1][1] * dL_dc;
            dcov[2] = 2.f * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db + 2.f * Tm[1][0] * Tm[1][2] * dL_dc;
            dcov[4] = 2.f * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db + 2.f * Tm[1][1] * Tm[1][2] * dL_dc;
        }
        const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
        float TS[2][3], dT[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) TS[i][j] = Tm[i][0] * S[0][j] + Tm[i][1] * S[1][j] + Tm[i][2] * S[2][j];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            dT[0][j] = 2.f * TS[0][j] * dL_da + TS[1][j] * dL_db;
            dT[1][j] = 2.f * TS[1][j] * dL_dc + TS[0][j] * dL_db;
        }
        const float *V = cam.V;
        const float dJ00 = dT[0][0] * V[0] + dT[0][1] * V[4] + dT[0][2] * V[8];
        const float dJ02 = dT[0][0] * V[2] + dT[0][1] * V[6] + dT[0][2] * V[10];
        const float dJ11 = dT[1][0] * V[1] + dT[1][1] * V[5] + dT[1][2] * V[9];
        const float dJ12 = dT[1][0] * V[2] + dT[1][1] * V[6] + dT[1][2] * V[10];
        const float fx = (float)v.image_width / (2.f * v.tanfovx), fy = (float)v.image_height / (2.f * v.tanfovy);
        const float tz = 1.f / pv[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = (geo.clx ? 0.f : 1.f) * -fx * tz2 * dJ02;
        const float dty = (geo.cly ? 0.f : 1.f) * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * geo.txc) * tz3 * dJ02 + (2.f * fy * geo.tyc) * tz3 * dJ12;
        dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
        // (ii) screen-space mean -> 3-D mean through the perspective divide
        const float *M = cam.M;
        float mh[4];
        xform4x4(M, p, mh);
        const float mw = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        dmean[0] += (M[0] * mw - M[3] * mul1) * dm2[0] + (M[1] * mw - M[3] * mul2) * dm2[1];
        dmean[1] += (M[4] * mw - M[7] * mul1) * dm2[0] + (M[5] * mw - M[7] * mul2) * dm2[1];
        dmean[2] += (M[8] * mw - M[11] * mul1) * dm2[0] + (M[9] * mw - M[11] * mul2) * dm2[1];
        // (iii) depth output
        dmean[0] += V[2] * ddepth; dmean[1] += V[6] * ddepth; dmean[2] += V[10] * ddepth;
        // (iv) colour
        if (HAS_SH) {
            const int deg = v.sh_degree;
            float sh[48];
            if (RAW) load_sh(deg, nullptr, shs + 3 * (size_t)g, raw.f_rest + (size_t)g * (Mc - 1) * 3, sh);
            else load_sh(deg, shs + (size_t)g * Mc * 3, nullptr, nullptr, sh);
            const uint32_t cl = clamped[g];
            const float dRGB[3] = {(cl & 1u) ? 0.f : dcol[0], (cl & 2u) ? 0.f : dcol[1], (cl & 4u) ? 0.f : dcol[2]};
            const float d0 = p[0] - cam.campos[0], d1 = p[1] - cam.campos[1], d2 = p[2] - cam.campos[2];
            const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            const float x = d0 / len, y = d1 / len, z = d2 / len;
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float ddir[3] = {0.f, 0.f, 0.f};
            // basis values (shared by the three channels)
            float B[16];
            B[0] = SH_C0;
            B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
            B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
            B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
            B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z; B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
            B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
            B[14] = SH_C3[5] * z * (xx - yy); B[15] = SH_C3[6] * x * (xx - 3.f * yy);
            const int ncoef = (deg + 1) * (deg + 1);
            dRGB_out[0] = dRGB[0]; dRGB_out[1] = dRGB[1]; dRGB_out[2] = dRGB[2];
            if (!LOWRANK) {
#pragma unroll
                for (int k = 0; k < 16; k++) {   // static indices keep B[] in registers
                    if (k < Mc) {
                        const float b = (k < ncoef) ? B[k] : 0.f;
                        DSH(k, 0) = b * dRGB[0]; DSH(k, 1) = b * dRGB[1]; DSH(k, 2) = b * dRGB[2];
                    }
                }
                for (int k = 16; k < Mc; k++) { DSH(k, 0) = 0.f; DSH(k, 1) = 0.f; DSH(k, 2) = 0.f; }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k)*3 + ch]
                float ddx = 0.f, ddy = 0.f, ddz = 0.f;
                if (deg > 0) {
                    ddx = -SH_C1 * SH(3); ddy = -SH_C1 * SH(1); ddz = SH_C1 * SH(2);
                    if (deg > 1) {
                        ddx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                        ddy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                        ddz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (deg > 2) {
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
                ddir[0] += ddx * dRGB[ch]; ddir[1] += ddy * dRGB[ch]; ddir[2] += ddz * dRGB[ch];
            }
            const float sum2 = d0 * d0 + d1 * d1 + d2 * d2;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - d0 * d0) * ddir[0] - d1 * d0 * ddir[1] - d2 * d0 * ddir[2]) * inv32;
            dmean[1] += (-d0 * d1 * ddir[0] + (sum2 - d1 * d1) * ddir[1] - d2 * d1 * ddir[2]) * inv32;
            dmean[2] += (-d0 * d2 * ddir[0] - d1 * d2 * ddir[1] + (sum2 - d2 * d2) * ddir[2]) * inv32;
        }
        // (v) cov3D -> scale, quaternion (as given)
        if (HAS_SCALE_ROT) {
            float R[3][3], L[3][3];
            quat_to_R(q, R);
            const float mod = v.scale_modifier;
            const float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * sc[j];
            const float Gs[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dLm[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) dLm[i][j] = 2.f * (Gs[i][0] * L[0][j] + Gs[i][1] * L[1][j] + Gs[i][2] * L[2][j]);
#pragma unroll
            for (int j = 0; j < 3; j++) dscale[j] = mod * (dLm[0][j] * R[0][j] + dLm[1][j] * R[1][j] + dLm[2][j] * R[2][j]);
            float GR[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) GR[i][j] = dLm[i][j] * sc[j];
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            drot[0] = 2.f * (-z * GR[0][1] + y * GR[0][2] + z * GR[1][0] - x * GR[1][2] - y * GR[2][0] + x * GR[2][1]);
            drot[1] = 2.f * (y * GR[0][1] + z * GR[0][2] + y * GR[1][0] - 2.f * x * GR[1][1] - r * GR[1][2] + z * GR[2][0] + r * GR[2][1] - 2.f * x * GR[2][2]);
            drot[2] = 2.f * (-2.f * y * GR[0][0] + x * GR[0][1] + r * GR[0][2] + x * GR[1][0] + z * GR[1][2] - r * GR[2][0] + z * GR[2][1] - 2.f * y * GR[2][2]);
            drot[3] = 2.f * (-2.f * z * GR[0][0] - r * GR[0][1] + x * GR[0][2] + r * GR[1][0] - 2.f * z * GR[1][1] + y * GR[1][2] + x * GR[2][0] + y * GR[2][1]);
        }
    }
#undef DSH
    if (STAGED) {
        // coalesced write-out of the block's SH gradient rows
        __syncthreads();
        const size_t g0 = (size_t)blk * 256;
        const int rows = (int)min((size_t)256, (size_t)P - g0);
        if (RAW && ADAM) {
            if (adam_ok && !(raw.skip_mask & 2u)) coop_adam<3, 1>(raw, 1, g0, rows, sh_stage, W3D_SHROW, 0);
            if (adam_ok && !(raw.skip_mask & 4u)) coop_adam<45, 4>(raw, 2, g0, rows, sh_stage, W3D_SHROW, 3);
        } else if (RAW) {
            float *ddc = dL_dshs + g0 * 3, *drest = raw.dL_df_rest + g0 * 45;
            for (int e = threadIdx.x; e < rows * 3; e += 256) ddc[e] = sh_stage[(e / 3) * W3D_SHROW + e % 3];
            for (int e = threadIdx.x; e < rows * 45; e += 256) drest[e] = sh_stage[(e / 45) * W3D_SHROW + 3 + e % 45];
        } else {
            float *dall = dL_dshs + g0 * 48;
            for (int e = threadIdx.x; e < rows * 48; e += 256) dall[e] = sh_stage[(e / 48) * W3D_SHROW + e % 48];
        }
        if (!ADAM && !active) return;      // (ADAM: every thread stays for the cooperative phases below)
    }
    if (RAW) {
        // chain through the activations: s = exp(ls), o = sigmoid(lo), q = r / |r|
        dscale[0] *= s_act[0]; dscale[1] *= s_act[1]; dscale[2] *= s_act[2];
        if (vis) {
            const float o = act_sigmoid(raw.opacity_logit[g]);
            dop *= o * (1.f - o);
        }
        const float qd = q_act[0] * drot[0] + q_act[1] * drot[1] + q_act[2] * drot[2] + q_act[3] * drot[3];
#pragma unroll
        for (int i = 0; i < 4; i++) drot[i] = (drot[i] - q_act[i] * qd) * inv_qnorm;
        const float gn = sqrtf(dm2[0] * dm2[0] + dm2[1] * dm2[1]);
        if (raw.gnorm_out && active) raw.gnorm_out[g] = vis ? gn : 0.f;
        // (ADAM: only when this backward is final — the host does not touch the statistics in that mode)
        if (raw.accum && vis && active && (!ADAM || adam_ok)) {
            // add_densification_stats + max_radii2D update (scene/gaussian_model.py:461-463,
            // train_vanilla_3dgs.py:102) fused for the single-GPU step
            raw.accum[g] += gn;
            raw.denom[g] += 1.f;
            raw.max_radii[g] = fmaxf(raw.max_radii[g], (float)raw.radii[g]);
        }
    }
    if (dL_dmeans2D && active) { dL_dmeans2D[3 * (size_t)g] = dm2[0]; dL_dmeans2D[3 * (size_t)g + 1] = dm2[1]; dL_dmeans2D[3 * (size_t)g + 2] = 0.f; }
    if (MODE == 3) {
        // append this Gaussian's row if any of its 14 gradient floats (or its norm) is non-zero: the selection of pack_rows_kernel
        const float gnr = (RAW && raw.norm_scale != 0.f && vis) ? sqrtf(dm2[0] * dm2[0] + dm2[1] * dm2[1]) * raw.norm_scale : 0.f;
        const uint32_t bits = __float_as_uint(dRGB_out[0]) | __float_as_uint(dRGB_out[1]) | __float_as_uint(dRGB_out[2]) |
                              __float_as_uint(dmean[0]) | __float_as_uint(dmean[1]) | __float_as_uint(dmean[2]) | __float_as_uint(dop) |
                              __float_as_uint(dscale[0]) | __float_as_uint(dscale[1]) | __float_as_uint(dscale[2]) |
                              __float_as_uint(drot[0]) | __float_as_uint(drot[1]) | __float_as_uint(drot[2]) | __float_as_uint(drot[3]) |
                              __float_as_uint(gnr);
        const bool any = active && (bits & 0x7FFFFFFFu) != 0u;
        const uint64_t bal = w3d_ballot(any);
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        if (lane == 0) s_rowtot[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = s_rowtot[0] + s_rowtot[1] + s_rowtot[2] + s_rowtot[3];
            s_rowtot[4] = tot ? atomicAdd(raw.rows_count, tot) : 0u;
        }
        __syncthreads();
        if (any) {
            uint32_t r = s_rowtot[4] + (uint32_t)__popcll(bal & (lane ? (~0ull >> (64u - lane)) : 0ull));
            for (uint32_t w = 0; w < wv; w++) r += s_rowtot[w];
            if (r < raw.rows_cap) {
                float4 *row = raw.rows_out + 4 * (size_t)r;
                row[0] = make_float4(__uint_as_float((uint32_t)g), gnr, dRGB_out[0], dRGB_out[1]);
                row[1] = make_float4(dRGB_out[2], dmean[0], dmean[1], dmean[2]);
                row[2] = make_float4(dop, dscale[0], dscale[1], dscale[2]);
                row[3] = make_float4(drot[0], drot[1], drot[2], drot[3]);
            }
        }
        return;
    }
    if (ADAM) {
        // the narrow blocks go through the (now free) LDS stage as well: row = [dmean 3 | dop 1 | dscale 3 | drot 4], stride 15
        __syncthreads();
        float *st2 = sh_stage + threadIdx.x * 15;
        st2[0] = dmean[0]; st2[1] = dmean[1]; st2[2] = dmean[2]; st2[3] = dop;
        st2[4] = dscale[0]; st2[5] = dscale[1]; st2[6] = dscale[2];
        st2[7] = drot[0]; st2[8] = drot[1]; st2[9] = drot[2]; st2[10] = drot[3];
        __syncthreads();
        if (!adam_ok) return;
        const size_t g0 = (size_t)blk * 256;
        const int rows = (int)min((size_t)256, (size_t)P - g0);
        if (!(raw.skip_mask & 1u)) coop_adam<3, 1>(raw, 0, g0, rows, sh_stage, 15, 0);
        if (!(raw.skip_mask & 8u)) coop_adam<1, 1>(raw, 3, g0, rows, sh_stage, 15, 3);
        if (!(raw.skip_mask & 16u)) coop_adam<3, 1>(raw, 4, g0, rows, sh_stage, 15, 4);
        if (!(raw.skip_mask & 32u)) coop_adam<4, 1>(raw, 5, g0, rows, sh_stage, 15, 7);
        return;
    }
    if (MODE == 2 && raw.dcolor_out) {       // (NULL: dcolor_extract_kernel already produced it, see w3d_backward_blend_dcolor)
        raw.dcolor_out[3 * (size_t)g] = dRGB_out[0]; raw.dcolor_out[3 * (size_t)g + 1] = dRGB_out[1];
        raw.dcolor_out[3 * (size_t)g + 2] = dRGB_out[2];
    }
    dL_dmeans3D[3 * (size_t)g] = dmean[0]; dL_dmeans3D[3 * (size_t)g + 1] = dmean[1]; dL_dmeans3D[3 * (size_t)g + 2] = dmean[2];
    dL_dopacity[g] = dop;
    if (!HAS_SH && dL_dcolors) {
        dL_dcolors[3 * (size_t)g] = dcol[0]; dL_dcolors[3 * (size_t)g + 1] = dcol[1]; dL_dcolors[3 * (size_t)g + 2] = dcol[2];
    }
    if (HAS_SCALE_ROT) {
        dL_dscales[3 * (size_t)g] = dscale[0]; dL_dscales[3 * (size_t)g + 1] = dscale[1]; dL_dscales[3 * (size_t)g + 2] = dscale[2];
        reinterpret_cast<float4 *>(dL_drots)[g] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
    if (dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)g + i] = dcov[i];
    }
}

// The clamp-masked dL/dRGB per Gaussian straight from the blend backward's records (what MODE 2 of preprocess_bwd_kernel
// writes as dcolor_out), so that the ranks' all-gather of it can start BEFORE the per-Gaussian backward runs.
__global__ void __launch_bounds__(256)
dcolor_extract_kernel(int P, const ushort4 *__restrict__ rect, const uint8_t *__restrict__ clamped,
                      const float *__restrict__ grad2d, float *__restrict__ dcolor_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const ushort4 rc = rect[g];
    const bool vis = ((int)rc.z - (int)rc.x) * ((int)rc.w - (int)rc.y) > 0;
    float d[3] = {0.f, 0.f, 0.f};
    if (vis) {
        const float *rec = grad2d + (size_t)g * W3D_G2D_STRIDE;
        const uint32_t cl = clamped[g];
        d[0] = (cl & 1u) ? 0.f : rec[6]; d[1] = (cl & 2u) ? 0.f : rec[7]; d[2] = (cl & 4u) ? 0.f : rec[8];
    }
    dcolor_out[3 * (size_t)g] = d[0]; dcolor_out[3 * (size_t)g + 1] = d[1]; dcolor_out[3 * (size_t)g + 2] = d[2];
}

// View-parallel optimizer step of the SH blocks (f_dc, f_rest) from the EXCHANGED colour gradients: for every Gaussian
//   dL/dSH[k][c] = sum over views v of  basis_k(normalize(xyz - campos_v)) * dcolor_v[c]      (k < (deg+1)^2, else 0)
// (dcolor_v = what preprocess_bwd_kernel<MODE 2> wrote on rank v, already scaled by 1/world), summed in view order so
// that every rank computes bit-identical values, then torch.optim.Adam's update in place through the LDS stage exactly
// as in the fused single-GPU backward.  14 instead of 59 floats per Gaussian and view cross the links, and because the
// update is replicated no parameter all-gather follows.
__global__ void __launch_bounds__(256)
sh_adam_lowrank_kernel(int P, int nviews, int deg, const float *__restrict__ campos_all, const float *__restrict__ xyz,
                       const float *__restrict__ dcolor_all, RawBwd raw) {
    __shared__ float sh_stage[256 * W3D_SHROW];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = gid < P;
    const int g = active ? gid : P - 1;
    const int ncoef = (deg + 1) * (deg + 1);
    float acc[48];
#pragma unroll
    for (int i = 0; i < 48; i++) acc[i] = 0.f;
    const float p[3] = {xyz[3 * (size_t)g], xyz[3 * (size_t)g + 1], xyz[3 * (size_t)g + 2]};
    for (int vw = 0; vw < nviews; vw++) {
        const float *dc = dcolor_all + ((size_t)vw * P + g) * 3;
        const float dr = dc[0], dg = dc[1], db = dc[2];
        if (dr == 0.f && dg == 0.f && db == 0.f) continue;      // culled (or fully clamped) in that view
        const float d0 = p[0] - campos_all[3 * vw], d1 = p[1] - campos_all[3 * vw + 1], d2 = p[2] - campos_all[3 * vw + 2];
        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float x = d0 / len, y = d1 / len, z = d2 / len;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float B[16];
        B[0] = SH_C0;
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
        B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
        B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z; B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        B[14] = SH_C3[5] * z * (xx - yy); B[15] = SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float b = (k < ncoef) ? B[k] : 0.f;
            acc[3 * k] += b * dr; acc[3 * k + 1] += b * dg; acc[3 * k + 2] += b * db;
        }
    }
#pragma unroll
    for (int i = 0; i < 48; i++) sh_stage[threadIdx.x * W3D_SHROW + i] = acc[i];
    __syncthreads();
    const size_t g0 = (size_t)blockIdx.x * 256;
    const int rows = (int)min((size_t)256, (size_t)P - g0);
    if (!(raw.skip_mask & 2u)) coop_adam<3, 1>(raw, 1, g0, rows, sh_stage, W3D_SHROW, 0);
    if (!(raw.skip_mask & 4u)) coop_adam<45, 4>(raw, 2, g0, rows, sh_stage, W3D_SHROW, 3);
}

// The replicated optimizer step of the SPARSE exchange (w3d_exchange.hip): the views' gradients arrive as packed 64-B rows
// (rows_all: (nviews, cap, 16) floats), indexed per Gaussian by a view bit mask and, for every set bit, the row's position
// (slots: (nviews, P)).  One lane per Gaussian walks its set bits in VIEW ORDER — the same additions in the same order on
// every rank — rebuilding the SH gradient as sh_adam_lowrank_kernel does and summing the 11 geometry gradients; then the
// workgroup applies torch.optim.Adam's update to all six parameter blocks through the LDS stage, exactly as the fused
// single-GPU backward does.  Gaussians no view touched (most of them) cost one mask word on top of their optimizer traffic;
// no dense per-view plane is zero-filled, scattered into or read.
__global__ void __launch_bounds__(256)
rows_adam_kernel(int P, int nviews, int deg, const float *__restrict__ campos_all, const float4 *__restrict__ rows_all, uint32_t cap,
                 const uint32_t *__restrict__ viewmask, const uint32_t *__restrict__ slots, RawBwd raw) {
    __shared__ float sh_stage[256 * W3D_SHROW];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = gid < P;
    const int g = active ? gid : P - 1;
    const int ncoef = (deg + 1) * (deg + 1);
    float acc[48], geo[11];
#pragma unroll
    for (int i = 0; i < 48; i++) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 11; i++) geo[i] = 0.f;
    uint32_t mask = active ? viewmask[g] : 0u;
    // (the pre-update position: read before this kernel's own Adam phase for the xyz block)
    const float *xyz = raw.pw[0];
    const float p[3] = {xyz[3 * (size_t)g], xyz[3 * (size_t)g + 1], xyz[3 * (size_t)g + 2]};
    while (mask) {
        const int vw = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        if (vw >= nviews) break;
        const uint32_t r = slots[(size_t)vw * P + g];
        const float4 *row = rows_all + 4 * ((size_t)vw * cap + r);
        const float4 a = row[0], b = row[1], c = row[2], d = row[3];
        geo[0] += b.y; geo[1] += b.z; geo[2] += b.w; geo[3] += c.x;
        geo[4] += c.y; geo[5] += c.z; geo[6] += c.w;
        geo[7] += d.x; geo[8] += d.y; geo[9] += d.z; geo[10] += d.w;
        const float dr = a.z, dg = a.w, db = b.x;
        if (dr == 0.f && dg == 0.f && db == 0.f) continue;      // (fully clamped in that view)
        const float d0 = p[0] - campos_all[3 * vw], d1 = p[1] - campos_all[3 * vw + 1], d2 = p[2] - campos_all[3 * vw + 2];
        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float x = d0 / len, y = d1 / len, z = d2 / len;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float B[16];
        B[0] = SH_C0;
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
        B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
        B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z; B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        B[14] = SH_C3[5] * z * (xx - yy); B[15] = SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float bk = (k < ncoef) ? B[k] : 0.f;
            acc[3 * k] += bk * dr; acc[3 * k + 1] += bk * dg; acc[3 * k + 2] += bk * db;
        }
    }
#pragma unroll
    for (int i = 0; i < 48; i++) sh_stage[threadIdx.x * W3D_SHROW + i] = acc[i];
    __syncthreads();
    const size_t g0 = (size_t)blockIdx.x * 256;
    const int rows = (int)min((size_t)256, (size_t)P - g0);
    if (!(raw.skip_mask & 2u)) coop_adam<3, 1>(raw, 1, g0, rows, sh_stage, W3D_SHROW, 0);
    if (!(raw.skip_mask & 4u)) coop_adam<45, 4>(raw, 2, g0, rows, sh_stage, W3D_SHROW, 3);
    // the narrow blocks through the (now free) stage: row = [dxyz 3 | dopacity 1 | dscaling 3 | drotation 4], stride 15
    __syncthreads();
    float *st2 = sh_stage + threadIdx.x * 15;
#pragma unroll
    for (int i = 0; i < 11; i++) st2[i] = geo[i];
    __syncthreads();
    if (!(raw.skip_mask & 1u)) coop_adam<3, 1>(raw, 0, g0, rows, sh_stage, 15, 0);
    if (!(raw.skip_mask & 8u)) coop_adam<1, 1>(raw, 3, g0, rows, sh_stage, 15, 3);
    if (!(raw.skip_mask & 16u)) coop_adam<3, 1>(raw, 4, g0, rows, sh_stage, 15, 4);
    if (!(raw.skip_mask & 32u)) coop_adam<4, 1>(raw, 5, g0, rows, sh_stage, 15, 7);
}

}  // namespace

int w3d_launch_rows_adam(int32_t P, int32_t nviews, int32_t sh_degree, const float *campos_all, const float *rows_all, uint32_t cap,
                         const uint32_t *viewmask, const uint32_t *slots, const w3d_raw_blocks &pw, const w3d_adam_fused &a,
                         hipStream_t stream) {
    if (P == 0) return W3D_OK;
    RawBwd raw = {};
    float *const pws[6] = {pw.xyz, pw.f_dc, pw.f_rest, pw.opacity, pw.scaling, pw.rotation};
    float *const ms[6] = {a.exp_avg.xyz, a.exp_avg.f_dc, a.exp_avg.f_rest, a.exp_avg.opacity, a.exp_avg.scaling, a.exp_avg.rotation};
    float *const vs[6] = {a.exp_avg_sq.xyz, a.exp_avg_sq.f_dc, a.exp_avg_sq.f_rest, a.exp_avg_sq.opacity, a.exp_avg_sq.scaling,
                          a.exp_avg_sq.rotation};
    for (int i = 0; i < 6; i++) {
        if (!pws[i] || !ms[i] || !vs[i]) { w3d_set_error("rows_adam: NULL parameter / moment block"); return W3D_ERR_INVALID; }
        raw.pw[i] = pws[i]; raw.m[i] = ms[i]; raw.v[i] = vs[i];
        if (a.skip[i]) { raw.skip_mask |= 1u << i; continue; }
        raw.step_size[i] = a.lr[i] / a.bias_correction1[i];
        raw.inv_sqrt_bc2[i] = 1.0f / sqrtf(a.bias_correction2[i]);
    }
    raw.b1 = a.beta1; raw.b2 = a.beta2; raw.eps = a.eps;
    W3D_PROF("rows_adam", stream);
    hipLaunchKernelGGL(rows_adam_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, nviews, sh_degree, campos_all,
                       reinterpret_cast<const float4 *>(rows_all), cap, viewmask, slots, raw);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

int w3d_launch_dcolor_extract(const W3DLayout &L, const char *state, const float *grad2d, float *dcolor_out, hipStream_t stream) {
    if (L.P == 0) return W3D_OK;
    hipLaunchKernelGGL(dcolor_extract_kernel, dim3((L.P + 255) / 256), dim3(256), 0, stream, L.P,
                       reinterpret_cast<const ushort4 *>(state + L.o_rect), reinterpret_cast<const uint8_t *>(state + L.o_clamped),
                       grad2d, dcolor_out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

int w3d_launch_sh_adam_lowrank(int32_t P, int32_t nviews, int32_t sh_degree, const float *campos_all, const float *xyz,
                               const float *dcolor_all, float *f_dc, float *f_rest, float *m_dc, float *v_dc, float *m_rest,
                               float *v_rest, float lr_dc, float lr_rest, int skip_dc, int skip_rest, float beta1, float beta2,
                               float eps, float bc1, float bc2, hipStream_t stream) {
    if (P == 0 || nviews == 0) return W3D_OK;
    RawBwd raw = {};
    raw.pw[1] = f_dc; raw.m[1] = m_dc; raw.v[1] = v_dc; raw.step_size[1] = lr_dc / bc1;
    raw.pw[2] = f_rest; raw.m[2] = m_rest; raw.v[2] = v_rest; raw.step_size[2] = lr_rest / bc1;
    raw.skip_mask = (skip_dc ? 2u : 0u) | (skip_rest ? 4u : 0u);
    raw.b1 = beta1; raw.b2 = beta2; raw.eps = eps; raw.inv_sqrt_bc2[1] = raw.inv_sqrt_bc2[2] = 1.0f / sqrtf(bc2);
    W3D_PROF("sh_adam_lowrank", stream);
    hipLaunchKernelGGL(sh_adam_lowrank_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, nviews, sh_degree, campos_all, xyz,
                       dcolor_all, raw);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

int w3d_launch_preprocess(const W3DLayout &L, const w3d_view &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          const float *rotations, const float *cov3D_precomp, int32_t *radii, char *state,
                          char *scratch, const float *f_rest_raw, const uint8_t *used_mask, hipStream_t stream) {
    // counters: [0] is written by the depth sort's last pass, [1] by the tile scan
    if (L.P == 0) {
        W3D_HIP_CHECK(hipMemsetAsync(state + L.o_counters, 0, 64, stream));
        return W3D_OK;
    }
    const int block = W3D_PRE_BLOCK, grid = (L.P + block - 1) / block;
    W3D_PROF("preprocess_fwd", stream);
#define ARGS                                                                                                          \
    v, L.P, L.gx, L.gy, L.lsx, L.lsy, means3D, shs, f_rest_raw, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, \
        reinterpret_cast<float4 *>(state + L.o_grec), reinterpret_cast<ushort4 *>(state + L.o_rect),                 \
        reinterpret_cast<uint8_t *>(state + L.o_clamped), reinterpret_cast<uint32_t *>(scratch + L.s_keys0),         \
        reinterpret_cast<uint32_t *>(scratch + L.s_vals0), reinterpret_cast<uint32_t *>(state + L.o_counters),       \
        reinterpret_cast<uint4 *>(state + L.o_tile_mask), used_mask, reinterpret_cast<uint4 *>(scratch + L.s_minmax)
    if (f_rest_raw) hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3(grid), dim3(block), 0, stream, ARGS);
    else hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3(grid), dim3(block), 0, stream, ARGS);
#undef ARGS
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_launch_flash_extras(const W3DLayout &L, const w3d_view &v, const int32_t *radii, char *state, float *proj_xy,
                            float *gs_depth, hipStream_t stream) {
    if (L.P == 0 || (!proj_xy && !gs_depth)) return W3D_OK;
    const int block = 256, grid = (L.P + block - 1) / block;
    hipLaunchKernelGGL(flash_extras_kernel, dim3(grid), dim3(block), 0, stream, L.P, (const uint32_t *)nullptr,
                       reinterpret_cast<const float4 *>(state + L.o_grec),
                       reinterpret_cast<const ushort4 *>(state + L.o_rect), radii, proj_xy, gs_depth);
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_launch_preprocess_backward(const W3DLayout &L, const w3d_view &v, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *scales, const float *rotations,
                                   const float *cov3D_precomp, const char *state, float *grad2d,
                                   float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dshs,
                                   float *dL_dopacity, float *dL_dscales, float *dL_drots, float *dL_dcov3D,
                                   const W3DRawBwdArgs *rawargs, hipStream_t stream) {
    if (L.P == 0) return W3D_OK;
    const int block = 256, grid = (L.P + block - 1) / block;
    const bool has_sh = (shs != nullptr), has_sr = (scales != nullptr);
    (void)colors_precomp;
    RawBwd raw = {};
    if (rawargs) {
        raw.f_rest = rawargs->f_rest; raw.opacity_logit = rawargs->opacity_logit; raw.dL_df_rest = rawargs->dL_df_rest;
        raw.gnorm_out = rawargs->gnorm_out; raw.radii = rawargs->radii; raw.accum = rawargs->accum;
        raw.denom = rawargs->denom; raw.max_radii = rawargs->max_radii;
    }
    const bool fused_adam = rawargs && rawargs->adam;
    if (fused_adam) {
        if (v.sh_coeffs != 16) { w3d_set_error("fused Adam needs 16 SH coefficients"); return W3D_ERR_INVALID; }
        const w3d_adam_fused &a = *rawargs->adam;
        const w3d_raw_blocks &pw = *rawargs->params_rw;
        float *const pws[6] = {pw.xyz, pw.f_dc, pw.f_rest, pw.opacity, pw.scaling, pw.rotation};
        float *const ms[6] = {a.exp_avg.xyz, a.exp_avg.f_dc, a.exp_avg.f_rest, a.exp_avg.opacity, a.exp_avg.scaling, a.exp_avg.rotation};
        float *const vs[6] = {a.exp_avg_sq.xyz, a.exp_avg_sq.f_dc, a.exp_avg_sq.f_rest, a.exp_avg_sq.opacity, a.exp_avg_sq.scaling,
                              a.exp_avg_sq.rotation};
        for (int i = 0; i < 6; i++) {
            if (!pws[i] || !ms[i] || !vs[i]) { w3d_set_error("fused Adam: NULL parameter / moment block"); return W3D_ERR_INVALID; }
            raw.pw[i] = pws[i]; raw.m[i] = ms[i]; raw.v[i] = vs[i];
            raw.step_size[i] = a.lr[i] / a.bias_correction1[i];
            raw.inv_sqrt_bc2[i] = 1.0f / sqrtf(a.bias_correction2[i]);
            if (a.skip[i]) raw.skip_mask |= 1u << i;
        }
        raw.b1 = a.beta1; raw.b2 = a.beta2; raw.eps = a.eps;
        raw.counters = reinterpret_cast<const uint32_t *>(state + L.o_counters);
    }
#define LAUNCH(A, B, C)                                                                                                  \
    if (A && v.sh_coeffs == 16)                                                                                          \
        LAUNCH2(A, B, C, true);                                                                                          \
    else                                                                                                                 \
        LAUNCH2(A, B, C, false)
#define LAUNCH2(A, B, C, D)                                                                                              \
    hipLaunchKernelGGL((preprocess_bwd_kernel<A, B, C, D>), dim3(grid), dim3(block), 0, stream, v, L.P, means3D, shs, scales, \
                       rotations, cov3D_precomp, raw, reinterpret_cast<const ushort4 *>(state + L.o_rect),               \
                       reinterpret_cast<const uint8_t *>(state + L.o_clamped), grad2d, dL_dmeans3D, dL_dmeans2D,         \
                       dL_dcolors, dL_dshs, dL_dopacity, dL_dscales, dL_drots, dL_dcov3D)
    W3D_PROF("preprocess_bwd", stream);
    if (rawargs && rawargs->lowrank == 2) {
        if (v.sh_coeffs != 16) { w3d_set_error("gradient rows need 16 SH coefficients"); return W3D_ERR_INVALID; }
        raw.rows_out = reinterpret_cast<float4 *>(rawargs->rows_out); raw.rows_cap = rawargs->rows_cap;
        raw.rows_count = rawargs->rows_count; raw.norm_scale = rawargs->norm_scale;
        W3D_HIP_CHECK(hipMemsetAsync(raw.rows_count, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, true, true, 3>), dim3(grid), dim3(block), 0, stream, v, L.P, means3D, shs,
                           scales, rotations, cov3D_precomp, raw, reinterpret_cast<const ushort4 *>(state + L.o_rect),
                           reinterpret_cast<const uint8_t *>(state + L.o_clamped), grad2d, dL_dmeans3D, dL_dmeans2D, dL_dcolors,
                           dL_dshs, dL_dopacity, dL_dscales, dL_drots, dL_dcov3D);
    } else if (rawargs && rawargs->lowrank) {
        if (v.sh_coeffs != 16) { w3d_set_error("low-rank colour-gradient output needs 16 SH coefficients"); return W3D_ERR_INVALID; }
        raw.dcolor_out = rawargs->dcolor_out;
        hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, true, true, 2>), dim3(grid), dim3(block), 0, stream, v, L.P, means3D, shs,
                           scales, rotations, cov3D_precomp, raw, reinterpret_cast<const ushort4 *>(state + L.o_rect),
                           reinterpret_cast<const uint8_t *>(state + L.o_clamped), grad2d, dL_dmeans3D, dL_dmeans2D, dL_dcolors,
                           dL_dshs, dL_dopacity, dL_dscales, dL_drots, dL_dcov3D);
    } else if (fused_adam)
        hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, true, true, 1>), dim3(grid), dim3(block), 0, stream, v, L.P, means3D, shs,
                           scales, rotations, cov3D_precomp, raw, reinterpret_cast<const ushort4 *>(state + L.o_rect),
                           reinterpret_cast<const uint8_t *>(state + L.o_clamped), grad2d, dL_dmeans3D, dL_dmeans2D, dL_dcolors,
                           dL_dshs, dL_dopacity, dL_dscales, dL_drots, dL_dcov3D);
    else if (rawargs) LAUNCH(true, true, true);
    else if (has_sh && has_sr) LAUNCH(true, true, false);
    else if (has_sh) LAUNCH(true, false, false);
    else if (has_sr) LAUNCH(false, true, false);
    else LAUNCH(false, false, false);
#undef LAUNCH
#undef LAUNCH2
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}
