// w3d_common.h — shared declarations of the gfx950 rasterizer kernels (internal; the public
// boundary is include/w3d.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/w3d.h"

#define W3D_TILE 16
#define W3D_LOG2E 1.4426950408889634f
#define W3D_WAVE 64
#define W3D_NEAR 0.2f
#define W3D_INVALID_KEY 0xFFFFFFFFu

// Upper bounds of the binning geometry (see w3d_binning.hip).
#ifndef W3D_MAX_CHUNKS
#define W3D_MAX_CHUNKS 2048       // depth-contiguous chunks of Gaussians, one wave each
#endif
#define W3D_CHUNK_MAX 65472       // per-chunk Gaussian count must fit a u16 counter (multiple of 64)
#define W3D_SCAN_SEGS 16
// depth sort (w3d_binning.hip): the view's depth interval is cut into W3D_DB_BINS buckets; every workgroup of the preprocess kernel
// (W3D_PRE_BLOCK Gaussians) leaves {min, max} of its visible depth keys in s_minmax — plain stores, nothing to initialise
#define W3D_DB_BITS 10
#define W3D_DB_BINS (1 << W3D_DB_BITS)
#define W3D_PRE_BLOCK 256

static inline uint64_t w3d_align_up(uint64_t x, uint64_t a = 256) { return (x + a - 1) / a * a; }

// Everything derived from (P, H, W) that host and kernels agree on.
struct W3DLayout {
    int32_t P, H, W, gx, gy, T;
    // the grid the per-tile LISTS live on (w3d_view.list_share): list cell (lx, ly) serves the tiles (lx << lsx .. , ly << lsy ..);
    // lsx = lsy = 0: one list per tile.  Binning (tile masks, count, scan, fill) works on this grid, the blend on the tile grid.
    int32_t lsx, lsy, lgx, lgy, LT;
    uint32_t chunk;   // Gaussians per binning chunk (multiple of 64)
    uint32_t C;       // number of chunks
    uint32_t seg;     // chunks per scan segment
    // ---- state buffer (kept until backward)
    uint64_t o_counters;   // u32[16]: [0]=num_visible [1]=num_rendered [3]=capacity of the list buffer given to stage 2
                           //          [4]=min visible depth key [5]=bucket shift of the depth sort (w3d_binning.hip)

    uint64_t o_grec;       // float4[4P]: ONE 64-B line per Gaussian, in the layout the blend kernels stage (w3d_render.hip StagedLDS):
                           //   [0] x, y, rect lo (minx | miny << 16), rect hi (maxx | maxy << 16) — the PUBLISHED tile rect: with shared
                           //       lists (w3d_view.list_share) a tile also sees entries of its neighbours and drops those whose rect it
                           //       is not in (the reference blends a Gaussian in the tiles of its 3-sigma bounding square only);
                           //       the staging lane replaces the two words by pmin * log2e and the Gaussian's index
                           //   [1] conic.x, conic.y, conic.z, opacity   [2] r, g, b, depth
                           //   [3] -0.5 log2e conic.x, -log2e conic.y, -0.5 log2e conic.z, opacity
                           // READERS MUST GATE ON o_rect (or radii): a culled Gaussian's record is NOT written by the forward — the line
                           // keeps whatever a recycled state buffer held; nothing reads the record of a Gaussian that is in no
                           // list (the blend gathers through the lists; flash_extras, det_gather and w3d_debug_gaussian_records'
                           // callers test the rect first)
                           // (rounds 1-4 kept xy / conic+opacity / rgb+depth in three arrays: three cache lines per gathered list
                           //  entry — 375 B through the fabric per walked entry of the blend backward, PMC — instead of one)
    uint64_t o_rect;       // ushort4[P] (minx,miny,maxx,maxy) tile units
    uint64_t o_clamped;    // u8[P] bit c set: SH colour channel c clamped at 0
    uint64_t o_tile_mask;  // uint4[P] {rect lo, rect hi, mask lo, mask hi}; mask bit k: k-th tile of the rect (row-major) is
                           // reachable (tile_cull) — rect and mask in ONE 16-B record so that the depth-order gather reads one line
    uint64_t o_tile_start; // u32[T+1]
    uint64_t o_final_T;    // float[HW]
    uint64_t o_n_contrib;  // u32[HW]
    uint64_t o_tile_walk;  // u32[T]: entries of the tile's list the forward blended at all (max n_contrib of its pixels)
    uint64_t o_tile_order; // u32[8 * order_cap]: the blend kernels' block -> (tile, part) map (w3d_render.hip tile_schedule_kernel)
    uint64_t state_bytes;
    // ---- scratch buffer (forward temporaries)
    uint64_t s_keys0, s_keys1, s_vals0, s_vals1; // u32[P] each (depth keys, Gaussian ids)
    uint64_t s_hist;       // u32[W3D_DB_BINS * pitch] per-run bucket histograms of the depth sort (pitch = sort_waves rounded up to 4)
    uint64_t s_rowtot;     // u32[W3D_DB_BINS] bucket totals
    uint64_t s_bstart;     // u32[W3D_DB_BINS + 1] bucket boundaries in the depth order
    uint64_t s_minmax;     // uint4[ceil(P / W3D_PRE_BLOCK)] {min, max} of the visible depth keys of every preprocess workgroup + two sample keys
    uint64_t s_grid;       // uint4[1 + 256]: the depth sort's bucket grid of this view (w3d_binning.hip DepthGridHead + segment table),
                           // followed by uint2[W3D_DB_BINS]: every bucket's {first key offset, width}
    uint64_t s_cnt;        // u16[C*T] per-chunk per-tile counts
    uint64_t s_off;        // u32[C*T] per-chunk per-tile list offsets
    uint64_t s_part;       // u32[SEGS*T]
    uint64_t s_rec, s_rec_mask; // uint4[P], uint2[P]: depth-ordered {id, rect} records and tile masks
    uint64_t scratch_bytes;
    uint32_t order_cap;    // block -> tile map: entries per XCD
    uint32_t sort_waves;   // waves used by the radix passes
    uint32_t sort_items;   // keys per wave per pass (multiple of 64)
};

// entries per XCD of the blend kernels' block -> (tile, part) map: frames of up to W3D_SCHED_MAX_TILES tiles get 1.5x their share of the
// tiles (work-balanced XCD ranges are uneven in tile count, and long tiles are cut into 2 or 4 part-waves) — small frames, whose tiles
// number fewer than a quarter of the chip's wave slots, 4x: every tile can run as four quadrant waves; larger frames the share
#define W3D_SCHED_MAX_TILES 8192u
static inline uint32_t w3d_order_cap(uint32_t T) {
    if (T > W3D_SCHED_MAX_TILES) return (T + 7u) / 8u;
    if (T <= 1024u) return 4u * ((T + 7u) / 8u) + 4u;        // (an even range of whole-frame quarters always fits)
    return (T + T / 2u + 7u) / 8u + 1u;
}

static inline int w3d_make_layout(int32_t P, int32_t H, int32_t W, W3DLayout *L) {
    if (P < 0 || H <= 0 || W <= 0) return W3D_ERR_INVALID;
    L->P = P; L->H = H; L->W = W;
    L->gx = (W + W3D_TILE - 1) / W3D_TILE;
    L->gy = (H + W3D_TILE - 1) / W3D_TILE;
    L->T = L->gx * L->gy;
    if (L->gx > 65535 || L->gy > 65535) return W3D_ERR_UNSUPPORTED;
    L->lsx = L->lsy = 0; L->lgx = L->gx; L->lgy = L->gy; L->LT = L->T;
    uint64_t Pp = P > 0 ? (uint64_t)P : 1;
    const uint64_t max_chunks = W3D_MAX_CHUNKS;
    uint64_t chunk = (Pp + max_chunks - 1) / max_chunks;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk < 64) chunk = 64;
    if (chunk > W3D_CHUNK_MAX) chunk = W3D_CHUNK_MAX;
    L->chunk = (uint32_t)chunk;
    L->C = (uint32_t)((Pp + chunk - 1) / chunk);
    L->seg = (L->C + W3D_SCAN_SEGS - 1) / W3D_SCAN_SEGS;
    uint64_t HW = (uint64_t)H * W, T = (uint64_t)L->T;
    uint64_t o = 0;
    L->o_counters = o;   o += w3d_align_up(16 * 4);
    L->o_grec = o;       o += w3d_align_up(Pp * 64);
    L->o_rect = o;       o += w3d_align_up(Pp * 8);
    L->o_clamped = o;    o += w3d_align_up(Pp);
    L->o_tile_mask = o;  o += w3d_align_up(Pp * 16);
    L->o_tile_start = o; o += w3d_align_up((T + 1) * 4);
    L->o_final_T = o;    o += w3d_align_up(HW * 4);
    L->o_n_contrib = o;  o += w3d_align_up(HW * 4);
    L->o_tile_walk = o;  o += w3d_align_up(T * 4);
    L->order_cap = w3d_order_cap((uint32_t)T);
    L->o_tile_order = o; o += w3d_align_up((uint64_t)8 * L->order_cap * 4);
    L->state_bytes = o;
    // depth sort geometry: one wave per contiguous run of sort_items keys
    // (rounds 1-5, the 12-launch radix sort, at P = 2 M: 512 runs 0.216 ms, 1024 0.161, 2048 0.177, 4096 0.214; the bucket sort's
    //  histogram + scatter: 1024 runs 52 us, 2048 runs 44 us, 4096 runs of 512 keys 47 us)
#ifndef W3D_SORT_RUNS
#define W3D_SORT_RUNS 2048
#endif
    const uint64_t max_runs = W3D_SORT_RUNS;
    uint64_t items = (Pp + max_runs - 1) / max_runs;
    items = (items + 63) / 64 * 64;
#ifndef W3D_SORT_ITEMS_MIN
#define W3D_SORT_ITEMS_MIN 1024
#endif
    if (items < W3D_SORT_ITEMS_MIN) items = W3D_SORT_ITEMS_MIN;
    L->sort_items = (uint32_t)items;
    L->sort_waves = (uint32_t)((Pp + items - 1) / items);
    o = 0;
    L->s_keys0 = o; o += w3d_align_up(Pp * 4);
    L->s_keys1 = o; o += w3d_align_up(Pp * 4);
    L->s_vals0 = o; o += w3d_align_up(Pp * 4);
    L->s_vals1 = o; o += w3d_align_up(Pp * 4);
    L->s_hist = o;  o += w3d_align_up((uint64_t)W3D_DB_BINS * ((L->sort_waves + 3) / 4 * 4) * 4);
    L->s_rowtot = o; o += w3d_align_up(W3D_DB_BINS * 4);
    L->s_bstart = o; o += w3d_align_up((W3D_DB_BINS + 1) * 4);
    L->s_minmax = o; o += w3d_align_up((Pp + W3D_PRE_BLOCK - 1) / W3D_PRE_BLOCK * 16);
    L->s_grid = o;  o += w3d_align_up(257 * 16 + W3D_DB_BINS * 8);
    L->s_cnt = o;   o += w3d_align_up((uint64_t)L->C * T * 2);
    L->s_off = o;   o += w3d_align_up((uint64_t)L->C * T * 4);
    L->s_part = o;  o += w3d_align_up((uint64_t)W3D_SCAN_SEGS * T * 4);
    L->s_rec = o;   o += w3d_align_up(Pp * 16);
    L->s_rec_mask = o; o += w3d_align_up(Pp * 8);
    L->scratch_bytes = o;
    return W3D_OK;
}

// The list grid a view asks for (w3d.h: list_share needs tile_cull and the atomic backward).  Buffer sizes do not depend on it
// (every per-list array is sized for the tile grid, which is never smaller).
static inline void w3d_set_list_share(W3DLayout *L, const w3d_view *v) {
    int mode = (v && v->tile_cull && !v->deterministic) ? v->list_share : 0;
#ifdef W3D_FORCE_LIST_SHARE        // (A/B builds: profiles/build_variant.sh <name> all -DW3D_FORCE_LIST_SHARE=k)
    if (v && v->tile_cull && !v->deterministic) mode = W3D_FORCE_LIST_SHARE;
#endif
    L->lsx = mode >= 1 ? 1 : 0;
    L->lsy = mode >= 2 ? 1 : 0;
    L->lgx = (L->gx + (1 << L->lsx) - 1) >> L->lsx;
    L->lgy = (L->gy + (1 << L->lsy) - 1) >> L->lsy;
    L->LT = L->lgx * L->lgy;
}

// error plumbing (w3d_api.hip)
void w3d_set_error(const char *fmt, ...);
#define W3D_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            w3d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return W3D_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)
#define W3D_LAUNCH_CHECK(view_debug, stream)                                                 \
    do {                                                                                     \
        W3D_HIP_CHECK(hipGetLastError());                                                    \
        if (view_debug) W3D_HIP_CHECK(hipStreamSynchronize(stream));                         \
    } while (0)

// optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg);
// off by default, enabled per kernel-name substring through w3d_profile_enable().
void w3d_prof_begin(const char *name, hipStream_t stream);
void w3d_prof_end(hipStream_t stream);
struct W3DProfScope {
    hipStream_t s;
    W3DProfScope(const char *name, hipStream_t stream) : s(stream) { w3d_prof_begin(name, stream); }
    ~W3DProfScope() { w3d_prof_end(s); }
};
#define W3D_PROF(name, stream) W3DProfScope w3d_prof_scope_(name, stream)

// extra arguments of the raw-parameter backward (NULL for the activated-parameter API)
struct W3DRawBwdArgs {
    const float *f_rest, *opacity_logit;
    float *dL_df_rest, *gnorm_out;
    const int32_t *radii;
    float *accum, *denom, *max_radii;
    int lowrank;                      // view-parallel flavour: geometry gradients only, the SH gradient stays implicit in dL/dRGB;
                                      // 2: ... and neither is written densely — the non-zero 64-B rows are appended to rows_out
    float *rows_out; uint32_t rows_cap; uint32_t *rows_count; float norm_scale;
    float *dcolor_out;                // ... which is written here as (P,3) (NULL: already extracted by w3d_launch_dcolor_extract)
    const w3d_adam_fused *adam;       // non-NULL: apply Adam in place instead of writing gradients
    const w3d_raw_blocks *params_rw;  // ... to these parameter blocks
};

// wave64 ballot of a predicate.  HIP's __ballot(int) makes the backend materialise the bool as 0/1 in a VGPR and compare it
// again (v_cndmask + v_cmp per call); the builtin consumes the lane mask the comparison already produced.
__device__ __forceinline__ uint64_t w3d_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Footprint culling against an exponent that is itself evaluated in fp32.  The blend computes q(d) = 0.5 (A dx^2 + C dy^2) +
// B dx dy per pixel in fp32; its rounding error is bounded by a few ulps of the SUM OF THE TERMS' MAGNITUDES, which for the
// needle-shaped Gaussians of a trained scene (conic condition 1e4 ... 1e7, centres hundreds of pixels away) reaches 0.01 ... 1
// in absolute terms — a pair whose exact q lies outside the ellipse q <= ln(255 o) can evaluate inside it, and the reference
// (which tests every pixel of the bounding square) blends it.  Every geometric cull therefore widens its threshold by
// w3d_q_noise() over the region it decides on: 16 x 2^-24 (dx, dy and the pre-scaled coefficients carry a rounding each, the
// three products and two sums another five) times the largest possible term magnitude there.  For an ordinary Gaussian this is
// 1e-6 ... 1e-4, below the fixed 1e-3 margin.
#define W3D_Q_NOISE 1.0e-6f
__device__ __forceinline__ float w3d_q_noise(float A, float B, float C, float Dx, float Dy) {
    return W3D_Q_NOISE * (0.5f * (fabsf(A) * Dx * Dx + fabsf(C) * Dy * Dy) + fabsf(B) * Dx * Dy);
}
// A C - B^2 of a conic to ~1.5 ulp (Kahan's difference of products): the plain expression loses log2(cond / 4) bits, i.e.
// everything for a needle with axis ratio 1e3.5, and the ellipse's extents sqrt(2 tau C / det) with it.
__device__ __forceinline__ float w3d_conic_det(float A, float B, float C) {
    const float w = B * B;
    const float e = fmaf(-B, B, w);       // w - B*B, exactly
    const float f = fmaf(A, C, -w);       // A*C - w, rounded once
    return f + e;
}

// torch.optim.Adam's element update (no weight decay / amsgrad), shared by the sweep kernel and the fused backward;
// contraction off so that both compile to the same roundings
__device__ __forceinline__ void w3d_adam1(float &p, float g, float &m, float &v, float step_size, float b1, float b2,
                                          float eps, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

// kernels' host launchers (one per .hip file)
int w3d_launch_preprocess(const W3DLayout &L, const w3d_view &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          const float *rotations, const float *cov3D_precomp, int32_t *radii, char *state,
                          char *scratch, const float *f_rest_raw, const uint8_t *used_mask, hipStream_t stream);
int w3d_launch_depth_sort(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, hipStream_t stream);
int w3d_launch_tile_count(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, hipStream_t stream);
int w3d_launch_fill_lists(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, uint32_t *point_list,
                          uint64_t list_capacity, hipStream_t stream);
int w3d_launch_render(const W3DLayout &L, const w3d_view &v, char *state, const uint32_t *point_list, uint64_t list_capacity,
                      float *out_color, float *out_depth, float *out_alpha, const float *gt_mask, int32_t num_obj,
                      float *used_count, int32_t *contrib_num, hipStream_t stream);
int w3d_launch_flash_extras(const W3DLayout &L, const w3d_view &v, const int32_t *radii_unused, char *state,
                            float *proj_xy, float *gs_depth, hipStream_t stream);
int w3d_launch_render_backward(const W3DLayout &L, const w3d_view &v, const char *state, const uint32_t *point_list,
                               const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha, float *grad2d,
                               hipStream_t stream);
int w3d_launch_preprocess_backward(const W3DLayout &L, const w3d_view &v, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *scales, const float *rotations,
                                   const float *cov3D_precomp, const char *state, float *grad2d,
                                   float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dshs,
                                   float *dL_dopacity, float *dL_dscales, float *dL_drots, float *dL_dcov3D,
                                   const struct W3DRawBwdArgs *rawargs, hipStream_t stream);
int w3d_launch_mask_binarize(int H, int W, int C, const uint8_t *pixels, float *out, hipStream_t stream);
int w3d_launch_mask_iou(int H, int W, int K, const float *alpha, float thresh, const uint8_t *masks, uint32_t *out,
                        hipStream_t stream);
int w3d_launch_knn(int32_t N, const float *points, float *out, hipStream_t stream);
uint64_t w3d_knn_scratch_bytes(int32_t N);
int w3d_launch_knn_grid(int32_t N, const float *points, float *out, char *scratch, hipStream_t stream);
int w3d_launch_dcolor_extract(const W3DLayout &L, const char *state, const float *grad2d, float *dcolor_out, hipStream_t stream);
int w3d_launch_rows_adam(int32_t P, int32_t nviews, int32_t sh_degree, const float *campos_all, const float *rows_all, uint32_t cap,
                         const uint32_t *viewmask, const uint32_t *slots, const w3d_raw_blocks &pw, const w3d_adam_fused &a,
                         hipStream_t stream);
int w3d_launch_sh_adam_lowrank(int32_t P, int32_t nviews, int32_t sh_degree, const float *campos_all, const float *xyz,
                               const float *dcolor_all, float *f_dc, float *f_rest, float *m_dc, float *v_dc, float *m_rest,
                               float *v_rest, float lr_dc, float lr_rest, int skip_dc, int skip_rest, float beta1, float beta2,
                               float eps, float bc1, float bc2, hipStream_t stream);

// per-Gaussian 2-D gradient record accumulated by the blend backward (16 floats = one 64-B line)
//  [0] dL/dmean2D.x  [1] dL/dmean2D.y  [2] dL/dconic.x  [3] dL/dconic.y(half)  [4] dL/dconic.z
//  [5] dL/dopacity   [6..8] dL/drgb    [9] dL/ddepth_g   [10..15] unused
#define W3D_G2D_STRIDE 16
