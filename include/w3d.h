/*
 * w3d.h — C-ABI of the MI355X-native Gaussian rasterizer (libw3d_hip.so).
 *
 * This is the drop-in boundary underneath Wheat-3DGS's Python rasterizer modules.  Each entry
 * point replaces one native call of the (un-vendored) CUDA submodules the reference binds:
 *
 *   w3d_forward_stage1 + w3d_forward_stage2
 *        <- diff_gaussian_rasterization._C.rasterize_gaussians, reached from
 *           GaussianRasterizer(...)(means3D, means2D, shs, colors_precomp, opacities, scales,
 *           rotations, cov3D_precomp)          reference gaussian_renderer/__init__.py:55,89-97
 *        <- flashsplat_rasterization's forward (8 outputs; gt_mask / num_obj)
 *                                              reference gaussian_renderer/__init__.py:149,194-204
 *   w3d_backward
 *        <- diff_gaussian_rasterization._C.rasterize_gaussians_backward, reached from
 *           loss.backward()                    reference train_vanilla_3dgs.py:80
 *   w3d_knn_dist2
 *        <- simple_knn._C.distCUDA2            reference scene/gaussian_model.py:20,148
 *   w3d_l1_ssim_fwd_bwd (next-row N1)
 *        <- utils/loss_utils.py:17-63 (l1_loss, ssim) as used at train_vanilla_3dgs.py:77-80
 *   w3d_flash_reblend (next-row N4)
 *        <- the per-mask inner call of run_3d_seg.py:88-97 / :127-134 (same view, another gt_mask)
 *   w3d_backward_raw_adam (next-row N2, single GPU)
 *        <- train_vanilla_3dgs.py:80 loss.backward() + :113-115 optimizer.step() / zero_grad()
 *   w3d_backward_raw_lowrank, w3d_sh_adam_lowrank, w3d_pack_gradient_rows, w3d_backward_raw_rows, w3d_apply_gradient_rows,
 *   w3d_index_gradient_rows, w3d_rows_norm_sum, w3d_rows_adam (row e, view-parallel exchange)
 *        <- no reference counterpart (the reference is single-GPU); same arithmetic as
 *           w3d_backward_raw + w3d_adam_step on the mean gradient of the views
 *   w3d_densify_compact (next-row N3)
 *        <- scene/gaussian_model.py:332-397,441-455 (optimizer-state surgery of densify / prune)
 *   w3d_adam_step (next-row N2)
 *        <- torch.optim.Adam over the 6 parameter groups, scene/gaussian_model.py:172-182,
 *           stepped at train_vanilla_3dgs.py:113-115
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every pointer named in a
 * signature is a DEVICE pointer unless its name ends in _host.  All arrays are fp32, dense,
 * row-major with the shapes the reference's Python passes (means3D (P,3), shs (P,M,3),
 * opacities (P,1), scales (P,3), rotations (P,4), cov3D_precomp (P,6), images (C,H,W)).
 * viewmatrix / projmatrix are the TRANSPOSED matrices exactly as scene/cameras.py:56-58 builds
 * them.  The caller owns every buffer (outputs, state, scratch, lists) and allocates them with
 * its own allocator (torch's, in the Python host); the library never allocates device memory
 * and keeps no global mutable state besides the last-error string (thread-local) and the opt-in, mutex-guarded event
 * records of w3d_profile_enable / w3d_profile_collect.
 * All work is enqueued on `stream`; nothing synchronises unless stated.
 *
 * Return value: 0 on success, a W3D_ERR_* code otherwise (w3d_last_error() has the text).
 */
#ifndef W3D_H_
#define W3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *w3d_stream_t; /* hipStream_t */

enum {
    W3D_OK = 0,
    W3D_ERR_INVALID = 1,  /* bad argument (null pointer, non-positive size, unsupported degree...) */
    W3D_ERR_CAPACITY = 2, /* a caller-provided buffer is too small */
    W3D_ERR_HIP = 3,      /* a HIP runtime call or kernel launch failed */
    W3D_ERR_UNSUPPORTED = 4
};

/* GaussianRasterizationSettings of the reference (gaussian_renderer/__init__.py:40-53), flattened. */
typedef struct w3d_view {
    uint32_t struct_size;    /* = sizeof(w3d_view) of the header the CALLER was built against.  Every entry point that takes a
                              * view refuses one whose size differs from the library's own (W3D_ERR_INVALID, "w3d_view size"):
                              * fields were appended to this struct in the past (tile_walk_hint, records_kept_clean), and a
                              * client compiled against a shorter struct would make the library read past it. */
    int32_t image_height, image_width;
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t sh_degree;       /* ACTIVE degree 0..3 */
    int32_t sh_coeffs;       /* coefficients stored per Gaussian in `shs` (16 for max degree 3) */
    int32_t prefiltered;     /* accepted, unused (always False in the reference) */
    int32_t debug;           /* 1: synchronise + check after every kernel */
    const float *bg;         /* device (3,) */
    const float *viewmatrix; /* device (4,4) transposed world->view */
    const float *projmatrix; /* device (4,4) transposed full projection */
    const float *campos;     /* device (3,) */
    int32_t tile_cull;       /* 0: per-tile lists = every tile of the 3-sigma bounding square (the published
                              * rule; lists comparable entry by entry with a reference implementation);
                              * 1: additionally drop (Gaussian, tile) instances whose footprint provably
                              * cannot reach alpha >= 1/255 on any pixel of the tile under the blend's own fp32
                              * evaluation of the exponent (the threshold carries that evaluation's rounding
                              * bound over the tile) — bit-identical images, identical gradients, 40-60 % of
                              * the list entries */
    int32_t deterministic;   /* backward only.  0: the blend backward adds every (tile, Gaussian) contribution to the
                              * Gaussian's record with float atomics — fastest, but the order of the additions (hence the
                              * last bits of every gradient) differs from run to run, as in the reference's CUDA kernels.
                              * 1: each contribution is stored in the slot of its list entry and one thread per Gaussian
                              * adds its slots in ascending tile order: bit-identical gradients run to run (the sanitizer
                              * mode of SURVEY.md section 5).  Needs scratch of w3d_backward_det_sizes() bytes. */
    uint64_t det_list_capacity; /* deterministic = 1: entries of point_list (= the capacity stage 2 was given) */
    uint32_t *tile_walk_hint;   /* device, u32[tiles], nullable; speed only, never results.  How long a tile's wave takes — how far
                                 * into its list it blends before its pixels saturate — is only known afterwards; a training loop
                                 * renders the same camera again and again, so the caller may keep one such array per camera: stage 2
                                 * builds its block -> (tile, part) schedule from the values it finds (work-balanced XCD ranges,
                                 * longest first, tiles far longer than the chip's per-slot share cut into 2 or 4 part-waves; zeros:
                                 * image order) and overwrites them with this render's walk lengths (for a tile that ran as
                                 * part-waves: the length one of its parts reported — a lower bound).  Any contents are safe. */
    int32_t records_kept_clean; /* backward only, deterministic = 0.  The first P * 64 bytes of the backward scratch are the
                                 * per-Gaussian records the blend backward adds to; they must start at zero.  0: the backward
                                 * zeroes the visible Gaussians' records itself (a pass of its own).  1: the caller promises they
                                 * ARE zero on entry — a scratch buffer it keeps from call to call, zero-filled once — and the
                                 * per-Gaussian backward, which consumes every record it reads, writes zeros back: no zeroing
                                 * pass, and the buffer is clean again when the call (for the two-call form: the second call)
                                 * has run. */
    int32_t list_share;         /* speed only, never results; honoured with tile_cull = 1 and deterministic = 0 (otherwise 0 is
                                 * used).  0: one depth-ordered list per 16x16 tile (the published layout).  1: ONE list per
                                 * 32x16 pair of horizontally adjacent tiles, 2: per 32x32 block of four — the exact union of the
                                 * member tiles' culled lists, in the same (depth, index) order; every tile still blends on its own
                                 * wave and skips the entries that cannot touch it, so images, gradients and per-pixel outputs are
                                 * the ones of mode 0 (float-atomic order aside), while the binning stage counts, scans and fills
                                 * 40 % / 64 % fewer instances.  num_rendered is then the length of the shared lists.  The SAME
                                 * value (and the same tile_cull / deterministic) must be given to stage 1, stage 2 and the
                                 * backward of a view; w3d_debug_tile_ranges reports, per 16x16 tile, the range of the list it reads. */
} w3d_view;

/* Version of this ABI: major * 100 + minor.  The major number changes whenever a struct of this header changes its layout or
 * an entry point its signature; a binding must refuse a library whose major number differs from the header it mirrors. */
#define W3D_ABI_VERSION 306
int w3d_version(void);
const char *w3d_last_error(void);

/* Bytes of the per-call `state` buffer (kept alive until backward has run) and of the
 * temporary `scratch` buffer (may be released after stage 2).  Both must be 256-B aligned. */
int w3d_forward_sizes(int32_t P, int32_t H, int32_t W, uint64_t *state_bytes, uint64_t *scratch_bytes);

/* Stage 1: preprocess (cull, project, covariance, SH->RGB), global depth sort of the Gaussians,
 * per-tile counting and scan.  Writes radii (P,) int32.  If counts_host is non-NULL the two
 * counters {num_visible, num_rendered} are copied there and the stream is synchronised (the one
 * host sync of a forward pass: the per-tile list length R = num_rendered sizes stage 2's list). */
int w3d_forward_stage1(const w3d_view *view, int32_t P, const float *means3D, const float *shs,
                       const float *colors_precomp, const float *opacities, const float *scales,
                       const float *rotations, const float *cov3D_precomp, int32_t *radii, void *state,
                       void *scratch, uint32_t *counts_host, w3d_stream_t stream);

/* Stage 2: fill the per-tile depth-ordered lists (point_list, capacity in entries) and blend
 * front-to-back.  The capacity must be >= num_rendered for correct output; if it is smaller (a caller
 * that sized the buffer speculatively to avoid stage 1's host sync) nothing is written or read beyond
 * it, the outputs of that view are incomplete, and the caller must repeat the stage with a larger list
 * once it has seen num_rendered (the counters are the first two uint32 of `state`).  out_color (3,H,W), out_depth (1,H,W), out_alpha
 * (1,H,W); out_depth and out_alpha may BOTH be NULL when no FlashSplat output is asked for (a training step whose loss only reads the
 * colour image: the blend then skips the two channels).  FlashSplat extras are all nullable: gt_mask (H,W) fp32 labels in [0,num_obj],
 * used_count (num_obj+1,P) is ACCUMULATED into (caller zero-fills), contrib_num (H,W) int32,
 * proj_xy (P,2), gs_depth (P,).
 * The FlashSplat outputs are FORWARD-ONLY by design: the fork's `mask_grad` setting (a gradient through used_count) is hard-coded
 * to False at the reference's only construction site (gaussian_renderer/__init__.py:145) and every FlashSplat call site runs
 * under no_grad (run_3d_seg.py:91,130,362; eval_wheatgs.py), so no entry point of this ABI differentiates them; the Python
 * host refuses mask_grad=True (NotImplementedError) rather than returning an un-differentiated result silently. */
int w3d_forward_stage2(const w3d_view *view, int32_t P, void *state, void *scratch, uint32_t *point_list,
                       uint64_t list_capacity, float *out_color, float *out_depth, float *out_alpha,
                       const float *gt_mask, int32_t num_obj, float *used_count, int32_t *contrib_num,
                       float *proj_xy, float *gs_depth, w3d_stream_t stream);

int w3d_backward_sizes(int32_t P, uint64_t *scratch_bytes);
/* scratch of a backward call with view->deterministic = 1 (64 B per Gaussian + 64 B per list entry) */
int w3d_backward_det_sizes(int32_t P, uint64_t list_capacity, uint64_t *scratch_bytes);

/* Backward of stage1+stage2.  dL_dcolor (3,H,W) required; dL_ddepth / dL_dalpha (H,W) nullable
 * (Wheat-3DGS's loss never feeds them).  Outputs are OVERWRITTEN (zero rows for culled
 * Gaussians): dL_dmeans3D (P,3), dL_dmeans2D (P,3) [x,y scaled by W/2,H/2; z = 0 — the
 * densification statistic of scene/gaussian_model.py:462], dL_dopacity (P,1), and per input
 * variant dL_dshs (P,M,3) | dL_dcolors (P,3), dL_dscales (P,3) + dL_drots (P,4) | dL_dcov3D (P,6).
 * Unused variant outputs may be NULL. */
int w3d_backward(const w3d_view *view, int32_t P, const float *means3D, const float *shs,
                 const float *colors_precomp, const float *opacities, const float *scales,
                 const float *rotations, const float *cov3D_precomp, const void *state,
                 const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth,
                 const float *dL_dalpha, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors,
                 float *dL_dshs, float *dL_dopacity, float *dL_dscales, float *dL_drots, float *dL_dcov3D,
                 void *scratch, w3d_stream_t stream);

/* FlashSplat's per-mask loop (run_3d_seg.py:88-97 renders the SAME view once per object mask): re-run only the blend on
 * the state and lists a completed w3d_forward_stage2 of this view left behind, with another label image.  Outputs as in
 * stage 2 (colour / depth / alpha are rewritten with identical values; used_count must be zeroed by the caller). */
int w3d_flash_reblend(const w3d_view *view, int32_t P, void *state, const uint32_t *point_list, uint64_t list_capacity,
                      float *out_color, float *out_depth, float *out_alpha, const float *gt_mask, int32_t num_obj,
                      float *used_count, int32_t *contrib_num, w3d_stream_t stream);

/* ---- next-row N2 (fused activations): the same two calls on the PRE-ACTIVATION parameters exactly as
 * GaussianModel stores them (reference scene/gaussian_model.py:101-121 applies exp / sigmoid /
 * F.normalize / cat(dc, rest) on every render call).  The kernels apply the activations and chain
 * their derivatives, so gradients land directly in the caller's parameter-gradient blocks
 * (OVERWRITTEN) — no activation kernels, no cat/split, no autograd accumulation. */
typedef struct w3d_raw_params {
    const float *xyz;      /* (P,3) */
    const float *f_dc;     /* (P,1,3) */
    const float *f_rest;   /* (P,sh_coeffs-1,3) */
    const float *opacity;  /* (P,1) logits */
    const float *scaling;  /* (P,3) log-scales */
    const float *rotation; /* (P,4) un-normalised quaternions */
} w3d_raw_params;
typedef struct w3d_raw_grads {
    float *xyz, *f_dc, *f_rest, *opacity, *scaling, *rotation; /* same shapes as w3d_raw_params */
} w3d_raw_grads;
/* Optional by-products of the backward pass that the training loop needs (all nullable):
 * dL_dmeans2D (P,3); grad2d_norm (P,) = ||dL_dmeans2D[:, :2]|| (0 for culled); and, when
 * xyz_gradient_accum is given, the in-place updates of add_densification_stats + max_radii2D
 * (scene/gaussian_model.py:461-463, train_vanilla_3dgs.py:102) for visible Gaussians. */
typedef struct w3d_densify_stats {
    float *dL_dmeans2D;
    float *grad2d_norm;
    const int32_t *radii;       /* (P,) from the forward */
    float *xyz_gradient_accum;  /* (P,1) += norm */
    float *denom;               /* (P,1) += 1   */
    float *max_radii2D;         /* (P,)  = max(., radii) */
} w3d_densify_stats;
int w3d_forward_stage1_raw(const w3d_view *view, int32_t P, const w3d_raw_params *params, int32_t *radii,
                           void *state, void *scratch, uint32_t *counts_host, w3d_stream_t stream);
int w3d_backward_raw(const w3d_view *view, int32_t P, const w3d_raw_params *params, const void *state,
                     const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth,
                     const float *dL_dalpha, const w3d_raw_grads *grads, const w3d_densify_stats *stats,
                     void *scratch, w3d_stream_t stream);
/* Stage 1 on a SUBSET of the Gaussians <- flashsplat_render(..., used_mask=obj_used_mask), reference
 * gaussian_renderer/__init__.py:151-156,168-170,186-187 (means3D[used_mask], opacity[used_mask], scales / rotations /
 * shs[used_mask]), the most frequent rasterizer call of run_3d_seg.py (:130-134 find_match, :362).  used_mask: (P,) bytes
 * (torch.bool), non-zero = render this Gaussian; NULL = all.  A Gaussian that is left out is culled before any of its
 * parameters is read: radii 0, no tile instances.  Everything downstream (stage 2, the FlashSplat extras) is unchanged
 * and keeps P rows, so row i of every per-Gaussian output still belongs to Gaussian i of the parameter blocks; the
 * reference's outputs are those rows gathered with the same mask. */
int w3d_forward_stage1_raw_subset(const w3d_view *view, int32_t P, const w3d_raw_params *params, const uint8_t *used_mask,
                                  int32_t *radii, void *state, void *scratch, uint32_t *counts_host, w3d_stream_t stream);

/* Single-GPU fusion of the optimizer into the backward pass (next-row N2): the per-Gaussian stage of the backward has
 * the complete gradient of its Gaussian in registers / LDS, so it applies torch.optim.Adam's update
 * (scene/gaussian_model.py:172-182, stepped at train_vanilla_3dgs.py:113-115) to the 59 parameters and their moments
 * right there — the 472 MB gradient bucket is neither written nor read back.  `params` are updated IN PLACE.
 * lr[i] / skip[i] in block order xyz, f_dc, f_rest, opacity, scaling, rotation; a skipped block keeps parameters and
 * moments untouched (the reference's replaced nn.Parameters have .grad None in that step).  If the forward's speculative
 * list buffer was too small (counters on the device say so) NOTHING is updated, so the caller can repeat the view.
 * Needs sh_coeffs == 16.  grad2d_norm of `stats` is always produced; its in-place densification statistics are, like the
 * parameters, only updated when the lists fitted. */
typedef struct w3d_raw_blocks {
    float *xyz, *f_dc, *f_rest, *opacity, *scaling, *rotation;
} w3d_raw_blocks;
typedef struct w3d_adam_fused {
    w3d_raw_blocks exp_avg, exp_avg_sq;
    float lr[6];
    int32_t skip[6];
    float beta1, beta2, eps;
    float bias_correction1[6], bias_correction2[6]; /* per block: torch.optim.Adam keeps one step counter per parameter,
                                                     * and a parameter whose update was skipped does not advance */
} w3d_adam_fused;
int w3d_backward_raw_adam(const w3d_view *view, int32_t P, const w3d_raw_blocks *params, const void *state,
                          const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth,
                          const float *dL_dalpha, const w3d_adam_fused *adam, const w3d_densify_stats *stats,
                          void *scratch, w3d_stream_t stream);

/* View-parallel training: what the ranks exchange.  The SH gradient of ONE view is rank one per Gaussian —
 * dL/dSH[k][c] = basis_k(view direction) * dL/dRGB[c] — so a rank does not ship its 48-float SH gradient rows:
 * w3d_backward_raw_lowrank writes the clamp-masked dL/dRGB (P,3) into dcolor_out (zeros for culled Gaussians) and the
 * gradients of the other four blocks into grads->{xyz, opacity, scaling, rotation} (grads->f_dc / f_rest are not
 * touched and may be NULL); the ranks all-gather the (P,3) arrays, all-reduce the 11 geometry floats, and every rank
 * then runs w3d_sh_adam_lowrank: it rebuilds  sum_v basis_k(normalize(xyz - campos_v)) * dcolor_v  for all n_views in
 * view order (bit-identical on every rank) and applies Adam to f_dc / f_rest and their moments in place.
 * campos_all: (n_views,3) device array; dcolor_all: (n_views,P,3); sh_degree = active degree; 16 coefficients. */
int w3d_backward_raw_lowrank(const w3d_view *view, int32_t P, const w3d_raw_params *params, const void *state,
                             const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth,
                             const float *dL_dalpha, const w3d_raw_grads *grads, float *dcolor_out,
                             const w3d_densify_stats *stats, void *scratch, w3d_stream_t stream);
/* The same backward in two calls, so that the all-gather of dcolor_out can be issued between them and travel while the
 * per-Gaussian backward runs: w3d_backward_blend_dcolor = blend backward + dcolor_out; then w3d_backward_raw_lowrank with
 * dL_dcolor = NULL (blend already done into the same scratch) and dcolor_out = NULL. */
int w3d_backward_blend_dcolor(const w3d_view *view, int32_t P, const void *state, const uint32_t *point_list,
                              const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha, float *dcolor_out,
                              void *scratch, w3d_stream_t stream);
int w3d_sh_adam_lowrank(int32_t P, int32_t n_views, int32_t sh_degree, const float *campos_all, const float *xyz,
                        const float *dcolor_all, float *f_dc, float *f_rest, float *exp_avg_dc, float *exp_avg_sq_dc,
                        float *exp_avg_rest, float *exp_avg_sq_rest, float lr_dc, float lr_rest, int32_t skip_dc,
                        int32_t skip_rest, float beta1, float beta2, float eps, float bias_correction1,
                        float bias_correction2, w3d_stream_t stream);

/* The sparse ("rows") form of the same exchange.  One view gives a gradient only to the Gaussians its pixels blended (a
 * small part of P), every other row of dcolor_out and of the geometry gradients is exactly zero, and adding zeros changes no
 * sum — so a rank ships only its non-zero rows, 16 floats (64 B) each:
 *     {index (u32 bits), grad2d_norm * norm_scale, dL/dRGB[3], d xyz[3], d opacity, d scaling[3], d rotation[4]}.
 * w3d_pack_gradient_rows appends one row for every Gaussian whose 14 gradient floats and grad2d_norm (may be NULL) are not all
 * zero to rows_out (capacity_rows rows, 16-byte aligned; order unspecified) and leaves their number in *count (device memory,
 * zeroed by the call; rows beyond the capacity are counted but not written — P rows always suffice).  The ranks all-gather
 * counts and rows; w3d_apply_gradient_rows, called once per view IN VIEW ORDER on buffers zeroed beforehand, stores the view's
 * colour rows into dcolor_view (P,3) (one plane of w3d_sh_adam_lowrank's dcolor_all), ADDS its geometry rows into
 * sums->{xyz, opacity, scaling, rotation} and its norms into norm_sum (P,) (may be NULL).  min(*count, max_rows) rows are read;
 * count is a device pointer, so no host round trip separates the collective from the kernels.  Replaces, on the reference
 * side, nothing: reference train_vanilla_3dgs.py is single-GPU (SURVEY.md §8e defines the view-parallel step). */
int w3d_pack_gradient_rows(int32_t P, const float *dcolor, const w3d_raw_grads *grads, const float *grad2d_norm,
                           float norm_scale, float *rows_out, uint32_t capacity_rows, uint32_t *count, w3d_stream_t stream);
/* w3d_backward_raw_lowrank + w3d_pack_gradient_rows in one pass (ABI 301): the per-Gaussian backward appends the non-zero rows
 * itself — the same rows, selected by the same rule, in unspecified order — and writes no dense gradient array at all (the 14
 * floats per Gaussian of the two-call form are written and read again only to be packed: 120 MB at 2 M Gaussians).
 * norm_scale = 0: no ||dL/dmean2D|| in the rows (densification over).  count is zeroed by the call. */
int w3d_backward_raw_rows(const w3d_view *view, int32_t P, const w3d_raw_params *params, const void *state,
                          const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                          float norm_scale, float *rows_out, uint32_t capacity_rows, uint32_t *count, void *scratch,
                          w3d_stream_t stream);
int w3d_apply_gradient_rows(int32_t P, const float *rows, const uint32_t *count, uint32_t max_rows, float *dcolor_view,
                            const w3d_raw_grads *sums, float *norm_sum, w3d_stream_t stream);
/* The optimizer step straight from the gathered rows, without dense per-view arrays.  rows_all: (n_views, cap_rows, 16) as
 * all-gathered, counts: (n_views) device.  w3d_index_gradient_rows zeroes viewmask (P u32) and, for every row r <
 * min(counts[v], cap_rows) of view v with Gaussian index g, sets bit v of viewmask[g] and slots[v*P + g] = r (slots: n_views*P
 * u32, only the marked entries are written or ever read; at most 32 views).  w3d_rows_norm_sum writes norm_sum[g] = sum in view
 * order of the rows' norms (0 where no view holds a row).  w3d_rows_adam walks every Gaussian's set bits in view order, rebuilds
 * the SH gradient (as w3d_sh_adam_lowrank) and the sum of the geometry gradients, and applies torch.optim.Adam's update to all
 * six parameter blocks and their moments in place — one pass; `adam` as in w3d_backward_raw_adam (per-block step sizes, skips
 * and bias corrections).  Same additions in the same order on every rank: replicas stay bit-identical. */
int w3d_index_gradient_rows(int32_t P, int32_t n_views, const float *rows_all, const uint32_t *counts, uint32_t cap_rows,
                            uint32_t *viewmask, uint32_t *slots, w3d_stream_t stream);
int w3d_rows_norm_sum(int32_t P, int32_t n_views, const float *rows_all, uint32_t cap_rows, const uint32_t *viewmask,
                      const uint32_t *slots, float *norm_sum, w3d_stream_t stream);
/* ... and the same sum ADDED to accum[g] (P,) in one addition per Gaussian: xyz_gradient_accum += sum over the views of the rows'
 * ||dL/dmean2D|| (reference scene/gaussian_model.py:462 once per view), without materialising the sum. */
int w3d_rows_norm_accumulate(int32_t P, int32_t n_views, const float *rows_all, uint32_t cap_rows, const uint32_t *viewmask,
                             const uint32_t *slots, float *accum, w3d_stream_t stream);
int w3d_rows_adam(int32_t P, int32_t n_views, int32_t sh_degree, const float *campos_all, const float *rows_all, uint32_t cap_rows,
                  const uint32_t *viewmask, const uint32_t *slots, const w3d_raw_blocks *params, const w3d_adam_fused *adam,
                  w3d_stream_t stream);

/* mean squared distance to the 3 nearest other points; points (N,3) -> out (N,) */
/* View-parallel bookkeeping between two densifications (per rank; reduced over the ranks when a densification or a checkpoint
 * reads them): vis_count[g] += radii[g] > 0, radii_max[g] = max(radii_max[g], radii[g]) — the per-view halves of reference
 * train_vanilla_3dgs.py:102 (max_radii2D) and scene/gaussian_model.py:463 (denom), all int32 (P,). */
int w3d_track_visibility(int32_t P, const int32_t *radii, int32_t *vis_count, int32_t *radii_max, w3d_stream_t stream);

int w3d_knn_dist2(int32_t N, const float *points, float *out, w3d_stream_t stream);
/* The same result (bit for bit) through a uniform grid built on the device: O(N) instead of O(N^2) for the million-point
 * initialisations of the 2 M-Gaussian configurations.  scratch: w3d_knn_sizes(N) bytes. */
int w3d_knn_sizes(int32_t N, uint64_t *scratch_bytes);
int w3d_knn_dist2_grid(int32_t N, const float *points, float *out, void *scratch, w3d_stream_t stream);

/* ---- next-row N1: fused photometric loss 0.8*L1 + 0.2*(1-SSIM) (lambda_dssim = 0.2), value and
 * gradient in one call.  image, gt, dL_dimage: (C,H,W); loss_out: device scalar (overwritten). */
int w3d_l1_ssim_sizes(int32_t C, int32_t H, int32_t W, uint64_t *scratch_bytes);
int w3d_l1_ssim_fwd_bwd(int32_t C, int32_t H, int32_t W, const float *image, const float *gt,
                        float lambda_dssim, float *loss_out, float *dL_dimage, void *scratch,
                        w3d_stream_t stream);

/* The same two passes as separate calls, for a caller whose loss lines are the reference's own
 * (train_vanilla_3dgs.py:77-79:  Ll1 = l1_loss(image, gt);  loss = (1-l)*Ll1 + l*(1 - ssim(image, gt))  with
 * utils/loss_utils.py:17-18,39-63): w3d_l1_ssim_values runs pass A and writes *l1_out = mean|image - gt| and
 * *ssim_out = ssim(image, gt) (device scalars); w3d_l1_ssim_grad runs pass B on the maps pass A left in `scratch` (same
 * buffer, untouched in between) with the two upstream gradients read from the DEVICE, w_l1 = dL/dLl1 and
 * w_ssim = dL/dssim (NULL = 0):  dL_dimage = w_l1 * dLl1/dimage + w_ssim * dssim/dimage  — no host sync in between. */
int w3d_l1_ssim_values(int32_t C, int32_t H, int32_t W, const float *image, const float *gt, float *l1_out,
                       float *ssim_out, void *scratch, w3d_stream_t stream);
int w3d_l1_ssim_grad(int32_t C, int32_t H, int32_t W, const float *image, const float *gt, const float *w_l1,
                     const float *w_ssim, float *dL_dimage, void *scratch, w3d_stream_t stream);

/* add_densification_stats of scene/gaussian_model.py:461-463 in one pass: for every row with update_filter != 0
 * (a (P,) array of bytes, torch.bool):  xyz_gradient_accum += ||dL_dmeans2D[:, :2]||,  denom += 1. */
int w3d_add_densification_stats(int32_t P, const float *dL_dmeans2D, const uint8_t *update_filter,
                                float *xyz_gradient_accum, float *denom, w3d_stream_t stream);

/* ---- next-row N2: one Adam step over n contiguous fp32 elements (torch.optim.Adam semantics,
 * no weight decay / amsgrad): m,v updated in place, param -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
 * The four arrays must share their 16-B misalignment.  zero_grad != 0 also clears grad. */
int w3d_adam_step(uint64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, float bias_correction1, float bias_correction2,
                  int32_t zero_grad, w3d_stream_t stream);

/* Next-row N3: one-pass row compaction of the flat parameter buffer and both Adam moments for densify / prune
 * <- scene/gaussian_model.py:332-397 (_prune_optimizer, cat_tensors_to_optimizer, densification_postfix,
 *    prune_points), called three to four times per densify_and_prune :441-455.
 * The flat buffers are block-wise: block b holds P rows of block_dims_host[b] floats; the blocks lie back to back except that
 * every block STARTS on a multiple of 4 floats (16 B; at most 3 floats of padding in front of a block, never read or written
 * here) — so that the streaming kernels find every block 16-B aligned whatever P is (ABI 201; ABI 200 packed them tightly).
 * Output row r takes source row src_rows[r] (< P_old).  Rows [0, n_keep) keep their Adam moments; rows
 * [n_keep, P_new) are new (clones, then split children from n_child0 on) and get zero moments; split children
 * take their xyz_block / scaling_block rows from child_xyz / child_scaling ((P_new - n_child0, 3) each).
 * exp_avg_* may all be NULL (no optimizer state).  block_dims_host is a HOST array. */
int w3d_densify_compact(int32_t n_blocks, const int32_t *block_dims_host, int32_t xyz_block, int32_t scaling_block,
                        uint64_t P_old, uint64_t P_new, uint64_t n_keep, uint64_t n_child0, const int32_t *src_rows,
                        const float *param_old, const float *exp_avg_old, const float *exp_avg_sq_old,
                        float *param_new, float *exp_avg_new, float *exp_avg_sq_new, const float *child_xyz,
                        const float *child_scaling, w3d_stream_t stream);

/* ---- next-row N4: the mask work around the FlashSplat render, on the device.
 * w3d_mask_binarize <- utils/wheatgs_utils.py:26-37 binarize_mask(PILtoTorch(png)) as used at run_3d_seg.py:88-89:
 *   pixels (H,W,C) uint8 as decoded (C = 1 or 3), out (H,W) fp32: 1 where any channel is non-zero, else 0.
 * w3d_mask_iou <- run_3d_seg.py:127-163 find_match's scoring: pred = alpha > thresh; out (2K+5) uint32:
 *   [2k] = |mask_k AND pred|, [2k+1] = |mask_k OR pred|  (utils/wheatgs_utils.py:94-103 calculate_seg_iou),
 *   [2K..2K+3] = bounding box of pred {x_min, y_min, x_max, y_max} (get_bbox_from_mask :45-53; x_min = 0xFFFFFFFF when
 *   pred is empty), [2K+4] = |pred|.  masks (K,H,W) uint8, non-zero = inside.  K may be 0. */
int w3d_mask_binarize(int32_t H, int32_t W, int32_t C, const uint8_t *pixels, float *out, w3d_stream_t stream);
int w3d_mask_iou(int32_t H, int32_t W, int32_t K, const float *alpha, float thresh, const uint8_t *masks, uint32_t *out,
                 w3d_stream_t stream);

/* Diagnostics for bench.py's roofline leg: time the launches whose stage name contains
 * `kernel_substr` ("*" = all, NULL/"" = off) with HIP events on their launch stream;
 * w3d_profile_collect waits for them and writes "name count total_ms" lines into out. */
int w3d_profile_enable(const char *kernel_substr);
int w3d_profile_collect(char *out, uint64_t cap);

/* Debug/inspection: copies of internal per-tile ranges (T,2) uint32 laid out as [start,end). */
int w3d_debug_tile_ranges(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *ranges_out, w3d_stream_t stream);
/* Debug/inspection: the per-Gaussian 16-B rect / tile-mask records of the forward (P,4) uint32 {rect lo, rect hi, mask lo, mask hi}
 * (lo = minx | miny << 16, hi = maxx | maxy << 16; written only when view->tile_cull was set).  UNITS: cells of the LIST grid the
 * forward ran on — 16x16 tiles with list_share = 0, 32x16 / 32x32 cells with list_share = 1 / 2 (a cell's mask bit is the OR of
 * its tiles' bits).  Culled Gaussians hold an all-zero record. */
int w3d_debug_tile_rects(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *rects_out, w3d_stream_t stream);
/* Debug/inspection: the block -> (tile, part) schedule the LAST blend kernel on this state ran (forward: built from the view's
 * tile_walk_hint; backward: from the forward's walk lengths).  *cap_out = entries per XCD range; order_out (HOST memory, 8 * cap
 * u32, may be NULL to query cap only; synchronous copy): entry = tile | part << 29 (part 0 the whole tile, 1 / 2 its upper /
 * lower pair of 8x8 quadrants, 3..6 one quadrant), 0xFFFFFFFF = no work.  Meaningless when the kernel ran without a schedule
 * (a forward without hint). */
int w3d_debug_tile_schedule(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *order_out, uint32_t *cap_out);
/* Debug/inspection: the 64-B per-Gaussian records the blend kernels gather, (P,16) f32: {x, y, rect lo bits, rect hi bits |
 * conic.x, conic.y, conic.z, opacity | r, g, b, depth | the conic scaled into the log2 domain, opacity}.  Records of culled
 * Gaussians (radii == 0) are NOT written: they hold whatever the buffer held before — gate on radii. */
int w3d_debug_gaussian_records(int32_t H, int32_t W, int32_t P, const void *state, float *records_out, w3d_stream_t stream);
/* Debug/inspection: the depth sort's bucket grid of the forward that last used this SCRATCH buffer (the caller must have kept it
 * alive and the stream must be idle: synchronous copies).  bstart_out: HOST memory, 1025 u32 — bucket b holds the positions
 * [bstart[b], bstart[b + 1]) of the depth order, bstart[1024] = visible Gaussians; brange_out: HOST memory, 2 x 1024 u32 —
 * {lo, width} of the key offsets (key - smallest visible key) bucket b covers, {0, 0} past the grid's last bucket. */
int w3d_debug_depth_buckets(int32_t H, int32_t W, int32_t P, const void *scratch, uint32_t *bstart_out, uint32_t *brange_out);
/* Debug/inspection of the per-pixel state kept for backward: final_T (H,W) f32, n_contrib (H,W) u32. */
int w3d_debug_pixel_state(int32_t H, int32_t W, int32_t P, const void *state, float *final_T_out,
                          uint32_t *n_contrib_out, w3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* W3D_H_ */
