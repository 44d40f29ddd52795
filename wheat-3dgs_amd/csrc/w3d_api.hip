// w3d_api.hip — the extern "C" boundary declared in include/w3d.h: argument checks, buffer
// layout, kernel sequencing.  No device allocation, no global mutable state except the
// thread-local last-error text.
#include <stdarg.h>
#include <stdio.h>

#include "w3d_common.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

// ---- per-kernel event timing: opt-in diagnostic state, the only non-error global in the library.  Off unless
// w3d_profile_enable() was called; the record list is guarded by a mutex (launches from several host threads), the
// "scope is open" flag is per thread.
namespace {
struct ProfRec { std::string name; hipEvent_t a, b; };
std::mutex g_prof_mutex;
std::vector<ProfRec> g_prof;
std::string g_prof_filter;
bool g_prof_on = false;
thread_local hipEvent_t g_prof_open_end = nullptr;
}  // namespace

void w3d_prof_begin(const char *name, hipStream_t stream) {
    g_prof_open_end = nullptr;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof_on) return;
    if (g_prof_filter != "*" && std::string(name).find(g_prof_filter) == std::string::npos) return;
    ProfRec r;
    r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
    g_prof_open_end = r.b;
}
void w3d_prof_end(hipStream_t stream) {
    if (g_prof_open_end) (void)hipEventRecord(g_prof_open_end, stream);
    g_prof_open_end = nullptr;
}

void w3d_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int w3d_debug_tile_ranges_impl(const W3DLayout &L, const char *state, uint32_t *ranges_out, hipStream_t stream);
int w3d_debug_pixel_state_impl(const W3DLayout &L, const char *state, float *final_T_out, uint32_t *n_contrib_out,
                               hipStream_t stream);

static int check_view(const w3d_view *v) {
    if (!v) { w3d_set_error("view is NULL"); return W3D_ERR_INVALID; }
    if (v->struct_size != sizeof(w3d_view)) {
        w3d_set_error("w3d_view size %u, this library expects %zu: the caller was built against another include/w3d.h "
                      "(ABI %d)", v->struct_size, sizeof(w3d_view), W3D_ABI_VERSION);
        return W3D_ERR_INVALID;
    }
    if (v->image_height <= 0 || v->image_width <= 0) { w3d_set_error("image size must be positive"); return W3D_ERR_INVALID; }
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos) { w3d_set_error("view holds a NULL device pointer"); return W3D_ERR_INVALID; }
    if (v->sh_degree < 0 || v->sh_degree > 3) { w3d_set_error("sh_degree %d unsupported (0..3)", v->sh_degree); return W3D_ERR_UNSUPPORTED; }
    if (v->list_share < 0 || v->list_share > 2) { w3d_set_error("list_share %d unsupported (0..2)", v->list_share); return W3D_ERR_UNSUPPORTED; }
    return W3D_OK;
}

static int check_variants(const w3d_view *v, const float *shs, const float *colors_precomp, const float *scales,
                          const float *rotations, const float *cov3D_precomp) {
    if ((shs == nullptr) == (colors_precomp == nullptr)) {
        w3d_set_error("provide exactly one of shs / colors_precomp");
        return W3D_ERR_INVALID;
    }
    const bool sr = scales && rotations;
    if ((scales == nullptr) != (rotations == nullptr) || sr == (cov3D_precomp != nullptr)) {
        w3d_set_error("provide exactly one of (scales, rotations) / cov3D_precomp");
        return W3D_ERR_INVALID;
    }
    if (shs && v->sh_coeffs < (v->sh_degree + 1) * (v->sh_degree + 1)) {
        w3d_set_error("shs holds %d coefficients, degree %d needs %d", v->sh_coeffs, v->sh_degree, (v->sh_degree + 1) * (v->sh_degree + 1));
        return W3D_ERR_INVALID;
    }
    return W3D_OK;
}

// depth sort, tile counting of the first (or only) layer, optional host read-back of the counters
static int finish_stage1(const W3DLayout &L, const w3d_view &view, char *st, char *sc, uint32_t *counts_host, hipStream_t stream) {
    int rc = w3d_launch_depth_sort(L, view, st, sc, stream);
    if (rc) return rc;
    rc = w3d_launch_tile_count(L, view, st, sc, stream);
    if (rc) return rc;
    if (counts_host) {
        W3D_HIP_CHECK(hipMemcpyAsync(counts_host, st + L.o_counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        W3D_HIP_CHECK(hipStreamSynchronize(stream));
    }
    return W3D_OK;
}

extern "C" {

int w3d_version(void) { return W3D_ABI_VERSION; }

int w3d_profile_enable(const char *kernel_substr) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = kernel_substr && kernel_substr[0];
    g_prof_filter = g_prof_on ? kernel_substr : "";
    return W3D_OK;
}

// Waits for the recorded events, writes "name count total_ms\n" lines into out (NUL-terminated),
// and clears the records.
int w3d_profile_collect(char *out, uint64_t cap) {
    std::map<std::string, std::pair<int, double>> agg;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto &r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            agg[r.name].first += 1;
            agg[r.name].second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    std::string txt;
    for (auto &kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        txt += line;
    }
    if (out && cap > 0) {
        snprintf(out, cap, "%s", txt.c_str());
    }
    return W3D_OK;
}

const char *w3d_last_error(void) { return g_err; }

int w3d_forward_sizes(int32_t P, int32_t H, int32_t W, uint64_t *state_bytes, uint64_t *scratch_bytes) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc) { w3d_set_error("bad sizes P=%d H=%d W=%d", P, H, W); return rc; }
    if (state_bytes) *state_bytes = L.state_bytes;
    if (scratch_bytes) *scratch_bytes = L.scratch_bytes;
    return W3D_OK;
}

int w3d_forward_stage1(const w3d_view *view, int32_t P, const float *means3D, const float *shs,
                       const float *colors_precomp, const float *opacities, const float *scales,
                       const float *rotations, const float *cov3D_precomp, int32_t *radii, void *state,
                       void *scratch, uint32_t *counts_host, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (!state || !scratch) { w3d_set_error("state/scratch is NULL"); return W3D_ERR_INVALID; }
    if (P > 0) {
        if (!means3D || !opacities || !radii) { w3d_set_error("means3D/opacities/radii is NULL"); return W3D_ERR_INVALID; }
        rc = check_variants(view, shs, colors_precomp, scales, rotations, cov3D_precomp);
        if (rc) return rc;
    }
    char *st = static_cast<char *>(state), *sc = static_cast<char *>(scratch);
    rc = w3d_launch_preprocess(L, *view, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii,
                               st, sc, nullptr, nullptr, stream);
    if (rc) return rc;
    rc = finish_stage1(L, *view, st, sc, counts_host, stream);
    return rc;
}

int w3d_forward_stage2(const w3d_view *view, int32_t P, void *state, void *scratch, uint32_t *point_list,
                       uint64_t list_capacity, float *out_color, float *out_depth, float *out_alpha,
                       const float *gt_mask, int32_t num_obj, float *used_count, int32_t *contrib_num,
                       float *proj_xy, float *gs_depth, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (!state || !scratch || !out_color) { w3d_set_error("NULL buffer"); return W3D_ERR_INVALID; }
    const bool flash_out = gt_mask || used_count || contrib_num;
    if ((out_depth == nullptr) != (out_alpha == nullptr) || (flash_out && !out_depth)) {
        w3d_set_error("out_depth / out_alpha: both or neither (neither only without the FlashSplat outputs)");
        return W3D_ERR_INVALID;
    }
    if (list_capacity > 0 && !point_list) { w3d_set_error("point_list is NULL"); return W3D_ERR_INVALID; }
    if (gt_mask && used_count && num_obj < 0) { w3d_set_error("num_obj must be >= 0"); return W3D_ERR_INVALID; }
    char *st = static_cast<char *>(state), *sc = static_cast<char *>(scratch);
    rc = w3d_launch_fill_lists(L, *view, st, sc, point_list, list_capacity, stream);
    if (rc) return rc;
    rc = w3d_launch_render(L, *view, st, point_list, list_capacity, out_color, out_depth, out_alpha, gt_mask, num_obj,
                           used_count, contrib_num, stream);
    if (rc) return rc;
    if (proj_xy || gs_depth) {
        // radii are not kept in the state; visibility is re-derived from the tile rectangle
        rc = w3d_launch_flash_extras(L, *view, nullptr, st, proj_xy, gs_depth, stream);
        if (rc) return rc;
    }
    return W3D_OK;
}

int w3d_flash_reblend(const w3d_view *view, int32_t P, void *state, const uint32_t *point_list, uint64_t list_capacity,
                      float *out_color, float *out_depth, float *out_alpha, const float *gt_mask, int32_t num_obj,
                      float *used_count, int32_t *contrib_num, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (!state || !out_color || !out_depth || !out_alpha) { w3d_set_error("NULL buffer"); return W3D_ERR_INVALID; }
    if (list_capacity > 0 && !point_list) { w3d_set_error("point_list is NULL"); return W3D_ERR_INVALID; }
    if (gt_mask && used_count && num_obj < 0) { w3d_set_error("num_obj must be >= 0"); return W3D_ERR_INVALID; }
    return w3d_launch_render(L, *view, static_cast<char *>(state), point_list, list_capacity, out_color, out_depth, out_alpha,
                             gt_mask, num_obj, used_count, contrib_num, stream);
}

int w3d_backward_sizes(int32_t P, uint64_t *scratch_bytes) {
    if (P < 0) { w3d_set_error("P < 0"); return W3D_ERR_INVALID; }
    if (scratch_bytes) *scratch_bytes = w3d_align_up((uint64_t)(P > 0 ? P : 1) * W3D_G2D_STRIDE * sizeof(float));
    return W3D_OK;
}

int w3d_backward_det_sizes(int32_t P, uint64_t list_capacity, uint64_t *scratch_bytes) {
    if (P < 0) { w3d_set_error("P < 0"); return W3D_ERR_INVALID; }
    if (scratch_bytes)
        *scratch_bytes = w3d_align_up((uint64_t)(P > 0 ? P : 1) * W3D_G2D_STRIDE * sizeof(float)) +
                         w3d_align_up((list_capacity > 0 ? list_capacity : 1) * W3D_G2D_STRIDE * sizeof(float));
    return W3D_OK;
}

int w3d_backward(const w3d_view *view, int32_t P, const float *means3D, const float *shs,
                 const float *colors_precomp, const float *opacities, const float *scales,
                 const float *rotations, const float *cov3D_precomp, const void *state,
                 const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth,
                 const float *dL_dalpha, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors,
                 float *dL_dshs, float *dL_dopacity, float *dL_dscales, float *dL_drots, float *dL_dcov3D,
                 void *scratch, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    (void)opacities;
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (P == 0) return W3D_OK;
    if (!state || !scratch || !dL_dcolor || !means3D || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity) {
        w3d_set_error("NULL buffer");
        return W3D_ERR_INVALID;
    }
    rc = check_variants(view, shs, colors_precomp, scales, rotations, cov3D_precomp);
    if (rc) return rc;
    if (shs && !dL_dshs) { w3d_set_error("dL_dshs is NULL"); return W3D_ERR_INVALID; }
    if (scales && (!dL_dscales || !dL_drots)) { w3d_set_error("dL_dscales/dL_drots is NULL"); return W3D_ERR_INVALID; }
    if (cov3D_precomp && !dL_dcov3D) { w3d_set_error("dL_dcov3D is NULL"); return W3D_ERR_INVALID; }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
    if (rc) return rc;
    return w3d_launch_preprocess_backward(L, *view, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, st,
                                          grad2d, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dshs, dL_dopacity, dL_dscales,
                                          dL_drots, dL_dcov3D, nullptr, stream);
}

// ---- raw-parameter path: activations + dc/rest split fused into the kernels
int w3d_forward_stage1_raw(const w3d_view *view, int32_t P, const w3d_raw_params *prm, int32_t *radii, void *state,
                           void *scratch, uint32_t *counts_host, w3d_stream_t stream_) {
    return w3d_forward_stage1_raw_subset(view, P, prm, nullptr, radii, state, scratch, counts_host, stream_);
}

int w3d_forward_stage1_raw_subset(const w3d_view *view, int32_t P, const w3d_raw_params *prm, const uint8_t *used_mask,
                                  int32_t *radii, void *state, void *scratch, uint32_t *counts_host, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (!state || !scratch) { w3d_set_error("state/scratch is NULL"); return W3D_ERR_INVALID; }
    if (P > 0) {
        if (!prm || !prm->xyz || !prm->f_dc || !prm->f_rest || !prm->opacity || !prm->scaling || !prm->rotation || !radii) {
            w3d_set_error("raw parameter block holds a NULL pointer");
            return W3D_ERR_INVALID;
        }
        if (view->sh_coeffs < 2 || view->sh_coeffs > 16 || view->sh_coeffs < (view->sh_degree + 1) * (view->sh_degree + 1)) {
            w3d_set_error("raw path needs 2..16 SH coefficients covering the active degree");
            return W3D_ERR_INVALID;
        }
    }
    char *st = static_cast<char *>(state), *sc = static_cast<char *>(scratch);
    rc = w3d_launch_preprocess(L, *view, prm ? prm->xyz : nullptr, prm ? prm->f_dc : nullptr, nullptr,
                               prm ? prm->opacity : nullptr, prm ? prm->scaling : nullptr, prm ? prm->rotation : nullptr,
                               nullptr, radii, st, sc, prm ? prm->f_rest : nullptr, used_mask, stream);
    if (rc) return rc;
    return finish_stage1(L, *view, st, sc, counts_host, stream);
}

int w3d_backward_raw(const w3d_view *view, int32_t P, const w3d_raw_params *prm, const void *state,
                     const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                     const w3d_raw_grads *grads, const w3d_densify_stats *stats, void *scratch, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (P == 0) return W3D_OK;
    if (!state || !scratch || !dL_dcolor || !prm || !grads || !grads->xyz || !grads->f_dc || !grads->f_rest ||
        !grads->opacity || !grads->scaling || !grads->rotation) {
        w3d_set_error("NULL buffer");
        return W3D_ERR_INVALID;
    }
    if (stats && stats->xyz_gradient_accum && (!stats->denom || !stats->max_radii2D || !stats->radii)) {
        w3d_set_error("fused statistics need accum, denom, max_radii2D and radii together");
        return W3D_ERR_INVALID;
    }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
    if (rc) return rc;
    W3DRawBwdArgs ra = {};
    ra.f_rest = prm->f_rest; ra.opacity_logit = prm->opacity; ra.dL_df_rest = grads->f_rest;
    if (stats) {
        ra.gnorm_out = stats->grad2d_norm; ra.radii = stats->radii; ra.accum = stats->xyz_gradient_accum;
        ra.denom = stats->denom; ra.max_radii = stats->max_radii2D;
    }
    return w3d_launch_preprocess_backward(L, *view, prm->xyz, prm->f_dc, nullptr, prm->scaling, prm->rotation, nullptr, st,
                                          grad2d, grads->xyz, stats ? stats->dL_dmeans2D : nullptr, nullptr, grads->f_dc,
                                          grads->opacity, grads->scaling, grads->rotation, nullptr, &ra, stream);
}

int w3d_backward_raw_adam(const w3d_view *view, int32_t P, const w3d_raw_blocks *prm, const void *state,
                          const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                          const w3d_adam_fused *adam, const w3d_densify_stats *stats, void *scratch, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (P == 0) return W3D_OK;
    if (!state || !scratch || !dL_dcolor || !prm || !adam) { w3d_set_error("NULL buffer"); return W3D_ERR_INVALID; }
    for (int i = 0; i < 6; i++) {
        if (!adam->skip[i] && (!(adam->bias_correction1[i] > 0.f) || !(adam->bias_correction2[i] > 0.f))) {
            w3d_set_error("fused Adam: bias corrections must be positive (step >= 1)");
            return W3D_ERR_INVALID;
        }
    }
    if (stats && stats->xyz_gradient_accum && (!stats->denom || !stats->max_radii2D || !stats->radii)) {
        w3d_set_error("fused statistics need accum, denom, max_radii2D and radii together");
        return W3D_ERR_INVALID;
    }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
    if (rc) return rc;
    W3DRawBwdArgs ra = {};
    ra.f_rest = prm->f_rest; ra.opacity_logit = prm->opacity;
    ra.adam = adam; ra.params_rw = prm;
    if (stats) {
        // (the statistics, like the parameters, are only updated when the forward's lists fitted their buffer)
        ra.gnorm_out = stats->grad2d_norm; ra.radii = stats->radii; ra.accum = stats->xyz_gradient_accum;
        ra.denom = stats->denom; ra.max_radii = stats->max_radii2D;
    }
    return w3d_launch_preprocess_backward(L, *view, prm->xyz, prm->f_dc, nullptr, prm->scaling, prm->rotation, nullptr, st,
                                          grad2d, nullptr, stats ? stats->dL_dmeans2D : nullptr, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, &ra, stream);
}

int w3d_backward_raw_lowrank(const w3d_view *view, int32_t P, const w3d_raw_params *prm, const void *state,
                             const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                             const w3d_raw_grads *grads, float *dcolor_out, const w3d_densify_stats *stats, void *scratch,
                             w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (P == 0) return W3D_OK;
    // dL_dcolor == NULL: the blend backward of this view already ran into `scratch` (w3d_backward_blend_dcolor); then
    // dcolor_out may be NULL as well
    if (!state || !scratch || !prm || !grads || !grads->xyz || !grads->opacity || !grads->scaling || !grads->rotation ||
        (dL_dcolor && !dcolor_out)) {
        w3d_set_error("NULL buffer");
        return W3D_ERR_INVALID;
    }
    if (stats && stats->xyz_gradient_accum) {
        w3d_set_error("view-parallel statistics are reduced across ranks by the caller");
        return W3D_ERR_INVALID;
    }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    if (dL_dcolor) {
        rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
        if (rc) return rc;
    }
    W3DRawBwdArgs ra = {};
    ra.f_rest = prm->f_rest; ra.opacity_logit = prm->opacity; ra.dcolor_out = dcolor_out; ra.lowrank = 1;
    if (stats) { ra.gnorm_out = stats->grad2d_norm; ra.radii = stats->radii; }
    return w3d_launch_preprocess_backward(L, *view, prm->xyz, prm->f_dc, nullptr, prm->scaling, prm->rotation, nullptr, st,
                                          grad2d, grads->xyz, stats ? stats->dL_dmeans2D : nullptr, nullptr, nullptr,
                                          grads->opacity, grads->scaling, grads->rotation, nullptr, &ra, stream);
}

int w3d_backward_raw_rows(const w3d_view *view, int32_t P, const w3d_raw_params *prm, const void *state,
                          const uint32_t *point_list, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                          float norm_scale, float *rows_out, uint32_t capacity_rows, uint32_t *count, void *scratch,
                          w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (!count) { w3d_set_error("NULL buffer"); return W3D_ERR_INVALID; }
    if (P == 0) { W3D_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), stream)); return W3D_OK; }
    if (!state || !scratch || !prm || !dL_dcolor || !rows_out || (reinterpret_cast<uintptr_t>(rows_out) & 15)) {
        w3d_set_error("NULL or misaligned buffer");
        return W3D_ERR_INVALID;
    }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
    if (rc) return rc;
    W3DRawBwdArgs ra = {};
    ra.f_rest = prm->f_rest; ra.opacity_logit = prm->opacity; ra.lowrank = 2;
    ra.rows_out = rows_out; ra.rows_cap = capacity_rows; ra.rows_count = count; ra.norm_scale = norm_scale;
    return w3d_launch_preprocess_backward(L, *view, prm->xyz, prm->f_dc, nullptr, prm->scaling, prm->rotation, nullptr, st, grad2d,
                                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ra, stream);
}

int w3d_backward_blend_dcolor(const w3d_view *view, int32_t P, const void *state, const uint32_t *point_list,
                              const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha, float *dcolor_out,
                              void *scratch, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int rc = check_view(view);
    if (rc) return rc;
    W3DLayout L;
    rc = w3d_make_layout(P, view->image_height, view->image_width, &L);
    if (rc) { w3d_set_error("bad sizes"); return rc; }
    w3d_set_list_share(&L, view);
    if (P == 0) return W3D_OK;
    if (!state || !scratch || !dL_dcolor || !dcolor_out) { w3d_set_error("NULL buffer"); return W3D_ERR_INVALID; }
    const char *st = static_cast<const char *>(state);
    float *grad2d = static_cast<float *>(scratch);
    rc = w3d_launch_render_backward(L, *view, st, point_list, dL_dcolor, dL_ddepth, dL_dalpha, grad2d, stream);
    if (rc) return rc;
    return w3d_launch_dcolor_extract(L, st, grad2d, dcolor_out, stream);
}

int w3d_sh_adam_lowrank(int32_t P, int32_t n_views, int32_t sh_degree, const float *campos_all, const float *xyz,
                        const float *dcolor_all, float *f_dc, float *f_rest, float *m_dc, float *v_dc, float *m_rest,
                        float *v_rest, float lr_dc, float lr_rest, int32_t skip_dc, int32_t skip_rest, float beta1, float beta2,
                        float eps, float bc1, float bc2, w3d_stream_t stream_) {
    if (P < 0 || n_views < 0 || sh_degree < 0 || sh_degree > 3) { w3d_set_error("sh_adam_lowrank: bad sizes"); return W3D_ERR_INVALID; }
    if (P == 0 || n_views == 0) return W3D_OK;
    if (!campos_all || !xyz || !dcolor_all || !f_dc || !f_rest || !m_dc || !v_dc || !m_rest || !v_rest) {
        w3d_set_error("sh_adam_lowrank: NULL buffer");
        return W3D_ERR_INVALID;
    }
    if (!(bc1 > 0.f) || !(bc2 > 0.f)) { w3d_set_error("sh_adam_lowrank: bias corrections must be positive"); return W3D_ERR_INVALID; }
    return w3d_launch_sh_adam_lowrank(P, n_views, sh_degree, campos_all, xyz, dcolor_all, f_dc, f_rest, m_dc, v_dc, m_rest, v_rest,
                                      lr_dc, lr_rest, skip_dc, skip_rest, beta1, beta2, eps, bc1, bc2,
                                      reinterpret_cast<hipStream_t>(stream_));
}

int w3d_knn_dist2(int32_t N, const float *points, float *out, w3d_stream_t stream_) {
    if (N < 0 || (N > 0 && (!points || !out))) { w3d_set_error("bad knn arguments"); return W3D_ERR_INVALID; }
    return w3d_launch_knn(N, points, out, reinterpret_cast<hipStream_t>(stream_));
}

int w3d_knn_sizes(int32_t N, uint64_t *scratch_bytes) {
    if (N < 0) { w3d_set_error("bad knn arguments"); return W3D_ERR_INVALID; }
    if (scratch_bytes) *scratch_bytes = w3d_knn_scratch_bytes(N);
    return W3D_OK;
}

int w3d_knn_dist2_grid(int32_t N, const float *points, float *out, void *scratch, w3d_stream_t stream_) {
    if (N < 0 || (N > 0 && (!points || !out || !scratch))) { w3d_set_error("bad knn arguments"); return W3D_ERR_INVALID; }
    return w3d_launch_knn_grid(N, points, out, static_cast<char *>(scratch), reinterpret_cast<hipStream_t>(stream_));
}

int w3d_mask_binarize(int32_t H, int32_t W, int32_t C, const uint8_t *pixels, float *out, w3d_stream_t stream_) {
    if (H <= 0 || W <= 0 || (C != 1 && C != 3) || !pixels || !out) { w3d_set_error("mask_binarize: bad arguments"); return W3D_ERR_INVALID; }
    return w3d_launch_mask_binarize(H, W, C, pixels, out, reinterpret_cast<hipStream_t>(stream_));
}

int w3d_mask_iou(int32_t H, int32_t W, int32_t K, const float *alpha, float thresh, const uint8_t *masks, uint32_t *out,
                 w3d_stream_t stream_) {
    if (H <= 0 || W <= 0 || K < 0 || K > 2000 || !alpha || !out || (K > 0 && !masks)) { w3d_set_error("mask_iou: bad arguments"); return W3D_ERR_INVALID; }
    return w3d_launch_mask_iou(H, W, K, alpha, thresh, masks, out, reinterpret_cast<hipStream_t>(stream_));
}

int w3d_debug_tile_ranges(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *ranges_out, w3d_stream_t stream_) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !state || !ranges_out) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    return w3d_debug_tile_ranges_impl(L, static_cast<const char *>(state), ranges_out, reinterpret_cast<hipStream_t>(stream_));
}

int w3d_debug_tile_rects(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *rects_out, w3d_stream_t stream_) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !state || (P > 0 && !rects_out)) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    if (P > 0)
        W3D_HIP_CHECK(hipMemcpyAsync(rects_out, static_cast<const char *>(state) + L.o_tile_mask, (size_t)P * 16, hipMemcpyDeviceToDevice,
                                     reinterpret_cast<hipStream_t>(stream_)));
    return W3D_OK;
}

int w3d_debug_tile_schedule(int32_t H, int32_t W, int32_t P, const void *state, uint32_t *order_out, uint32_t *cap_out) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !cap_out) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    *cap_out = L.order_cap;
    if (order_out) {
        if (!state) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
        W3D_HIP_CHECK(hipMemcpy(order_out, static_cast<const char *>(state) + L.o_tile_order, (size_t)8 * L.order_cap * 4, hipMemcpyDeviceToHost));
    }
    return W3D_OK;
}

int w3d_debug_gaussian_records(int32_t H, int32_t W, int32_t P, const void *state, float *records_out, w3d_stream_t stream_) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !state || (P > 0 && !records_out)) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    if (P > 0)
        W3D_HIP_CHECK(hipMemcpyAsync(records_out, static_cast<const char *>(state) + L.o_grec, (size_t)P * 64, hipMemcpyDeviceToDevice,
                                     reinterpret_cast<hipStream_t>(stream_)));
    return W3D_OK;
}

int w3d_debug_depth_buckets(int32_t H, int32_t W, int32_t P, const void *scratch, uint32_t *bstart_out, uint32_t *brange_out) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !scratch || !bstart_out || !brange_out) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    const char *s = static_cast<const char *>(scratch);
    W3D_HIP_CHECK(hipMemcpy(bstart_out, s + L.s_bstart, (size_t)(W3D_DB_BINS + 1) * 4, hipMemcpyDeviceToHost));
    W3D_HIP_CHECK(hipMemcpy(brange_out, s + L.s_grid + 257 * 16, (size_t)W3D_DB_BINS * 8, hipMemcpyDeviceToHost));
    return W3D_OK;
}

int w3d_debug_pixel_state(int32_t H, int32_t W, int32_t P, const void *state, float *final_T_out,
                          uint32_t *n_contrib_out, w3d_stream_t stream_) {
    W3DLayout L;
    int rc = w3d_make_layout(P, H, W, &L);
    if (rc || !state) { w3d_set_error("bad arguments"); return W3D_ERR_INVALID; }
    return w3d_debug_pixel_state_impl(L, static_cast<const char *>(state), final_T_out, n_contrib_out,
                                      reinterpret_cast<hipStream_t>(stream_));
}

}  // extern "C"
