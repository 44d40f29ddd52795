// w3d_exchange.hip — the sparse ("rows") form of the view-parallel gradient exchange (the reference is
// single-GPU, reference train_vanilla_3dgs.py:65 picks one camera per iteration — here every rank renders its own).
//
// One view gives a gradient to the Gaussians its pixels actually blended: on the SURVEY §8d scene 0.1 M of 2 M, on a trained
// scene a few hundred thousand.  Every other row of the (P,3) colour gradient and of the 11 geometry gradients is exactly
// zero, and adding zeros changes no sum — so a rank ships ONLY the non-zero rows, 64 B each:
//     {index, ||dL/dmean2D|| * norm_scale, dL/dRGB[3], d xyz[3], d opacity, d scaling[3], d rotation[4]}
// (all-gather of the packed rows, train.Trainer.exchange_rows).  Nothing dense is rebuilt from them: index_rows_kernel leaves,
// per Gaussian, a bit mask of the views that hold a row of it and the row's position in each, and the replicated optimizer
// step (rows_adam_kernel, w3d_preprocess.hip) walks every Gaussian's set bits in VIEW ORDER — the same additions in the same
// order on every rank, so the replicas stay bit-identical without any parameter traffic.  apply_rows_kernel is the dense
// materialisation of the same sums (one launch per view): the checker of the indexed path and the form a caller uses who
// wants the (V,P,3) colour planes and the summed geometry gradients themselves.
#include "w3d_common.h"

namespace {

constexpr int PACK_ITEMS = 8;                 // Gaussians per thread: 2048 per workgroup -> one counter atomic per 2048

__global__ void __launch_bounds__(256)
pack_rows_kernel(int P, const float *__restrict__ dcolor, const float *__restrict__ gxyz, const float *__restrict__ gop,
                 const float *__restrict__ gsc, const float *__restrict__ grot, const float *__restrict__ gnorm,
                 float norm_scale, float4 *__restrict__ rows, uint32_t capacity, uint32_t *__restrict__ count) {
    __shared__ uint32_t wave_total[4];
    __shared__ uint32_t block_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t lt = lane ? (~0ull >> (64u - lane)) : 0ull;
    const int base_g = blockIdx.x * (256 * PACK_ITEMS);
    uint64_t bal[PACK_ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PACK_ITEMS; i++) {
        const int g = base_g + i * 256 + (int)threadIdx.x;
        // all 15 loads are independent (no short-circuit between them): OR of the raw bits, sign bits dropped at the end
        uint32_t bits = 0u;
        if (g < P) {
            const uint32_t *dc = reinterpret_cast<const uint32_t *>(dcolor) + 3 * (size_t)g;
            const uint32_t *x = reinterpret_cast<const uint32_t *>(gxyz) + 3 * (size_t)g;
            const uint32_t *sc = reinterpret_cast<const uint32_t *>(gsc) + 3 * (size_t)g;
            const uint32_t *q = reinterpret_cast<const uint32_t *>(grot) + 4 * (size_t)g;
            bits = dc[0] | dc[1] | dc[2] | x[0] | x[1] | x[2] | sc[0] | sc[1] | sc[2] | q[0] | q[1] | q[2] | q[3] |
                   reinterpret_cast<const uint32_t *>(gop)[g];
            if (gnorm) bits |= reinterpret_cast<const uint32_t *>(gnorm)[g];
        }
        const bool any = (bits & 0x7FFFFFFFu) != 0u;
        bal[i] = w3d_ballot(any);
        mine += (uint32_t)__popcll(bal[i]);                      // (wave-uniform)
    }
    if (lane == 0) wave_total[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base;
    for (uint32_t w = 0; w < wv; w++) pos += wave_total[w];
#pragma unroll
    for (int i = 0; i < PACK_ITEMS; i++) {
        if ((bal[i] >> lane) & 1ull) {
            const int g = base_g + i * 256 + (int)threadIdx.x;
            const uint32_t r = pos + (uint32_t)__popcll(bal[i] & lt);
            if (r < capacity) {
                rows[4 * (size_t)r + 0] = make_float4(__uint_as_float((uint32_t)g), gnorm ? gnorm[g] * norm_scale : 0.f,
                                                      dcolor[3 * g], dcolor[3 * g + 1]);
                rows[4 * (size_t)r + 1] = make_float4(dcolor[3 * g + 2], gxyz[3 * g], gxyz[3 * g + 1], gxyz[3 * g + 2]);
                rows[4 * (size_t)r + 2] = make_float4(gop[g], gsc[3 * g], gsc[3 * g + 1], gsc[3 * g + 2]);
                rows[4 * (size_t)r + 3] = make_float4(grot[4 * g], grot[4 * g + 1], grot[4 * g + 2], grot[4 * g + 3]);
            }
        }
        pos += (uint32_t)__popcll(bal[i]);
    }
}

// one thread per row of ONE view; the indices of a view are distinct, so the read-modify-writes do not collide
__global__ void __launch_bounds__(256)
apply_rows_kernel(int P, const float4 *__restrict__ rows, const uint32_t *__restrict__ count, uint32_t max_rows,
                  float *__restrict__ dcolor_view, float *__restrict__ sxyz, float *__restrict__ sop,
                  float *__restrict__ ssc, float *__restrict__ srot, float *__restrict__ norm_sum) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = min(*count, max_rows);
    if (r >= n) return;
    const float4 a = rows[4 * (size_t)r], b = rows[4 * (size_t)r + 1], c = rows[4 * (size_t)r + 2], d = rows[4 * (size_t)r + 3];
    const uint32_t g = __float_as_uint(a.x);
    if (g >= (uint32_t)P) return;
    if (norm_sum) norm_sum[g] += a.y;
    dcolor_view[3 * (size_t)g] = a.z; dcolor_view[3 * (size_t)g + 1] = a.w; dcolor_view[3 * (size_t)g + 2] = b.x;
    sxyz[3 * (size_t)g] += b.y; sxyz[3 * (size_t)g + 1] += b.z; sxyz[3 * (size_t)g + 2] += b.w;
    sop[g] += c.x;
    ssc[3 * (size_t)g] += c.y; ssc[3 * (size_t)g + 1] += c.z; ssc[3 * (size_t)g + 2] += c.w;
    srot[4 * (size_t)g] += d.x; srot[4 * (size_t)g + 1] += d.y; srot[4 * (size_t)g + 2] += d.z; srot[4 * (size_t)g + 3] += d.w;
}

// (view, row) -> "Gaussian g has row r in view v": slots[v][g] = r and bit v of viewmask[g].  The indices of one view are
// distinct; different views meet in the mask word only (atomic OR: order-independent).
__global__ void __launch_bounds__(256)
index_rows_kernel(int P, const float4 *__restrict__ rows_all, const uint32_t *__restrict__ counts, uint32_t cap,
                  uint32_t *__restrict__ viewmask, uint32_t *__restrict__ slots) {
    const uint32_t v = blockIdx.y, r = blockIdx.x * 256u + threadIdx.x;
    if (r >= min(counts[v], cap)) return;
    const uint32_t g = __float_as_uint(rows_all[4 * ((size_t)v * cap + r)].x);
    if (g >= (uint32_t)P) return;
    slots[(size_t)v * P + g] = r;
    atomicOr(&viewmask[g], 1u << v);
}

// norm_sum[g] = sum over the views that hold a row of g, in view order, of the row's ||dL/dmean2D|| (0 where none does)
__global__ void __launch_bounds__(256)
rows_norm_sum_kernel(int P, const float4 *__restrict__ rows_all, uint32_t cap, const uint32_t *__restrict__ viewmask,
                     const uint32_t *__restrict__ slots, float *__restrict__ norm_sum, int accumulate) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    uint32_t mask = viewmask[g];
    float s = 0.f;
    while (mask) {
        const uint32_t v = (uint32_t)__ffs((int)mask) - 1u;
        mask &= mask - 1u;
        s += rows_all[4 * ((size_t)v * cap + slots[(size_t)v * P + g])].y;
    }
    // accumulate: norm_sum IS the running statistic (xyz_gradient_accum += the views' sum: the same single addition torch's `+=`
    // of the summed array performs, without the array)
    norm_sum[g] = accumulate ? norm_sum[g] + s : s;
}

bool geo_ok(const w3d_raw_grads *g) {
    return g && g->xyz && g->opacity && g->scaling && g->rotation;
}

}  // namespace

extern "C" int w3d_pack_gradient_rows(int32_t P, const float *dcolor, const w3d_raw_grads *grads, const float *grad2d_norm,
                                      float norm_scale, float *rows_out, uint32_t capacity_rows, uint32_t *count,
                                      w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || !count) { w3d_set_error("pack_gradient_rows: bad arguments"); return W3D_ERR_INVALID; }
    W3D_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), stream));
    if (P == 0) return W3D_OK;
    if (!dcolor || !geo_ok(grads) || !rows_out || (reinterpret_cast<uintptr_t>(rows_out) & 15)) {
        w3d_set_error("pack_gradient_rows: NULL buffer, or rows not 16-byte aligned");
        return W3D_ERR_INVALID;
    }
    W3D_PROF("pack_gradient_rows", stream);
    const unsigned blocks = (unsigned)((P + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS));
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, stream, P, dcolor, grads->xyz, grads->opacity, grads->scaling,
                       grads->rotation, grad2d_norm, norm_scale, reinterpret_cast<float4 *>(rows_out), capacity_rows, count);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

extern "C" int w3d_apply_gradient_rows(int32_t P, const float *rows, const uint32_t *count, uint32_t max_rows,
                                       float *dcolor_view, const w3d_raw_grads *sums, float *norm_sum, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || !count) { w3d_set_error("apply_gradient_rows: bad arguments"); return W3D_ERR_INVALID; }
    if (P == 0 || max_rows == 0) return W3D_OK;
    if (!rows || (reinterpret_cast<uintptr_t>(rows) & 15) || !dcolor_view || !geo_ok(sums)) {
        w3d_set_error("apply_gradient_rows: NULL buffer, or rows not 16-byte aligned");
        return W3D_ERR_INVALID;
    }
    W3D_PROF("apply_gradient_rows", stream);
    hipLaunchKernelGGL(apply_rows_kernel, dim3((max_rows + 255u) / 256u), dim3(256), 0, stream, P,
                       reinterpret_cast<const float4 *>(rows), count, max_rows, dcolor_view, sums->xyz, sums->opacity,
                       sums->scaling, sums->rotation, norm_sum);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

extern "C" int w3d_index_gradient_rows(int32_t P, int32_t n_views, const float *rows_all, const uint32_t *counts,
                                       uint32_t cap_rows, uint32_t *viewmask, uint32_t *slots, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || n_views < 1 || n_views > 32) { w3d_set_error("index_gradient_rows: 1..32 views"); return W3D_ERR_INVALID; }
    if (P == 0) return W3D_OK;
    if (!viewmask || !slots || !counts || (cap_rows && (!rows_all || (reinterpret_cast<uintptr_t>(rows_all) & 15)))) {
        w3d_set_error("index_gradient_rows: NULL buffer, or rows not 16-byte aligned");
        return W3D_ERR_INVALID;
    }
    W3D_HIP_CHECK(hipMemsetAsync(viewmask, 0, (size_t)P * sizeof(uint32_t), stream));
    if (cap_rows == 0) return W3D_OK;
    W3D_PROF("index_gradient_rows", stream);
    hipLaunchKernelGGL(index_rows_kernel, dim3((cap_rows + 255u) / 256u, (unsigned)n_views), dim3(256), 0, stream, P,
                       reinterpret_cast<const float4 *>(rows_all), counts, cap_rows, viewmask, slots);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

extern "C" int w3d_rows_norm_sum(int32_t P, int32_t n_views, const float *rows_all, uint32_t cap_rows, const uint32_t *viewmask,
                                 const uint32_t *slots, float *norm_sum, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || n_views < 1 || n_views > 32) { w3d_set_error("rows_norm_sum: 1..32 views"); return W3D_ERR_INVALID; }
    if (P == 0) return W3D_OK;
    if (!viewmask || !slots || !norm_sum || !rows_all) { w3d_set_error("rows_norm_sum: NULL buffer"); return W3D_ERR_INVALID; }
    hipLaunchKernelGGL(rows_norm_sum_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P,
                       reinterpret_cast<const float4 *>(rows_all), cap_rows, viewmask, slots, norm_sum, 0);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

extern "C" int w3d_rows_norm_accumulate(int32_t P, int32_t n_views, const float *rows_all, uint32_t cap_rows, const uint32_t *viewmask,
                                        const uint32_t *slots, float *accum, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || n_views < 1 || n_views > 32) { w3d_set_error("rows_norm_accumulate: 1..32 views"); return W3D_ERR_INVALID; }
    if (P == 0) return W3D_OK;
    if (!viewmask || !slots || !accum || !rows_all) { w3d_set_error("rows_norm_accumulate: NULL buffer"); return W3D_ERR_INVALID; }
    hipLaunchKernelGGL(rows_norm_sum_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P,
                       reinterpret_cast<const float4 *>(rows_all), cap_rows, viewmask, slots, accum, 1);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

extern "C" int w3d_rows_adam(int32_t P, int32_t n_views, int32_t sh_degree, const float *campos_all, const float *rows_all,
                             uint32_t cap_rows, const uint32_t *viewmask, const uint32_t *slots, const w3d_raw_blocks *params,
                             const w3d_adam_fused *adam, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || n_views < 1 || n_views > 32 || sh_degree < 0 || sh_degree > 3) { w3d_set_error("rows_adam: bad sizes"); return W3D_ERR_INVALID; }
    if (P == 0) return W3D_OK;
    if (!campos_all || !rows_all || !viewmask || !slots || !params || !adam) { w3d_set_error("rows_adam: NULL buffer"); return W3D_ERR_INVALID; }
    for (int i = 0; i < 6; i++)
        if (!adam->skip[i] && (!(adam->bias_correction1[i] > 0.f) || !(adam->bias_correction2[i] > 0.f))) {
            w3d_set_error("rows_adam: bias corrections must be positive (step >= 1)");
            return W3D_ERR_INVALID;
        }
    return w3d_launch_rows_adam(P, n_views, sh_degree, campos_all, rows_all, cap_rows, viewmask, slots, *params, *adam, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Per-rank visibility bookkeeping of the view-parallel step (train.Trainer.track_local): count[g] += radii[g] > 0,
// rmax[g] = max(rmax[g], radii[g]) — one launch instead of five torch kernels (compare, two casts, add, maximum) in every step.
namespace {
__global__ void __launch_bounds__(256)
track_visibility_kernel(int P, const int32_t *__restrict__ radii, int32_t *__restrict__ count, int32_t *__restrict__ rmax) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const int32_t r = radii[g];
    if (r > 0) { count[g] += 1; rmax[g] = max(rmax[g], r); }
}
}  // namespace

extern "C" int w3d_track_visibility(int32_t P, const int32_t *radii, int32_t *vis_count, int32_t *radii_max, w3d_stream_t stream_) {
    if (P < 0 || (P > 0 && (!radii || !vis_count || !radii_max))) { w3d_set_error("track_visibility: bad arguments"); return W3D_ERR_INVALID; }
    if (P == 0) return W3D_OK;
    hipLaunchKernelGGL(track_visibility_kernel, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), P, radii,
                       vis_count, radii_max);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
