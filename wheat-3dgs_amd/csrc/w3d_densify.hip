// w3d_densify.hip — one-pass row compaction of the flat parameter buffer and both Adam moments for
// densify / prune (replaces the boolean-mask indexing + torch.cat of every
// parameter group and of its optimizer state that reference scene/gaussian_model.py:332-397
// (_prune_optimizer, cat_tensors_to_optimizer, densification_postfix, prune_points) performs three to four
// times per densification).  The host decides WHICH rows survive / are cloned / are split children
// (gaussian_model.py densify_and_prune: a handful of P-sized boolean ops) and passes one source-row index per
// output row; this kernel then moves every block of the three flat buffers exactly once:
//   out rows [0, n_keep)        : surviving originals        — parameters AND moments copied
//   out rows [n_keep, n_child0) : clones                     — parameters copied, moments zero
//   out rows [n_child0, P_new)  : split children             — parameters copied, moments zero, the
//                                 xyz / scaling blocks taken from the host-provided child arrays
// Pure HBM streaming: 12 B read + 12 B written per element of a surviving row, coalesced writes.
#include "w3d_common.h"

namespace {

struct DensifyBlocks {
    int32_t n;             // number of blocks (<= 8)
    int32_t dim[8];        // floats per row in each block
    int32_t child_xyz;     // block index overridden by child_xyz for split children (-1: none)
    int32_t child_scaling; // block index overridden by child_scaling
};

__global__ void __launch_bounds__(256)
densify_compact_kernel(DensifyBlocks B, uint64_t P_old, uint64_t P_new, uint64_t n_keep, uint64_t n_child0,
                       const int32_t *__restrict__ src, const float *__restrict__ p_old, const float *__restrict__ m_old,
                       const float *__restrict__ v_old, float *__restrict__ p_new, float *__restrict__ m_new,
                       float *__restrict__ v_new, const float *__restrict__ child_xyz,
                       const float *__restrict__ child_scaling) {
    const int b = blockIdx.y;
    const uint32_t dim = (uint32_t)B.dim[b];
    uint64_t off_old = 0, off_new = 0;
    // (every block starts on a multiple of 4 floats: include/w3d.h, w3d_densify_compact)
    for (int i = 0; i < b; i++) {
        off_old = ((off_old + 3ull) & ~3ull) + P_old * (uint64_t)B.dim[i];
        off_new = ((off_new + 3ull) & ~3ull) + P_new * (uint64_t)B.dim[i];
    }
    off_old = (off_old + 3ull) & ~3ull;
    off_new = (off_new + 3ull) & ~3ull;
    const uint64_t n = P_new * dim;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = e / dim;
        const uint32_t col = (uint32_t)(e - row * dim);
        const uint64_t s = (uint64_t)(uint32_t)src[row];
        float pv;
        if (row >= n_child0 && b == B.child_xyz) pv = child_xyz[(row - n_child0) * 3 + col];
        else if (row >= n_child0 && b == B.child_scaling) pv = child_scaling[(row - n_child0) * 3 + col];
        else pv = p_old[off_old + s * dim + col];
        p_new[off_new + e] = pv;
        if (m_new) {
            const bool keep = row < n_keep;
            m_new[off_new + e] = keep ? m_old[off_old + s * dim + col] : 0.f;
            v_new[off_new + e] = keep ? v_old[off_old + s * dim + col] : 0.f;
        }
    }
}

}  // namespace

extern "C" int w3d_densify_compact(int32_t n_blocks, const int32_t *block_dims_host, int32_t xyz_block, int32_t scaling_block,
                                   uint64_t P_old, uint64_t P_new, uint64_t n_keep, uint64_t n_child0, const int32_t *src_rows,
                                   const float *param_old, const float *exp_avg_old, const float *exp_avg_sq_old,
                                   float *param_new, float *exp_avg_new, float *exp_avg_sq_new, const float *child_xyz,
                                   const float *child_scaling, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (n_blocks <= 0 || n_blocks > 8 || !block_dims_host) { w3d_set_error("densify: 1..8 blocks expected"); return W3D_ERR_INVALID; }
    if (P_new == 0) return W3D_OK;
    if (!src_rows || !param_old || !param_new) { w3d_set_error("densify: NULL buffer"); return W3D_ERR_INVALID; }
    if ((exp_avg_new != nullptr) != (exp_avg_sq_new != nullptr) || (exp_avg_new && (!exp_avg_old || !exp_avg_sq_old))) {
        w3d_set_error("densify: moments must be given together");
        return W3D_ERR_INVALID;
    }
    if (n_keep > n_child0 || n_child0 > P_new || P_old >= (1ull << 31) || P_new >= (1ull << 31)) {
        w3d_set_error("densify: inconsistent row counts");
        return W3D_ERR_INVALID;
    }
    if (n_child0 < P_new && (!child_xyz || !child_scaling || xyz_block < 0 || scaling_block < 0 || xyz_block >= n_blocks ||
                             scaling_block >= n_blocks || block_dims_host[xyz_block] != 3 || block_dims_host[scaling_block] != 3)) {
        w3d_set_error("densify: split children need child_xyz / child_scaling and their (3-wide) blocks");
        return W3D_ERR_INVALID;
    }
    DensifyBlocks B{};
    B.n = n_blocks;
    int32_t maxdim = 1;
    for (int i = 0; i < n_blocks; i++) {
        if (block_dims_host[i] <= 0) { w3d_set_error("densify: bad block width"); return W3D_ERR_INVALID; }
        B.dim[i] = block_dims_host[i];
        maxdim = block_dims_host[i] > maxdim ? block_dims_host[i] : maxdim;
    }
    B.child_xyz = xyz_block; B.child_scaling = scaling_block;
    const uint64_t nmax = P_new * (uint64_t)maxdim;
    uint64_t gx = (nmax + 255) / 256;
    if (gx > 65536) gx = 65536;
    hipLaunchKernelGGL(densify_compact_kernel, dim3((uint32_t)gx, (uint32_t)n_blocks), dim3(256), 0, stream, B, P_old, P_new, n_keep,
                       n_child0, src_rows, param_old, exp_avg_old, exp_avg_sq_old, param_new, exp_avg_new, exp_avg_sq_new,
                       child_xyz, child_scaling);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// add_densification_stats (scene/gaussian_model.py:461-463) for a boolean update filter, one pass over P rows
namespace {
__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const float *__restrict__ g2d, const uint8_t *__restrict__ filter, float *__restrict__ accum,
                     float *__restrict__ denom) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P || !filter[g]) return;
    const float x = g2d[3 * (size_t)g], y = g2d[3 * (size_t)g + 1];
    accum[g] += sqrtf(x * x + y * y);
    denom[g] += 1.f;
}
}  // namespace

extern "C" int w3d_add_densification_stats(int32_t P, const float *dL_dmeans2D, const uint8_t *update_filter,
                                           float *xyz_gradient_accum, float *denom, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || (P > 0 && (!dL_dmeans2D || !update_filter || !xyz_gradient_accum || !denom))) {
        w3d_set_error("add_densification_stats: bad arguments");
        return W3D_ERR_INVALID;
    }
    if (P == 0) return W3D_OK;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, dL_dmeans2D, update_filter,
                       xyz_gradient_accum, denom);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
