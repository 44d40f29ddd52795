// w3d_mask.hip — the per-view mask work around the FlashSplat contribution render, on the device:
//   * binarising an object-mask image (reference run_3d_seg.py:88-89: binarize_mask(PILtoTorch(png)) on the CPU, then a
//     7.7-MB float upload per mask): the decoded 8-bit pixels are uploaded as they are (1 or 3 bytes per pixel) and turned
//     into the float label image the rasterizer takes here;
//   * find_match's scoring (run_3d_seg.py:127-163): pred = alpha > 0.5 of a subset render, its bounding box, and the
//     intersection / union counts of pred with K candidate masks — one pass over the alpha image instead of a
//     .cpu().numpy() copy of it per view plus K numpy logical_and / logical_or passes.
#include "w3d_common.h"

namespace {

// out[h][w] = any channel of pixels[h][w][:] > 0 ? 1 : 0      (utils/wheatgs_utils.py:26-37 binarize_mask)
__global__ void __launch_bounds__(256)
mask_binarize_kernel(size_t n, int C, const uint8_t *__restrict__ pixels, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool any = false;
    for (int c = 0; c < C; c++) any = any || pixels[i * C + c] != 0;
    out[i] = any ? 1.0f : 0.0f;
}

// out: [0 .. 2K) = {intersection, union} per mask, [2K .. 2K+4) = bbox of pred {x_min, y_min, x_max, y_max} (x_min > x_max
// when pred is empty), [2K+4] = pixels of pred.  The caller presets out to zeros and the bbox minima to 0xFFFFFFFF.
__global__ void __launch_bounds__(256)
mask_iou_kernel(int H, int W, int K, const float *__restrict__ alpha, float thresh, const uint8_t *__restrict__ masks,
                uint32_t *__restrict__ out) {
    extern __shared__ uint32_t sh[];          // 2K + 5 block-level accumulators
    const int nacc = 2 * K + 5;
    for (int i = threadIdx.x; i < nacc; i += blockDim.x) sh[i] = (i == 2 * K || i == 2 * K + 1) ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    const size_t HW = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) {
        const bool pred = alpha[i] > thresh;
        const uint64_t pb = w3d_ballot(pred);
        if (pred) {
            const uint32_t x = (uint32_t)(i % W), y = (uint32_t)(i / W);
            atomicMin(&sh[2 * K], x); atomicMin(&sh[2 * K + 1], y);
            atomicMax(&sh[2 * K + 2], x); atomicMax(&sh[2 * K + 3], y);
        }
        if ((threadIdx.x & 63) == 0 && pb) atomicAdd(&sh[2 * K + 4], (uint32_t)__popcll(pb));
        for (int k = 0; k < K; k++) {
            const bool m = masks[(size_t)k * HW + i] != 0;
            const uint64_t bi = w3d_ballot(m && pred), bu = w3d_ballot(m || pred);
            if ((threadIdx.x & 63) == 0) {
                if (bi) atomicAdd(&sh[2 * k], (uint32_t)__popcll(bi));
                if (bu) atomicAdd(&sh[2 * k + 1], (uint32_t)__popcll(bu));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nacc; i += blockDim.x) {
        const uint32_t v = sh[i];
        if (i == 2 * K || i == 2 * K + 1) atomicMin(&out[i], v);
        else if (i == 2 * K + 2 || i == 2 * K + 3) atomicMax(&out[i], v);
        else if (v) atomicAdd(&out[i], v);
    }
}

__global__ void mask_iou_init_kernel(int K, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * K + 5) out[i] = (i == 2 * K || i == 2 * K + 1) ? 0xFFFFFFFFu : 0u;
}

}  // namespace

int w3d_launch_mask_binarize(int H, int W, int C, const uint8_t *pixels, float *out, hipStream_t stream) {
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(mask_binarize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, C, pixels, out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}

int w3d_launch_mask_iou(int H, int W, int K, const float *alpha, float thresh, const uint8_t *masks, uint32_t *out,
                        hipStream_t stream) {
    hipLaunchKernelGGL(mask_iou_init_kernel, dim3((2 * K + 5 + 255) / 256), dim3(256), 0, stream, K, out);
    const size_t n = (size_t)H * W;
    const unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(mask_iou_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), (size_t)(2 * K + 5) * 4, stream, H, W, K, alpha, thresh,
                       masks, out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
