// w3d_knn.hip — mean squared distance to the 3 nearest other points (SURVEY.md Appendix A.7;
// replaces simple_knn._C.distCUDA2, reference scene/gaussian_model.py:20,148 — a one-shot call
// at scene initialisation).  Exact brute force: every workgroup stages 256 candidate points in
// LDS (SoA, broadcast reads) and each lane keeps a branch-free sorted triple of the smallest
// distances.  Distances are formed with FP contraction off so the result is bit-identical to the
// CPU oracle's.  O(N^2): fine for SfM-sized initialisations (<= a few 1e5 points).
#include "w3d_common.h"

namespace {

__global__ void __launch_bounds__(256)
knn3_kernel(int N, const float *__restrict__ pts, float *__restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < N;
    const float qx = act ? pts[3 * (size_t)i] : 0.f, qy = act ? pts[3 * (size_t)i + 1] : 0.f, qz = act ? pts[3 * (size_t)i + 2] : 0.f;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    for (int base = 0; base < N; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < N) { sx[threadIdx.x] = pts[3 * (size_t)j]; sy[threadIdx.x] = pts[3 * (size_t)j + 1]; sz[threadIdx.x] = pts[3 * (size_t)j + 2]; }
        __syncthreads();
        const int n = min(256, N - base);
        for (int k = 0; k < n; k++) {
            const float dx = qx - sx[k], dy = qy - sy[k], dz = qz - sz[k];
            float d = dx * dx + dy * dy + dz * dz;
            d = (base + k == i) ? INFINITY : d;
            const float t = fmaxf(b0, d);
            b0 = fminf(b0, d);
            const float t2 = fmaxf(b1, t);
            b1 = fminf(b1, t);
            b2 = fminf(b2, t2);
        }
    }
    if (act) {
        const int n = N - 1 < 3 ? N - 1 : 3;
        float sum = 0.f;
        if (n > 0) sum += b0;
        if (n > 1) sum += b1;
        if (n > 2) sum += b2;
        out[i] = n > 0 ? sum / 3.0f : 0.f;
    }
}

}  // namespace

int w3d_launch_knn(int32_t N, const float *points, float *out, hipStream_t stream) {
    if (N <= 0) return W3D_OK;
    hipLaunchKernelGGL(knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, points, out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
