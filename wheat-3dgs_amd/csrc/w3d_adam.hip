// w3d_adam.hip — one-sweep Adam over a block of the flat parameter buffer (SURVEY.md §8f row
// N2; replaces torch.optim.Adam's multi-kernel step over the 6 parameter groups of reference
// scene/gaussian_model.py:172-182, eps 1e-15).  Pure HBM streaming: reads p,g,m,v and writes
// p,m,v (+ g = 0 when asked) = 28|32 B per element, 16-B vector accesses, grid-stride.
#include "w3d_common.h"

namespace {

__device__ __forceinline__ void adam1(float &p, float &g, float &m, float &v, float step_size, float b1, float b2,
                                      float eps, float inv_sqrt_bc2) {
    w3d_adam1(p, g, m, v, step_size, b1, b2, eps, inv_sqrt_bc2);
}

// NT (bit 0 loads, bit 1 stores): nontemporal accesses — the sweep touches 3.3 GB once per step, caching any of it
// only evicts useful lines.
// U: float4 groups per thread and iteration (4 x U independent 16-B loads in flight).
template <bool ZERO, int NT, int U>
__global__ void __launch_bounds__(256)
adam_kernel(uint64_t n, float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
            float step_size, float b1, float b2, float eps, float inv_sqrt_bc2, uint64_t head) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    // unaligned head / tail elements (all four arrays share the same misalignment)
    const uint64_t nvec = (n - head) / 4, tail = head + nvec * 4;
    if (tid < head) {
        float pp = p[tid], gg = g[tid], mm = m[tid], vv = v[tid];
        adam1(pp, gg, mm, vv, step_size, b1, b2, eps, inv_sqrt_bc2);
        p[tid] = pp; m[tid] = mm; v[tid] = vv;
        if (ZERO) g[tid] = 0.f;
    }
    if (tid < n - tail) {
        const uint64_t i = tail + tid;
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam1(pp, gg, mm, vv, step_size, b1, b2, eps, inv_sqrt_bc2);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (ZERO) g[i] = 0.f;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *p4 = reinterpret_cast<f4 *>(p + head), *g4 = reinterpret_cast<f4 *>(g + head);
    f4 *m4 = reinterpret_cast<f4 *>(m + head), *v4 = reinterpret_cast<f4 *>(v + head);
    auto ld = [](const f4 *a) -> f4 { return (NT & 1) ? __builtin_nontemporal_load(a) : *a; };
    auto st = [](f4 *a, f4 x) { if (NT & 2) __builtin_nontemporal_store(x, a); else *a = x; };
    for (uint64_t i0 = tid; i0 < nvec; i0 += nthreads * U) {
        f4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            if (i < nvec) { pp[u] = ld(p4 + i); gg[u] = ld(g4 + i); mm[u] = ld(m4 + i); vv[u] = ld(v4 + i); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            if (i < nvec) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float a = pp[u][c], b = gg[u][c], cm = mm[u][c], cv = vv[u][c];
                    adam1(a, b, cm, cv, step_size, b1, b2, eps, inv_sqrt_bc2);
                    pp[u][c] = a; mm[u][c] = cm; vv[u][c] = cv;
                }
                st(p4 + i, pp[u]); st(m4 + i, mm[u]); st(v4 + i, vv[u]);
                if (ZERO) st(g4 + i, f4{0.f, 0.f, 0.f, 0.f});
            }
        }
    }
}

}  // namespace

extern "C" int w3d_adam_step(uint64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                             float beta1, float beta2, float eps, float bias_correction1, float bias_correction2,
                             int32_t zero_grad, w3d_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (n == 0) return W3D_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) { w3d_set_error("adam: NULL buffer"); return W3D_ERR_INVALID; }
    const uintptr_t a = reinterpret_cast<uintptr_t>(param);
    if ((a & 3) || ((reinterpret_cast<uintptr_t>(grad) ^ a) & 15) || ((reinterpret_cast<uintptr_t>(exp_avg) ^ a) & 15) ||
        ((reinterpret_cast<uintptr_t>(exp_avg_sq) ^ a) & 15)) {
        w3d_set_error("adam: the four arrays must share their 16-B misalignment");
        return W3D_ERR_INVALID;
    }
    uint64_t head = ((16 - (a & 15)) & 15) / 4;
    if (head > n) head = n;
    const float step_size = lr / bias_correction1, inv_sqrt_bc2 = 1.0f / sqrtf(bias_correction2);
    // one iteration per thread (no grid-stride persistence) and nontemporal accesses measured best on MI355X:
    // 6.0 TB/s against 4.9 TB/s for 16 persistent blocks per CU with cached accesses (118 M elements)
    constexpr int U = 2;
    uint64_t blocks = (((n + 3) / 4 + U - 1) / U + 255) / 256;
    if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
    if (blocks < 1) blocks = 1;
    if (zero_grad)
        hipLaunchKernelGGL((adam_kernel<true, 3, U>), dim3((unsigned)blocks), dim3(256), 0, stream, n, param, grad, exp_avg,
                           exp_avg_sq, step_size, beta1, beta2, eps, inv_sqrt_bc2, head);
    else
        hipLaunchKernelGGL((adam_kernel<false, 3, U>), dim3((unsigned)blocks), dim3(256), 0, stream, n, param, grad, exp_avg,
                           exp_avg_sq, step_size, beta1, beta2, eps, inv_sqrt_bc2, head);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
