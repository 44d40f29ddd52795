// Reference point for the depth sort (profiles/HISTORY.md section D): rocPRIM's device radix sort of (u32 depth bits, u32 id) pairs on
// the same GPU.  Not part of the product (the sort of w3d_binning.hip is hand-written); build and run by hand:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/rocprim_sort_probe.hip -o /tmp/rpsort && /tmp/rpsort
// MI355X, ROCm 7.2: 1.22 M pairs 139 us, 2 M pairs 150 us (ours: 160 us for 2 M keys in, 1.22 M out, culled keys dropped).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    for (size_t n : {1220000ul, 2000000ul}) {
        std::vector<uint32_t> hk(n), hv(n);
        std::mt19937 rng(1);
        for (size_t i = 0; i < n; i++) { float d = 1.7f + 1.5f * (rng() / 4294967296.0f); memcpy(&hk[i], &d, 4); hv[i] = (uint32_t)i; }
        uint32_t *k0, *k1, *v0, *v1; void *tmp = nullptr; size_t tb = 0;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
        CK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
        CK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, 32, 0));
        CK(hipMalloc(&tmp, tb));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int bits : {32, 27}) {
            for (int w = 0; w < 5; w++) CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0));
            hipEventRecord(a, 0);
            for (int w = 0; w < 50; w++) CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0));
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n %zu bits %d: %.1f us per sort (temp %zu KB)\n", n, bits, 1000.f * ms / 50, tb / 1024);
        }
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
    }
    return 0;
}
