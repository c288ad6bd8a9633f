// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for 16-byte GATHERS. The guide's x2 correction of FETCH_SIZE
// is measured for wide coalesced streaming reads only; k_prep_contacts reads its contact points as 16-byte gathers through the
// colour-sorted order (one float4 of one manifold per lane and array), so its counter figure needs its own factor.
//
// Each kernel below makes every lane load ONE float4 from an array far larger than L2 + Infinity Cache (so that nothing is served on-die)
// with a known pattern - and therefore a known number of distinct 64-byte sectors and useful bytes per launch:
//   stream     lane i reads element i                      (wide coalesced: the guide's case; 16 B useful = 16 B of sectors per lane)
//   stride4    lane i reads element 4 i                    (every lane its own 64-byte sector, neighbouring sectors; 16 B useful / 64 B)
//   stride8    lane i reads element 8 i                    (every lane its own 128-byte line; 16 B useful / 128 B)
//   permuted   lane i reads element perm[i], perm a random permutation of the whole array (the prep pattern: a 16-byte gather whose
//              sector neighbours are read by unrelated lanes at unrelated times)
//   store16    lane i writes element perm[i] (scattered 16-byte stores: WRITE_SIZE's granularity)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); scripts/pmc_kernel_traffic.py-style post-processing:
// counter bytes per launch / lanes = what the counter charges per 16-byte access. Prints the useful and the sector bytes per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_gather_stream(const float4 *__restrict__ a, uint32_t n, float *sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const float4 v = a[i]; if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ void k_gather_stride4(const float4 *__restrict__ a, uint32_t n, float *sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const float4 v = a[(size_t)4 * i]; if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ void k_gather_stride8(const float4 *__restrict__ a, uint32_t n, float *sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const float4 v = a[(size_t)8 * i]; if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ void k_gather_permuted(const float4 *__restrict__ a, const uint32_t *__restrict__ perm, uint32_t n, float *sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const float4 v = a[perm[i]]; if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ void k_scatter_store16(float4 *__restrict__ a, const uint32_t *__restrict__ perm, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    a[perm[i]] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ void k_stream_store16(float4 *__restrict__ a, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    a[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
int main() {
    const uint32_t n = 8u << 20;                       // 8 Mi lanes per launch
    const size_t elems = (size_t)8 * n;                // 64 Mi float4 = 1 GiB: four times L2 + Infinity Cache
    float4 *a; uint32_t *perm; float *sink;
    CK(hipMalloc(&a, elems * sizeof(float4))); CK(hipMalloc(&perm, (size_t)n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, elems * sizeof(float4)));
    std::vector<uint32_t> h(n);
    {   // a random sample of n DISTINCT elements out of `elems`: shuffle the sector-granular index space so that no two lanes share a sector
        std::mt19937_64 rng(12345);
        std::vector<uint32_t> sectors(elems / 4);
        std::iota(sectors.begin(), sectors.end(), 0u);
        std::shuffle(sectors.begin(), sectors.end(), rng);
        for (uint32_t i = 0; i < n; ++i) h[i] = 4u * sectors[i] + (uint32_t)(rng() & 3);
    }
    CK(hipMemcpy(perm, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    const dim3 g((n + 255) / 256), b(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch, double useful, double sectors64) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s lanes %u  useful %.1f MB  distinct 64-B sectors %.1f MB  %.1f us per launch\n", name, n, useful / 1e6, sectors64 / 1e6, 1e3 * ms / 3);
    };
    timed("k_gather_stream", [&] { hipLaunchKernelGGL(k_gather_stream, g, b, 0, 0, a, n, sink); }, 16.0 * n, 16.0 * n);
    timed("k_gather_stride4", [&] { hipLaunchKernelGGL(k_gather_stride4, g, b, 0, 0, a, n, sink); }, 16.0 * n, 64.0 * n);
    timed("k_gather_stride8", [&] { hipLaunchKernelGGL(k_gather_stride8, g, b, 0, 0, a, n, sink); }, 16.0 * n, 64.0 * n);
    timed("k_gather_permuted", [&] { hipLaunchKernelGGL(k_gather_permuted, g, b, 0, 0, a, perm, n, sink); }, 16.0 * n + 4.0 * n, 64.0 * n + 4.0 * n);
    timed("k_scatter_store16", [&] { hipLaunchKernelGGL(k_scatter_store16, g, b, 0, 0, a, perm, n); }, 16.0 * n, 64.0 * n);
    timed("k_stream_store16", [&] { hipLaunchKernelGGL(k_stream_store16, g, b, 0, 0, a, n); }, 16.0 * n, 16.0 * n);
    return 0;
}
