// How should k_prep_contacts read a manifold's contact points? Every lane owns one manifold reached through a
// permutation (the colour-sorted order) and needs its 4 points x 5 float4 (pivot A, pivot B, normal, local normal, impulses) = 320 B.
//   soa        what runs: five arrays indexed [k * cap + m] - 20 gathers of 16 B, each from its own 128-byte line
//   aos_lane   one 320-byte record per manifold, every lane reads its own 20 float4 (round 5's experiment: each instruction still touches
//              64 different lines; the reuse of a line by the lane's next loads depends on the L1)
//   aos_coop   the same records, loaded COOPERATIVELY: 20 lanes read one record's 20 float4 - contiguous - three records per pass, through
//              LDS to the owner lane (3 x 2.5 lines per instruction instead of 64)
// Prints the time per launch; under rocprofv3 --pmc FETCH_SIZE the bytes each form pulls over the fabric.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_points_soa(const float4 *__restrict__ a, size_t cap, const uint32_t *__restrict__ perm, uint32_t n, float *sink) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; if (p >= n) return;
    const uint32_t m = perm[p];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int f = 0; f < 5; ++f) { const float4 v = a[((size_t)f * 4 + k) * cap + m]; acc += v.x + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_points_aos_lane(const float4 *__restrict__ a, const uint32_t *__restrict__ perm, uint32_t n, float *sink) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; if (p >= n) return;
    const float4 *r = a + (size_t)perm[p] * 20;
    float acc = 0;
#pragma unroll
    for (int j = 0; j < 20; ++j) { const float4 v = r[j]; acc += v.x + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
constexpr int kPad = 21;   // float4 per record in LDS (one of padding: the owner lanes' reads then spread over the banks)
__global__ void __launch_bounds__(256) k_points_aos_coop(const float4 *__restrict__ a, const uint32_t *__restrict__ perm, uint32_t n, float *sink) {
    __shared__ float4 lds[4][64 * kPad];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t m = p < n ? perm[p] : 0xFFFFFFFFu;
    float4 *mine = lds[wave];
    const uint32_t sub = lane / 20u, j = lane % 20u;   // lanes 0-59: three records per pass
#pragma unroll 1
    for (uint32_t g = 0; g < 22; ++g) {
        const uint32_t src = g * 3 + sub;
        const uint32_t ms = __shfl(m, src < 64 ? src : 0);
        if (lane < 60 && src < 64 && ms != 0xFFFFFFFFu) mine[src * kPad + j] = a[(size_t)ms * 20 + j];
    }
    __builtin_amdgcn_wave_barrier();
    float acc = 0;
    if (m != 0xFFFFFFFFu) {
#pragma unroll
        for (int jj = 0; jj < 20; ++jj) { const float4 v = mine[lane * kPad + jj]; acc += v.x + v.w; }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
int main() {
    const uint32_t n = 2u << 20;                // 2 Mi manifolds per launch (the headline pile has 0.15 Mi active, C4 0.8 Mi)
    const size_t cap = n;
    float4 *soa, *aos; uint32_t *perm; float *sink;
    CK(hipMalloc(&soa, cap * 20 * sizeof(float4))); CK(hipMalloc(&aos, cap * 20 * sizeof(float4))); CK(hipMalloc(&perm, (size_t)n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(soa, 0, cap * 20 * sizeof(float4))); CK(hipMemset(aos, 0, cap * 20 * sizeof(float4)));
    std::vector<uint32_t> h(n);
    std::iota(h.begin(), h.end(), 0u);
    // the colour-sorted order: ~18 colours, ascending manifold index inside a colour - a stride-18 interleave, not a random shuffle
    std::vector<uint32_t> colour(n);
    std::mt19937 rng(7);
    for (uint32_t i = 0; i < n; ++i) colour[i] = rng() % 18u;
    std::stable_sort(h.begin(), h.end(), [&](uint32_t x, uint32_t y) { return colour[x] < colour[y]; });
    CK(hipMemcpy(perm, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    const dim3 g((n + 255) / 256), b(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s manifolds %u  useful %.0f MB  %.1f us per launch  (%.2f TB/s of useful bytes)\n", name, n, 320.0 * n / 1e6, 1e3 * ms / 5, 320.0 * n / (ms / 5 * 1e-3) / 1e12);
    };
    timed("k_points_soa", [&] { hipLaunchKernelGGL(k_points_soa, g, b, 0, 0, soa, cap, perm, n, sink); });
    timed("k_points_aos_lane", [&] { hipLaunchKernelGGL(k_points_aos_lane, g, b, 0, 0, aos, perm, n, sink); });
    timed("k_points_aos_coop", [&] { hipLaunchKernelGGL(k_points_aos_coop, g, b, 0, 0, aos, perm, n, sink); });
    return 0;
}
