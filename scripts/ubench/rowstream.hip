// Micro-benchmark: how fast can N lanes stream their solver rows (60 float4 per lane, layout rw[plane*cap + p])
// as a function of resident waves and of the layout (plane-major as in the solver vs. one contiguous block per wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NPL = 60;
template <int LAYOUT>   // 0: plane-major, 1: wave blocks [wave][plane][lane]
__global__ void __launch_bounds__(256) k_stream(const float4 *__restrict__ rw, size_t cap, uint32_t n, uint32_t stride, float *sink, int reps) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    for (int rep = 0; rep < reps; ++rep)
        for (uint32_t p = t; p < n; p += stride) {
            float4 v[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const size_t idx = LAYOUT == 0 ? (size_t)k * cap + p : ((size_t)(p >> 6) * NPL + k) * 64 + (p & 63);
                v[k] = rw[idx];
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) acc += v[k].x + v[k].w;
        }
    if (acc == 12345.678f) sink[0] = acc;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const uint32_t n = 147456;            // lanes' worth of rows (~ the pile's active manifolds), multiple of 64
    const size_t cap = 525328;
    float4 *rw; float *sink;
    CK(hipMalloc(&rw, cap * NPL * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(rw, 0, cap * NPL * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)n * NPL * 16;
    for (int layout = 0; layout < 2; ++layout)
        for (uint32_t lanes : {16384u, 32768u, 65536u, 131072u, 147456u}) {
            const int reps = 20;
            auto launch = [&] {
                if (layout == 0) hipLaunchKernelGGL(k_stream<0>, dim3(lanes / 256), dim3(256), 0, s, rw, cap, n, lanes, sink, reps);
                else hipLaunchKernelGGL(k_stream<1>, dim3(lanes / 256), dim3(256), 0, s, rw, cap, n, lanes, sink, reps);
            };
            launch(); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("layout %s  resident lanes %6u : %.2f TB/s  (%.1f us per pass over %.0f MB)\n", layout ? "wave-block " : "plane-major", lanes,
                   bytes * reps / (ms * 1e-3) / 1e12, 1e3 * ms / reps, bytes / 1e6);
        }
    return 0;
}
