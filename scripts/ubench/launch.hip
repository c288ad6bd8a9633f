// Micro-benchmark: cost of a chain of dependent small kernels on one stream, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_empty(float *p) { if (p == nullptr) p[0] = 1; }
__global__ void k_touch(float4 *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float4 v = p[i]; v.x += 1; p[i] = v; } }
__global__ void k_gather(const uint32_t *idx, float4 *body, const float4 *rows, int n, int rcap, int nrow) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    uint32_t a = idx[i];
    float4 acc = body[a];
    for (int r = 0; r < nrow; ++r) { float4 v = rows[(size_t)r * rcap + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; }
    body[a] = acc;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 9500, nb = 32768, rcap = 16384, nrow = 60;
    float4 *buf, *body, *rows; uint32_t *idx;
    CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&body, nb * 16)); CK(hipMalloc(&rows, (size_t)rcap * nrow * 16 * 16)); CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(buf, 0, n * 16)); CK(hipMemset(body, 0, nb * 16)); CK(hipMemset(rows, 0, (size_t)rcap * nrow * 16 * 16));
    std::vector<uint32_t> h(n); for (int i = 0; i < n; ++i) h[i] = (uint32_t)((i * 2654435761u) % nb);
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 2000;
    auto run = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 100; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < L; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s eager  %.2f us/launch\n", name, 1e3 * ms / L);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s graph  %.2f us/launch\n", name, 1e3 * ms / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return 0;
    };
    run("empty <<<1,64>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float *)buf); });
    run("empty <<<148,64>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(148), dim3(64), 0, s, (float *)buf); });
    run("touch 9.5k float4 <<<148,64>>>", [&](int) { hipLaunchKernelGGL(k_touch, dim3(148), dim3(64), 0, s, buf, n); });
    run("gather+60 rows (different slice each) 64", [&](int i) { hipLaunchKernelGGL(k_gather, dim3(148), dim3(64), 0, s, idx, body, rows + (size_t)(i % 15) * rcap * nrow, n, rcap, nrow); });
    run("gather+60 rows (same slice) 64", [&](int) { hipLaunchKernelGGL(k_gather, dim3(148), dim3(64), 0, s, idx, body, rows, n, rcap, nrow); });
    run("gather+20 rows (different slice) 64", [&](int i) { hipLaunchKernelGGL(k_gather, dim3(148), dim3(64), 0, s, idx, body, rows + (size_t)(i % 15) * rcap * nrow, n, rcap, 20); });
    run("gather+60 rows (different slice) 256", [&](int i) { hipLaunchKernelGGL(k_gather, dim3(38), dim3(256), 0, s, idx, body, rows + (size_t)(i % 15) * rcap * nrow, n, rcap, nrow); });
    return 0;
}
