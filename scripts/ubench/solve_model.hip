// Micro-model of one colour launch of the SI solve: which part of the ~7.5 us is index->gather dependency, row
// streaming, stores, or fixed launch cost? Variants toggle one ingredient at a time on realistic sizes
// (13 colour slices x 14k lanes x 60 float4 of rows = 175 MB working set cycling, 32k bodies).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int GATHER, bool ROWS, bool STORES, int NROW>
__global__ void __launch_bounds__(64) k_model(const uint32_t *__restrict__ idx, float4 *__restrict__ body, float4 *__restrict__ rows, int n, size_t rcap, int pass) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    uint32_t a, b;
    if (GATHER == 1) { a = idx[2 * i]; b = idx[2 * i + 1]; }
    else if (GATHER == 0) { a = (uint32_t)(2 * i) & 32767u; b = (uint32_t)(2 * i + 1) & 32767u; }
    else if (GATHER == 2) { a = ((uint32_t)(2 * i) * 2654435761u >> 7) & 32767u; b = ((uint32_t)(2 * i + 1) * 2654435761u >> 7) & 32767u; }   // random, no dependent load
    else { a = idx[2 * i]; b = idx[2 * i + 1]; }
    float4 r[NROW];
    if (ROWS) {
#pragma unroll
        for (int k = 0; k < NROW; ++k) r[k] = rows[(size_t)k * rcap + i];
    } else {
#pragma unroll
        for (int k = 0; k < NROW; ++k) r[k] = make_float4(k, i, pass, 1);
    }
    uint32_t ra = a, rb = b;
    if (GATHER == 3) { ra = (uint32_t)(2 * i) & 32767u; rb = (uint32_t)(2 * i + 1) & 32767u; }   // reads coalesced, writes scattered
    float4 va = body[2 * ra], wa = body[2 * ra + 1], vb = body[2 * rb], wb = body[2 * rb + 1];
    float acc = va.x + wa.y + vb.z + wb.w;
#pragma unroll
    for (int k = 0; k < NROW; ++k) { acc = acc * 0.999f + r[k].x * va.x + r[k].y * wa.y + r[k].z * vb.z + r[k].w; va.x += acc * 1e-6f; }
    if (STORES) {
#pragma unroll
        for (int k = 2; k < NROW; k += 5) { r[k].w = acc; rows[(size_t)k * rcap + i] = r[k]; }
    }
    va.y = acc; vb.y = acc;
    if (GATHER == 3) {   // push model: coalesced reads happened above from p-indexed slots? emulate: read p-indexed, write scattered
        body[2 * a] = va; body[2 * a + 1] = wa; body[2 * b] = vb; body[2 * b + 1] = wb;
    } else { body[2 * a] = va; body[2 * a + 1] = wa; body[2 * b] = vb; body[2 * b + 1] = wb; }
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 14000, nb = 32768, ncol = 13; const size_t rcap = 16384; const int NROW = 60;
    float4 *body, *rows; uint32_t *idx;
    CK(hipMalloc(&body, (size_t)nb * 2 * 16)); CK(hipMalloc(&rows, rcap * NROW * ncol * 16)); CK(hipMalloc(&idx, (size_t)n * 2 * 4 * ncol));
    CK(hipMemset(body, 0, (size_t)nb * 2 * 16)); CK(hipMemset(rows, 0, rcap * NROW * ncol * 16));
    std::vector<uint32_t> h((size_t)n * 2 * ncol);
    for (int c = 0; c < ncol; ++c) { std::vector<uint32_t> perm(nb); for (int i = 0; i < nb; ++i) perm[i] = i;
        uint64_t st = 88172645463325252ull + c; for (int i = nb - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; std::swap(perm[i], perm[st % (i + 1)]); }
        for (int i = 0; i < 2 * n; ++i) h[(size_t)c * 2 * n + i] = perm[i]; }   // distinct bodies within a colour, like a real colouring
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 50; ++i) launch(i);
        CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
        const int L = 1300;
        for (int i = 0; i < L; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %.2f us/launch\n", name, 1e3 * ms / L); return 0; };
    dim3 g((n + 63) / 64), b(64);
#define L(G, R, S, N) [&](int i) { int c = i % ncol; hipLaunchKernelGGL((k_model<G, R, S, N>), g, b, 0, s, idx + (size_t)c * 2 * n, body, rows + (size_t)c * rcap * NROW, n, rcap, i); }
    run("full: idx-gather random bodies + 60 rows", L(1, true, true, 60));
    run("coalesced bodies (p-indexed R+W) + 60 rows", L(0, true, true, 60));
    run("random bodies, no dependent idx load", L(2, true, true, 60));
    run("push model: coalesced reads, scattered writes", L(3, true, true, 60));
    run("full, no row loads", L(1, false, true, 60));
    run("nothing but coalesced body RMW", L(0, false, false, 60));
    return 0;
}
