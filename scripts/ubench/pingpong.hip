// Micro-benchmark: latency of handing a 16-byte value from one resident wave to another through device memory
// (the hop cost of a dataflow / persistent-kernel solver), for workgroups on the same and on different XCDs, with
// (a) agent-scope atomics on ordinary memory, (b) plain volatile accesses on uncached (fine-grained) memory,
// (c) 16-byte sc1 loads/stores carrying a sequence tag. Also: cost of a grid-wide barrier built from atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)); }

// (a) two blocks bounce a counter; a = block `who0`, b = block `who1`; other blocks exit.
__global__ void k_pp_atomic(uint32_t *fa, uint32_t *fb, int n, int who0, int who1, uint32_t *xcc) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == who0) {
        xcc[0] = xcc_id();
        for (int i = 1; i <= n; ++i) {
            __hip_atomic_store(fa, (uint32_t)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(fb, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
        }
    } else if ((int)blockIdx.x == who1) {
        xcc[1] = xcc_id();
        for (int i = 1; i <= n; ++i) {
            while (__hip_atomic_load(fa, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
            __hip_atomic_store(fb, (uint32_t)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// relaxed variant: no fences, only the scoped accesses
__global__ void k_pp_relaxed(uint32_t *fa, uint32_t *fb, int n, int who0, int who1) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == who0) {
        for (int i = 1; i <= n; ++i) {
            __hip_atomic_store(fa, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
        }
    } else if ((int)blockIdx.x == who1) {
        for (int i = 1; i <= n; ++i) {
            while (__hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
            __hip_atomic_store(fb, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// (b) plain volatile on uncached memory
__global__ void k_pp_volatile(volatile uint32_t *fa, volatile uint32_t *fb, int n, int who0, int who1) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == who0) {
        for (int i = 1; i <= n; ++i) { *fa = i; while (*fb != (uint32_t)i) {} }
    } else if ((int)blockIdx.x == who1) {
        for (int i = 1; i <= n; ++i) { while (*fa != (uint32_t)i) {} *fb = i; }
    }
}
// (c) 16-byte tagged values with sc1 (agent-coherent) loads and stores
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load16_sc1(const float4 *p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store16_sc1(float4 *p, float4 f) {
    v4f v = {f.x, f.y, f.z, f.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__global__ void k_pp_16(float4 *fa, float4 *fb, int n, int who0, int who1) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == who0) {
        for (int i = 1; i <= n; ++i) {
            store16_sc1(fa, make_float4(1.f, 2.f, 3.f, __int_as_float(i)));
            while (__float_as_int(load16_sc1(fb).w) != i) {}
        }
    } else if ((int)blockIdx.x == who1) {
        for (int i = 1; i <= n; ++i) {
            float4 v;
            do { v = load16_sc1(fa); } while (__float_as_int(v.w) != i);
            v.x += 1.f;
            store16_sc1(fb, v);
        }
    }
}
// (d) round 5: would a hand-off that stays inside one XCD be faster if the producer left the line in that XCD's L2? Producer: PLAIN (or sc0)
// 16-byte store instead of the write-through sc1 store; consumer: sc1 (or sc0) polls. A poll that never sees the value gives up after
// `limit` tries and the pair reports the failure (cross-XCD pairs are expected to: a plain store is not visible outside its XCD's L2).
__device__ __forceinline__ void store16_plain(float4 *p, float4 f) {
    v4f v = {f.x, f.y, f.z, f.w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float4 load16_sc0(const float4 *p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int STORE, int LOAD>   // STORE 0 plain, 1 sc1; LOAD 0 sc0, 1 sc1
__global__ void k_pp_16x(float4 *fa, float4 *fb, int n, int who0, int who1, uint32_t limit, uint32_t *fail) {
    if (threadIdx.x != 0) return;
    auto st = [](float4 *p, float4 f) { if (STORE) store16_sc1(p, f); else store16_plain(p, f); };
    auto ld = [](const float4 *p) { return LOAD ? load16_sc1(p) : load16_sc0(p); };
    if ((int)blockIdx.x == who0) {
        for (int i = 1; i <= n; ++i) {
            st(fa, make_float4(1.f, 2.f, 3.f, __int_as_float(i)));
            uint32_t tries = 0;
            while (__float_as_int(ld(fb).w) != i) if (++tries > limit) { fail[0] = (uint32_t)i; return; }
        }
    } else if ((int)blockIdx.x == who1) {
        for (int i = 1; i <= n; ++i) {
            float4 v; uint32_t tries = 0;
            do { v = ld(fa); if (++tries > limit) { fail[1] = (uint32_t)i; return; } } while (__float_as_int(v.w) != i);
            v.x += 1.f;
            st(fb, v);
        }
    }
}
// chain: block k waits for block k-1's tag then publishes its own; measures the per-hop cost over many XCD crossings
__global__ void k_chain_16(float4 *slots, int n_iter, int nblocks) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    for (int it = 1; it <= n_iter; ++it) {
        const int src = b == 0 ? nblocks - 1 : b - 1;
        const int want = b == 0 ? it - 1 : it;
        float4 v = make_float4(0, 0, 0, 0);
        if (!(b == 0 && it == 1)) do { v = load16_sc1(slots + src); } while (__float_as_int(v.w) != want);
        v.x += 1.f; v.w = __int_as_float(it);
        store16_sc1(slots + b, v);
    }
}
// grid barrier: one counter, monotone target
__global__ void k_barrier(uint32_t *counter, int n, uint32_t nblocks, float *sink) {
    float acc = threadIdx.x;
    for (int i = 1; i <= n; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const uint32_t target = (uint32_t)i * nblocks;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
        acc += 1.f;
    }
    if (acc < 0) sink[0] = acc;
}
// two-level barrier: 16 group counters, then one top counter, release through a generation word
__global__ void k_barrier2(uint32_t *grp, uint32_t *top, uint32_t *gen, int n, uint32_t nblocks, float *sink) {
    float acc = threadIdx.x;
    const uint32_t g = blockIdx.x & 15u, per = nblocks / 16u;
    for (int i = 1; i <= n; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const uint32_t old = atomicAdd(grp + 32 * g, 1u);
            if (old + 1 == (uint32_t)i * per) {
                const uint32_t o2 = atomicAdd(top, 1u);
                if (o2 + 1 == (uint32_t)i * 16u) __hip_atomic_store(gen, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)i) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
        acc += 1.f;
    }
    if (acc < 0) sink[0] = acc;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *flags, *uflags, *xcc; float4 *slots; float *sink;
    CK(hipMalloc(&flags, 1 << 16)); CK(hipMalloc(&xcc, 64)); CK(hipMalloc(&slots, 1 << 20)); CK(hipMalloc(&sink, 64));
    CK(hipExtMallocWithFlags((void **)&uflags, 1 << 16, hipDeviceMallocUncached));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 20000;
    auto timeit = [&](auto launch) -> float {
        CK(hipMemsetAsync(flags, 0, 1 << 16, s)); CK(hipMemsetAsync(uflags, 0, 1 << 16, s)); CK(hipMemsetAsync(slots, 0, 1 << 20, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
    };
    for (int other : {1, 8, 4, 3}) {
        uint32_t hx[2] = {99, 99};
        float ms = timeit([&] { hipLaunchKernelGGL(k_pp_atomic, dim3(16), dim3(64), 0, s, flags, flags + 64, N, 0, other, xcc); return 0; });
        CK(hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost));
        printf("blocks 0<->%d (xcc %u,%u): acq/rel atomics       %.0f ns/hop\n", other, hx[0], hx[1], 1e6 * ms / N / 2);
        ms = timeit([&] { hipLaunchKernelGGL(k_pp_relaxed, dim3(16), dim3(64), 0, s, flags, flags + 64, N, 0, other); return 0; });
        printf("blocks 0<->%d             : relaxed agent atomics %.0f ns/hop\n", other, 1e6 * ms / N / 2);
        ms = timeit([&] { hipLaunchKernelGGL(k_pp_volatile, dim3(16), dim3(64), 0, s, uflags, uflags + 64, N, 0, other); return 0; });
        printf("blocks 0<->%d             : volatile on uncached  %.0f ns/hop\n", other, 1e6 * ms / N / 2);
        ms = timeit([&] { hipLaunchKernelGGL(k_pp_16, dim3(16), dim3(64), 0, s, slots, slots + 16, N, 0, other); return 0; });
        printf("blocks 0<->%d             : 16-B tagged sc1       %.0f ns/hop\n", other, 1e6 * ms / N / 2);
    }
    {   // (d): same-XCD pair (blocks 0 and 8) and cross-XCD pair (0 and 1), every store / load flavour, with a give-up limit
        uint32_t *fail; CK(hipMalloc(&fail, 64));
        for (int other : {8, 1}) {
            auto run = [&](const char *what, auto kern) {
                CK(hipMemset(fail, 0, 64));
                float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(16), dim3(64), 0, s, slots, slots + 16, N, 0, other, 200000u, fail); return 0; });
                uint32_t hf[2]; CK(hipMemcpy(hf, fail, 8, hipMemcpyDeviceToHost));
                if (hf[0] || hf[1]) printf("blocks 0<->%d: %-28s NOT VISIBLE (gave up at hand-off %u / %u)\n", other, what, hf[0], hf[1]);
                else printf("blocks 0<->%d: %-28s %.0f ns/hop\n", other, what, 1e6 * ms / N / 2);
                return 0;
            };
            run("sc1 store, sc1 poll", k_pp_16x<1, 1>);
            run("plain store, sc1 poll", k_pp_16x<0, 1>);
            run("plain store, sc0 poll", k_pp_16x<0, 0>);
            run("sc1 store, sc0 poll", k_pp_16x<1, 0>);
        }
    }
    for (int nb : {8, 64, 256}) {
        const int it = 2000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_chain_16, dim3(nb), dim3(64), 0, s, slots, it, nb); return 0; });
        printf("chain over %3d blocks: %.0f ns/hop\n", nb, 1e6 * ms / ((double)it * nb));
    }
    for (int nb : {64, 128, 256, 512, 1024}) {
        const int it = 2000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_barrier, dim3(nb), dim3(256), 0, s, flags, it, (uint32_t)nb, sink); return 0; });
        printf("grid barrier, %4d blocks x 256: flat %.2f us", nb, 1e3 * ms / it);
        ms = timeit([&] { hipLaunchKernelGGL(k_barrier2, dim3(nb), dim3(256), 0, s, flags, flags + 1024, flags + 2048, it, (uint32_t)nb, sink); return 0; });
        printf("   two-level %.2f us\n", 1e3 * ms / it);
    }
    return 0;
}
