// C-ABI implementation (include/edynhip.h): context lifetime, scene upload, step orchestration,
// state read-back. Host-side logic only; every simulation stage runs in the HIP kernels of
// broadphase.hip / narrowphase.hip / solver.hip. There is no CPU fallback.
#include "ctx.hpp"
#include "dcollide.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace eh {
using namespace dm;

static std::string g_create_error;

int set_error(edynhip_ctx *c, int code, const char *what, hipError_t e) {
    std::string msg = what ? what : "";
    if (e != hipSuccess) { msg += ": "; msg += hipGetErrorString(e); }
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// Host <- device counters without a stream synchronisation: a one-workgroup kernel copies the counter block into pinned
// host memory, fences, and then publishes a sequence number; the host spins on that number. hipStreamSynchronize wakes
// the host through an interrupt (~25-50 us of idle GPU per sync, scripts/prof_timeline.py); the spin sees the write
// within a few microseconds. If the number does not arrive (a kernel faulted), a real synchronise reports the error.
__global__ void k_publish_counters(const uint32_t *__restrict__ src, uint32_t *dst, uint32_t words, volatile uint32_t *seq, uint32_t value) {
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store((uint32_t *)seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int fetch_counters(edynhip_ctx *c, size_t bytes) {
    const uint32_t value = ++c->cnt_seq_next;
    hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(256), 0, c->stream, (const uint32_t *)c->cnt, (uint32_t *)c->cnt_host,
                       (uint32_t)(bytes / sizeof(uint32_t)), c->cnt_seq, value);
    EH_HIP(c, hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *c->cnt_seq != value; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            EH_HIP(c, hipStreamSynchronize(c->stream));   // surfaces a device fault; otherwise the counters are there now
            if (*c->cnt_seq != value) return set_error(c, EDYNHIP_ERR_INTERNAL, "fetch_counters: the published sequence number never arrived");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return EDYNHIP_OK;
}

template <typename T>
static int dalloc(edynhip_ctx *c, T *&p, size_t count) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    EH_HIP(c, hipMalloc(&q, bytes));
    EH_HIP(c, hipMemsetAsync(q, 0, bytes, c->stream));
    c->allocs.push_back(q);
    p = (T *)q;
    return EDYNHIP_OK;
}

static int alloc_manifolds(edynhip_ctx *c, Manifolds &m, uint32_t cap, uint32_t nb) {
    m.cap = cap;
    EH_TRY(dalloc(c, m.seg_start, nb)); EH_TRY(dalloc(c, m.seg_end, nb)); EH_TRY(dalloc(c, m.prev_idx, cap));
    EH_TRY(dalloc(c, m.skey, cap)); EH_TRY(dalloc(c, m.bodyA, cap)); EH_TRY(dalloc(c, m.bodyB, cap)); EH_TRY(dalloc(c, m.info, cap));
    EH_TRY(dalloc(c, m.pA, (size_t)cap * kMaxPts)); EH_TRY(dalloc(c, m.pB, (size_t)cap * kMaxPts));
    EH_TRY(dalloc(c, m.nrm, (size_t)cap * kMaxPts)); EH_TRY(dalloc(c, m.lnrm, (size_t)cap * kMaxPts));
    EH_TRY(dalloc(c, m.imp, (size_t)cap * kMaxPts));
    return EDYNHIP_OK;
}

static int allocate(edynhip_ctx *c) {
    const uint32_t nb = c->cfg.max_bodies, M = c->cfg.max_manifolds, nj = c->cfg.max_joints;
    Bodies &b = c->b;
    b.cap = nb;
    EH_TRY(dalloc(c, b.xf, (size_t)nb * 8)); EH_TRY(dalloc(c, b.dvw, (size_t)nb * 2)); EH_TRY(dalloc(c, b.linvel, nb)); EH_TRY(dalloc(c, b.angvel, nb));
    EH_TRY(dalloc(c, b.amin, nb)); EH_TRY(dalloc(c, b.amax, nb)); EH_TRY(dalloc(c, b.shape, nb)); EH_TRY(dalloc(c, b.grav, nb));
    EH_TRY(dalloc(c, b.mat, nb)); EH_TRY(dalloc(c, b.flags, nb)); EH_TRY(dalloc(c, b.group, nb)); EH_TRY(dalloc(c, b.mask, nb));
    EH_TRY(dalloc(c, b.island, nb));
    EH_TRY(alloc_manifolds(c, c->m[0], M, nb));
    EH_TRY(alloc_manifolds(c, c->m[1], M, nb));
    Rows &r = c->rows;
    EH_TRY(dalloc(c, r.order, M)); EH_TRY(dalloc(c, r.bA, M)); EH_TRY(dalloc(c, r.bB, M)); EH_TRY(dalloc(c, r.np, M)); EH_TRY(dalloc(c, r.label, M));
    EH_TRY(dalloc(c, r.rw, (size_t)M * kMaxPts * kRowsPerPoint * kRowF));
    EH_TRY(dalloc(c, r.dslot, ((size_t)M + 64) * 4));   // whole 64-lane blocks (dslot_at)
    EH_TRY(dalloc(c, r.pslot, ((size_t)M + 32) * 6));   // whole 32-lane blocks (pslot_at)
    EH_TRY(dalloc(c, r.next, (size_t)M * 2)); EH_TRY(dalloc(c, r.im, (size_t)M * 2));
    EH_TRY(dalloc(c, r.slot_of, (size_t)nb * kMaxColours)); EH_TRY(dalloc(c, r.first_slot, nb));
    LBVH &t = c->bvh;
    EH_TRY(dalloc(c, t.keys, nb)); EH_TRY(dalloc(c, t.keys_sorted, nb));
    EH_TRY(dalloc(c, t.parent, (size_t)2 * nb)); EH_TRY(dalloc(c, t.left, nb)); EH_TRY(dalloc(c, t.right, nb));
    EH_TRY(dalloc(c, t.nmin, (size_t)2 * nb)); EH_TRY(dalloc(c, t.nmax, (size_t)2 * nb)); EH_TRY(dalloc(c, t.visit, nb));
    EH_TRY(dalloc(c, t.np_list, nb));
    EH_TRY(dalloc(c, c->pair_keys, M)); EH_TRY(dalloc(c, c->pair_keys_sorted, M)); EH_TRY(dalloc(c, c->new_edges, M));
    EH_TRY(dalloc(c, c->own_keys, (size_t)nb * 32)); EH_TRY(dalloc(c, c->own_count, (size_t)nb + 1)); EH_TRY(dalloc(c, c->own_offset, (size_t)nb + 1));
    EH_TRY(dalloc(c, c->col_keys, M)); EH_TRY(dalloc(c, c->col_keys_sorted, M));
    EH_TRY(dalloc(c, c->col_unc, kColUncCap));
    { const size_t cs = 256 * (((size_t)M + 1023) / 1024) + 1; EH_TRY(dalloc(c, c->cs_hist, cs)); EH_TRY(dalloc(c, c->cs_start, cs)); }
    EH_TRY(dalloc(c, c->used, nb)); EH_TRY(dalloc(c, c->best[0], nb)); EH_TRY(dalloc(c, c->best[1], nb));
    EH_TRY(dalloc(c, c->isl_err, nb)); EH_TRY(dalloc(c, c->isl_done, nb));
    EH_TRY(dalloc(c, c->state_dev, (size_t)nb * 13));
    EH_HIP(c, hipHostMalloc((void **)&c->state_host, (size_t)nb * 13 * sizeof(float), hipHostMallocDefault));
    EH_TRY(dalloc(c, c->sleep_state, nb)); EH_TRY(dalloc(c, c->sleep_action, nb)); EH_TRY(dalloc(c, c->sleep_since, nb));
    EH_HIP(c, hipMemsetAsync(c->sleep_since, 0xFF, (size_t)nb * sizeof(int32_t), c->stream));   // -1: no timer running
    Joints &j = c->j;
    j.cap = nj;
    EH_TRY(dalloc(c, j.orig, nj)); EH_TRY(dalloc(c, j.type, nj)); EH_TRY(dalloc(c, j.bodyA, nj)); EH_TRY(dalloc(c, j.bodyB, nj));
    EH_TRY(dalloc(c, j.pivA, nj)); EH_TRY(dalloc(c, j.pivB, nj)); EH_TRY(dalloc(c, j.axA, nj)); EH_TRY(dalloc(c, j.pA, nj));
    EH_TRY(dalloc(c, j.qA, nj)); EH_TRY(dalloc(c, j.axB, nj)); EH_TRY(dalloc(c, j.impulse, (size_t)nj * 5));
    EH_TRY(dalloc(c, j.rA, nj)); EH_TRY(dalloc(c, j.rB, nj)); EH_TRY(dalloc(c, j.wp, nj)); EH_TRY(dalloc(c, j.wq, nj));
    EH_TRY(dalloc(c, j.eff, (size_t)nj * 5)); EH_TRY(dalloc(c, j.rhs, (size_t)nj * 5));
    c->sort_tmp_bytes = sort_temp_bytes(std::max(std::max(M, nb + 1), 256u * ((M + 1023u) / 1024u) + 1u));
    { void *q = nullptr; EH_HIP(c, hipMalloc(&q, c->sort_tmp_bytes)); c->allocs.push_back(q); c->sort_tmp = q; }
    EH_TRY(dalloc(c, c->cnt, 1));
    EH_HIP(c, hipHostMalloc((void **)&c->cnt_host, sizeof(Counters), hipHostMallocDefault));
    std::memset(c->cnt_host, 0, sizeof(Counters));
    { void *q = nullptr; EH_HIP(c, hipHostMalloc(&q, 64, hipHostMallocDefault)); std::memset(q, 0, 64); c->cnt_seq = (volatile uint32_t *)q; }
    EH_HIP(c, hipStreamSynchronize(c->stream));
    return EDYNHIP_OK;
}

// Scene upload: raw packed arrays -> float4 SoA + derived quantities (rigidbody.cpp:47-131).
struct RawBodies {
    const int32_t *kind; const float *pos, *orn, *linvel, *angvel, *mass, *inertia; const uint8_t *has_inertia;
    const int32_t *shape_type; const float *shape_param, *friction, *restitution; const uint64_t *group, *mask; const float *gravity; const uint8_t *sleeping_disabled;
};
__global__ void k_init_bodies(uint32_t first, uint32_t n, RawBodies r, Bodies b, float3 default_gravity) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;   // index into the caller's arrays
    if (l >= n) return;
    const uint32_t i = first + l;                                // body index
    const int kind = r.kind[l], st = r.shape_type[l];
    const f3 pos = mk3(r.pos[3 * l], r.pos[3 * l + 1], r.pos[3 * l + 2]);
    const q4 orn{r.orn[4 * l], r.orn[4 * l + 1], r.orn[4 * l + 2], r.orn[4 * l + 3]};
    const float4 sp = make_float4(r.shape_param[4 * l], r.shape_param[4 * l + 1], r.shape_param[4 * l + 2], r.shape_param[4 * l + 3]);
    float inv_m = 0;
    m3 il = m3_zero(), iw = m3_zero();
    if (kind == EDYNHIP_KIND_DYNAMIC) {
        const float mass = r.mass[l];
        inv_m = 1.0f / mass;
        m3 I;
        if (r.has_inertia && r.has_inertia[l]) {
            const float *p = r.inertia + 9 * l;
            I = {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}};
        } else if (st == dc::SHAPE_BOX) {   // moment_of_inertia.cpp:11-17,179-181
            f3 ext = from4(sp) * 2.0f;
            f3 d = 1.0f / 12.0f * mass * mk3(ext.y * ext.y + ext.z * ext.z, ext.z * ext.z + ext.x * ext.x, ext.x * ext.x + ext.y * ext.y);
            I = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}};
        } else if (st == dc::SHAPE_SPHERE) {   // :19-21,163-165
            float s = 0.4f * mass * sp.x * sp.x;
            I = {{1 * s, 0 * s, 0 * s}, {0 * s, 1 * s, 0 * s}, {0 * s, 0 * s, 1 * s}};
        } else {
            I = {{kScalarMax, 0, 0}, {0, kScalarMax, 0}, {0, 0, kScalarMax}};
        }
        // inverse_matrix_symmetric, matrix3x3.hpp:190-218
        float det = dot(I.r0, cross(I.r1, I.r2));
        float di = 1.0f / det;
        float a11 = I.r0.x, a12 = I.r0.y, a13 = I.r0.z, a22 = I.r1.y, a23 = I.r1.z, a33 = I.r2.z;
        il.r0.x = di * (a22 * a33 - a23 * a23);
        il.r0.y = di * (a13 * a23 - a12 * a33);
        il.r0.z = di * (a12 * a23 - a13 * a22);
        il.r1.x = il.r0.y;
        il.r1.y = di * (a11 * a33 - a13 * a13);
        il.r1.z = di * (a12 * a13 - a11 * a23);
        il.r2.x = il.r0.z;
        il.r2.y = il.r1.z;
        il.r2.z = di * (a11 * a22 - a12 * a12);
        m3 basis = to_m3(orn);
        iw = mul(mul(basis, il), transpose(basis));
    }
    B_POS(b, i) = to4(pos, inv_m);
    B_ORN(b, i) = to4(orn);
    const bool moving = kind != EDYNHIP_KIND_STATIC;
    b.linvel[i] = moving ? make_float4(r.linvel[3 * l], r.linvel[3 * l + 1], r.linvel[3 * l + 2], 0) : make_float4(0, 0, 0, 0);
    b.angvel[i] = moving ? make_float4(r.angvel[3 * l], r.angvel[3 * l + 1], r.angvel[3 * l + 2], 0) : make_float4(0, 0, 0, 0);
    B_DV(b, i) = make_float4(0, 0, 0, 0); B_DW(b, i) = make_float4(0, 0, 0, 0);
    B_IW(b, i, 0) = to4(iw.r0, 0); B_IW(b, i, 1) = to4(iw.r1, 0); B_IW(b, i, 2) = to4(iw.r2, 0);
    B_IL(b, i, 0) = to4(il.r0, 0); B_IL(b, i, 1) = to4(il.r1, 0); B_IL(b, i, 2) = to4(il.r2, 0);
    b.shape[i] = sp;
    f3 g = r.gravity ? mk3(r.gravity[3 * l], r.gravity[3 * l + 1], r.gravity[3 * l + 2]) : mk3(default_gravity.x, default_gravity.y, default_gravity.z);
    b.grav[i] = kind == EDYNHIP_KIND_DYNAMIC ? to4(g, 0) : make_float4(0, 0, 0, 0);
    b.mat[i] = make_float2(r.friction[l], r.restitution[l]);
    b.flags[i] = (uint32_t)kind | ((uint32_t)st << BF_SHAPE_SHIFT) | ((r.sleeping_disabled && r.sleeping_disabled[l]) ? BF_NOSLEEP : 0u);
    b.group[i] = r.group ? r.group[l] : ~0ull;
    b.mask[i] = r.mask ? r.mask[l] : ~0ull;
    b.island[i] = i;
    // shape_aabb (aabb_util.cpp:11-70)
    f3 mn = pos, mx = pos;
    if (st == dc::SHAPE_BOX) {
        const m3 basis = to_m3(orn);
        const f3 h = from4(sp);
        float lo[3] = {pos.x, pos.y, pos.z}, hi[3] = {pos.x, pos.y, pos.z};
        const f3 rws[3] = {basis.r0, basis.r1, basis.r2};
        for (int rr = 0; rr < 3; ++rr)
            for (int cc = 0; cc < 3; ++cc) {
                float e = comp(rws[rr], cc) * -comp(h, cc);
                float f = -e;
                if (e < f) { lo[rr] += e; hi[rr] += f; } else { lo[rr] += f; hi[rr] += e; }
            }
        mn = mk3(lo[0], lo[1], lo[2]); mx = mk3(hi[0], hi[1], hi[2]);
    } else if (st == dc::SHAPE_SPHERE) {
        mn = mk3(pos.x - sp.x, pos.y - sp.x, pos.z - sp.x); mx = mk3(pos.x + sp.x, pos.y + sp.x, pos.z + sp.x);
    } else if (st == dc::SHAPE_PLANE) {
        const f3 nrm = from4(sp);
        f3 umin = mk3(-1, -1, -1), umax = mk3(1, 1, 1);
        if (eq(nrm, mk3(1, 0, 0))) umax = mk3(0, 1, 1);
        else if (eq(nrm, mk3(-1, 0, 0))) umin = mk3(0, -1, -1);
        else if (eq(nrm, mk3(0, 1, 0))) umax = mk3(1, 0, 1);
        else if (eq(nrm, mk3(0, -1, 0))) umin = mk3(-1, 0, -1);
        else if (eq(nrm, mk3(0, 0, 1))) umax = mk3(1, 1, 0);
        else if (eq(nrm, mk3(0, 0, -1))) umin = mk3(-1, -1, 0);
        const f3 pw = nrm * sp.w;
        mn = umin * 99999.0f + pw; mx = umax * 99999.0f + pw;
    }
    b.amin[i] = to4(mn, 0); b.amax[i] = to4(mx, 0);
}

template <typename T>
static int upload(edynhip_ctx *c, const T *host, size_t count, const T *&dev, std::vector<void *> &tmp) {
    dev = nullptr;
    if (!host || count == 0) return EDYNHIP_OK;
    void *q = nullptr;
    EH_HIP(c, hipMalloc(&q, count * sizeof(T)));
    tmp.push_back(q);
    EH_HIP(c, hipMemcpyAsync(q, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    dev = (const T *)q;
    return EDYNHIP_OK;
}

__global__ void k_manifolds_to_records(uint32_t M, Manifolds mf, edynhip_manifold *out) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    edynhip_manifold r;
    memset(&r, 0, sizeof(r));
    r.body[0] = mf.bodyA[m]; r.body[1] = mf.bodyB[m];
    const uint32_t info = mf.info[m];
    r.num_points = info & 0xFF; r.colour = info >> 8;
    for (uint32_t k = 0; k < r.num_points; ++k) {
        const size_t s = (size_t)k * mf.cap + m;
        float4 a = mf.pA[s], b = mf.pB[s], n = mf.nrm[s], l = mf.lnrm[s], im = mf.imp[s];
        edynhip_point &p = r.pt[k];
        p.pivotA[0] = a.x; p.pivotA[1] = a.y; p.pivotA[2] = a.z; p.distance = a.w;
        p.pivotB[0] = b.x; p.pivotB[1] = b.y; p.pivotB[2] = b.z; p.friction = b.w;
        p.normal[0] = n.x; p.normal[1] = n.y; p.normal[2] = n.z; p.attachment = __float_as_int(n.w);
        p.local_normal[0] = l.x; p.local_normal[1] = l.y; p.local_normal[2] = l.z; p.restitution = l.w;
        p.normal_impulse = im.x; p.friction_impulse[0] = im.y; p.friction_impulse[1] = im.z; p.lifetime = __float_as_uint(im.w);
    }
    out[m] = r;
}
// Owner of a pair (see broadphase.hip): the procedural body, the higher index if both are.
__host__ __device__ inline uint32_t pair_owner(uint32_t a, uint32_t b, bool proc_a, bool proc_b) {
    if (proc_a && proc_b) return a > b ? a : b;
    return proc_a ? a : b;
}
__global__ void k_records_to_manifolds(uint32_t M, const edynhip_manifold *in, Manifolds mf, const uint32_t *__restrict__ flags) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const edynhip_manifold r = in[m];
    const uint32_t a = r.body[0], b = r.body[1];
    auto owner_of = [&](const edynhip_manifold &x) { return pair_owner(x.body[0], x.body[1], (flags[x.body[0]] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC, (flags[x.body[1]] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC); };
    const uint32_t hi = owner_of(r), lo = hi == a ? b : a;
    mf.skey[m] = ((((uint64_t)hi << 32) | lo) << 1) | (a == lo ? 1u : 0u);
    mf.bodyA[m] = a; mf.bodyB[m] = b;
    {
        const uint32_t ph = m > 0 ? owner_of(in[m - 1]) : 0xFFFFFFFFu;
        const uint32_t nh = m + 1 < M ? owner_of(in[m + 1]) : 0xFFFFFFFFu;
        if (ph != hi) mf.seg_start[hi] = m;
        if (nh != hi) mf.seg_end[hi] = m + 1;
    }
    mf.info[m] = (r.num_points & 0xFF) | ((r.colour & 0xFF) << 8);
    for (uint32_t k = 0; k < r.num_points; ++k) {
        const size_t s = (size_t)k * mf.cap + m;
        const edynhip_point &p = r.pt[k];
        mf.pA[s] = make_float4(p.pivotA[0], p.pivotA[1], p.pivotA[2], p.distance);
        mf.pB[s] = make_float4(p.pivotB[0], p.pivotB[1], p.pivotB[2], p.friction);
        mf.nrm[s] = make_float4(p.normal[0], p.normal[1], p.normal[2], __int_as_float(p.attachment));
        mf.lnrm[s] = make_float4(p.local_normal[0], p.local_normal[1], p.local_normal[2], p.restitution);
        mf.imp[s] = make_float4(p.normal_impulse, p.friction_impulse[0], p.friction_impulse[1], __uint_as_float(p.lifetime));
    }
}
__global__ void k_pack_state(uint32_t first, uint32_t count, Bodies b, float *dst) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t i = first + t;
    float4 p = B_POS(b, i), q = B_ORN(b, i), v = b.linvel[i], w = b.angvel[i];
    float *o = dst + (size_t)t * 13;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
}
__global__ void k_unpack_state(uint32_t n, const float *src, Bodies b) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *o = src + (size_t)i * 13;
    B_POS(b, i) = make_float4(o[0], o[1], o[2], B_POS(b, i).w);
    B_ORN(b, i) = make_float4(o[3], o[4], o[5], o[6]);
    b.linvel[i] = make_float4(o[7], o[8], o[9], 0);
    b.angvel[i] = make_float4(o[10], o[11], o[12], 0);
}
__global__ void k_pack_derived(uint32_t n, Bodies b, float *aabb, float *iw) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = b.amin[i], c = b.amax[i];
    aabb[6 * i] = a.x; aabb[6 * i + 1] = a.y; aabb[6 * i + 2] = a.z; aabb[6 * i + 3] = c.x; aabb[6 * i + 4] = c.y; aabb[6 * i + 5] = c.z;
    for (int r = 0; r < 3; ++r) { float4 x = B_IW(b, i, r); iw[9 * i + 3 * r] = x.x; iw[9 * i + 3 * r + 1] = x.y; iw[9 * i + 3 * r + 2] = x.z; }
}

constexpr uint32_t kMaxTimedSteps = 4096;
static int begin_timed_step(edynhip_ctx *c) {
    StageTimer &t = c->timer;
    t.e = nullptr;
    if (!(c->cfg.flags & (EDYNHIP_FLAG_TIMING | EDYNHIP_FLAG_TIMING_SOLVE)) || t.recorded >= kMaxTimedSteps) return EDYNHIP_OK;
    t.mask = (c->cfg.flags & EDYNHIP_FLAG_TIMING) ? 0x7FFu : ((1u << 5) | (1u << 6));
    if (t.recorded >= t.capacity) {
        for (int k = 0; k < StageTimer::kEvents; ++k) { hipEvent_t e; EH_HIP(c, hipEventCreate(&e)); t.ev.push_back(e); }
        t.capacity += 1;
    }
    t.e = &t.ev[(size_t)t.recorded * StageTimer::kEvents];
    return EDYNHIP_OK;
}
static void resolve_timings(edynhip_ctx *c) {
    StageTimer &tm = c->timer;
    edynhip_timings &t = c->timings;
    const uint32_t launches = t.solve_velocity_launches;
    t = edynhip_timings{};
    t.solve_velocity_launches = launches;
    if (tm.recorded == 0) return;
    const bool all = tm.mask == 0x7FFu;
    (void)hipEventSynchronize(tm.ev[(size_t)(tm.recorded - 1) * StageTimer::kEvents + (all ? 10 : 6)]);
    for (uint32_t s = 0; s < tm.recorded; ++s) {
        hipEvent_t *e = &tm.ev[(size_t)s * StageTimer::kEvents];
        auto el = [&](int a, int b) { float ms = 0; if (all || (a == 5 && b == 6)) (void)hipEventElapsedTime(&ms, e[a], e[b]); return ms; };
        t.broadphase_ms += el(0, 1); t.narrowphase_ms += el(1, 2); t.islands_ms += el(2, 3); t.colouring_ms += el(3, 4);
        t.prepare_ms += el(4, 5); t.solve_velocity_ms += el(5, 6); t.integrate_ms += el(6, 7); t.solve_position_ms += el(7, 8);
        t.finish_ms += el(8, 9); t.step_ms += el(0, 10);
    }
    t.steps = tm.recorded;
}

static int run_stages(edynhip_ctx *c, uint32_t mask) {
    c->timer.e = nullptr;
    c->full_step = mask == EDYNHIP_STAGE_ALL && c->clears_primed;
    if (mask == EDYNHIP_STAGE_ALL) EH_TRY(begin_timed_step(c));
    else c->force_islands = true;   // partial runs (tests) never rely on a previous step's labels
    auto rec = [&](int i) { if (c->timer.e && ((c->timer.mask >> i) & 1u)) (void)hipEventRecord(c->timer.e[i], c->stream); };
    // A stage that fails (capacity, colour limit, device-side invariant) leaves the step half done: k_finish did not run, so
    // nothing pre-cleared the next step's scratch and the island labels are stale. The next call must start from the
    // stand-alone path (memsets + k_step_reset, which also clears the sticky error counters) and relabel the islands.
    auto guarded = [&](int rc) {
        if (rc != EDYNHIP_OK) { c->clears_primed = false; c->full_step = false; c->force_islands = true; c->timer.e = nullptr; }
        return rc;
    };
    rec(0);
    if (mask & EDYNHIP_STAGE_BROADPHASE) EH_TRY(guarded(broadphase(c)));
    rec(1);
    if (mask & EDYNHIP_STAGE_NARROWPHASE) EH_TRY(guarded(narrowphase(c)));
    rec(2);
    if (mask & EDYNHIP_STAGE_ISLANDS) EH_TRY(guarded(islands(c)));
    if (mask & EDYNHIP_STAGE_SOLVE) EH_TRY(guarded(solve(c)));   // records events 3..9
    rec(10);
    c->clears_primed = (mask & EDYNHIP_STAGE_SOLVE) != 0;   // k_finish left the next step's scratch cleared
    if (c->timer.e) { c->timer.recorded += 1; c->timer.e = nullptr; }
    return EDYNHIP_OK;
}

}  // namespace eh

using namespace eh;

extern "C" {

uint32_t edynhip_abi_version(void) { return 2; }   // 2: edynhip_bodies.sleeping_disabled, edynhip_add_bodies, sleeping entry points

const char *edynhip_last_error(const edynhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

edynhip_ctx *edynhip_create(const edynhip_config *cfg, int *status_out) {
    auto fail = [&](int code, const char *msg, hipError_t e = hipSuccess) -> edynhip_ctx * {
        set_error(nullptr, code, msg, e);
        if (status_out) *status_out = code;
        return nullptr;
    };
    if (!cfg || cfg->max_bodies == 0) return fail(EDYNHIP_ERR_INVALID, "edynhip_create: bad config");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(EDYNHIP_ERR_NO_DEVICE, "edynhip_create: no HIP device (this library has no CPU fallback)", e);
    if (cfg->device < 0 || cfg->device >= ndev) return fail(EDYNHIP_ERR_INVALID, "edynhip_create: device ordinal out of range");
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(EDYNHIP_ERR_HIP, "hipSetDevice", e);
    edynhip_ctx *c = new edynhip_ctx();
    c->cfg = *cfg;
    c->sleeping = (cfg->flags & EDYNHIP_FLAG_SLEEPING) != 0;
    c->device = cfg->device;
    if (c->cfg.max_manifolds == 0) c->cfg.max_manifolds = 16 * c->cfg.max_bodies + 1024;
    if (c->cfg.fixed_dt <= 0) c->cfg.fixed_dt = 1.0f / 60.0f;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(EDYNHIP_ERR_HIP, "hipStreamCreate", e); }
    int rc = allocate(c);
    if (rc != EDYNHIP_OK) {
        g_create_error = c->err;
        if (status_out) *status_out = rc;
        edynhip_destroy(c);
        return nullptr;
    }
    if (status_out) *status_out = EDYNHIP_OK;
    return c;
}

void edynhip_destroy(edynhip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->cnt_host) (void)hipHostFree(c->cnt_host);
    if (c->cnt_seq) (void)hipHostFree((void *)c->cnt_seq);
    if (c->state_host) (void)hipHostFree(c->state_host);
    for (auto &e : c->timer.ev) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int edynhip_set_stream(edynhip_ctx *c, void *hip_stream) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else { EH_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    return EDYNHIP_OK;
}

int edynhip_synchronize(edynhip_ctx *c) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipStreamSynchronize(c->stream));
    return EDYNHIP_OK;
}

static int load_bodies(edynhip_ctx *c, uint32_t first, uint32_t n, const edynhip_bodies *in, const char *who) {
    if (!c || !in) return EDYNHIP_ERR_INVALID;
    if ((uint64_t)first + n > c->b.cap) return set_error(c, EDYNHIP_ERR_CAPACITY, (std::string(who) + ": more than max_bodies").c_str());
    if (n && (!in->kind || !in->pos || !in->orn || !in->linvel || !in->angvel || !in->mass || !in->shape_type || !in->shape_param ||
              !in->friction || !in->restitution))
        return set_error(c, EDYNHIP_ERR_INVALID, (std::string(who) + ": missing array").c_str());
    for (uint32_t i = 0; i < n; ++i)
        if (in->restitution[i] != 0.0f)
            return set_error(c, EDYNHIP_ERR_UNSUPPORTED, (std::string(who) + ": restitution > 0 needs the restitution solver (out of scope)").c_str());
    EH_HIP(c, hipSetDevice(c->device));
    std::vector<void *> tmp;
    RawBodies r{};
    int rc = EDYNHIP_OK;
    auto up = [&](auto host, size_t count, auto &dev) { if (rc == EDYNHIP_OK) rc = upload(c, host, count, dev, tmp); };
    up(in->kind, n, r.kind); up(in->pos, (size_t)n * 3, r.pos); up(in->orn, (size_t)n * 4, r.orn);
    up(in->linvel, (size_t)n * 3, r.linvel); up(in->angvel, (size_t)n * 3, r.angvel); up(in->mass, n, r.mass);
    up(in->inertia, in->has_inertia ? (size_t)n * 9 : 0, r.inertia); up(in->has_inertia, in->inertia ? n : 0, r.has_inertia);
    up(in->shape_type, n, r.shape_type); up(in->shape_param, (size_t)n * 4, r.shape_param);
    up(in->friction, n, r.friction); up(in->restitution, n, r.restitution);
    up(in->group, n, r.group); up(in->mask, n, r.mask); up(in->gravity, (size_t)n * 3, r.gravity);
    up(in->sleeping_disabled, n, r.sleeping_disabled);
    const uint32_t total = first + n;
    if (rc == EDYNHIP_OK) {
        c->b.n = total;
        if (n)
            hipLaunchKernelGGL(k_init_bodies, dim3((n + 255) / 256), dim3(256), 0, c->stream, first, n, r, c->b,
                               make_float3(c->cfg.gravity[0], c->cfg.gravity[1], c->cfg.gravity[2]));
        c->host_kind.resize(first); c->host_shape.resize(first);
        c->host_kind.insert(c->host_kind.end(), in->kind, in->kind + n);
        c->host_shape.insert(c->host_shape.end(), in->shape_type, in->shape_type + n);
    }
    // broadphase participants: [shaped non-procedural ..., shaped procedural ...]
    std::vector<uint32_t> np_list, proc_list;
    if (rc == EDYNHIP_OK)
        for (uint32_t i = 0; i < total; ++i) {
            if (c->host_shape[i] == EDYNHIP_SHAPE_NONE) continue;
            (c->host_kind[i] == EDYNHIP_KIND_DYNAMIC ? proc_list : np_list).push_back(i);
        }
    // only procedural shaped bodies own pairs; everybody else's count must read 0 in the scan
    if (rc == EDYNHIP_OK && hipMemsetAsync(c->own_count, 0, ((size_t)c->b.cap + 1) * sizeof(uint32_t), c->stream) != hipSuccess)
        rc = set_error(c, EDYNHIP_ERR_HIP, "clear pair counts");
    c->all_asleep = false;
    c->bvh.age = 0;   // the tree topology is rebuilt on the next step
    c->bvh.num_np = (uint32_t)np_list.size();
    c->bvh.num_proc = (uint32_t)proc_list.size();
    np_list.insert(np_list.end(), proc_list.begin(), proc_list.end());
    if (rc == EDYNHIP_OK && !np_list.empty())
        rc = hipMemcpyAsync(c->bvh.np_list, np_list.data(), np_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess
                 ? EDYNHIP_OK : set_error(c, EDYNHIP_ERR_HIP, "upload broadphase lists");
    (void)hipStreamSynchronize(c->stream);
    for (void *p : tmp) (void)hipFree(p);
    if (first == 0) {   // a new world: no manifolds, no running sleep timers (appended bodies keep every index stable instead)
        c->num_manifolds = 0;
        c->prev_num_manifolds = 0;
        c->step_index = 0;
        c->num_colours = 0;
        (void)hipMemsetAsync(c->sleep_since, 0xFF, (size_t)c->b.cap * sizeof(int32_t), c->stream);
        (void)hipMemsetAsync(c->sleep_state, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream);
        (void)hipMemsetAsync(c->sleep_action, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream);
        (void)hipStreamSynchronize(c->stream);
    }
    c->force_islands = true;
    c->all_asleep = false;
    c->clears_primed = false;
    c->stats.num_bodies = total;
    if (rc == EDYNHIP_OK) EH_HIP(c, hipGetLastError());
    return rc;
}

int edynhip_set_bodies(edynhip_ctx *c, uint32_t n, const edynhip_bodies *in) { return load_bodies(c, 0, n, in, "edynhip_set_bodies"); }

int edynhip_add_bodies(edynhip_ctx *c, uint32_t n, const edynhip_bodies *in) {
    if (!c) return EDYNHIP_ERR_INVALID;
    return load_bodies(c, c->b.n, n, in, "edynhip_add_bodies");
}

int edynhip_set_joints(edynhip_ctx *c, uint32_t n, const edynhip_joints *in) {
    if (!c || (n && !in)) return EDYNHIP_ERR_INVALID;
    if (n > c->j.cap) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_set_joints: n > max_joints");
    EH_HIP(c, hipSetDevice(c->device));
    Joints &j = c->j;
    j.n = n; j.num_colours = 0; j.rows = 0;
    c->force_islands = true;
    c->all_asleep = false;
    std::memset(j.colour_start, 0, sizeof(j.colour_start));
    if (n == 0) return EDYNHIP_OK;
    // body kinds are needed for the colouring (only procedural endpoints constrain a colour)
    std::vector<uint32_t> flags(c->b.n);
    EH_HIP(c, hipMemcpy(flags.data(), c->b.flags, flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    auto dyn = [&](uint32_t b) { return (flags[b] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC; };
    // Deterministic edge colouring, identical to the per-step contact colouring kernels (solver.hip k_col_*).
    std::vector<uint32_t> colour(n, kNoColour);
    std::vector<uint64_t> used(c->b.n, 0), best(c->b.n, 0);
    for (;;) {
        bool any = false;
        for (uint32_t e = 0; e < n; ++e) {
            if (colour[e] != kNoColour) continue;
            any = true;
            uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);
            uint32_t a = in->body[2 * e], b = in->body[2 * e + 1];
            if (a >= c->b.n || b >= c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_joints: body index out of range");
            if (dyn(a)) best[a] = std::max(best[a], pr);
            if (dyn(b)) best[b] = std::max(best[b], pr);
        }
        if (!any) break;
        for (uint32_t e = 0; e < n; ++e) {
            if (colour[e] != kNoColour) continue;
            uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);
            uint32_t a = in->body[2 * e], b = in->body[2 * e + 1];
            bool da = dyn(a), db = dyn(b);
            if ((da && best[a] != pr) || (db && best[b] != pr)) continue;
            uint64_t busy = (da ? used[a] : 0) | (db ? used[b] : 0);
            uint32_t col = 0;
            while (col < kMaxColours && (busy >> col & 1)) ++col;
            if (col >= kMaxColours) return set_error(c, EDYNHIP_ERR_COLOURS, "edynhip_set_joints: more than 64 joint colours");
            colour[e] = col;
            if (da) used[a] |= 1ull << col;
            if (db) used[b] |= 1ull << col;
        }
        std::fill(best.begin(), best.end(), 0);
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return colour[x] < colour[y]; });
    std::vector<uint32_t> type(n), bA(n), bB(n);
    std::vector<float4> pivA(n), pivB(n), axA(n), pA(n), qA(n), axB(n);
    auto plane_space_h = [](const float *nn, float *p, float *q) {   // geom.cpp:730-754 (host, fp32)
        if (std::fabs(nn[2]) > dm::kHalfSqrt2) {
            float a = nn[1] * nn[1] + nn[2] * nn[2]; float k = 1.0f / std::sqrt(a);
            p[0] = 0; p[1] = -nn[2] * k; p[2] = nn[1] * k; q[0] = a * k; q[1] = -nn[0] * p[2]; q[2] = nn[0] * p[1];
        } else {
            float a = nn[0] * nn[0] + nn[1] * nn[1]; float k = 1.0f / std::sqrt(a);
            p[0] = -nn[1] * k; p[1] = nn[0] * k; p[2] = 0; q[0] = -nn[2] * p[1]; q[1] = nn[2] * p[0]; q[2] = a * k;
        }
    };
    uint32_t ncol = 0, rows = 0;
    for (uint32_t p = 0; p < n; ++p) {
        const uint32_t e = order[p];
        type[p] = (uint32_t)in->type[e]; bA[p] = in->body[2 * e]; bB[p] = in->body[2 * e + 1];
        const float *pv = in->pivot + 6 * e;
        pivA[p] = make_float4(pv[0], pv[1], pv[2], 0); pivB[p] = make_float4(pv[3], pv[4], pv[5], 0);
        axA[p] = pA[p] = qA[p] = axB[p] = make_float4(0, 0, 0, 0);
        if (in->type[e] == EDYNHIP_JOINT_HINGE) {
            if (!in->axis) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_joints: hinge needs axes");
            const float *ax = in->axis + 6 * e;
            float p3[3], q3[3];
            plane_space_h(ax, p3, q3);
            axA[p] = make_float4(ax[0], ax[1], ax[2], 0); pA[p] = make_float4(p3[0], p3[1], p3[2], 0); qA[p] = make_float4(q3[0], q3[1], q3[2], 0);
            axB[p] = make_float4(ax[3], ax[4], ax[5], 0);
            rows += 5;
        } else rows += 3;
        ncol = std::max(ncol, colour[e] + 1);
    }
    for (uint32_t k = 0; k <= ncol; ++k) {
        uint32_t s = 0;
        while (s < n && colour[order[s]] < k) ++s;
        j.colour_start[k] = s;
    }
    j.num_colours = ncol; j.rows = rows;
    hipStream_t s = c->stream;
    EH_HIP(c, hipMemcpyAsync(j.orig, order.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.type, type.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.bodyA, bA.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.bodyB, bB.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pivA, pivA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pivB, pivB.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.axA, axA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pA, pA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.qA, qA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.axB, axB.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemsetAsync(j.impulse, 0, (size_t)j.cap * 5 * sizeof(float), s));
    EH_HIP(c, hipStreamSynchronize(s));
    c->stats.num_joints = n; c->stats.num_joint_rows = rows; c->stats.num_joint_colours = ncol;
    return EDYNHIP_OK;
}

int edynhip_run_stages(edynhip_ctx *c, uint32_t mask) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    return run_stages(c, mask);
}

int edynhip_step(edynhip_ctx *c, uint32_t nsteps) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    c->timings = edynhip_timings{};
    c->timer.recorded = 0;
    for (uint32_t i = 0; i < nsteps; ++i) {
        // every procedural body asleep and nothing edited since: the step changes nothing (each stage excludes sleeping
        // entities), so it is not run at all - a world at rest costs no GPU time, as in the reference
        if (c->all_asleep) { ++c->step_index; continue; }
        EH_TRY(run_stages(c, EDYNHIP_STAGE_ALL));
    }
    return EDYNHIP_OK;
}

int edynhip_get_state(edynhip_ctx *c, float *pos, float *orn, float *linvel, float *angvel) {
    if (!c) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    float *d = c->state_dev;
    const float *h = c->state_host;
    hipLaunchKernelGGL(k_pack_state, dim3((n + 255) / 256), dim3(256), 0, c->stream, 0u, n, c->b, d);
    hipError_t e = hipMemcpyAsync(c->state_host, d, (size_t)n * 13 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_get_state", e);
    for (uint32_t i = 0; i < n; ++i) {
        const float *o = &h[(size_t)i * 13];
        if (pos) std::memcpy(pos + 3 * i, o, 12);
        if (orn) std::memcpy(orn + 4 * i, o + 3, 16);
        if (linvel) std::memcpy(linvel + 3 * i, o + 7, 12);
        if (angvel) std::memcpy(angvel + 3 * i, o + 10, 12);
    }
    return EDYNHIP_OK;
}

int edynhip_set_state(edynhip_ctx *c, const float *pos, const float *orn, const float *linvel, const float *angvel) {
    if (!c || !pos || !orn || !linvel || !angvel) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    float *h = c->state_host, *d = c->state_dev;
    for (uint32_t i = 0; i < n; ++i) {
        float *o = &h[(size_t)i * 13];
        std::memcpy(o, pos + 3 * i, 12); std::memcpy(o + 3, orn + 4 * i, 16);
        std::memcpy(o + 7, linvel + 3 * i, 12); std::memcpy(o + 10, angvel + 3 * i, 12);
    }
    hipError_t e = hipMemcpyAsync(d, h, (size_t)n * 13 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_unpack_state, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d, c->b);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the staging buffers are reused by the next call
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_set_state", e);
    if (c->sleeping) return edynhip_wake_all(c);   // an edited body wakes its island (wake_up_entity); all of them here
    return EDYNHIP_OK;
}

// ---- collision exclusion lists (util/exclude_collision.cpp:9-71)
static int upload_exclusion_rows(edynhip_ctx *c, uint32_t a, uint32_t b) {
    if (!c->excl) {
        EH_TRY(dalloc(c, c->excl, (size_t)c->b.cap * 16));
        EH_HIP(c, hipMemsetAsync(c->excl, 0xFF, (size_t)c->b.cap * 16 * sizeof(uint32_t), c->stream));
    }
    for (uint32_t x : {a, b})
        EH_HIP(c, hipMemcpyAsync(c->excl + (size_t)x * 16, c->host_excl.data() + (size_t)x * 16, 16 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    return EDYNHIP_OK;
}
int edynhip_exclude_collision(edynhip_ctx *c, uint32_t a, uint32_t b) {
    if (!c || a >= c->b.n || b >= c->b.n || a == b) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    if (c->host_excl.empty()) c->host_excl.assign((size_t)c->b.cap * 16, 0xFFFFFFFFu);
    auto add = [&](uint32_t x, uint32_t y) -> int {   // exclude_collision_one_way
        uint32_t *l = &c->host_excl[(size_t)x * 16];
        uint32_t k = 0;
        for (; k < 16 && l[k] != 0xFFFFFFFFu; ++k) if (l[k] == y) return EDYNHIP_OK;
        if (k == 16) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_exclude_collision: more than 16 exclusions on one body (collision_exclusion::max_exclusions)");
        l[k] = y;
        return EDYNHIP_OK;
    };
    EH_TRY(add(a, b)); EH_TRY(add(b, a));
    return upload_exclusion_rows(c, a, b);
}
int edynhip_remove_collision_exclusion(edynhip_ctx *c, uint32_t a, uint32_t b) {
    if (!c || a >= c->b.n || b >= c->b.n) return EDYNHIP_ERR_INVALID;
    if (c->host_excl.empty()) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    auto drop = [&](uint32_t x, uint32_t y) {   // remove_collision_exclusion_one_way: the last entry takes the hole
        uint32_t *l = &c->host_excl[(size_t)x * 16];
        uint32_t size = 0;
        while (size < 16 && l[size] != 0xFFFFFFFFu) ++size;
        for (uint32_t i = size; i; --i)
            if (l[i - 1] == y) { l[i - 1] = l[size - 1]; l[size - 1] = 0xFFFFFFFFu; break; }
    };
    drop(a, b); drop(b, a);
    return upload_exclusion_rows(c, a, b);
}

int edynhip_refresh_derived(edynhip_ctx *c) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    return refresh_derived(c);
}

__global__ void k_wake_all(uint32_t n, uint32_t *flags, int32_t *since) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] &= ~BF_ASLEEP;
    since[i] = -1;
}
int edynhip_wake_all(edynhip_ctx *c) {
    if (!c) return EDYNHIP_ERR_INVALID;
    c->all_asleep = false;
    if (c->b.n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_wake_all, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b.flags, c->sleep_since);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}
int edynhip_get_asleep(edynhip_ctx *c, uint8_t *asleep) {
    if (!c || !asleep) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    std::vector<uint32_t> fl(n);
    EH_HIP(c, hipMemcpyAsync(fl.data(), c->b.flags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i) asleep[i] = (fl[i] & BF_ASLEEP) ? 1 : 0;
    return EDYNHIP_OK;
}

int edynhip_pack_state_device(edynhip_ctx *c, void *dst, uint32_t first, uint32_t count) {
    if (!c || !dst || (uint64_t)first + count > c->b.n) return EDYNHIP_ERR_INVALID;
    if (count == 0) return EDYNHIP_OK;
    hipLaunchKernelGGL(k_pack_state, dim3((count + 255) / 256), dim3(256), 0, c->stream, first, count, c->b, (float *)dst);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

int edynhip_get_derived(edynhip_ctx *c, float *aabb, float *iw, uint32_t *island) {
    if (!c) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    float *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, (size_t)n * 15 * sizeof(float)));
    hipLaunchKernelGGL(k_pack_derived, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->b, d, d + (size_t)n * 6);
    hipError_t e = hipSuccess;
    if (aabb) e = hipMemcpyAsync(aabb, d, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && iw) e = hipMemcpyAsync(iw, d + (size_t)n * 6, (size_t)n * 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && island) e = hipMemcpyAsync(island, c->b.island, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_get_derived", e);
    return EDYNHIP_OK;
}

int edynhip_num_manifolds(edynhip_ctx *c, uint32_t *n) {
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    *n = c->num_manifolds;
    return EDYNHIP_OK;
}

int edynhip_get_manifolds(edynhip_ctx *c, edynhip_manifold *out, uint32_t capacity, uint32_t *n) {
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    const uint32_t M = c->num_manifolds;
    *n = M;
    if (M == 0 || !out) return EDYNHIP_OK;
    if (capacity < M) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_manifolds: capacity too small");
    EH_HIP(c, hipSetDevice(c->device));
    edynhip_manifold *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, (size_t)M * sizeof(edynhip_manifold)));
    hipLaunchKernelGGL(k_manifolds_to_records, dim3((M + 127) / 128), dim3(128), 0, c->stream, M, c->m[c->cur], d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)M * sizeof(edynhip_manifold), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_get_manifolds", e);
    return EDYNHIP_OK;
}

int edynhip_set_manifolds(edynhip_ctx *c, const edynhip_manifold *in, uint32_t n) {
    if (!c || (n && !in)) return EDYNHIP_ERR_INVALID;
    if (n > c->m[c->cur].cap) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_set_manifolds: n > max_manifolds");
    EH_HIP(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < n; ++i)
        if (in[i].body[0] >= c->b.n || in[i].body[1] >= c->b.n || in[i].num_points > (uint32_t)kMaxPts)
            return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_manifolds: body index or point count out of range");
    // records must arrive in ascending canonical key order (the order edynhip_get_manifolds returns)
    for (uint32_t i = 1; i < n; ++i) {
        auto key = [&](const edynhip_manifold &m) -> uint64_t {
            const uint32_t a = m.body[0], b = m.body[1];
            if (a >= c->b.n || b >= c->b.n) return (uint64_t)~0ull;
            const uint32_t o = pair_owner(a, b, c->host_kind[a] == EDYNHIP_KIND_DYNAMIC, c->host_kind[b] == EDYNHIP_KIND_DYNAMIC);
            return ((uint64_t)o << 32) | (o == a ? b : a);
        };
        if (!(key(in[i - 1]) < key(in[i]))) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_manifolds: records not sorted by canonical pair key");
    }
    c->num_manifolds = n;
    c->force_islands = true;
    c->all_asleep = false;
    c->clears_primed = false;
    EH_HIP(c, hipMemsetAsync(c->m[c->cur].seg_start, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream));
    EH_HIP(c, hipMemsetAsync(c->m[c->cur].seg_end, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream));
    if (n == 0) return EDYNHIP_OK;
    edynhip_manifold *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, (size_t)n * sizeof(edynhip_manifold)));
    hipError_t e = hipMemcpyAsync(d, in, (size_t)n * sizeof(edynhip_manifold), hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_records_to_manifolds, dim3((n + 127) / 128), dim3(128), 0, c->stream, n, d, c->m[c->cur], c->b.flags);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_set_manifolds", e);
    return EDYNHIP_OK;
}

int edynhip_get_pairs(edynhip_ctx *c, uint64_t *keys, uint32_t capacity, uint32_t *n) {
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    const uint32_t M = c->num_manifolds;
    *n = M;
    if (M == 0 || !keys) return EDYNHIP_OK;
    if (capacity < M) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_pairs: capacity too small");
    EH_HIP(c, hipSetDevice(c->device));
    EH_HIP(c, hipMemcpyAsync(keys, c->m[c->cur].skey, (size_t)M * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < M; ++i) {   // drop the orientation bit; (owner, other) -> the documented (max, min)
        const uint32_t a = (uint32_t)(keys[i] >> 33), b = (uint32_t)(keys[i] >> 1);
        keys[i] = ((uint64_t)std::max(a, b) << 32) | std::min(a, b);
    }
    std::sort(keys, keys + M);
    return EDYNHIP_OK;
}

int edynhip_get_joint_impulses(edynhip_ctx *c, float *out) {
    if (!c || !out) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->j.n;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    std::vector<float> imp((size_t)c->j.cap * 5);
    std::vector<uint32_t> orig(n);
    EH_HIP(c, hipMemcpyAsync(imp.data(), c->j.impulse, imp.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipMemcpyAsync(orig.data(), c->j.orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t p = 0; p < n; ++p)
        for (int r = 0; r < 5; ++r) out[5 * orig[p] + r] = imp[(size_t)r * c->j.cap + p];
    return EDYNHIP_OK;
}

int edynhip_debug_collide(edynhip_ctx *c, uint32_t n, const int32_t *shape_type, const float *shape_param, const float *pos,
                          const float *orn, float threshold, float *out_points, uint32_t *out_count) {
    if (!c || (n && (!shape_type || !shape_param || !pos || !orn || !out_points || !out_count))) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    return debug_collide(c, n, shape_type, shape_param, pos, orn, threshold, out_points, out_count);
}

int edynhip_get_timings(edynhip_ctx *c, edynhip_timings *out) {
    if (!c || !out) return EDYNHIP_ERR_INVALID;
    resolve_timings(c);
    *out = c->timings;
    return EDYNHIP_OK;
}

int edynhip_get_stats(edynhip_ctx *c, edynhip_stats *out) {
    if (!c || !out) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(count_points(c));
    EH_HIP(c, hipMemcpyAsync(c->cnt_host, c->cnt, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    c->stats.num_bodies = c->b.n;
    c->stats.num_manifolds = c->num_manifolds;
    c->stats.num_points = c->cnt_host->num_points;
    c->stats.num_active_manifolds = c->cnt_host->num_active;
    c->stats.num_islands = c->cnt_host->num_islands;
    c->stats.num_colours = c->num_colours;
    for (uint32_t k = 0; k < kMaxColours; ++k) c->stats.colour_size[k] = k < c->num_colours ? c->colour_end[k] - c->colour_start[k] : 0;
    *out = c->stats;
    return EDYNHIP_OK;
}

}  // extern "C"
