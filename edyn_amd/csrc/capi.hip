// C-ABI implementation (include/edynhip.h): context lifetime, scene upload, step orchestration,
// state read-back. Host-side logic only; every simulation stage runs in the HIP kernels of
// broadphase.hip / narrowphase.hip / solver.hip. There is no CPU fallback.
#include "ctx.hpp"
#include "dpolyhedron.hpp"
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
// The two halves of a fetch: the publish kernel is enqueued, then the caller may enqueue work that does not need the host's answer
// (speculative launches whose kernels read their element counts from device memory) before it waits - the GPU then runs that work
// while the sequence number travels to the host and the next launches travel back, instead of idling (~11-14 us per fetch).
int publish_counters(edynhip_ctx *c, size_t bytes, uint32_t *ticket) {
    const uint32_t value = ++c->cnt_seq_next;
    c->last_fetch_step = c->step_index;
    hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(256), 0, c->stream, (const uint32_t *)c->cnt, (uint32_t *)c->cnt_host,
                       (uint32_t)(bytes / sizeof(uint32_t)), c->cnt_seq, value);
    EH_HIP(c, hipGetLastError());
    *ticket = value;
    return EDYNHIP_OK;
}
int wait_counters(edynhip_ctx *c, uint32_t value) {
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
int fetch_counters(edynhip_ctx *c, size_t bytes) {
    uint32_t ticket = 0;
    EH_TRY(publish_counters(c, bytes, &ticket));
    return wait_counters(c, ticket);
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
    if (c->cfg.flags & EDYNHIP_FLAG_CONTACT_EVENTS) EH_TRY(dalloc(c, m.pid, (size_t)cap * kMaxPts));
    EH_TRY(dalloc(c, m.seg_start, nb)); EH_TRY(dalloc(c, m.seg_end, nb)); EH_TRY(dalloc(c, m.prev_idx, cap)); EH_TRY(dalloc(c, m.tree, cap));
    EH_TRY(dalloc(c, m.skey, cap)); EH_TRY(dalloc(c, m.bodyA, cap)); EH_TRY(dalloc(c, m.bodyB, cap)); EH_TRY(dalloc(c, m.info, cap));
    if (kPointRecords) {   // one allocation, five interleaved base pointers: element pt_at(cap, k, m) of each is a field of point k of manifold m
        float4 *pts = nullptr;
        EH_TRY(dalloc(c, pts, (size_t)cap * kMaxPts * kPointF));
        m.pA = pts; m.pB = pts + 1; m.nrm = pts + 2; m.lnrm = pts + 3; m.imp = pts + 4;
    } else {
        EH_TRY(dalloc(c, m.pA, (size_t)cap * kMaxPts)); EH_TRY(dalloc(c, m.pB, (size_t)cap * kMaxPts));
        EH_TRY(dalloc(c, m.nrm, (size_t)cap * kMaxPts)); EH_TRY(dalloc(c, m.lnrm, (size_t)cap * kMaxPts));
        EH_TRY(dalloc(c, m.imp, (size_t)cap * kMaxPts));
    }
    return EDYNHIP_OK;
}

static int allocate(edynhip_ctx *c) {
    const uint32_t nb = c->cfg.max_bodies, M = c->cfg.max_manifolds, nj = c->cfg.max_joints;
    Bodies &b = c->b;
    b.cap = nb;
    EH_TRY(dalloc(c, b.xf, (size_t)nb * 8)); EH_TRY(dalloc(c, b.dvw, (size_t)nb * 2)); EH_TRY(dalloc(c, b.linvel, nb)); EH_TRY(dalloc(c, b.angvel, nb));
    EH_TRY(dalloc(c, b.amin, nb)); EH_TRY(dalloc(c, b.amax, nb)); EH_TRY(dalloc(c, b.shape, nb)); EH_TRY(dalloc(c, b.grav, nb));
    EH_TRY(dalloc(c, b.mat, nb)); EH_TRY(dalloc(c, b.mat2, nb)); EH_TRY(dalloc(c, b.flags, nb)); EH_TRY(dalloc(c, b.group, nb)); EH_TRY(dalloc(c, b.mask, nb));
    EH_TRY(dalloc(c, b.island, nb));
    EH_TRY(alloc_manifolds(c, c->m[0], M, nb));
    EH_TRY(alloc_manifolds(c, c->m[1], M, nb));
    Rows &r = c->rows;
    EH_TRY(dalloc(c, r.order, M)); EH_TRY(dalloc(c, r.bA, M)); EH_TRY(dalloc(c, r.bB, M)); EH_TRY(dalloc(c, r.np, M)); EH_TRY(dalloc(c, r.label, M));
    EH_TRY(dalloc(c, r.rw, (size_t)M * kMaxPts * kRowsPerPoint * kRowF));
    EH_TRY(dalloc(c, r.dslot, ((size_t)M + 64) * 4));   // whole 64-lane blocks (dslot_at)
    EH_TRY(dalloc(c, r.pslot, ((size_t)M + 32) * 6));   // whole 32-lane blocks (pslot_at)
    EH_TRY(dalloc(c, r.next, (size_t)M * 2)); EH_TRY(dalloc(c, r.im, (size_t)M * 2));
    EH_TRY(dalloc(c, r.pw, (size_t)M * kMaxPts * kPosF));
    EH_TRY(dalloc(c, r.slot_of, (size_t)nb * kMaxColours)); EH_TRY(dalloc(c, r.first_slot, nb)); EH_TRY(dalloc(c, r.skip, M));
    LBVH &t = c->bvh;
    EH_TRY(dalloc(c, t.keys, nb)); EH_TRY(dalloc(c, t.keys_sorted, nb));
    EH_TRY(dalloc(c, t.parent, (size_t)2 * nb)); EH_TRY(dalloc(c, t.left, nb)); EH_TRY(dalloc(c, t.right, nb));
    EH_TRY(dalloc(c, t.nmin, (size_t)2 * nb)); EH_TRY(dalloc(c, t.nmax, (size_t)2 * nb)); EH_TRY(dalloc(c, t.visit, nb));
    EH_TRY(dalloc(c, t.np_list, nb)); EH_TRY(dalloc(c, t.rope, (size_t)2 * nb)); EH_TRY(dalloc(c, t.split, 8));
    EH_TRY(dalloc(c, t.cand_list, (size_t)nb * kListCap)); EH_TRY(dalloc(c, t.cand_count, nb)); EH_TRY(dalloc(c, t.ref_min, nb)); EH_TRY(dalloc(c, t.ref_max, nb));
    EH_TRY(dalloc(c, c->np_ra, (size_t)M * kMaxPts)); EH_TRY(dalloc(c, c->np_rb, (size_t)M * kMaxPts)); EH_TRY(dalloc(c, c->np_rn, (size_t)M * kMaxPts));
    EH_TRY(dalloc(c, c->np_rnum, M));
    if (c->cfg.flags & EDYNHIP_FLAG_CONTACT_EVENTS) {
        c->event_cap = 5u * M + 1024u;
        EH_TRY(dalloc(c, c->events, c->event_cap)); EH_TRY(dalloc(c, c->event_count, 4)); EH_TRY(dalloc(c, c->prev_matched, M));
    }
    EH_TRY(dalloc(c, c->pair_keys, M)); EH_TRY(dalloc(c, c->pair_keys_sorted, M)); EH_TRY(dalloc(c, c->new_edges, M)); EH_TRY(dalloc(c, c->new_edge_m, M));
    EH_TRY(dalloc(c, c->isl_top, nb)); EH_TRY(dalloc(c, c->rot_off, nb)); EH_HIP(c, hipMemsetAsync(c->rot_off, 0xFF, (size_t)nb * sizeof(uint32_t), c->stream));
    EH_TRY(dalloc(c, c->own_keys, (size_t)nb * kOwnCap)); EH_TRY(dalloc(c, c->own_count, (size_t)nb + 1)); EH_TRY(dalloc(c, c->own_offset, (size_t)nb + 1));
    EH_TRY(dalloc(c, c->col_keys, M)); EH_TRY(dalloc(c, c->col_keys_sorted, M));
    EH_TRY(dalloc(c, c->col_unc, kColUncCap));
    { const size_t cs = 256 * (((size_t)M + 1023) / 1024) + 1; EH_TRY(dalloc(c, c->cs_hist, cs)); EH_TRY(dalloc(c, c->cs_start, cs)); EH_TRY(dalloc(c, c->cs_sup, 16 * 256)); EH_HIP(c, hipMemset(c->cs_sup, 0, 16 * 256 * sizeof(uint32_t))); }
    EH_TRY(dalloc(c, c->used, nb)); EH_TRY(dalloc(c, c->best[0], nb)); EH_TRY(dalloc(c, c->best[1], nb));
    EH_TRY(dalloc(c, c->com_store, nb)); EH_TRY(dalloc(c, c->origin_store, nb));
    EH_TRY(dalloc(c, c->isl_cnt, (size_t)nb + 1)); EH_TRY(dalloc(c, c->isl_off, (size_t)nb + 1)); EH_TRY(dalloc(c, c->isl_list, nb));
    EH_TRY(dalloc(c, c->isl_items, (size_t)M + nj)); EH_TRY(dalloc(c, c->isl_sorted, (size_t)M + nj)); EH_TRY(dalloc(c, c->isl_joint, nb));
    EH_TRY(dalloc(c, c->isl_err, nb)); EH_TRY(dalloc(c, c->isl_done, nb)); EH_TRY(dalloc(c, c->pos_err, (size_t)nb * kMaxDfPosIters));
    EH_TRY(dalloc(c, c->state_dev, (size_t)nb * 13));
    EH_HIP(c, hipHostMalloc((void **)&c->state_host, (size_t)nb * 13 * sizeof(float), hipHostMallocDefault));
    EH_TRY(dalloc(c, c->sleep_state, nb)); EH_TRY(dalloc(c, c->sleep_action, nb)); EH_TRY(dalloc(c, c->sleep_since, nb));
    EH_TRY(dalloc(c, c->sleep_old_label, nb)); EH_TRY(dalloc(c, c->sleep_size, nb)); EH_TRY(dalloc(c, c->sleep_best, nb)); EH_TRY(dalloc(c, c->sleep_carried, nb));
    EH_HIP(c, hipMemsetAsync(c->sleep_since, 0xFF, (size_t)nb * sizeof(double), c->stream));   // all ones (a NaN): no timer running
    Joints &j = c->j;
    j.cap = nj;
    EH_TRY(dalloc(c, j.orig, nj)); EH_TRY(dalloc(c, j.type, nj)); EH_TRY(dalloc(c, j.bodyA, nj)); EH_TRY(dalloc(c, j.bodyB, nj));
    EH_TRY(dalloc(c, j.pivA, nj)); EH_TRY(dalloc(c, j.pivB, nj)); EH_TRY(dalloc(c, j.axA, nj)); EH_TRY(dalloc(c, j.pA, nj));
    EH_TRY(dalloc(c, j.qA, nj)); EH_TRY(dalloc(c, j.axB, nj)); EH_TRY(dalloc(c, j.pB, nj)); EH_TRY(dalloc(c, j.impulse, (size_t)nj * kJointSlots));
    EH_TRY(dalloc(c, j.wbx, nj)); EH_TRY(dalloc(c, j.gJ, (size_t)nj * 18)); EH_TRY(dalloc(c, j.params, (size_t)nj * kJointParams)); EH_TRY(dalloc(c, j.angle, nj)); EH_TRY(dalloc(c, j.rmask, nj));
    EH_TRY(dalloc(c, j.rA, nj)); EH_TRY(dalloc(c, j.rB, nj)); EH_TRY(dalloc(c, j.wp, nj)); EH_TRY(dalloc(c, j.wq, nj)); EH_TRY(dalloc(c, j.wax, nj));
    EH_TRY(dalloc(c, j.eff, (size_t)nj * kJointSlots)); EH_TRY(dalloc(c, j.rhs, (size_t)nj * kJointSlots));
    EH_TRY(dalloc(c, j.lo, (size_t)nj * kJointSlots)); EH_TRY(dalloc(c, j.hi, (size_t)nj * kJointSlots));
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
    const int32_t *shape_type; const float *shape_param, *friction, *restitution; const uint64_t *group, *mask; const float *gravity; const uint8_t *sleeping_disabled; const float *com;
};
__global__ void k_init_bodies(uint32_t first, uint32_t n, RawBodies r, Bodies b, float3 default_gravity, dc::Meshes meshes) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;   // index into the caller's arrays
    if (l >= n) return;
    const uint32_t i = first + l;                                // body index
    const int kind = r.kind[l], st = r.shape_type[l];
    const f3 pos = mk3(r.pos[3 * l], r.pos[3 * l + 1], r.pos[3 * l + 2]);
    const q4 orn{r.orn[4 * l], r.orn[4 * l + 1], r.orn[4 * l + 2], r.orn[4 * l + 3]};
    const float4 sp = make_float4(r.shape_param[4 * l], r.shape_param[4 * l + 1], r.shape_param[4 * l + 2], r.shape_param[4 * l + 3]);
    float inv_m = 0;
    m3 il = m3_zero(), iw = m3_zero();
    const f3 com = r.com ? mk3(r.com[3 * l], r.com[3 * l + 1], r.com[3 * l + 2]) : mk3(0, 0, 0);
    const bool has_com = !(com.x == 0 && com.y == 0 && com.z == 0);
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
        } else if (st == dc::SHAPE_CAPSULE) {   // moment_of_inertia.cpp:65-90,171-173, shape_volume.cpp:10-16
            const float kPiF = 3.1415926535897932384626433832795029f;
            const float len = sp.y * 2, radius = sp.x;
            const float cyl_vol = kPiF * radius * radius * len;
            const float sph_vol = kPiF * radius * radius * radius * 4.0f / 3.0f;
            const float total_vol = cyl_vol + sph_vol;
            const float cyl_mass = mass * cyl_vol / total_vol, sph_mass = mass * sph_vol / total_vol;
            const float cyl_xx = 0.5f * cyl_mass * radius * radius;
            const float cyl_yy = 1.0f / 12.0f * cyl_mass * (3.0f * radius * radius + len * len);
            const float sph_inertia = 0.4f * sph_mass * radius * radius;
            // the capsule formula reads the cylinder's vector as (.x axial, .y transverse) although it arrives permuted for
            // the axis (:77-81): reproduced - it is the inertia the reference simulates with
            const f3 cyl = sp.z == 0.0f ? mk3(cyl_xx, cyl_yy, cyl_yy) : (sp.z == 1.0f ? mk3(cyl_yy, cyl_xx, cyl_yy) : mk3(cyl_yy, cyl_yy, cyl_xx));
            const float xx = sph_inertia + cyl.x;
            const float yy_zz = sph_inertia + sph_mass * square(4.0f * len + 3.0f * radius) / 64.0f + cyl.y;
            const f3 d = sp.z == 0.0f ? mk3(xx, yy_zz, yy_zz) : (sp.z == 1.0f ? mk3(yy_zz, xx, yy_zz) : mk3(yy_zz, yy_zz, xx));
            I = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}};
        } else if (st == dc::SHAPE_CYLINDER) {   // moment_of_inertia.cpp:27-44,167-169
            const f3 d = dc::cylinder_inertia_diag(dc::cyl_of(sp), mass);
            I = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}};
        } else if (st == dc::SHAPE_POLYHEDRON) {   // moment_of_inertia.cpp:93-157,183-185
            I = dc::polyhedron_inertia(meshes, sp, mass);
        } else {
            I = {{kScalarMax, 0, 0}, {0, kScalarMax, 0}, {0, 0, kScalarMax}};
        }
        if (has_com && !(r.has_inertia && r.has_inertia[l])) {   // shift_moment_of_inertia (moment_of_inertia.cpp:217-220): I + (d^T d) m, d = skew(com)
            const m3 d = {{0, -com.z, com.y}, {com.z, 0, -com.x}, {-com.y, com.x, 0}};
            const m3 dd = mul(transpose(d), d);
            I = {I.r0 + dd.r0 * mass, I.r1 + dd.r1 * mass, I.r2 + dd.r2 * mass};
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
    const bool moving = kind != EDYNHIP_KIND_STATIC;
    f3 lv = moving ? mk3(r.linvel[3 * l], r.linvel[3 * l + 1], r.linvel[3 * l + 2]) : mk3(0, 0, 0);
    const f3 av = moving ? mk3(r.angvel[3 * l], r.angvel[3 * l + 1], r.angvel[3 * l + 2]) : mk3(0, 0, 0);
    f3 pos_com = pos;
    if (has_com) {   // apply_center_of_mass (rigidbody.cpp:517-548): the given position is the origin
        pos_com = to_world(com, pos, orn);
        if (moving) lv += cross(av, pos_com - pos);
    }
    if (b.com) { b.com[i] = to4(com, has_com ? 1.0f : 0.0f); b.origin[i] = to4(pos, 0); }
    B_POS(b, i) = to4(pos_com, inv_m);
    B_ORN(b, i) = to4(orn);
    b.linvel[i] = to4(lv, 0);
    b.angvel[i] = to4(av, 0);
    B_DV(b, i) = make_float4(0, 0, 0, 0); B_DW(b, i) = make_float4(0, 0, 0, 0);
    B_IW(b, i, 0) = to4(iw.r0, 0); B_IW(b, i, 1) = to4(iw.r1, 0); B_IW(b, i, 2) = to4(iw.r2, 0);
    B_IL(b, i, 0) = to4(il.r0, 0); B_IL(b, i, 1) = to4(il.r1, 0); B_IL(b, i, 2) = to4(il.r2, 0);
    b.shape[i] = sp;
    f3 g = r.gravity ? mk3(r.gravity[3 * l], r.gravity[3 * l + 1], r.gravity[3 * l + 2]) : mk3(default_gravity.x, default_gravity.y, default_gravity.z);
    b.grav[i] = kind == EDYNHIP_KIND_DYNAMIC ? to4(g, 0) : make_float4(0, 0, 0, 0);
    b.mat[i] = make_float2(r.friction[l], r.restitution[l]);
    b.mat2[i] = make_float4(0.0f, 0.0f, kLarge, kLarge);   // material defaults (comp/material.hpp:15-22); edynhip_set_material_extras changes them
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
    } else if (st == dc::SHAPE_CAPSULE) {   // aabb_util.cpp:81-88
        const f3 v = rotate(orn, dc::axis_vector(sp.z)) * sp.y;
        const f3 p0 = pos - v, p1 = pos + v;
        mn = mk3(fminf(p0.x, p1.x) - sp.x, fminf(p0.y, p1.y) - sp.x, fminf(p0.z, p1.z) - sp.x);
        mx = mk3(fmaxf(p0.x, p1.x) + sp.x, fmaxf(p0.y, p1.y) + sp.x, fmaxf(p0.z, p1.z) + sp.x);
    } else if (st == dc::SHAPE_CYLINDER) {   // aabb_util.cpp:72-79
        const box3 bb = dc::cylinder_aabb(dc::cyl_of(sp), pos, orn);
        mn = bb.mn; mx = bb.mx;
    } else if (st == dc::SHAPE_POLYHEDRON) {   // aabb_util.cpp:141-164,195-197
        const box3 bb = dc::polyhedron_aabb(meshes, sp, pos, orn);
        mn = bb.mn; mx = bb.mx;
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
        const size_t s = pt_at(mf.cap, k, m);
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
__global__ void k_records_to_manifolds(uint32_t M, const edynhip_manifold *in, Manifolds mf, const uint32_t *__restrict__ flags,
                                       const float4 *__restrict__ mat2) {
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
    mf.tree[m] = 0;   // (set_manifolds forces a full island update, which rebuilds the certificate)
    for (uint32_t k = 0; k < r.num_points; ++k) {
        const size_t t = pt_at(mf.cap, k, m), s = slot_at(mf.cap, k, m);
        const edynhip_point &p = r.pt[k];
        mf.pA[t] = make_float4(p.pivotA[0], p.pivotA[1], p.pivotA[2], p.distance);
        mf.pB[t] = make_float4(p.pivotB[0], p.pivotB[1], p.pivotB[2], p.friction);
        mf.nrm[t] = make_float4(p.normal[0], p.normal[1], p.normal[2], __int_as_float(p.attachment));
        mf.lnrm[t] = make_float4(p.local_normal[0], p.local_normal[1], p.local_normal[2], p.restitution);
        mf.imp[t] = make_float4(p.normal_impulse, p.friction_impulse[0], p.friction_impulse[1], __uint_as_float(p.lifetime));
        if (mf.pid) mf.pid[s] = ((uint64_t)m << 2) | k;   // injected points: high word 0
        if (mf.xmat) {   // the record carries no extras: the materials are mixed as for a new point, the extras impulses start at 0
            const float4 ma = mat2[a], mb = mat2[b];
            float stiff = kLarge, damp = kLarge;
            if (ma.z < kLarge || mb.z < kLarge) { stiff = 1.0f / (1.0f / ma.z + 1.0f / mb.z); damp = 1.0f / (1.0f / ma.w + 1.0f / mb.w); }
            mf.xmat[s] = make_float4(fmaxf(ma.y, mb.y), fmaxf(ma.x, mb.x), stiff, damp);
            mf.ximp[s] = make_float4(0, 0, 0, 0);
        }
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
    if (c->evp_now && c->events) {   // every event of a step is emitted by the broadphase and the narrowphase: the list of this call is complete
        EH_HIP(c, hipEventRecord(c->evp_np_done, c->stream));
        EH_HIP(c, hipStreamWaitEvent(c->snap_stream, c->evp_np_done, 0));
        EH_HIP(c, hipMemcpyAsync(c->evp_host, c->event_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->snap_stream));
        EH_HIP(c, hipMemcpyAsync(c->evp_host + 64, c->events, (size_t)std::min(c->evp_max, c->event_cap) * sizeof(eh::ContactEvent), hipMemcpyDeviceToHost, c->snap_stream));
        EH_HIP(c, hipEventRecord(c->evp_ready, c->snap_stream));
        c->evp_state = 1;
    }
    if (mask & EDYNHIP_STAGE_ISLANDS) EH_TRY(guarded(islands(c)));
    if (mask & EDYNHIP_STAGE_SOLVE) EH_TRY(guarded(solve(c)));   // records events 3..9
    rec(10);
    c->clears_primed = (mask & EDYNHIP_STAGE_SOLVE) != 0;   // k_finish left the next step's scratch cleared
    if (c->timer.e) { c->timer.recorded += 1; c->timer.e = nullptr; }
    return EDYNHIP_OK;
}

// ---- removal (registry.destroy on a rigid body: island_manager.cpp:47-115)
__global__ void k_mark_removed(uint32_t n, const uint32_t *__restrict__ list, Bodies b, uint32_t *wake) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t i = list[t];
    const uint32_t fl = b.flags[i];
    if ((fl & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) wake[b.island[i]] = 1u;
    b.flags[i] = EDYNHIP_KIND_STATIC | BF_REMOVED;     // shapeless, static, awake: no stage touches it any more
    b.linvel[i] = make_float4(0, 0, 0, 0); b.angvel[i] = make_float4(0, 0, 0, 0);
    b.grav[i] = make_float4(0, 0, 0, 0);
    float4 p = B_POS(b, i); p.w = 0; B_POS(b, i) = p;
    B_DV(b, i) = make_float4(0, 0, 0, 0); B_DW(b, i) = make_float4(0, 0, 0, 0);
    b.island[i] = i;
}
__global__ void k_mark_touching(uint32_t n, const uint32_t *__restrict__ list, Bodies b, uint32_t *wake) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t i = list[t];
    if ((b.flags[i] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) wake[b.island[i]] = 1u;
}
__global__ void k_wake_partners(uint32_t M, Manifolds mf, Bodies b, uint32_t *wake) {   // islands resting on a removed static/kinematic body
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t a = mf.bodyA[m], bb = mf.bodyB[m];
    const bool ra = b.flags[a] & BF_REMOVED, rb = b.flags[bb] & BF_REMOVED;
    if (ra && (b.flags[bb] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) wake[b.island[bb]] = 1u;
    if (rb && (b.flags[a] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) wake[b.island[a]] = 1u;
}
__global__ void k_wake_marked(uint32_t n, Bodies b, uint32_t *wake, double *since) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t fl = b.flags[i];
    if ((fl & BF_KIND_MASK) != EDYNHIP_KIND_DYNAMIC) return;
    const uint32_t l = b.island[i];
    if (wake[l]) { b.flags[i] = fl & ~BF_ASLEEP; if (l == i) since[i] = -1.0; }
}
// wake the islands of the listed bodies (wake_up_island, island_manager.cpp:541-571)
// Index lists handed to small kernels (bodies to wake, bodies to remove): one persistent device buffer that grows on demand
// and is released with the context - no hipMalloc / hipFree per call, nothing to leak on an error path.
static int index_scratch(edynhip_ctx *c, size_t count, uint32_t *&out) {
    if (count > c->idx_scratch_cap) {
        if (c->idx_scratch) (void)hipFree(c->idx_scratch);
        c->idx_scratch = nullptr; c->idx_scratch_cap = 0;
        const size_t cap = std::max<size_t>(count * 2, 1024);
        EH_HIP(c, hipMalloc((void **)&c->idx_scratch, cap * sizeof(uint32_t)));
        c->idx_scratch_cap = cap;
    }
    out = c->idx_scratch;
    return EDYNHIP_OK;
}
int wake_islands_of(edynhip_ctx *c, const std::vector<uint32_t> &bodies) {
    if (!c->sleeping || bodies.empty() || c->b.n == 0) { c->all_asleep = false; return EDYNHIP_OK; }
    uint32_t *list = nullptr;
    EH_TRY(index_scratch(c, bodies.size(), list));
    EH_HIP(c, hipMemcpyAsync(list, bodies.data(), bodies.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipMemsetAsync(c->sleep_action, 0, (size_t)c->b.n * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_mark_touching, dim3(((uint32_t)bodies.size() + 127) / 128), dim3(128), 0, c->stream, (uint32_t)bodies.size(), list, c->b, c->sleep_action);
    hipLaunchKernelGGL(k_wake_marked, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b, c->sleep_action, c->sleep_since);
    hipError_t e = hipStreamSynchronize(c->stream);   // the scratch list may be refilled by the next call
    c->all_asleep = false;
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "wake_islands_of", e);
    return EDYNHIP_OK;
}

}  // namespace eh

using namespace eh;

extern "C" {

uint32_t edynhip_abi_version(void) { return 15; }   // 15: edynhip_snapshot_records / edynhip_snapshot_map (the registry write-back read in place), edynhip_set_event_prefetch / edynhip_prefetched_events; 14: EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / EDYNHIP_FLAG_BLOCK_POSITION (the default contact arithmetic is the reference's); 13: edynhip_set_pair_filter (settings.should_collide_func); 12: multi-GPU world (edynhip_world_*, multi.hip); 11: polyhedron shapes (edynhip_create_convex_mesh); 10: edynhip_stats::solve_schedule, edynhip_measure_bandwidth; 9: edynhip_set_center_of_mass; 8: edynhip_bodies::center_of_mass; 7: edynhip_wake_bodies; 6: every constraint type, capsules, material mix table; 5: contact_extras materials; 4: contact events + point ids, double-buffered snapshots;   // 3: joint slots/params, add/remove joints, remove bodies, params, timed steps, exclusions

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
    if (getenv("EDYNHIP_TREE_STATS"))   // developer knob: how the island labels were kept up to date (solver.hip islands)
        fprintf(stderr, "[edynhip] island labels: %llu steps relabelled in full, %llu incrementally\n", (unsigned long long)c->cc_full_steps, (unsigned long long)c->cc_incremental_steps);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->idx_scratch) (void)hipFree(c->idx_scratch);
    for (void *p : c->mesh_allocs) (void)hipFree(p);
    if (c->rot) (void)hipFree(c->rot);
    if (c->poly_work) (void)hipFree(c->poly_work);
    if (c->cnt_host) (void)hipHostFree(c->cnt_host);
    if (c->cnt_seq) (void)hipHostFree((void *)c->cnt_seq);
    if (c->state_host) (void)hipHostFree(c->state_host);
    for (int k = 0; k < 2; ++k) {
        if (c->snap_host[k]) (void)hipHostFree(c->snap_host[k]);
        if (c->snap_event[k]) (void)hipEventDestroy(c->snap_event[k]);
    }
    if (c->evp_host) (void)hipHostFree(c->evp_host);
    if (c->evp_np_done) (void)hipEventDestroy(c->evp_np_done);
    if (c->evp_ready) (void)hipEventDestroy(c->evp_ready);
    for (int k = 0; k < 2; ++k) {
        if (c->rec_host[k]) (void)hipHostFree(c->rec_host[k]);
        if (c->rec_event[k]) (void)hipEventDestroy(c->rec_event[k]);
    }
    if (c->snap_ready) (void)hipEventDestroy(c->snap_ready);
    if (c->snap_stream) (void)hipStreamDestroy(c->snap_stream);
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

// broadphase participants from the host mirror of (kind, shape): [shaped non-procedural ..., shaped procedural ...]
static int rebuild_broadphase_lists(edynhip_ctx *c) {
    std::vector<uint32_t> np_list, proc_list;
    for (uint32_t i = 0; i < (uint32_t)c->host_kind.size(); ++i) {
        if (c->host_shape[i] == EDYNHIP_SHAPE_NONE) continue;
        (c->host_kind[i] == EDYNHIP_KIND_DYNAMIC ? proc_list : np_list).push_back(i);
    }
    // only procedural shaped bodies own pairs; everybody else's count must read 0 in the scan
    EH_HIP(c, hipMemsetAsync(c->own_count, 0, ((size_t)c->b.cap + 1) * sizeof(uint32_t), c->stream));
    c->all_asleep = false;
    c->bvh.age = 0;   // the tree topology is rebuilt on the next step
    c->bvh.lists_dirty = true;
    c->bvh.num_np = (uint32_t)np_list.size();
    c->bvh.num_proc = (uint32_t)proc_list.size();
    np_list.insert(np_list.end(), proc_list.begin(), proc_list.end());
    if (!np_list.empty())
        EH_HIP(c, hipMemcpyAsync(c->bvh.np_list, np_list.data(), np_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));   // np_list (a local) must outlive the copy
    return EDYNHIP_OK;
}

static int rebuild_mix_table(edynhip_ctx *c);
static int load_bodies(edynhip_ctx *c, uint32_t first, uint32_t n, const edynhip_bodies *in, const char *who) {
    if (c) ++c->topology_epoch;
    if (!c || !in) return EDYNHIP_ERR_INVALID;
    if ((uint64_t)first + n > c->b.cap) return set_error(c, EDYNHIP_ERR_CAPACITY, (std::string(who) + ": more than max_bodies").c_str());
    if (n && (!in->kind || !in->pos || !in->orn || !in->linvel || !in->angvel || !in->mass || !in->shape_type || !in->shape_param ||
              !in->friction || !in->restitution))
        return set_error(c, EDYNHIP_ERR_INVALID, (std::string(who) + ": missing array").c_str());
    for (uint32_t i = 0; i < n; ++i)
        if (in->shape_type[i] < EDYNHIP_SHAPE_NONE || in->shape_type[i] > EDYNHIP_SHAPE_POLYHEDRON)
            return set_error(c, EDYNHIP_ERR_UNSUPPORTED, (std::string(who) + ": shape type not on this path (box, sphere, plane, capsule, cylinder, polyhedron)").c_str());
    EH_HIP(c, hipSetDevice(c->device));   // before anything allocates: a process may hold contexts on several GPUs
    EH_TRY(mesh_bind_bodies(c, first, n, in->shape_type, in->shape_param));
    if (first == 0) c->has_cylinder = false;
    for (uint32_t i = 0; i < n; ++i) if (in->shape_type[i] == EDYNHIP_SHAPE_CYLINDER) c->has_cylinder = true;   // narrowphase.hip launches k_np_detect_ext
    if (first == 0) { c->has_restitution = false; c->extras = false; c->host_mat_id.clear(); c->b.mix_K = 0; for (auto &kv : c->host_mix) if (kv.second[0] > 0) c->has_restitution = true; }
    for (uint32_t i = 0; i < n; ++i)
        if (in->restitution[i] > 0.0f) c->has_restitution = true;   // turns the restitution solver on (restitution.hip)
    if (first == 0) { c->host_joints.clear(); c->j.n = 0; c->j.num_colours = 0; c->j.rows = 0; c->host_excl.clear(); c->excl_dirty_lo = 0xFFFFFFFFu; c->excl_dirty_hi = 0; if (c->excl) (void)hipMemsetAsync(c->excl, 0xFF, (size_t)c->b.cap * 16 * sizeof(uint32_t), c->stream); }
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
    {   // host mirror of the collision groups / masks (edynhip_default_should_collide)
        c->host_group.resize((size_t)first + n, ~0ull); c->host_mask.resize((size_t)first + n, ~0ull);
        for (uint32_t i = 0; i < n; ++i) { c->host_group[first + i] = in->group ? in->group[i] : ~0ull; c->host_mask[first + i] = in->mask ? in->mask[i] : ~0ull; }
    }
    up(in->sleeping_disabled, n, r.sleeping_disabled);
    if (first == 0) c->num_sleepable = 0;
    for (uint32_t i = 0; i < n; ++i) if (in->kind[i] == EDYNHIP_KIND_DYNAMIC && !(in->sleeping_disabled && in->sleeping_disabled[i])) ++c->num_sleepable;
    {   // centre-of-mass offsets: the origin arrays are attached to the body set with the first body that has one
        bool any = false;
        if (in->center_of_mass) for (size_t k = 0; k < (size_t)n * 3; ++k) any = any || in->center_of_mass[k] != 0.0f;
        if (first == 0) { c->b.com = nullptr; c->b.origin = nullptr; }
        if (any && !c->b.com) {
            if (first != 0 && rc == EDYNHIP_OK && hipMemsetAsync(c->com_store, 0, (size_t)c->b.cap * sizeof(float4), c->stream) != hipSuccess) rc = EDYNHIP_ERR_HIP;
            c->b.com = c->com_store; c->b.origin = c->origin_store;
        }
        if (c->b.com) up(in->center_of_mass, in->center_of_mass ? (size_t)n * 3 : 0, r.com);
    }
    const uint32_t total = first + n;
    if (rc == EDYNHIP_OK) {
        c->b.n = total;
        if (n)
            hipLaunchKernelGGL(k_init_bodies, dim3((n + 255) / 256), dim3(256), 0, c->stream, first, n, r, c->b,
                               make_float3(c->cfg.gravity[0], c->cfg.gravity[1], c->cfg.gravity[2]), c->meshes);
        c->host_kind.resize(first); c->host_shape.resize(first);
        c->host_kind.insert(c->host_kind.end(), in->kind, in->kind + n);
        c->host_shape.insert(c->host_shape.end(), in->shape_type, in->shape_type + n);
    }
    if (rc == EDYNHIP_OK) rc = rebuild_broadphase_lists(c);
    if (rc == EDYNHIP_OK && first != 0 && c->b.mix_K) rc = rebuild_mix_table(c);   // appended bodies carry no material id yet
    (void)hipStreamSynchronize(c->stream);
    for (void *p : tmp) (void)hipFree(p);
    if (first == 0) {   // a new world: no manifolds, no running sleep timers (appended bodies keep every index stable instead)
        c->num_manifolds = 0;
        c->prev_num_manifolds = 0;
        c->step_index = 0;
        c->num_colours = 0;
        (void)hipMemsetAsync(c->sleep_since, 0xFF, (size_t)c->b.cap * sizeof(double), c->stream);
        c->sim_clock = 0;
        (void)hipMemsetAsync(c->sleep_state, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream);
        (void)hipMemsetAsync(c->sleep_action, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream);
        (void)hipMemsetAsync(c->sleep_size, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream);
        (void)hipMemsetAsync(c->sleep_best, 0, (size_t)c->b.cap * sizeof(unsigned long long), c->stream);
        c->sleep_prev_n = 0;   // no islands of a previous step
        c->island_labels_valid = false;
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

// Host joint list -> device arrays: deterministic edge colouring over the live joints in caller-index order (the rule of the
// per-step contact colouring, solver.hip k_col_*), colour-sorted upload, applied impulses and hinge angles carried along.
// fetch = first read the current impulses / angles back from the device (the joints were stepped since the last rebuild).
static int rebuild_joints(edynhip_ctx *c, bool fetch) {
    Joints &j = c->j;
    ++c->topology_epoch;
    hipStream_t s = c->stream;
    std::vector<HostJoint> &hj = c->host_joints;
    if (fetch && j.n) {
        std::vector<float> imp((size_t)j.cap * kJointSlots), ang(j.n);
        std::vector<uint32_t> orig(j.n);
        EH_HIP(c, hipMemcpyAsync(imp.data(), j.impulse, imp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        EH_HIP(c, hipMemcpyAsync(ang.data(), j.angle, (size_t)j.n * sizeof(float), hipMemcpyDeviceToHost, s));
        EH_HIP(c, hipMemcpyAsync(orig.data(), j.orig, (size_t)j.n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        EH_HIP(c, hipStreamSynchronize(s));
        for (uint32_t p = 0; p < j.n; ++p) {
            HostJoint &h = hj[orig[p]];
            for (int r = 0; r < kJointSlots; ++r) h.impulse[r] = imp[(size_t)r * j.cap + p];
            h.angle = ang[p];
        }
    }
    std::vector<uint32_t> live;
    c->has_generic = false;
    for (uint32_t e = 0; e < hj.size(); ++e) if (hj[e].alive) { live.push_back(e); if (hj[e].type == EDYNHIP_JOINT_GENERIC) c->has_generic = true; }
    const uint32_t n = (uint32_t)live.size();
    if (n > j.cap) return set_error(c, EDYNHIP_ERR_CAPACITY, "joints: more live joints than max_joints");
    j.n = n; j.num_colours = 0; j.rows = 0;
    c->force_islands = true;
    c->all_asleep = false;
    std::memset(j.colour_start, 0, sizeof(j.colour_start));
    c->stats.num_joints = n; c->stats.num_joint_rows = 0; c->stats.num_joint_colours = 0;
    if (n == 0) return EDYNHIP_OK;
    auto dyn = [&](uint32_t b) { return c->host_kind[b] == EDYNHIP_KIND_DYNAMIC; };
    std::vector<uint32_t> colour(n, kNoColour);
    std::vector<uint64_t> used(c->b.n, 0), best(c->b.n, 0);
    for (;;) {
        bool any = false;
        for (uint32_t e = 0; e < n; ++e) {
            if (colour[e] != kNoColour) continue;
            any = true;
            const uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);
            const uint32_t a = hj[live[e]].body[0], b = hj[live[e]].body[1];
            if (dyn(a)) best[a] = std::max(best[a], pr);
            if (dyn(b)) best[b] = std::max(best[b], pr);
        }
        if (!any) break;
        for (uint32_t e = 0; e < n; ++e) {
            if (colour[e] != kNoColour) continue;
            const uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);
            const uint32_t a = hj[live[e]].body[0], b = hj[live[e]].body[1];
            const bool da = dyn(a), db = dyn(b);
            if ((da && best[a] != pr) || (db && best[b] != pr)) continue;
            const uint64_t busy = (da ? used[a] : 0) | (db ? used[b] : 0);
            uint32_t col = 0;
            while (col < kMaxColours && (busy >> col & 1)) ++col;
            if (col >= kMaxColours) return set_error(c, EDYNHIP_ERR_COLOURS, "joints: more than 64 joint colours");
            colour[e] = col;
            if (da) used[a] |= 1ull << col;
            if (db) used[b] |= 1ull << col;
        }
        std::fill(best.begin(), best.end(), 0);
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return colour[x] < colour[y]; });
    std::vector<uint32_t> orig(n), type(n), bA(n), bB(n);
    std::vector<float4> pivA(n), pivB(n), axA(n), pA(n), qA(n), axB(n), pB(n);
    std::vector<float> params((size_t)j.cap * kJointParams, 0.0f), impulse((size_t)j.cap * kJointSlots, 0.0f), angle(n);
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
        const HostJoint &h = hj[live[order[p]]];
        orig[p] = live[order[p]];
        type[p] = (uint32_t)h.type; bA[p] = h.body[0]; bB[p] = h.body[1];
        pivA[p] = make_float4(h.pivot[0], h.pivot[1], h.pivot[2], 0); pivB[p] = make_float4(h.pivot[3], h.pivot[4], h.pivot[5], 0);
        axA[p] = pA[p] = qA[p] = axB[p] = pB[p] = make_float4(0, 0, 0, 0);
        if (h.type == EDYNHIP_JOINT_HINGE) {   // hinge_constraint::set_axes, hinge_constraint.cpp:11-17
            float p3[3], q3[3];
            plane_space_h(h.axis, p3, q3);
            axA[p] = make_float4(h.axis[0], h.axis[1], h.axis[2], 0); pA[p] = make_float4(p3[0], p3[1], p3[2], 0); qA[p] = make_float4(q3[0], q3[1], q3[2], 0);
            plane_space_h(h.axis + 3, p3, q3);
            axB[p] = make_float4(h.axis[3], h.axis[4], h.axis[5], 0); pB[p] = make_float4(p3[0], p3[1], p3[2], 0);
            rows += 5;
        } else if (h.has_frames) {   // cone / cvjoint: full frames (column k of a row-major 3x3 = elements k, 3 + k, 6 + k)
            auto col = [&](int f, int k) { return make_float4(h.frame[9 * f + k], h.frame[9 * f + 3 + k], h.frame[9 * f + 6 + k], 0); };
            axA[p] = col(0, 0); pA[p] = col(0, 1); qA[p] = col(0, 2); axB[p] = col(1, 0); pB[p] = col(1, 1);
            rows += h.type == EDYNHIP_JOINT_CVJOINT ? 9 : (h.type == EDYNHIP_JOINT_GENERIC ? 24 : 2);
        } else rows += 3;
        for (int k = 0; k < kJointParams; ++k) params[(size_t)k * j.cap + p] = h.params[k];
        for (int r = 0; r < kJointSlots; ++r) impulse[(size_t)r * j.cap + p] = h.impulse[r];
        angle[p] = h.angle;
        ncol = std::max(ncol, colour[order[p]] + 1);
    }
    for (uint32_t k = 0; k <= ncol; ++k) {
        uint32_t q = 0;
        while (q < n && colour[order[q]] < k) ++q;
        j.colour_start[k] = q;
    }
    j.num_colours = ncol; j.rows = rows;
    EH_HIP(c, hipMemcpyAsync(j.orig, orig.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.type, type.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.bodyA, bA.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.bodyB, bB.data(), n * 4, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pivA, pivA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pivB, pivB.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.axA, axA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pA, pA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.qA, qA.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.axB, axB.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.pB, pB.data(), n * 16, hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.params, params.data(), params.size() * sizeof(float), hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.impulse, impulse.data(), impulse.size() * sizeof(float), hipMemcpyHostToDevice, s));
    EH_HIP(c, hipMemcpyAsync(j.angle, angle.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
    EH_HIP(c, hipStreamSynchronize(s));
    c->stats.num_joints = n; c->stats.num_joint_rows = rows; c->stats.num_joint_colours = ncol;
    return EDYNHIP_OK;
}
static int append_host_joints(edynhip_ctx *c, uint32_t n, const edynhip_joints *in, const char *who) {
    for (uint32_t e = 0; e < n; ++e) {
        HostJoint h;
        h.type = in->type[e]; h.body[0] = in->body[2 * e]; h.body[1] = in->body[2 * e + 1];
        if (h.body[0] >= c->b.n || h.body[1] >= c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, (std::string(who) + ": body index out of range").c_str());
        if (h.type < EDYNHIP_JOINT_POINT || h.type > EDYNHIP_JOINT_NULL) return set_error(c, EDYNHIP_ERR_UNSUPPORTED, (std::string(who) + ": joint type").c_str());
        std::memcpy(h.pivot, in->pivot + 6 * e, sizeof(h.pivot));
        if (h.type == EDYNHIP_JOINT_HINGE) {
            if (!in->axis) return set_error(c, EDYNHIP_ERR_INVALID, (std::string(who) + ": hinge needs axes").c_str());
            std::memcpy(h.axis, in->axis + 6 * e, sizeof(h.axis));
        }
        if (in->params) std::memcpy(h.params, in->params + (size_t)kJointApiParams * e, sizeof(float) * kJointApiParams);
        if (h.type == EDYNHIP_JOINT_CONE || h.type == EDYNHIP_JOINT_CVJOINT || h.type == EDYNHIP_JOINT_GENERIC) {   // identity frames, an open cone / free dofs, until defined
            const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(h.frame, id, sizeof(id)); std::memcpy(h.frame + 9, id, sizeof(id));
            h.has_frames = true;
            if (h.type == EDYNHIP_JOINT_CONE && !(h.params[0] > 0 && h.params[1] > 0)) { h.params[0] = 1; h.params[1] = 1; }
        }
        c->host_joints.push_back(h);
    }
    return EDYNHIP_OK;
}
// hinges whose parameters enable the angle-dependent rows start from the angle of the current pose (reset_angle)
static int reset_new_angles(edynhip_ctx *c, uint32_t first_caller_index) {
    Joints &j = c->j;
    if (j.n == 0) return EDYNHIP_OK;
    std::vector<uint32_t> orig(j.n);
    EH_HIP(c, hipMemcpyAsync(orig.data(), j.orig, (size_t)j.n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint8_t> which(j.n, 0);
    bool any = false;
    for (uint32_t p = 0; p < j.n; ++p) if (orig[p] >= first_caller_index) { which[p] = 1; any = true; }
    if (!any) return EDYNHIP_OK;
    uint8_t *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, j.n));
    EH_HIP(c, hipMemcpyAsync(d, which.data(), j.n, hipMemcpyHostToDevice, c->stream));
    int rc = joint_reset_angles(c, d);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    return rc;
}

static int flush_joint_redefs(edynhip_ctx *c);
static int flush_exclusions(edynhip_ctx *c);
int edynhip_set_joints(edynhip_ctx *c, uint32_t n, const edynhip_joints *in) {
    if (!c || (n && !in)) return EDYNHIP_ERR_INVALID;
    if (n > c->j.cap) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_set_joints: n > max_joints");
    EH_HIP(c, hipSetDevice(c->device));
    c->host_joints.clear();
    c->pending_redefs.clear();
    if (n) EH_TRY(append_host_joints(c, n, in, "edynhip_set_joints"));
    EH_TRY(rebuild_joints(c, false));
    return reset_new_angles(c, 0);
}
int edynhip_add_joints(edynhip_ctx *c, uint32_t n, const edynhip_joints *in, uint32_t *first_index) {
    if (!c || (n && !in)) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    const uint32_t first = (uint32_t)c->host_joints.size();
    if (first_index) *first_index = first;
    if (n == 0) return EDYNHIP_OK;
    EH_TRY(flush_joint_redefs(c));
    EH_TRY(append_host_joints(c, n, in, "edynhip_add_joints"));
    int rc = rebuild_joints(c, true);
    if (rc != EDYNHIP_OK) { c->host_joints.resize(first); (void)rebuild_joints(c, false); return rc; }
    return reset_new_angles(c, first);
}
int edynhip_remove_joints(edynhip_ctx *c, uint32_t n, const uint32_t *indices) {
    if (!c || (n && !indices)) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    for (uint32_t k = 0; k < n; ++k) if (indices[k] >= c->host_joints.size()) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_remove_joints: index out of range");
    EH_TRY(flush_joint_redefs(c));
    // fetch first: the impulses of the survivors come from the device
    EH_TRY(rebuild_joints(c, true));
    std::vector<uint32_t> touched;
    for (uint32_t k = 0; k < n; ++k) {
        HostJoint &h = c->host_joints[indices[k]];
        if (!h.alive) continue;
        h.alive = false;
        touched.push_back(h.body[0]); touched.push_back(h.body[1]);
    }
    EH_TRY(rebuild_joints(c, false));
    return wake_islands_of(c, touched);   // destroying an edge wakes its island (island_manager.cpp:74-97)
}
static int redefine_joint(edynhip_ctx *c, uint32_t joint, const float *params, int nparams, const float *frameA, const float *frameB);
int edynhip_set_joint_params(edynhip_ctx *c, uint32_t joint, const float *params) {
    if (!c || !params || joint >= c->host_joints.size() || !c->host_joints[joint].alive) return EDYNHIP_ERR_INVALID;
    return redefine_joint(c, joint, params, kJointApiParams, nullptr, nullptr);
}
int edynhip_set_joint_definition(edynhip_ctx *c, uint32_t joint, const float *frameA, const float *frameB, const float *params16) {
    if (!c || !params16 || !frameA || !frameB || joint >= c->host_joints.size() || !c->host_joints[joint].alive) return EDYNHIP_ERR_INVALID;
    const int t = c->host_joints[joint].type;
    if (t != EDYNHIP_JOINT_CONE && t != EDYNHIP_JOINT_CVJOINT) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_joint_definition: cone and cvjoint constraints only");
    if (t == EDYNHIP_JOINT_CONE && !(params16[0] > 0 && params16[1] > 0)) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_joint_definition: cone span tangents must be positive");
    return redefine_joint(c, joint, params16, 16, frameA, frameB);
}
int edynhip_set_generic_definition(edynhip_ctx *c, uint32_t joint, const float *frameA, const float *frameB, const float *dof60) {
    if (!c || !dof60 || !frameA || !frameB || joint >= c->host_joints.size() || !c->host_joints[joint].alive) return EDYNHIP_ERR_INVALID;
    if (c->host_joints[joint].type != EDYNHIP_JOINT_GENERIC) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_generic_definition: generic constraints only");
    for (int d = 0; d < 6; ++d)
        if (dof60[10 * d] != 0 && dof60[10 * d + 1] > dof60[10 * d + 2]) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_generic_definition: a limit's minimum exceeds its maximum");
    return redefine_joint(c, joint, dof60, 60, frameA, frameB);
}
int edynhip_set_joint_warm_start(edynhip_ctx *c, const float *impulses24, const float *angles) {
    if (!c || !impulses24) return EDYNHIP_ERR_INVALID;
    const uint32_t total = (uint32_t)c->host_joints.size();
    if (total == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(flush_joint_redefs(c));
    for (uint32_t e = 0; e < total; ++e) {
        HostJoint &h = c->host_joints[e];
        if (!h.alive) continue;
        std::memcpy(h.impulse, impulses24 + (size_t)kJointSlots * e, sizeof(h.impulse));
        if (angles) h.angle = angles[e];
    }
    return rebuild_joints(c, false);   // host copies -> device (colouring unchanged: same joints)
}
__global__ void k_set_asleep(uint32_t n, Bodies b, const uint8_t *__restrict__ asleep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t fl = b.flags[i];
    if ((fl & BF_KIND_MASK) != EDYNHIP_KIND_DYNAMIC || (fl & (BF_REMOVED | BF_NOSLEEP))) return;
    if (asleep[i]) {   // put_to_sleep (island_manager.cpp:553-565): tag + zero velocities
        b.flags[i] = fl | BF_ASLEEP;
        b.linvel[i] = make_float4(0, 0, 0, 0); b.angvel[i] = make_float4(0, 0, 0, 0);
    } else b.flags[i] = fl & ~BF_ASLEEP;
}
int edynhip_set_asleep(edynhip_ctx *c, const uint8_t *asleep) {
    if (!c || !asleep) return EDYNHIP_ERR_INVALID;
    if (!c->sleeping || c->b.n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    uint32_t *buf = nullptr;
    EH_TRY(index_scratch(c, (c->b.n + 3) / 4, buf));
    EH_HIP(c, hipMemcpyAsync(buf, asleep, c->b.n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_asleep, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b, (const uint8_t *)buf);
    EH_HIP(c, hipStreamSynchronize(c->stream));
    c->all_asleep = false;   // decided again by the next step
    c->force_islands = true;
    return EDYNHIP_OK;
}
int edynhip_get_sleep_timers(edynhip_ctx *c, uint32_t *island_label, double *since, double *clock) {
    if (!c || !island_label || !since || !clock) return EDYNHIP_ERR_INVALID;
    *clock = c->sim_clock;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    // no island stage has run yet on this scene (a re-partition before the first step): every body is its own island, no timer runs
    if (!c->sleeping || !c->island_labels_valid) { for (uint32_t i = 0; i < n; ++i) { island_label[i] = i; since[i] = -1.0; } return EDYNHIP_OK; }
    EH_HIP(c, hipSetDevice(c->device));
    EH_HIP(c, hipMemcpyAsync(island_label, c->b.island, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipMemcpyAsync(since, c->sleep_since, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    return EDYNHIP_OK;
}
int edynhip_set_sleep_timers(edynhip_ctx *c, const uint32_t *island_label, const double *since, double clock) {
    if (!c || !island_label || !since) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    if (!c->sleeping || n == 0) return EDYNHIP_OK;
    for (uint32_t i = 0; i < n; ++i) if (island_label[i] >= n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_sleep_timers: island label out of range");
    EH_HIP(c, hipSetDevice(c->device));
    // The labels become "last step's labels": the next step's relabel (forced below) starts from them, so its merge / split rules
    // (k_sleep_sizes / k_sleep_carry, k_cc_flatten's split marks) see the islands the timers belong to.
    EH_HIP(c, hipMemcpyAsync(c->b.island, island_label, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipMemcpyAsync(c->sleep_since, since, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    c->sim_clock = clock;
    c->sleep_prev_n = n;
    c->force_islands = true;
    c->island_labels_valid = true;
    return EDYNHIP_OK;
}
int edynhip_get_joint_slot_impulses(edynhip_ctx *c, float *out) {
    if (!c || !out) return EDYNHIP_ERR_INVALID;
    const uint32_t total = (uint32_t)c->host_joints.size();
    if (total == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(flush_joint_redefs(c));
    std::memset(out, 0, (size_t)total * kJointSlots * sizeof(float));
    const uint32_t n = c->j.n;
    if (n == 0) return EDYNHIP_OK;
    std::vector<float> imp((size_t)c->j.cap * kJointSlots);
    std::vector<uint32_t> orig(n);
    EH_HIP(c, hipMemcpyAsync(imp.data(), c->j.impulse, imp.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipMemcpyAsync(orig.data(), c->j.orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t p = 0; p < n; ++p)
        for (int r = 0; r < kJointSlots; ++r) out[(size_t)kJointSlots * orig[p] + r] = imp[(size_t)r * c->j.cap + p];
    return EDYNHIP_OK;
}
// Redefinitions are applied lazily: the host copy is edited at once, the device arrays are rebuilt (and the angles of the
// edited joints reset against the current orientations, as hinge_constraint / cvjoint_constraint::reset_angle) by the next entry
// point that reads or advances them - a figure's thousands of definitions cost one rebuild, not one each.
static int redefine_joint(edynhip_ctx *c, uint32_t joint, const float *params, int nparams, const float *frameA, const float *frameB) {
    std::memcpy(c->host_joints[joint].params, params, sizeof(float) * nparams);
    if (frameA) { std::memcpy(c->host_joints[joint].frame, frameA, 9 * sizeof(float)); std::memcpy(c->host_joints[joint].frame + 9, frameB, 9 * sizeof(float)); c->host_joints[joint].has_frames = true; }
    c->pending_redefs.push_back(joint);
    return EDYNHIP_OK;
}
static int flush_joint_redefs(edynhip_ctx *c) {
    if (c->pending_redefs.empty()) return EDYNHIP_OK;
    std::vector<uint32_t> pending;
    pending.swap(c->pending_redefs);
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(rebuild_joints(c, true));    // applied impulses and angles of the others come from the device
    EH_TRY(rebuild_joints(c, false));
    if (c->j.n == 0) return EDYNHIP_OK;
    std::vector<uint8_t> edited(c->host_joints.size(), 0);
    for (uint32_t j : pending) if (j < edited.size()) edited[j] = 1;
    std::vector<uint32_t> orig(c->j.n);
    EH_HIP(c, hipMemcpyAsync(orig.data(), c->j.orig, (size_t)c->j.n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint8_t> which(c->j.n, 0);
    for (uint32_t p = 0; p < c->j.n; ++p) which[p] = edited[orig[p]];
    uint8_t *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, c->j.n));
    EH_HIP(c, hipMemcpyAsync(d, which.data(), c->j.n, hipMemcpyHostToDevice, c->stream));
    int rc = joint_reset_angles(c, d);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    return rc;
}

int edynhip_run_stages(edynhip_ctx *c, uint32_t mask) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(flush_joint_redefs(c));
    EH_TRY(flush_exclusions(c));
    return run_stages(c, mask);
}

static int step_stamped(edynhip_ctx *c, uint32_t nsteps, bool timed, double first_time, double step_dt) {
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(flush_joint_redefs(c));
    EH_TRY(flush_exclusions(c));
    c->timings = edynhip_timings{};
    c->timer.recorded = 0;
    if (c->events) EH_HIP(c, hipMemsetAsync(c->event_count, 0, sizeof(uint32_t), c->stream));   // the events of THIS call
    c->evp_state = c->evp_max ? 2 : 0;
    for (uint32_t i = 0; i < nsteps; ++i) {
        c->evp_now = c->evp_max != 0 && i + 1 == nsteps;
        // island_manager::update runs put_islands_to_sleep() against the PREVIOUS step's stamp and only then takes the new one
        // (island_manager.cpp:533-539): the kernels read c->sim_clock, which is advanced after the step
        const double stamp = timed ? first_time + step_dt * (double)i : c->sim_clock + (double)c->cfg.fixed_dt;
        // every procedural body asleep and nothing edited since: the step changes nothing (each stage excludes sleeping
        // entities), so it is not run at all - a world at rest costs no GPU time, as in the reference
        if (c->all_asleep) { ++c->step_index; c->sim_clock = stamp; continue; }
        const int rc = run_stages(c, EDYNHIP_STAGE_ALL);
        c->sim_clock = stamp;
        if (rc != EDYNHIP_OK) { c->evp_now = false; return rc; }
    }
    c->evp_now = false;
    return EDYNHIP_OK;
}
int edynhip_step(edynhip_ctx *c, uint32_t nsteps) {
    if (!c) return EDYNHIP_ERR_INVALID;
    return step_stamped(c, nsteps, false, 0, 0);
}
int edynhip_step_timed(edynhip_ctx *c, uint32_t nsteps, double first_step_time, double step_dt) {
    if (!c) return EDYNHIP_ERR_INVALID;
    return step_stamped(c, nsteps, true, first_step_time, step_dt);
}

int edynhip_get_params(edynhip_ctx *c, edynhip_params *out) {
    if (!c || !out) return EDYNHIP_ERR_INVALID;
    out->fixed_dt = c->cfg.fixed_dt;
    out->num_velocity_iterations = c->cfg.num_velocity_iterations;
    out->num_position_iterations = c->cfg.num_position_iterations;
    std::memcpy(out->gravity, c->cfg.gravity, sizeof(out->gravity));
    out->num_restitution_iterations = c->restitution_iterations;
    out->num_individual_restitution_iterations = c->individual_restitution_iterations;
    return EDYNHIP_OK;
}
__global__ void k_set_gravity(uint32_t n, Bodies b, float4 g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (b.flags[i] & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) b.grav[i] = g;
}
int edynhip_set_params(edynhip_ctx *c, const edynhip_params *p) {
    if (!c || !p || !(p->fixed_dt > 0)) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    c->cfg.fixed_dt = p->fixed_dt;
    c->cfg.num_velocity_iterations = p->num_velocity_iterations;
    c->cfg.num_position_iterations = p->num_position_iterations;
    c->restitution_iterations = p->num_restitution_iterations;
    c->individual_restitution_iterations = p->num_individual_restitution_iterations;
    if (std::memcmp(c->cfg.gravity, p->gravity, sizeof(p->gravity)) != 0) {   // set_gravity: the setting and every body's gravity
        std::memcpy(c->cfg.gravity, p->gravity, sizeof(p->gravity));
        if (c->b.n) hipLaunchKernelGGL(k_set_gravity, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b, make_float4(p->gravity[0], p->gravity[1], p->gravity[2], 0));
        EH_HIP(c, hipGetLastError());
        if (c->sleeping) EH_TRY(edynhip_wake_all(c));
    }
    return EDYNHIP_OK;
}

int edynhip_remove_bodies(edynhip_ctx *c, uint32_t n, const uint32_t *indices) {
    if (!c || (n && !indices)) return EDYNHIP_ERR_INVALID;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    for (uint32_t k = 0; k < n; ++k) if (indices[k] >= c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_remove_bodies: index out of range");
    // joints attached to a removed body go with it (the node's edges are destroyed, island_manager.cpp:56-62)
    std::vector<uint8_t> gone(c->b.n, 0);
    for (uint32_t k = 0; k < n; ++k) gone[indices[k]] = 1;
    bool joints_changed = false;
    std::vector<uint32_t> partners;
    if (!c->host_joints.empty()) {
        EH_TRY(flush_joint_redefs(c));
        EH_TRY(rebuild_joints(c, true));
        for (HostJoint &h : c->host_joints)
            if (h.alive && (gone[h.body[0]] || gone[h.body[1]])) { h.alive = false; joints_changed = true; partners.push_back(h.body[0]); partners.push_back(h.body[1]); }
    }
    uint32_t *list = nullptr;
    EH_TRY(index_scratch(c, n, list));
    EH_HIP(c, hipMemcpyAsync(list, indices, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipMemsetAsync(c->sleep_action, 0, (size_t)c->b.n * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_mark_removed, dim3((n + 127) / 128), dim3(128), 0, c->stream, n, list, c->b, c->sleep_action);
    if (c->num_manifolds) hipLaunchKernelGGL(k_wake_partners, dim3((c->num_manifolds + 255) / 256), dim3(256), 0, c->stream, c->num_manifolds, c->m[c->cur], c->b, c->sleep_action);
    hipLaunchKernelGGL(k_wake_marked, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b, c->sleep_action, c->sleep_since);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_remove_bodies", e);
    for (uint32_t k = 0; k < n; ++k) { c->host_kind[indices[k]] = EDYNHIP_KIND_STATIC; c->host_shape[indices[k]] = EDYNHIP_SHAPE_NONE; }
    if (!c->host_excl.empty())
        for (uint32_t k = 0; k < n; ++k) std::fill(c->host_excl.begin() + (size_t)indices[k] * 16, c->host_excl.begin() + (size_t)indices[k] * 16 + 16, 0xFFFFFFFFu);
    EH_TRY(rebuild_broadphase_lists(c));
    if (joints_changed) { EH_TRY(rebuild_joints(c, false)); EH_TRY(wake_islands_of(c, partners)); }
    c->force_islands = true;
    c->all_asleep = false;
    c->clears_primed = false;
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
    EH_TRY(flush_joint_redefs(c));   // pending angle resets refer to the orientations before this edit
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
    c->bvh.lists_dirty = true;   // moved behind the broadphase's back: the candidate lists are rebuilt at the next step
    if (c->sleeping) return edynhip_wake_all(c);   // an edited body wakes its island (wake_up_entity); all of them here
    return EDYNHIP_OK;
}

// ---- collision exclusion lists (util/exclude_collision.cpp:9-71)
// The host list is edited at once; the device rows follow lazily, with the next entry point that runs the broadphase (flush_exclusions):
// a figure's 21 exclusions per rag doll - 21 504 calls for a field of 1 024 - cost one upload, not two 64-byte copies and a stream
// synchronisation each (43 042 copy launches at scene set-up).
static int upload_exclusion_rows(edynhip_ctx *c, uint32_t a, uint32_t b) {
    c->excl_dirty_lo = std::min(c->excl_dirty_lo, std::min(a, b));
    c->excl_dirty_hi = std::max(c->excl_dirty_hi, std::max(a, b) + 1u);
    return EDYNHIP_OK;
}
static int flush_exclusions(edynhip_ctx *c) {
    if (c->excl_dirty_lo >= c->excl_dirty_hi) return EDYNHIP_OK;
    const uint32_t lo = c->excl_dirty_lo, hi = std::min(c->excl_dirty_hi, c->b.cap);
    c->excl_dirty_lo = 0xFFFFFFFFu; c->excl_dirty_hi = 0;
    if (c->host_excl.empty() || lo >= hi) return EDYNHIP_OK;
    if (!c->excl) {
        EH_TRY(dalloc(c, c->excl, (size_t)c->b.cap * 16));
        EH_HIP(c, hipMemsetAsync(c->excl, 0xFF, (size_t)c->b.cap * 16 * sizeof(uint32_t), c->stream));
    }
    EH_HIP(c, hipMemcpyAsync(c->excl + (size_t)lo * 16, c->host_excl.data() + (size_t)lo * 16, (size_t)(hi - lo) * 16 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));   // (the host list may be edited again right away)
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
int edynhip_set_pair_filter(edynhip_ctx *c, edynhip_pair_filter filter, void *user) {
    if (!c) return EDYNHIP_ERR_INVALID;
    c->pair_filter = filter; c->pair_filter_user = user;
    if (filter && !c->filter_new_idx) {
        EH_HIP(c, hipSetDevice(c->device));
        EH_TRY(dalloc(c, c->filter_new_idx, c->cfg.max_manifolds));
    }
    return EDYNHIP_OK;
}
int edynhip_default_should_collide(edynhip_ctx *c, uint32_t a, uint32_t b) {   // should_collide_default, should_collide.cpp:11-57
    if (!c || a >= c->b.n || b >= c->b.n) return EDYNHIP_ERR_INVALID;
    if (a == b) return 0;
    if (a < c->host_group.size() && b < c->host_group.size() && ((c->host_group[a] & c->host_mask[b]) == 0 || (c->host_group[b] & c->host_mask[a]) == 0)) return 0;
    if (!c->host_excl.empty()) {
        auto listed = [&](uint32_t x, uint32_t y) {
            const uint32_t *l = &c->host_excl[(size_t)x * 16];
            for (uint32_t k = 0; k < 16 && l[k] != 0xFFFFFFFFu; ++k) if (l[k] == y) return true;
            return false;
        };
        if (listed(a, b) || listed(b, a)) return 0;
    }
    return 1;
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
    c->bvh.lists_dirty = true;
    return refresh_derived(c);
}

__global__ void k_wake_all(uint32_t n, uint32_t *flags, double *since) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] &= ~BF_ASLEEP;
    since[i] = -1.0;
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
int edynhip_wake_bodies(edynhip_ctx *c, uint32_t n, const uint32_t *indices) {
    if (!c || (n && !indices)) return EDYNHIP_ERR_INVALID;
    for (uint32_t k = 0; k < n; ++k) if (indices[k] >= c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_wake_bodies: index out of range");
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    return wake_islands_of(c, std::vector<uint32_t>(indices, indices + n));
}
__global__ void k_move_com(Bodies b, uint32_t i, float3 com_new) {
    const f3 com = mk3(com_new.x, com_new.y, com_new.z);
    const float4 p4 = B_POS(b, i);
    const f3 pos = from4(p4);
    const q4 orn = q_from4(B_ORN(b, i));
    const f3 origin = b.com[i].w != 0.0f ? to_world(-from4(b.com[i]), pos, orn) : pos;
    const f3 com_world = to_world(com, origin, orn);
    if ((b.flags[i] & BF_KIND_MASK) != EDYNHIP_KIND_STATIC) b.linvel[i] = to4(from4(b.linvel[i]) + cross(from4(b.angvel[i]), com_world - pos), 0);
    B_POS(b, i) = to4(com_world, p4.w);
    const bool has = !(com.x == 0 && com.y == 0 && com.z == 0);
    b.com[i] = to4(has ? com : mk3(0, 0, 0), has ? 1.0f : 0.0f);
    b.origin[i] = to4(origin, 0);
}
int edynhip_set_center_of_mass(edynhip_ctx *c, uint32_t body, const float *com3) {
    if (!c || !com3 || body >= c->b.n) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    if (!c->b.com) {   // the first offset of this world: attach the (zeroed) origin arrays
        EH_HIP(c, hipMemsetAsync(c->com_store, 0, (size_t)c->b.cap * sizeof(float4), c->stream));
        c->b.com = c->com_store; c->b.origin = c->origin_store;
    }
    hipLaunchKernelGGL(k_move_com, dim3(1), dim3(1), 0, c->stream, c->b, body, make_float3(com3[0], com3[1], com3[2]));
    EH_HIP(c, hipGetLastError());
    c->bvh.lists_dirty = true;   // (the reference does not wake the body's island either)
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
    // colouring state that travels with the records: the number of colours in use (the next step releases the top one)
    c->num_colours = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (in[i].colour != kNoColour && in[i].num_points > 0 && in[i].colour + 1 > c->num_colours) c->num_colours = in[i].colour + 1;
    EH_HIP(c, hipMemsetAsync(c->m[c->cur].seg_start, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream));
    EH_HIP(c, hipMemsetAsync(c->m[c->cur].seg_end, 0, (size_t)c->b.cap * sizeof(uint32_t), c->stream));
    if (n == 0) return EDYNHIP_OK;
    edynhip_manifold *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, (size_t)n * sizeof(edynhip_manifold)));
    hipError_t e = hipMemcpyAsync(d, in, (size_t)n * sizeof(edynhip_manifold), hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k_records_to_manifolds, dim3((n + 127) / 128), dim3(128), 0, c->stream, n, d, c->m[c->cur], c->b.flags, c->b.mat2);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_set_manifolds", e);
    return EDYNHIP_OK;
}

// ---- contact_extras materials (comp/material.hpp:15-22; contact_extras_constraint.cpp)
static int ensure_extras_storage(edynhip_ctx *c) {   // the per-point storage and the extras rows come into existence with the first such material
    if (!c->m[0].xmat) {
        for (int k = 0; k < 2; ++k) {
            EH_TRY(dalloc(c, c->m[k].xmat, (size_t)c->m[k].cap * kMaxPts)); EH_TRY(dalloc(c, c->m[k].ximp, (size_t)c->m[k].cap * kMaxPts));
        }
        EH_TRY(dalloc(c, c->rows.rwx, (size_t)c->m[0].cap * kMaxPts * kXPoint));
        // points that already exist keep plain contact_constraint behaviour: default material, no impulses
        std::vector<float4> def((size_t)c->m[0].cap * kMaxPts, make_float4(0, 0, kLarge, kLarge));
        for (int k = 0; k < 2; ++k) EH_HIP(c, hipMemcpyAsync(c->m[k].xmat, def.data(), def.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        EH_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->extras = true;
    return EDYNHIP_OK;
}
int edynhip_set_material_extras(edynhip_ctx *c, uint32_t first, uint32_t n, const float *spin, const float *roll, const float *stiffness, const float *damping) {
    if (!c) return EDYNHIP_ERR_INVALID;
    if ((uint64_t)first + n > c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_material_extras: body range out of bounds");
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    std::vector<float4> h(n);
    bool any = false;
    for (uint32_t i = 0; i < n; ++i) {
        h[i] = make_float4(spin ? spin[i] : 0.0f, roll ? roll[i] : 0.0f, stiffness ? stiffness[i] : kLarge, damping ? damping[i] : kLarge);
        if (h[i].x < 0 || h[i].y < 0 || !(h[i].z > 0) || !(h[i].w > 0)) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_material_extras: negative friction or non-positive stiffness / damping");
        any = any || h[i].x > 0 || h[i].y > 0 || h[i].z < kLarge || h[i].w < kLarge;
    }
    if (any) EH_TRY(ensure_extras_storage(c));
    EH_HIP(c, hipMemcpyAsync(c->b.mat2 + first, h.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    return EDYNHIP_OK;
}
// ---- material ids and the mix table (material_mixing.hpp:36-82, util/insert_material_mixing.cpp)
static int rebuild_mix_table(edynhip_ctx *c) {
    const uint32_t nb = c->b.n;
    c->host_mat_id.resize(nb, 0xFFFFu);
    std::map<uint32_t, uint32_t> compact;   // material id -> compact index, ids in use only
    for (uint32_t i = 0; i < nb; ++i) if (c->host_mat_id[i] != 0xFFFFu) compact.emplace(c->host_mat_id[i], 0u);
    uint32_t K = 0;
    for (auto &kv : compact) kv.second = K++;
    if (c->host_mix.empty() || K == 0) { c->b.mix_K = 0; return EDYNHIP_OK; }
    std::vector<uint32_t> cid(nb, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < nb; ++i) if (c->host_mat_id[i] != 0xFFFFu) cid[i] = compact[c->host_mat_id[i]];
    std::vector<float> vals;
    std::map<const void *, int32_t> entry_index;
    bool extras = false;
    for (auto &kv : c->host_mix) {
        entry_index[&kv.second] = (int32_t)(vals.size() / 6);
        vals.insert(vals.end(), kv.second.begin(), kv.second.end());
        if (kv.second[0] > 0) c->has_restitution = true;
        extras = extras || kv.second[2] > 0 || kv.second[3] > 0 || kv.second[4] < kLarge || kv.second[5] < kLarge;
    }
    std::vector<int32_t> lut((size_t)K * K, -1);
    for (auto &a : compact)
        for (auto &b : compact) {   // the reference's own lookup, ordered pair (id of body[0], id of body[1])
            auto it = c->host_mix.find(edynhip_ctx::MixIdPair{a.first, b.first});
            if (it != c->host_mix.end()) lut[(size_t)a.second * K + b.second] = entry_index[&it->second];
        }
    if (extras) EH_TRY(ensure_extras_storage(c));
    if (!c->d_mat_cid) EH_TRY(dalloc(c, c->d_mat_cid, c->b.cap));
    if (c->mix_lut_cap < lut.size()) { c->d_mix_lut = nullptr; EH_TRY(dalloc(c, c->d_mix_lut, lut.size())); c->mix_lut_cap = (uint32_t)lut.size(); }
    if (c->mix_vals_cap < vals.size()) { c->d_mix_vals = nullptr; EH_TRY(dalloc(c, c->d_mix_vals, vals.size())); c->mix_vals_cap = (uint32_t)vals.size(); }
    EH_HIP(c, hipMemcpyAsync(c->d_mat_cid, cid.data(), (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipMemcpyAsync(c->d_mix_lut, lut.data(), lut.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipMemcpyAsync(c->d_mix_vals, vals.data(), vals.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    c->b.mat_cid = c->d_mat_cid; c->b.mix_lut = c->d_mix_lut; c->b.mix_vals = c->d_mix_vals; c->b.mix_K = K;
    return EDYNHIP_OK;
}
int edynhip_set_material_ids(edynhip_ctx *c, uint32_t first, uint32_t n, const uint32_t *ids) {
    if (!c || (n && !ids)) return EDYNHIP_ERR_INVALID;
    if ((uint64_t)first + n > c->b.n) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_material_ids: body range out of bounds");
    for (uint32_t i = 0; i < n; ++i) if (ids[i] > 0xFFFFu) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_set_material_ids: material::id_type is 16 bits (0xFFFF = unassigned)");
    EH_HIP(c, hipSetDevice(c->device));
    c->host_mat_id.resize(c->b.n, 0xFFFFu);
    for (uint32_t i = 0; i < n; ++i) c->host_mat_id[first + i] = ids[i];
    return rebuild_mix_table(c);
}
int edynhip_insert_material_mixing(edynhip_ctx *c, uint32_t id0, uint32_t id1, const float *material6) {
    if (!c || !material6) return EDYNHIP_ERR_INVALID;
    if (id0 >= 0xFFFFu || id1 >= 0xFFFFu) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_insert_material_mixing: unassigned material id");
    EH_HIP(c, hipSetDevice(c->device));
    c->host_mix[edynhip_ctx::MixIdPair{id0, id1}] = {material6[0], material6[1], material6[2], material6[3], material6[4], material6[5]};
    return rebuild_mix_table(c);
}
int edynhip_get_point_extras(edynhip_ctx *c, float *out7, uint32_t capacity, uint32_t *n) {
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    const uint32_t M = c->num_manifolds;
    *n = M;
    if (M == 0 || !out7) return EDYNHIP_OK;
    if (capacity < M) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_point_extras: capacity too small");
    EH_HIP(c, hipSetDevice(c->device));
    const Manifolds &mf = c->m[c->cur];
    std::vector<uint32_t> info(M);
    EH_HIP(c, hipMemcpyAsync(info.data(), mf.info, (size_t)M * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<float4> xm(M), xi(M);
    for (uint32_t k = 0; k < (uint32_t)kMaxPts; ++k) {
        if (mf.xmat) {
            EH_HIP(c, hipMemcpyAsync(xm.data(), mf.xmat + (size_t)k * mf.cap, (size_t)M * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
            EH_HIP(c, hipMemcpyAsync(xi.data(), mf.ximp + (size_t)k * mf.cap, (size_t)M * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
            EH_HIP(c, hipStreamSynchronize(c->stream));
        }
        for (uint32_t m = 0; m < M; ++m) {
            float *o = out7 + ((size_t)4 * m + k) * 7;
            const bool live = k < (info[m] & 0xFF);
            const float4 a = mf.xmat ? xm[m] : make_float4(0, 0, kLarge, kLarge), b = mf.xmat ? xi[m] : make_float4(0, 0, 0, 0);
            o[0] = live ? b.x : 0; o[1] = live ? b.y : 0; o[2] = live ? b.z : 0;
            o[3] = live ? a.x : 0; o[4] = live ? a.y : 0; o[5] = live ? a.z : 0; o[6] = live ? a.w : 0;
        }
    }
    return EDYNHIP_OK;
}

int edynhip_get_contact_events(edynhip_ctx *c, edynhip_contact_event *out, uint32_t capacity, uint32_t *n) {
    static_assert(sizeof(edynhip_contact_event) == sizeof(ContactEvent), "event record layout");
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    *n = 0;
    if (!c->events) return set_error(c, EDYNHIP_ERR_UNSUPPORTED, "edynhip_get_contact_events: create the context with EDYNHIP_FLAG_CONTACT_EVENTS");
    EH_HIP(c, hipSetDevice(c->device));
    uint32_t count = 0;
    EH_HIP(c, hipMemcpyAsync(&count, c->event_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t held = std::min(count, c->event_cap);
    *n = held;
    if (out) {
        if (capacity < held) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_contact_events: capacity too small");
        if (held) {
            EH_HIP(c, hipMemcpyAsync(out, c->events, (size_t)held * sizeof(ContactEvent), hipMemcpyDeviceToHost, c->stream));
            EH_HIP(c, hipStreamSynchronize(c->stream));
        }
    }
    if (count > c->event_cap) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_contact_events: more events than the context holds; resynchronise from edynhip_get_manifolds");
    return EDYNHIP_OK;
}
int edynhip_get_point_ids(edynhip_ctx *c, uint64_t *ids, uint32_t capacity, uint32_t *n) {
    if (!c || !n) return EDYNHIP_ERR_INVALID;
    const uint32_t M = c->num_manifolds;
    *n = M;
    if (!c->events) return set_error(c, EDYNHIP_ERR_UNSUPPORTED, "edynhip_get_point_ids: create the context with EDYNHIP_FLAG_CONTACT_EVENTS");
    if (M == 0 || !ids) return EDYNHIP_OK;
    if (capacity < M) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_point_ids: capacity too small");
    EH_HIP(c, hipSetDevice(c->device));
    const Manifolds &mf = c->m[c->cur];
    std::vector<uint64_t> col(M);
    std::vector<uint32_t> info(M);
    EH_HIP(c, hipMemcpyAsync(info.data(), mf.info, (size_t)M * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < (uint32_t)kMaxPts; ++k) {
        EH_HIP(c, hipMemcpyAsync(col.data(), mf.pid + (size_t)k * mf.cap, (size_t)M * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        EH_HIP(c, hipStreamSynchronize(c->stream));
        for (uint32_t m = 0; m < M; ++m) ids[(size_t)4 * m + k] = k < (info[m] & 0xFF) ? col[m] : 0;
    }
    return EDYNHIP_OK;
}

// ---- double-buffered read-back (simulation_worker.cpp:406-444: finished steps are handed over while the next one runs)
int edynhip_snapshot(edynhip_ctx *c) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    const uint32_t n = c->b.n;
    const int slot = (c->snap_last + 1) & 1;
    if (!c->snap_stream) {
        EH_HIP(c, hipStreamCreateWithFlags(&c->snap_stream, hipStreamNonBlocking));
        EH_HIP(c, hipEventCreateWithFlags(&c->snap_ready, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) {
            EH_HIP(c, hipEventCreateWithFlags(&c->snap_event[k], hipEventDisableTiming));
            EH_HIP(c, hipHostMalloc((void **)&c->snap_host[k], (size_t)c->b.cap * 13 * sizeof(float), hipHostMallocDefault));
            EH_TRY(dalloc(c, c->snap_dev[k], (size_t)c->b.cap * 13));
        }
    }
    // pack on the stepper's stream (it must see the finished step), copy on the side stream: the copy engine moves the
    // bytes while the stepper's next kernels already run
    if (n) hipLaunchKernelGGL(k_pack_state, dim3((n + 255) / 256), dim3(256), 0, c->stream, 0u, n, c->b, c->snap_dev[slot]);
    EH_HIP(c, hipEventRecord(c->snap_ready, c->stream));
    EH_HIP(c, hipStreamWaitEvent(c->snap_stream, c->snap_ready, 0));
    if (n) EH_HIP(c, hipMemcpyAsync(c->snap_host[slot], c->snap_dev[slot], (size_t)n * 13 * sizeof(float), hipMemcpyDeviceToHost, c->snap_stream));
    EH_HIP(c, hipEventRecord(c->snap_event[slot], c->snap_stream));
    c->snap_step[slot] = c->step_index; c->snap_bodies[slot] = n; c->snap_last = slot;
    return EDYNHIP_OK;
}
int edynhip_snapshot_read(edynhip_ctx *c, float *pos, float *orn, float *linvel, float *angvel, uint32_t *step_index) {
    if (!c) return EDYNHIP_ERR_INVALID;
    if (c->snap_last < 0) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_snapshot_read: no snapshot was taken");
    EH_HIP(c, hipSetDevice(c->device));
    const int slot = c->snap_last;
    EH_HIP(c, hipEventSynchronize(c->snap_event[slot]));   // that copy only - not the steps enqueued after it
    const float *h = c->snap_host[slot];
    for (uint32_t i = 0; i < c->snap_bodies[slot]; ++i) {
        const float *o = &h[(size_t)i * 13];
        if (pos) std::memcpy(pos + 3 * i, o, 12);
        if (orn) std::memcpy(orn + 4 * i, o + 3, 16);
        if (linvel) std::memcpy(linvel + 3 * i, o + 7, 12);
        if (angvel) std::memcpy(angvel + 3 * i, o + 10, 12);
    }
    if (step_index) *step_index = c->snap_step[slot];
    return EDYNHIP_OK;
}

// ---- record snapshots: the registry write-back read in place (edynhip.h, ABI 15)
static_assert(sizeof(edynhip_body_record) == 96, "record layout");
constexpr size_t kRecHeader = 64;   // [0] = events that occurred, [1] = events copied into this block
__global__ void k_pack_records(uint32_t n, Bodies b, float present_dt, float4 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p4 = B_POS(b, i), q4v = B_ORN(b, i), v4 = b.linvel[i], w4 = b.angvel[i];
    const uint32_t fl = b.flags[i];
    const f3 p = from4(p4), v = from4(v4), w = from4(w4);
    const q4 q{q4v.x, q4v.y, q4v.z, q4v.w};
    // update_presentation.cpp:73-79: pre = pos + vel * interpolation_dt; pre = integrate(orn, vel, interpolation_dt)
    const f3 pp = p + v * present_dt;
    const q4 po = integrate(q, w, present_dt);
    const bool has_origin = b.origin && b.com[i].w != 0.0f;
    const f3 org = has_origin ? from4(b.origin[i]) : p;
    uint32_t rf = 0;
    if ((fl & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC) rf |= EDYNHIP_RECORD_DYNAMIC;
    if (fl & BF_ASLEEP) rf |= EDYNHIP_RECORD_ASLEEP;
    if (has_origin) rf |= EDYNHIP_RECORD_HAS_ORIGIN;
    if (fl & BF_REMOVED) rf |= EDYNHIP_RECORD_REMOVED;
    float4 *o = dst + (size_t)i * 6;
    o[0] = make_float4(p.x, p.y, p.z, q.x);
    o[1] = make_float4(q.y, q.z, q.w, v.x);
    o[2] = make_float4(v.y, v.z, w.x, w.y);
    o[3] = make_float4(w.z, pp.x, pp.y, pp.z);
    o[4] = make_float4(po.x, po.y, po.z, po.w);
    o[5] = make_float4(org.x, org.y, org.z, __uint_as_float(rf));
}
__global__ void k_pack_events(const eh::ContactEvent *__restrict__ events, const uint32_t *__restrict__ count, uint32_t cap, uint32_t max_copy,
                              uint32_t *__restrict__ header, eh::ContactEvent *__restrict__ dst) {
    const uint32_t total = events ? *count : 0u;
    const uint32_t held = min(min(total, cap), max_copy);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { header[0] = total; header[1] = held; }
    for (uint32_t k = t; k < held; k += gridDim.x * blockDim.x) dst[k] = events[k];
}
int edynhip_snapshot_records(edynhip_ctx *c, float present_dt, uint32_t max_events, uint32_t flags) {
    if (!c) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    const uint32_t n = c->b.n;
    const int slot = (c->rec_last + 1) & 1;
    if (!c->snap_stream) {
        EH_HIP(c, hipStreamCreateWithFlags(&c->snap_stream, hipStreamNonBlocking));
        EH_HIP(c, hipEventCreateWithFlags(&c->snap_ready, hipEventDisableTiming));
    }
    if (!c->rec_dev[0]) {
        c->rec_event_cap = c->events ? std::min<uint32_t>(c->event_cap, std::max<uint32_t>(4096u, 2u * c->b.cap)) : 0u;
        const size_t bytes = kRecHeader + (size_t)c->rec_event_cap * sizeof(eh::ContactEvent) + (size_t)c->b.cap * sizeof(edynhip_body_record);
        for (int k = 0; k < 2; ++k) {
            EH_HIP(c, hipEventCreateWithFlags(&c->rec_event[k], hipEventDisableTiming));
            EH_HIP(c, hipHostMalloc((void **)&c->rec_host[k], bytes, hipHostMallocDefault));
            EH_TRY(dalloc(c, c->rec_dev[k], bytes));
        }
    }
    const uint32_t copy_events = std::min(max_events, c->rec_event_cap);
    const size_t ev_bytes = (size_t)c->rec_event_cap * sizeof(eh::ContactEvent);
    uint8_t *d = c->rec_dev[slot], *h = c->rec_host[slot];
    // EDYNHIP_SNAPSHOT_DIRECT: the pack kernels store straight into the pinned host slot (it is mapped into the device's address space) on the
    // stepper's stream - no copy engine, no second stream, no event between the two. A/B on one box (scripts/runs/r6l.sh): edyn::update in
    // sequential mode 705 -> 727 steps/s; in asynchronous mode, where nobody waits for the snapshot, the copy engine's overlap is worth as much
    // (785 / 759 against 764 / 765): the shim asks for it in its synchronous write-back only. (developer knob EDYNHIP_RECORDS_DIRECT=0 / 1 overrides)
    static const int direct_env = getenv("EDYNHIP_RECORDS_DIRECT") ? atoi(getenv("EDYNHIP_RECORDS_DIRECT")) : -1;
    const bool direct = direct_env >= 0 ? direct_env != 0 : (flags & EDYNHIP_SNAPSHOT_DIRECT) != 0;
    if (direct) {
        if (n) hipLaunchKernelGGL(k_pack_records, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->b, present_dt, (float4 *)(h + kRecHeader + ev_bytes));
        hipLaunchKernelGGL(k_pack_events, dim3(copy_events > 4096 ? 64 : 4), dim3(256), 0, c->stream, (const eh::ContactEvent *)c->events, (const uint32_t *)c->event_count,
                           c->event_cap, copy_events, (uint32_t *)h, (eh::ContactEvent *)(h + kRecHeader));
        EH_HIP(c, hipEventRecord(c->rec_event[slot], c->stream));
        c->rec_step[slot] = c->step_index; c->rec_bodies[slot] = n; c->rec_events_copied[slot] = copy_events; c->rec_last = slot;
        return EDYNHIP_OK;
    }
    // pack on the stepper's stream (it must see the finished steps), copy on the side stream: the copy engine moves the bytes
    // while the stepper's next kernels already run
    if (n) hipLaunchKernelGGL(k_pack_records, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->b, present_dt, (float4 *)(d + kRecHeader + ev_bytes));
    hipLaunchKernelGGL(k_pack_events, dim3(copy_events > 4096 ? 64 : 4), dim3(256), 0, c->stream, (const eh::ContactEvent *)c->events, (const uint32_t *)c->event_count,
                       c->event_cap, copy_events, (uint32_t *)d, (eh::ContactEvent *)(d + kRecHeader));
    EH_HIP(c, hipEventRecord(c->snap_ready, c->stream));
    EH_HIP(c, hipStreamWaitEvent(c->snap_stream, c->snap_ready, 0));
    EH_HIP(c, hipMemcpyAsync(h, d, kRecHeader + (size_t)copy_events * sizeof(eh::ContactEvent), hipMemcpyDeviceToHost, c->snap_stream));
    if (n) EH_HIP(c, hipMemcpyAsync(h + kRecHeader + ev_bytes, d + kRecHeader + ev_bytes, (size_t)n * sizeof(edynhip_body_record), hipMemcpyDeviceToHost, c->snap_stream));
    EH_HIP(c, hipEventRecord(c->rec_event[slot], c->snap_stream));
    c->rec_step[slot] = c->step_index; c->rec_bodies[slot] = n; c->rec_events_copied[slot] = copy_events; c->rec_last = slot;
    return EDYNHIP_OK;
}
int edynhip_snapshot_map(edynhip_ctx *c, edynhip_record_view *view) {
    if (!c || !view) return EDYNHIP_ERR_INVALID;
    if (c->rec_last < 0) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_snapshot_map: no record snapshot was taken");
    EH_HIP(c, hipSetDevice(c->device));
    const int slot = c->rec_last;
    EH_HIP(c, hipEventSynchronize(c->rec_event[slot]));   // that copy only - not the steps enqueued after it
    const uint8_t *h = c->rec_host[slot];
    const uint32_t *header = (const uint32_t *)h;
    view->records = (const edynhip_body_record *)(h + kRecHeader + (size_t)c->rec_event_cap * sizeof(eh::ContactEvent));
    view->num_bodies = c->rec_bodies[slot];
    view->step_index = c->rec_step[slot];
    view->events = c->events ? (const edynhip_contact_event *)(h + kRecHeader) : nullptr;
    view->total_events = header[0];
    view->num_events = header[1];
    return EDYNHIP_OK;
}

// ---- contact-event prefetch: the events of a step call handed over while its solve still runs
int edynhip_set_event_prefetch(edynhip_ctx *c, uint32_t max_events) {
    if (!c) return EDYNHIP_ERR_INVALID;
    if (!c->events) return set_error(c, EDYNHIP_ERR_UNSUPPORTED, "edynhip_set_event_prefetch: create the context with EDYNHIP_FLAG_CONTACT_EVENTS");
    EH_HIP(c, hipSetDevice(c->device));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->snap_stream) EH_HIP(c, hipStreamSynchronize(c->snap_stream));
    if (c->evp_host) { (void)hipHostFree(c->evp_host); c->evp_host = nullptr; }
    c->evp_max = std::min(max_events, c->event_cap); c->evp_state = 0;
    if (c->evp_max == 0) return EDYNHIP_OK;
    if (!c->snap_stream) {
        EH_HIP(c, hipStreamCreateWithFlags(&c->snap_stream, hipStreamNonBlocking));
        EH_HIP(c, hipEventCreateWithFlags(&c->snap_ready, hipEventDisableTiming));
    }
    if (!c->evp_np_done) { EH_HIP(c, hipEventCreateWithFlags(&c->evp_np_done, hipEventDisableTiming)); EH_HIP(c, hipEventCreateWithFlags(&c->evp_ready, hipEventDisableTiming)); }
    EH_HIP(c, hipHostMalloc((void **)&c->evp_host, 64 + (size_t)c->evp_max * sizeof(eh::ContactEvent), hipHostMallocDefault));
    std::memset(c->evp_host, 0, 64);
    return EDYNHIP_OK;
}
int edynhip_prefetched_events(edynhip_ctx *c, const edynhip_contact_event **events, uint32_t *num_events, uint32_t *total_events) {
    if (!c || !events || !num_events || !total_events) return EDYNHIP_ERR_INVALID;
    *events = nullptr; *num_events = 0; *total_events = 0;
    if (c->evp_max == 0 || c->evp_state == 0) return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_prefetched_events: no step call ran with the prefetch enabled");
    if (c->evp_state == 2) return EDYNHIP_OK;   // every island asleep: the call ran no step, nothing happened
    EH_HIP(c, hipSetDevice(c->device));
    EH_HIP(c, hipEventSynchronize(c->evp_ready));   // that copy only: the step's solve may still be running
    const uint32_t total = *(const uint32_t *)c->evp_host;
    *total_events = total;
    *num_events = std::min(std::min(total, c->event_cap), c->evp_max);
    *events = (const edynhip_contact_event *)(c->evp_host + 64);
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
    const uint32_t total = (uint32_t)c->host_joints.size();
    if (total == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    EH_TRY(flush_joint_redefs(c));
    std::memset(out, 0, (size_t)total * 10 * sizeof(float));
    const uint32_t n = c->j.n;
    if (n == 0) return EDYNHIP_OK;
    std::vector<float> imp((size_t)c->j.cap * kJointSlots), ang(n);
    std::vector<uint32_t> orig(n);
    EH_HIP(c, hipMemcpyAsync(imp.data(), c->j.impulse, imp.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipMemcpyAsync(ang.data(), c->j.angle, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipMemcpyAsync(orig.data(), c->j.orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t p = 0; p < n; ++p) {
        for (int r = 0; r < kJointApiSlots; ++r) out[10 * (size_t)orig[p] + r] = imp[(size_t)r * c->j.cap + p];
        out[10 * (size_t)orig[p] + 9] = ang[p];
    }
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

// ---- measurement aid: what this chip streams (the practical denominator printed next to the HBM spec peak)
__global__ void __launch_bounds__(256) k_bw_read(const float4 *__restrict__ src, size_t n, float4 *sink) {
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;   // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_bw_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
int edynhip_measure_bandwidth(edynhip_ctx *c, uint64_t bytes, float *read_gbs, float *copy_gbs) {
    if (!c || bytes < (1u << 20)) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)(bytes / sizeof(float4));
    float4 *a = nullptr, *b = nullptr;
    if (hipMalloc((void **)&a, n * sizeof(float4)) != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_measure_bandwidth: hipMalloc");
    if (hipMalloc((void **)&b, n * sizeof(float4)) != hipSuccess) { (void)hipFree(a); return set_error(c, EDYNHIP_ERR_HIP, "edynhip_measure_bandwidth: hipMalloc"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    if (err == hipSuccess) err = hipMemsetAsync(a, 0, n * sizeof(float4), c->stream);
    if (err == hipSuccess) err = hipMemsetAsync(b, 0, n * sizeof(float4), c->stream);
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    const dim3 grid((unsigned)ncu * 8u), block(256);
    float best_r = 0, best_c = 0;
    for (int rep = 0; rep < 6 && err == hipSuccess; ++rep) {   // the first pass of each is a warm-up
        float ms = 0;
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(k_bw_read, grid, block, 0, c->stream, a, n, b);
        (void)hipEventRecord(e1, c->stream);
        err = hipEventSynchronize(e1);
        if (err == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0 && rep > 0) best_r = std::max(best_r, (float)(n * sizeof(float4) / 1e6 / ms));
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(k_bw_copy, grid, block, 0, c->stream, a, b, n);
        (void)hipEventRecord(e1, c->stream);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        if (err == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0 && rep > 0) best_c = std::max(best_c, (float)(2.0 * n * sizeof(float4) / 1e6 / ms));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (err != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_measure_bandwidth", err);
    if (read_gbs) *read_gbs = best_r;
    if (copy_gbs) *copy_gbs = best_c;
    return EDYNHIP_OK;
}

}  // extern "C"
