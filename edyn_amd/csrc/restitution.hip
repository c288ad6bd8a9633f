// Restitution solver (src/edyn/dynamics/restitution_solver.cpp:31-408): shock propagation that runs before the
// constraint solver when any material has restitution > 0.
//
// Per island and iteration (settings.num_restitution_iterations, default 8): take the manifold (among those tagged
// contact_manifold_with_restitution) whose contact closes fastest; if it closes faster than 0.005 m/s, walk the island
// breadth-first from the faster of its two bodies and, at every procedural body reached, solve the manifolds around it
// that are still closing - normal rows carrying the contact's restitution plus their friction pairs, impulses from zero,
// num_individual_restitution_iterations Gauss-Seidel sweeps - and add the velocity changes to the bodies at once, so that
// bodies further along the walk already see the rebound of the ones before them.
//
// The walk is sequential by construction (that is what propagates the shock); islands are independent. So: one lane per
// island leader (label = lowest body index) runs the walk of its island, all islands in parallel; the per-manifold scan
// for the fastest closing contact and the adjacency build are manifold-parallel kernels. The reference walks its entity
// graph in adjacency-list order (an artefact of insertion history); every choice here is canonical instead - manifolds in
// ascending pair-key order (= manifold index), ties to the lower index, neighbours in that same order - which is what the
// CPU checker implements too. The row arithmetic is the reference's, operation for operation.
#include "ctx.hpp"
#include "dcollide.hpp"

namespace eh {
using namespace dm;

namespace {
inline uint32_t blocks(uint32_t n, uint32_t bs) { return (n + bs - 1) / bs; }
DI bool dyn(uint32_t flags) { return (flags & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC; }
constexpr float kRelvelThreshold = -0.005f;   // restitution_solver.cpp:138

struct RBody { f3 pos, org; q4 orn; f3 v, w; float inv_m; m3 inv_I; };   // org: where pivots are anchored (the origin of a body with a centre-of-mass offset)
DI RBody load_rbody(const Bodies &b, uint32_t i) {   // restitution_solver.cpp:166-220: kind decides velocity / mass / inertia
    RBody r;
    const float4 p = B_POS(b, i);
    const uint32_t fl = b.flags[i];
    r.pos = from4(p); r.org = B_ORG(b, i); r.orn = q_from4(B_ORN(b, i));
    if (dyn(fl)) { r.inv_m = p.w; r.inv_I = {from4(B_IW(b, i, 0)), from4(B_IW(b, i, 1)), from4(B_IW(b, i, 2))}; }
    else { r.inv_m = 0; r.inv_I = m3_zero(); }
    if ((fl & BF_KIND_MASK) == EDYNHIP_KIND_STATIC) { r.v = mk3(0, 0, 0); r.w = mk3(0, 0, 0); }
    else { r.v = from4(b.linvel[i]); r.w = from4(b.angvel[i]); }
    return r;
}
DI float eff_mass4(f3 J0, f3 J1, f3 J2, f3 J3, const RBody &A, const RBody &B) {
    const float s = dot(J0, J0) * A.inv_m + dot(mul(A.inv_I, J1), J1) + dot(J2, J2) * B.inv_m + dot(mul(B.inv_I, J3), J3);
    return 1.0f / s;
}
DI float rel_speed4(f3 J0, f3 J1, f3 J2, f3 J3, f3 vA, f3 wA, f3 vB, f3 wB) { return dot(J0, vA) + dot(J1, wA) + dot(J2, vB) + dot(J3, wB); }

// get_manifold_min_relvel, restitution_solver.cpp:31-81
DI float manifold_min_relvel(const Manifolds &mf, const Bodies &b, uint32_t m) {
    const uint32_t np = mf.info[m] & 0xFF;
    if (np == 0) return kScalarMax;
    const RBody A = load_rbody(b, mf.bodyA[m]), B = load_rbody(b, mf.bodyB[m]);
    float mn = kScalarMax;
    for (uint32_t k = 0; k < np; ++k) {
        const size_t pt = pt_at(mf.cap, k, m);
        const f3 pA = to_world(from4(mf.pA[pt]), A.org, A.orn), pB = to_world(from4(mf.pB[pt]), B.org, B.orn);
        const f3 rA = pA - A.pos, rB = pB - B.pos;
        const f3 velA = A.v + cross(A.w, rA), velB = B.v + cross(B.w, rB);
        mn = fminf(dot(velA - velB, from4(mf.nrm[pt])), mn);
    }
    return mn;
}
DI bool tagged(const Bodies &b, uint32_t a, uint32_t bb) {   // contact_manifold_with_restitution (constraint_util.cpp:83-101)
    if (const float *e = mix_lookup(b, a, bb)) return e[0] > kEps;
    return fminf(b.mat[a].y, b.mat[bb].y) > kEps;
}
DI bool manifold_asleep_r(const Bodies &b, uint32_t a, uint32_t bb) { return edge_asleep(b.flags[a], b.flags[bb]); }
DI uint32_t order_bits(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
DI float unorder_bits(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
}  // namespace

// ---- adjacency: every dynamic body's awake manifolds in ascending manifold index. Manifolds without points are edges of the
// graph too (the null_constraint of make_contact_manifold, constraint_util.cpp:76-78): the walk crosses them.
__global__ void k_radj_count(uint32_t M, Manifolds mf, Bodies b, uint32_t *deg) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t a = mf.bodyA[m], bb = mf.bodyB[m];
    if (manifold_asleep_r(b, a, bb)) return;   // island_view excludes sleeping islands (restitution_solver.cpp:390)
    if (dyn(b.flags[a])) atomicAdd(&deg[a], 1u);
    if (dyn(b.flags[bb])) atomicAdd(&deg[bb], 1u);
}
__global__ void k_radj_fill(uint32_t M, Manifolds mf, Bodies b, const uint32_t *__restrict__ off, uint32_t *cursor, uint32_t *adj) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t a = mf.bodyA[m], bb = mf.bodyB[m];
    if (manifold_asleep_r(b, a, bb)) return;
    if (dyn(b.flags[a])) adj[off[a] + atomicAdd(&cursor[a], 1u)] = m;
    if (dyn(b.flags[bb])) adj[off[bb] + atomicAdd(&cursor[bb], 1u)] = m;
}
__global__ void k_radj_sort(uint32_t n, const uint32_t *__restrict__ off, uint32_t *adj, uint32_t *cursor, uint64_t *best, uint32_t *visited) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    best[i] = ~0ull; visited[i] = 0;
    uint32_t *l = adj + off[i];
    const uint32_t d = off[i + 1] - off[i];
    for (uint32_t x = 1; x < d; ++x) {   // insertion sort: the lists are a dozen entries long
        const uint32_t v = l[x];
        uint32_t y = x;
        while (y > 0 && l[y - 1] > v) { l[y] = l[y - 1]; --y; }
        l[y] = v;
    }
    cursor[i] = 0;
}

// ---- per iteration: fastest closing tagged manifold of every island (ties: the lower manifold index)
__global__ void k_rest_find(uint32_t M, Manifolds mf, Bodies b, uint64_t *best) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || (mf.info[m] & 0xFF) == 0) return;
    const uint32_t a = mf.bodyA[m], bb = mf.bodyB[m];
    if (!tagged(b, a, bb) || manifold_asleep_r(b, a, bb)) return;
    const float r = manifold_min_relvel(mf, b, m);
    const uint32_t label = b.island[dyn(b.flags[a]) ? a : bb];
    atomicMin((unsigned long long *)&best[label], ((unsigned long long)order_bits(r) << 32) | m);
}

struct RestArgs {
    uint32_t n, individual_iterations, stamp;
    Manifolds mf; Bodies b;
    const uint32_t *off, *adj;
    uint8_t *star;          // per adjacency entry: belongs to the star being solved
    uint64_t *best;         // per island label: packed (relvel, manifold), reset after use
    uint32_t *visited;      // per body: stamp of the walk that reached it
    uint32_t *qnext;        // per body: next body in the walk's queue
    float4 *rimp;           // per point: restitution impulses (normal, friction 0, friction 1) - contact_point_impulse's separate set
};
// solve_manifolds (restitution_solver.cpp:149-314) for the star flagged around `node`
DI void solve_star(const RestArgs &a, uint32_t node) {
    const Manifolds &mf = a.mf;
    const Bodies &b = a.b;
    const uint32_t lo = a.off[node], hi = a.off[node + 1];
    for (uint32_t e = lo; e < hi; ++e) {   // rows start from zero impulses; the bodies' delta records are scratch here
        if (!a.star[e]) continue;
        const uint32_t m = a.adj[e], np = mf.info[m] & 0xFF;
        for (uint32_t k = 0; k < np; ++k) a.rimp[(size_t)k * mf.cap + m] = make_float4(0, 0, 0, 0);
        for (uint32_t side = 0; side < 2; ++side) {
            const uint32_t bi = side ? mf.bodyB[m] : mf.bodyA[m];
            B_DV(b, bi) = make_float4(0, 0, 0, B_DV(b, bi).w); B_DW(b, bi) = make_float4(0, 0, 0, 0);
        }
    }
    for (uint32_t it = 0; it < a.individual_iterations; ++it)
        for (uint32_t e = lo; e < hi; ++e) {
            if (!a.star[e]) continue;
            const uint32_t m = a.adj[e], np = mf.info[m] & 0xFF;
            const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
            const RBody A = load_rbody(b, ia), B = load_rbody(b, ib);   // velocities do not change until the star is done
            for (uint32_t k = 0; k < np; ++k) {
                const size_t s = slot_at(mf.cap, k, m), pt = pt_at(mf.cap, k, m);   // (s: the solver's own slot-major scratch; pt: the manifold's point record)
                const f3 n = from4(mf.nrm[pt]);
                const f3 pA = to_world(from4(mf.pA[pt]), A.org, A.orn), pB = to_world(from4(mf.pB[pt]), B.org, B.orn);
                const f3 rA = pA - A.pos, rB = pB - B.pos;
                f3 dvA = from4(B_DV(b, ia)), dwA = from4(B_DW(b, ia)), dvB = from4(B_DV(b, ib)), dwB = from4(B_DW(b, ib));
                if (!dyn(b.flags[ia])) { dvA = mk3(0, 0, 0); dwA = mk3(0, 0, 0); }   // dummy deltas of non-procedural bodies
                if (!dyn(b.flags[ib])) { dvB = mk3(0, 0, 0); dwB = mk3(0, 0, 0); }
                float4 imp = a.rimp[s];
                // normal row (prepare_row with options.restitution, solve, apply_row_impulse)
                {
                    const f3 J0 = n, J1 = cross(rA, n), J2 = -n, J3 = -cross(rB, n);
                    const float em = eff_mass4(J0, J1, J2, J3, A, B);
                    const float relvel = rel_speed4(J0, J1, J2, J3, A.v, A.w, B.v, B.w);
                    const float restitution = mf.lnrm[pt].w;
                    const float rhs = -(0.0f * 0.2f + relvel * (1 + restitution));
                    const float drel = rel_speed4(J0, J1, J2, J3, dvA, dwA, dvB, dwB);
                    float dimp = (rhs - drel) * em;
                    const float ni = imp.x + dimp;
                    if (ni < 0.0f) { dimp = 0.0f - imp.x; imp.x = 0.0f; }
                    else if (ni > kLarge) { dimp = kLarge - imp.x; imp.x = kLarge; }
                    else imp.x = ni;
                    dvA += A.inv_m * J0 * dimp; dvB += B.inv_m * J2 * dimp;
                    dwA += mul(A.inv_I, J1) * dimp; dwB += mul(B.inv_I, J3) * dimp;
                }
                // friction pair (solve_friction, constraint_row_friction.cpp:11-54)
                {
                    f3 t[2];
                    plane_space(n, t[0], t[1]);
                    const float mu = mf.pB[pt].w;
                    float di[2], ni[2];
                    f3 K1[2], K3[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        K1[q] = cross(rA, t[q]); K3[q] = -cross(rB, t[q]);
                        const float em = eff_mass4(t[q], K1[q], -t[q], K3[q], A, B);
                        const float rhs = -rel_speed4(t[q], K1[q], -t[q], K3[q], A.v, A.w, B.v, B.w);
                        const float drel = rel_speed4(t[q], K1[q], -t[q], K3[q], dvA, dwA, dvB, dwB);
                        di[q] = (rhs - drel) * em;
                        ni[q] = (q == 0 ? imp.y : imp.z) + di[q];
                    }
                    const float len2 = ni[0] * ni[0] + ni[1] * ni[1];
                    const float max_len = mu * imp.x;
                    if (len2 > square(max_len)) {
                        const float len = sqrtf(len2);
                        if (len > kEps) { ni[0] = ni[0] / len * max_len; ni[1] = ni[1] / len * max_len; }
                        else { ni[0] = 0; ni[1] = 0; }
                        di[0] = ni[0] - imp.y; di[1] = ni[1] - imp.z;
                    }
                    imp.y = ni[0]; imp.z = ni[1];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        dvA += A.inv_m * t[q] * di[q]; dwA += mul(A.inv_I, K1[q]) * di[q];
                        dvB += B.inv_m * (-t[q]) * di[q]; dwB += mul(B.inv_I, K3[q]) * di[q];
                    }
                }
                a.rimp[s] = imp;
                if (dyn(b.flags[ia])) { B_DV(b, ia) = to4(dvA, B_DV(b, ia).w); B_DW(b, ia) = to4(dwA, 0); }
                if (dyn(b.flags[ib])) { B_DV(b, ib) = to4(dvB, B_DV(b, ib).w); B_DW(b, ib) = to4(dwB, 0); }
            }
        }
    // apply delta velocities (restitution_solver.cpp:296-313); the delta records go back to zero
    for (uint32_t e = lo; e < hi; ++e) {
        if (!a.star[e]) continue;
        const uint32_t m = a.adj[e];
        for (uint32_t side = 0; side < 2; ++side) {
            const uint32_t bi = side ? mf.bodyB[m] : mf.bodyA[m];
            if (!dyn(b.flags[bi])) continue;
            f3 v = from4(b.linvel[bi]), w = from4(b.angvel[bi]);
            v += from4(B_DV(b, bi)); w += from4(B_DW(b, bi));
            b.linvel[bi] = to4(v, 0); b.angvel[bi] = to4(w, 0);
            B_DV(b, bi) = make_float4(0, 0, 0, B_DV(b, bi).w); B_DW(b, bi) = make_float4(0, 0, 0, 0);
        }
    }
}
// solve_restitution_iteration (restitution_solver.cpp:86-386): one lane per island leader
__global__ void k_rest_walk(RestArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t packed = a.best[i];
    a.best[i] = ~0ull;
    if (packed == ~0ull) return;                                   // not a leader, or no tagged manifold with points
    const float min_relvel = unorder_bits((uint32_t)(packed >> 32));
    if (min_relvel > kRelvelThreshold) return;                     // the island is solved
    const Bodies &b = a.b;
    const Manifolds &mf = a.mf;
    const uint32_t fm = (uint32_t)packed;
    const uint32_t fa = mf.bodyA[fm], fb = mf.bodyB[fm];
    const float sA = (b.flags[fa] & BF_KIND_MASK) == EDYNHIP_KIND_STATIC ? 0.0f : length_sqr(from4(b.linvel[fa]));
    const float sB = (b.flags[fb] & BF_KIND_MASK) == EDYNHIP_KIND_STATIC ? 0.0f : length_sqr(from4(b.linvel[fb]));
    uint32_t start;
    if (sA > sB) start = dyn(b.flags[fa]) ? fa : fb;
    else start = dyn(b.flags[fb]) ? fb : fa;
    // breadth-first over procedural bodies (entity_graph.hpp:356-422): intrusive FIFO through qnext
    uint32_t head = start, tail = start;
    a.visited[start] = a.stamp;
    a.qnext[start] = 0xFFFFFFFFu;
    while (head != 0xFFFFFFFFu) {
        const uint32_t node = head;
        const uint32_t lo = a.off[node], hi = a.off[node + 1];
        bool any = false;
        for (uint32_t e = lo; e < hi; ++e) {
            const bool closing = manifold_min_relvel(mf, b, a.adj[e]) < kRelvelThreshold;
            a.star[e] = closing ? 1 : 0;
            any |= closing;
        }
        if (any) solve_star(a, node);
        for (uint32_t e = lo; e < hi; ++e) {
            const uint32_t m = a.adj[e];
            const uint32_t o = mf.bodyA[m] == node ? mf.bodyB[m] : mf.bodyA[m];
            if (dyn(b.flags[o]) && a.visited[o] != a.stamp) {
                a.visited[o] = a.stamp;
                a.qnext[o] = 0xFFFFFFFFu;
                a.qnext[tail] = o; tail = o;
            }
        }
        head = a.qnext[node];
    }
}

int restitution(edynhip_ctx *c) {
    if (!c->has_restitution || c->restitution_iterations == 0) return EDYNHIP_OK;
    const uint32_t n = c->b.n, M = c->num_manifolds;
    if (n == 0 || M == 0) return EDYNHIP_OK;
    hipStream_t s = c->stream;
    Manifolds &mf = c->m[c->cur];
    if (!c->radj) {   // first use: scratch sized for the world's capacities
        auto alloc = [&](void **p, size_t bytes) -> int {
            EH_HIP(c, hipMalloc(p, bytes));
            EH_HIP(c, hipMemsetAsync(*p, 0, bytes, s));
            c->allocs.push_back(*p);
            return EDYNHIP_OK;
        };
        const size_t nb = c->b.cap, cap = mf.cap;
        EH_TRY(alloc((void **)&c->rdeg, (nb + 1) * sizeof(uint32_t))); EH_TRY(alloc((void **)&c->roff, (nb + 1) * sizeof(uint32_t)));
        EH_TRY(alloc((void **)&c->rcursor, nb * sizeof(uint32_t))); EH_TRY(alloc((void **)&c->radj, 2 * cap * sizeof(uint32_t)));
        EH_TRY(alloc((void **)&c->rstar, 2 * cap)); EH_TRY(alloc((void **)&c->rbest, nb * sizeof(uint64_t)));
        EH_TRY(alloc((void **)&c->rvisited, nb * sizeof(uint32_t))); EH_TRY(alloc((void **)&c->rqnext, nb * sizeof(uint32_t)));
        EH_TRY(alloc((void **)&c->rimp, cap * kMaxPts * sizeof(float4)));
    }
    EH_HIP(c, hipMemsetAsync(c->rdeg, 0, ((size_t)n + 1) * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_radj_count, dim3(blocks(M, 256)), dim3(256), 0, s, M, mf, c->b, c->rdeg);
    EH_TRY(scan_u32(c, c->rdeg, c->roff, n + 1));
    hipLaunchKernelGGL(k_radj_fill, dim3(blocks(M, 256)), dim3(256), 0, s, M, mf, c->b, c->roff, c->rcursor, c->radj);
    hipLaunchKernelGGL(k_radj_sort, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->roff, c->radj, c->rcursor, c->rbest, c->rvisited);
    // Every iteration re-examines every island; an island that is solved costs its leader one load. The reference stops
    // once an iteration finds all islands solved - running the remaining ones changes nothing (no island passes the test).
    for (uint32_t it = 0; it < c->restitution_iterations; ++it) {
        hipLaunchKernelGGL(k_rest_find, dim3(blocks(M, 256)), dim3(256), 0, s, M, mf, c->b, c->rbest);
        RestArgs a{n, c->individual_restitution_iterations, it + 1, mf, c->b, c->roff, c->radj, c->rstar, c->rbest, c->rvisited, c->rqnext, c->rimp};
        hipLaunchKernelGGL(k_rest_walk, dim3(blocks(n, 64)), dim3(64), 0, s, a);
    }
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

}  // namespace eh
