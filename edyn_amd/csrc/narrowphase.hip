// Narrowphase on the GPU, one contact manifold per lane:
//   update_contact_distances   src/edyn/util/collision_util.cpp:28-45
//   detect_collision           src/edyn/util/collision_util.cpp:440-475 (+ collide/*.cpp via dcollide.hpp)
//   process_collision          include/edyn/util/collision_util.hpp:104-276
//   find_nearest_contact[_rolling], merge_point, create_contact_point, should_remove_point
//                              src/edyn/util/collision_util.cpp:205-280,319-413
// The reference keeps every contact point as an EnTT entity in a linked list (newest first); here a
// manifold owns four fixed slots kept in that same list order, so creation/destruction is a register
// shuffle instead of entity churn.
#include "ctx.hpp"
#include "dpolyhedron.hpp"

namespace eh {
using namespace dc;

struct BodyIn { f3 pos; q4 orn; f3 angvel; float friction, restitution; bool rolling; };

DI f3 local_normal(const CPoint &rp, const BodyIn &A, const BodyIn &B) {
    return rp.attachment != NA_NONE ? rotate(conjugate(rp.attachment == NA_ON_A ? A.orn : B.orn), rp.normal) : mk3(0, 0, 0);
}
DI int find_nearest(const CPoint &cp, const CPoint (&rs)[4], int R) {
    float best = square(kCachingThreshold);
    int idx = R;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < R) {
            float dA = length_sqr(rs[i].pivotA - cp.pivotA);
            float dB = length_sqr(rs[i].pivotB - cp.pivotB);
            if (dA < best) { best = dA; idx = i; }
            if (dB < best) { best = dB; idx = i; }
        }
    }
    return idx;
}
DI int find_nearest_rolling(const CPoint (&rs)[4], int R, f3 cp_pivot, f3 origin, q4 orn, f3 angvel, float dt) {
    int idx = R;
    q4 prev_orn = integrate(orn, angvel, -dt);
    f3 prev_pivot = to_world(cp_pivot, origin, prev_orn);
    float best = square(kCachingThreshold);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < R) {
            f3 pA = to_world(rs[i].pivotA, origin, orn);   // pivotA for both bodies, as the reference does
            float d2 = distance_sqr(pA, prev_pivot);
            if (d2 < best) { best = d2; idx = i; }
        }
    }
    return idx;
}
DI bool should_remove(const CPoint &cp, const BodyIn &A, const BodyIn &B) {
    const float thr = kBreakingThreshold, thr2 = thr * thr;
    f3 d = to_world(cp.pivotA, A.pos, A.orn) - to_world(cp.pivotB, B.pos, B.orn);
    float nd = dot(d, cp.normal);
    f3 td = d - nd * cp.normal;
    return nd > thr || length_sqr(td) > thr2;
}

// Two kernels, so that neither has to hold the closest-feature search and the manifold bookkeeping in registers at once:
//   k_np_detect  detect_collision: AABB pre-check + collide() -> the raw result (<= 4 points) into a staging array
//   k_np_merge   update_contact_distances + process_collision: old points x result -> the manifold's new point list
// Staging layout per result point slot k of manifold m, at [k * cap + m]: ra = (pivotA, distance), rb = (pivotB, bits(attachment)),
// rn = (normal, -); rnum[m] = number of result points.
struct Staging { float4 *ra, *rb, *rn; uint32_t *rnum; };

__global__ void __launch_bounds__(64, 2)
k_np_detect(uint32_t M, Manifolds mf, Bodies b, bool sleeping, Staging st) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
    const uint32_t fa = b.flags[ia], fb = b.flags[ib];
    if (sleeping && edge_asleep(fa, fb)) { st.rnum[m] = 0; return; }
    CResult res;
    res.num = 0;
    const box3 ba{from4(b.amin[ia]), from4(b.amax[ia])}, bbx{from4(b.amin[ib]), from4(b.amax[ib])};
    if (intersect(inset(ba, -kBreakingThreshold), bbx)) {
        const int tA = (int)((fa & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT), tB = (int)((fb & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
        Ctx ctx{B_ORG(b, ia), q_from4(B_ORN(b, ia)), B_ORG(b, ib), q_from4(B_ORN(b, ib)), kCollisionThreshold};   // shapes sit at the origin (collision_util.cpp:451-465)
        collide(tA, b.shape[ia], tB, b.shape[ib], ctx, res);
    }
    st.rnum[m] = (uint32_t)res.num;
#pragma unroll
    for (int k = 0; k < kMaxPts; ++k) {
        if (k >= res.num) break;
        const size_t d = (size_t)k * mf.cap + m;
        st.ra[d] = to4(res.pt[k].pivotA, res.pt[k].distance);
        st.rb[d] = to4(res.pt[k].pivotB, __int_as_float(res.pt[k].attachment));
        st.rn[d] = to4(res.pt[k].normal, 0.0f);
    }
}

// The same for the manifolds that involve a cylinder (dcylinder.hpp): launched after k_np_detect - which leaves such pairs
// without points - and only in worlds that have a cylinder; every other lane returns at once.
__global__ void __launch_bounds__(64)
k_np_detect_ext(uint32_t M, Manifolds mf, Bodies b, bool sleeping, Staging st) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
    const uint32_t fa = b.flags[ia], fb = b.flags[ib];
    const int tA = (int)((fa & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT), tB = (int)((fb & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
    if (tA != SHAPE_CYLINDER && tB != SHAPE_CYLINDER) return;
    if (sleeping && edge_asleep(fa, fb)) return;
    const box3 ba{from4(b.amin[ia]), from4(b.amax[ia])}, bbx{from4(b.amin[ib]), from4(b.amax[ib])};
    if (!intersect(inset(ba, -kBreakingThreshold), bbx)) return;
    CResult res;
    res.num = 0;
    Ctx ctx{B_ORG(b, ia), q_from4(B_ORN(b, ia)), B_ORG(b, ib), q_from4(B_ORN(b, ib)), kCollisionThreshold};
    collide_ext(tA, b.shape[ia], tB, b.shape[ib], ctx, res);
    st.rnum[m] = (uint32_t)res.num;
    for (int k = 0; k < res.num; ++k) {
        const size_t d = (size_t)k * mf.cap + m;
        st.ra[d] = to4(res.pt[k].pivotA, res.pt[k].distance);
        st.rb[d] = to4(res.pt[k].pivotB, __int_as_float(res.pt[k].attachment));
        st.rn[d] = to4(res.pt[k].normal, 0.0f);
    }
}

// The same for the manifolds that involve a polyhedron (dpolyhedron.hpp), in worlds that have one; k_update_rotated ran before.
// Polyhedron pairs are binned by what they pair - (mesh, mesh), (mesh, other shape) - before the routines run: one lane per manifold
// in manifold order put a cube-wedge pair beside a prism-plane pair in the same wave, and a wave executes the union of its lanes' loops
// (SQ counters: 38k VALU instructions per wave for ~10k per lane). k_poly_count bins the manifolds that involve a polyhedron and pass
// the box test (LDS histogram per workgroup), k_poly_scan turns the counts into offsets, k_poly_scatter writes the manifold indices bin
// by bin (order inside a bin is arbitrary: every entry writes its own manifold's staging slot), k_np_detect_poly walks that list:
// 2.01 -> 1.20 ms on the 32k-polyhedron heap (+0.055 ms for the three binning kernels). Measured and dropped: a first pass that tries only
// the cheap separating axes and compacts the survivors (56 % of the listed pairs are separated along a face normal) so that the full
// routine runs in full waves - 0.25 + 1.10 ms: the survivors' waves diverge more (support polygons, clipping cases) and run one per SIMD.
// Round 5: the bins of polyhedron-polyhedron pairs come first in the list (keys below kPolyPairBins) and are walked by their own kernel,
// k_np_detect_pp, a GROUP of lanes per pair (dpolyhedron.hpp collide_polyhedron_polyhedron_group); the other pairs keep one lane each.
constexpr uint32_t kPolyBins = 1024;
constexpr uint32_t kPolyPairBins = 24 * 24;   // (mesh id mod 24) x (mesh id mod 24); then 2 x 24 x 8: (which side holds the polyhedron, its class, the other body's shape type)
struct PolyBins { uint32_t *count, *start, *cursor, *total, *key, *list, *hint; };   // count / start / cursor: [kPolyBins]; key, list, hint: [max_manifolds]
// hint: per manifold of THIS manifold array, the separating axis that decided the pair last time (dpolyhedron.hpp pp_hint_separates); one
// array per manifold array, carried over by k_poly_count through prev_idx when the broadphase rebuilt the manifolds.
DI uint32_t poly_bin(int tA, float4 sA, int tB, float4 sB) {
    const uint32_t cA = (uint32_t)sA.x % 24u, cB = (uint32_t)sB.x % 24u;
    if (tA == SHAPE_POLYHEDRON && tB == SHAPE_POLYHEDRON) return cA * 24u + cB;
    return tA == SHAPE_POLYHEDRON ? kPolyPairBins + cA * 8u + ((uint32_t)tB & 7u) : kPolyPairBins + 192u + cB * 8u + ((uint32_t)tA & 7u);
}
__global__ void __launch_bounds__(256) k_poly_count(uint32_t M, Manifolds mf, Bodies b, bool sleeping, PolyBins pb, Meshes meshes, const uint32_t *hint_prev, bool rebuilt, bool use_hints) {
    __shared__ uint32_t hist[kPolyBins];
    for (uint32_t k = threadIdx.x; k < kPolyBins; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = 0xFFFFFFFFu;
    if (m < M) {
        const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
        const uint32_t fa = b.flags[ia], fb = b.flags[ib];
        const int tA = (int)((fa & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT), tB = (int)((fb & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
        if ((tA == SHAPE_POLYHEDRON || tB == SHAPE_POLYHEDRON) && !(sleeping && edge_asleep(fa, fb))) {
            const box3 ba{from4(b.amin[ia]), from4(b.amax[ia])}, bbx{from4(b.amin[ib]), from4(b.amax[ib])};
            if (intersect(inset(ba, -kBreakingThreshold), bbx)) key = poly_bin(tA, b.shape[ia], tB, b.shape[ib]);
        }
        // the pair's hint follows it into a rebuilt manifold array; a polyhedron pair that its hinted axis still separates is done (k_np_detect
        // left it without points) and stays out of the list
        uint32_t hint = 0;
        if (rebuilt) { const uint32_t p = mf.prev_idx[m]; if (p < mf.cap) hint = hint_prev[p]; } else hint = pb.hint[m];
        pb.hint[m] = hint;
        if (use_hints && hint != 0 && key < kPolyPairBins &&
            pp_hint_separates(meshes, pp_side(meshes, b.shape[ia], meshes.rot + meshes.rot_off[ia]), pp_side(meshes, b.shape[ib], meshes.rot + meshes.rot_off[ib]),
                              B_ORG(b, ib) - B_ORG(b, ia), kCollisionThreshold, hint)) key = 0xFFFFFFFFu;
        pb.key[m] = key;
    }
    if (key != 0xFFFFFFFFu) atomicAdd(&hist[key], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kPolyBins; k += 256) if (hist[k]) atomicAdd(&pb.count[k], hist[k]);
}
__global__ void __launch_bounds__(1024) k_poly_scan(PolyBins pb) {   // exclusive scan of the 1024 bin counts, one workgroup
    __shared__ uint32_t part[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t v = pb.count[t];
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t u = __shfl_up(inc, off); if ((int)lane >= off) inc += u; }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += part[w];
    pb.start[t] = base + inc - v;
    pb.cursor[t] = 0;
    pb.count[t] = 0;   // for the next step
    if (t == 1023) *pb.total = base + inc;
}
__global__ void __launch_bounds__(256) k_poly_scatter(uint32_t M, PolyBins pb) {
    __shared__ uint32_t hist[kPolyBins];   // first the workgroup's count per bin, then the start of its reserved range
    for (uint32_t k = threadIdx.x; k < kPolyBins; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    const uint32_t key = m < M ? pb.key[m] : 0xFFFFFFFFu;
    uint32_t rank = 0;
    if (key != 0xFFFFFFFFu) rank = atomicAdd(&hist[key], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kPolyBins; k += 256) if (hist[k]) hist[k] = pb.start[k] + atomicAdd(&pb.cursor[k], hist[k]);
    __syncthreads();
    if (key != 0xFFFFFFFFu) pb.list[hist[key] + rank] = m;
}
// (register budget: 1 / 2 / 4 / 8 / 16 workgroups per CU as the launch bound gave 231 / 251 / 240 / 235 / 220 steps/s on polyheap32k -
// two leaves the allocator enough registers to keep the polygons' scalars out of scratch and still lets a second wave hide latency)
__global__ void __launch_bounds__(64, 2)
k_np_detect_poly(uint32_t M, Manifolds mf, Bodies b, Staging st, Meshes meshes, PolyBins pb, bool pp_elsewhere) {
    const uint32_t i = (pp_elsewhere ? pb.start[kPolyPairBins] : 0u) + blockIdx.x * blockDim.x + threadIdx.x;   // polyhedron pairs: k_np_detect_pp
    if (i >= M || i >= *pb.total) return;
    const uint32_t m = pb.list[i];
    const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
    const uint32_t fa = b.flags[ia], fb = b.flags[ib];
    const int tA = (int)((fa & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT), tB = (int)((fb & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
    CResult res;
    res.num = 0;
    Ctx ctx{B_ORG(b, ia), q_from4(B_ORN(b, ia)), B_ORG(b, ib), q_from4(B_ORN(b, ib)), kCollisionThreshold};
    collide_poly(meshes, tA, b.shape[ia], tA == SHAPE_POLYHEDRON ? meshes.rot + meshes.rot_off[ia] : nullptr, tB, b.shape[ib],
                 tB == SHAPE_POLYHEDRON ? meshes.rot + meshes.rot_off[ib] : nullptr, ctx, res);
    st.rnum[m] = (uint32_t)res.num;
    for (int k = 0; k < res.num; ++k) {
        const size_t d = (size_t)k * mf.cap + m;
        st.ra[d] = to4(res.pt[k].pivotA, res.pt[k].distance);
        st.rb[d] = to4(res.pt[k].pivotB, __int_as_float(res.pt[k].attachment));
        st.rn[d] = to4(res.pt[k].normal, 0.0f);
    }
}

// Polyhedron-polyhedron pairs, G lanes per pair, in two kernels (round 5). The binned list's first pb.start[kPolyPairBins] entries are such pairs
// (the count lives on the device: fixed grids stride over them).
//   k_np_pp_axes      the separating axes (dpolyhedron.hpp pp_group_axes): no LDS, few registers - many waves per SIMD hide its dependent
//                     loads (list -> bodies -> mesh tables -> rotated vertices). A pair within the threshold leaves its axis in its staging
//                     slot and its manifold index in the survivor list (the key array, free once the scatter ran; counter: the cursor of
//                     the unused bin 1023, zeroed by k_poly_scan); a separated pair is done (no points).
//   k_np_pp_contacts  support polygons, hulls, clipping of the survivors (pp_group_contacts), polygons in LDS.
// Profile that decided the split (one kernel, G = 16, polyheap32k: 300-400k pairs per step): faces 8.8 us + edges 8.1 us of dependent-load
// latency per pair-iteration at two waves per SIMD (251 VGPRs, the contact half's), polygons + hulls + contacts 3.3 us, 38 % of the pairs survive.
DI PPSide pp_side_of(const Meshes &meshes, const Bodies &b, uint32_t body) { return pp_side(meshes, b.shape[body], meshes.rot + meshes.rot_off[body]); }
template <int G, bool PROF>
__global__ void __launch_bounds__(64)
k_np_pp_axes(Manifolds mf, Bodies b, Staging st, Meshes meshes, PolyBins pb, unsigned long long *prof_out) {
    constexpr uint32_t kGroups = 64u / G;
    const uint32_t g = threadIdx.x / G;
    const uint32_t n = pb.start[kPolyPairBins];
    PPProf<PROF> prof;
    if (PROF) for (int k = 0; k < 8; ++k) prof.t[k] = 0;
    uint32_t iters = 0;
    for (uint32_t i = blockIdx.x * kGroups + g; i < n; i += gridDim.x * kGroups) {
        prof.stamp(0);
        const uint32_t m = pb.list[i];
        const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
        PPSeparation sep;
        const bool touching = pp_group_axes<G>(meshes, pp_side_of(meshes, b, ia), pp_side_of(meshes, b, ib), B_ORG(b, ib) - B_ORG(b, ia), kCollisionThreshold, sep, prof);
        if (threadIdx.x % G == 0) {
            pb.hint[m] = sep.hint;
            if (touching) {
                pb.key[atomicAdd(&pb.cursor[kPolyBins - 1], 1u)] = m;
                st.ra[m] = to4(sep.axis, sep.distance);
                st.rb[m] = make_float4(sep.projectionA, sep.projectionB, 0.0f, 0.0f);
            } else {
                st.rnum[m] = 0;
            }
        }
        prof.stamp(3);
        ++iters;
    }
    if (PROF && threadIdx.x % G == 0) {   // [0]: pairs; [1..3]: ticks of setup + faces, edges, stores + waiting for the wave's other groups
        atomicAdd(&prof_out[0], (unsigned long long)iters);
        for (int k = 1; k < 4; ++k) atomicAdd(&prof_out[k], (unsigned long long)(prof.t[k] - prof.t[k - 1]));
    }
}
template <int G, bool PROF>
__global__ void __launch_bounds__(64)
k_np_pp_contacts(Manifolds mf, Bodies b, Staging st, Meshes meshes, PolyBins pb, unsigned long long *prof_out) {
    constexpr uint32_t kGroups = 64u / G;
    __shared__ __align__(16) unsigned char lds_raw[kGroups * sizeof(PPLds)];
    const uint32_t g = threadIdx.x / G;
    PPLds &lds = reinterpret_cast<PPLds *>(lds_raw)[g];
    const uint32_t n = pb.cursor[kPolyBins - 1];
    PPProf<PROF> prof;
    if (PROF) for (int k = 0; k < 8; ++k) prof.t[k] = 0;
    uint32_t iters = 0;
    for (uint32_t i = blockIdx.x * kGroups + g; i < n; i += gridDim.x * kGroups) {
        prof.stamp(2);
        const uint32_t m = pb.key[i];
        const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
        const float4 r0 = st.ra[m], r1 = st.rb[m];
        const PPSeparation sep{from4(r0), r0.w, r1.x, r1.y};
        CResult res;
        res.num = 0;
        const Ctx ctx{B_ORG(b, ia), q_from4(B_ORN(b, ia)), B_ORG(b, ib), q_from4(B_ORN(b, ib)), kCollisionThreshold};
        pp_group_contacts<G>(pp_side_of(meshes, b, ia), pp_side_of(meshes, b, ib), ctx, sep, lds, res, prof);
        if (threadIdx.x % G == 0) {
            st.rnum[m] = (uint32_t)res.num;
            for (int k = 0; k < res.num; ++k) {
                const size_t d = (size_t)k * mf.cap + m;
                st.ra[d] = to4(res.pt[k].pivotA, res.pt[k].distance);
                st.rb[d] = to4(res.pt[k].pivotB, __int_as_float(res.pt[k].attachment));
                st.rn[d] = to4(res.pt[k].normal, 0.0f);
            }
        }
        pp_group_sync();   // the next pair's polygons overwrite this one's
        prof.stamp(6);
        ++iters;
    }
    if (PROF && threadIdx.x % G == 0) {   // [4]: survivors; [5..8]: ticks of setup + polygons, hulls, contacts, stores + waiting
        atomicAdd(&prof_out[4], (unsigned long long)iters);
        for (int k = 3; k < 7; ++k) atomicAdd(&prof_out[k + 2], (unsigned long long)(prof.t[k] - prof.t[k - 1]));
    }
}

__global__ void __launch_bounds__(64, 2)
k_np_merge(uint32_t M, Manifolds mf, Bodies b, float dt, bool sleeping, Manifolds old, bool points_in_old, Staging st, EventSink ev) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) {
        const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
        const uint32_t info = mf.info[m];
        const int n_old = (int)(info & 0xFF);
        const uint32_t fa = b.flags[ia], fb = b.flags[ib];
        // old contact points: in this array (copied by the broadphase), or still in the previous array
        const Manifolds &src = points_in_old ? old : mf;
        const uint32_t sidx = points_in_old ? mf.prev_idx[m] : m;
        if (sleeping && edge_asleep(fa, fb)) {   // narrowphase.cpp:31: sleeping manifolds are left as they are
            if (points_in_old)
                for (int k = 0; k < n_old; ++k) {
                    const size_t ts = pt_at(src.cap, (uint32_t)k, sidx), td = pt_at(mf.cap, (uint32_t)k, m);
                    mf.pA[td] = src.pA[ts]; mf.pB[td] = src.pB[ts]; mf.nrm[td] = src.nrm[ts]; mf.lnrm[td] = src.lnrm[ts]; mf.imp[td] = src.imp[ts];
                    const size_t s = slot_at(src.cap, (uint32_t)k, sidx), d = slot_at(mf.cap, (uint32_t)k, m);
                    if (mf.pid) mf.pid[d] = src.pid[s];
                    if (mf.xmat) { mf.xmat[d] = src.xmat[s]; mf.ximp[d] = src.ximp[s]; }
                }
            return;
        }
        const int tA = (int)((fa & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT), tB = (int)((fb & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
        BodyIn A, B;
        { A.pos = B_ORG(b, ia); A.orn = q_from4(B_ORN(b, ia)); A.angvel = from4(b.angvel[ia]);
          float2 mt = b.mat[ia]; A.friction = mt.x; A.restitution = mt.y;
          A.rolling = (fa & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC && (tA == SHAPE_SPHERE || tA == SHAPE_CAPSULE || tA == SHAPE_CYLINDER); }   // rolling_shapes_tuple_t
        { B.pos = B_ORG(b, ib); B.orn = q_from4(B_ORN(b, ib)); B.angvel = from4(b.angvel[ib]);
          float2 mt = b.mat[ib]; B.friction = mt.x; B.restitution = mt.y;
          B.rolling = (fb & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC && (tB == SHAPE_SPHERE || tB == SHAPE_CAPSULE || tB == SHAPE_CYLINDER); }

        // Old points: only what the matching needs is held across it (pivots, normal, distance); friction, restitution, the
        // local normal, the warm-start impulses and the lifetime are re-read just before the stores.
        CPoint pts[kMaxPts];
#pragma unroll
        for (int k = 0; k < kMaxPts; ++k) {
            if (k < n_old) {
                const size_t s = pt_at(src.cap, (uint32_t)k, sidx);
                const float4 a = src.pA[s], bb = src.pB[s], n = src.nrm[s];
                CPoint &p = pts[k];
                p.pivotA = from4(a); p.pivotB = from4(bb); p.normal = from4(n); p.attachment = __float_as_int(n.w);
                // update_contact_distances
                p.distance = dot(p.normal, to_world(p.pivotA, A.pos, A.orn) - to_world(p.pivotB, B.pos, B.orn));
            } else {
                pts[k] = CPoint{};
            }
        }

        const int R = (int)st.rnum[m];
        CPoint rs[kMaxPts];
#pragma unroll
        for (int k = 0; k < kMaxPts; ++k) {
            if (k < R) {
                const size_t d = (size_t)k * mf.cap + m;
                const float4 ra = st.ra[d], rb = st.rb[d], rn = st.rn[d];
                rs[k] = CPoint{from4(ra), from4(rb), from4(rn), ra.w, __float_as_int(rb.w)};
            } else {
                rs[k] = CPoint{};
            }
        }

        // ---- process_collision ----
        uint32_t merged = 0, dead = 0, changed = 0;   // bit r: result point r consumed; bit i: old point i removed / rewritten
        int num_points = n_old;
#pragma unroll
        for (int i = 0; i < kMaxPts; ++i) {
            if (i < n_old) {
                CPoint &cp = pts[i];
                int nearest = find_nearest(cp, rs, R);
                if (nearest == R && A.rolling) nearest = find_nearest_rolling(rs, R, cp.pivotA, A.pos, A.orn, A.angvel, dt);
                if (nearest == R && B.rolling) nearest = find_nearest_rolling(rs, R, cp.pivotB, B.pos, B.orn, B.angvel, dt);
                if (nearest < R && !((merged >> nearest) & 1u)) { cp = pick4(rs, nearest); changed |= 1u << i; merged |= 1u << nearest; }
                else if (should_remove(cp, A, B)) { dead |= 1u << i; --num_points; }
            }
        }
        const uint32_t all_r = (1u << R) - 1u;

        CPoint lp[kMaxPts];
        uint32_t create = 0;   // bit i: local slot i becomes a new contact point
        if ((merged & all_r) != all_r) {
            int lold[kMaxPts] = {-1, -1, -1, -1};
            int ltype[kMaxPts] = {INS_NONE, INS_NONE, INS_NONE, INS_NONE};
#pragma unroll
            for (int i = 0; i < kMaxPts; ++i) lp[i] = CPoint{};
            if (num_points > 0) {
                int k = 0;
#pragma unroll
                for (int i = 0; i < kMaxPts; ++i) {
                    if (i < n_old && !((dead >> i) & 1u)) {
                        CPoint loc = pts[i];
                        loc.attachment = NA_NONE;
                        put4(lp, k, loc);
                        put4(lold, k, i);
                        ++k;
                    }
                }
            } else {
                num_points = 1;
                lp[0] = rs[0];
                ltype[0] = INS_APPEND;
                merged |= 1u;
            }
#pragma unroll
            for (int r = 0; r < kMaxPts; ++r) {
                if (r < R && !((merged >> r) & 1u)) {
                    const CPoint rp = rs[r];
                    f3 piv[kMaxPts];
#pragma unroll
                    for (int i = 0; i < kMaxPts; ++i) piv[i] = lp[i].pivotA;
                    int ir = insertion_point_index(piv, num_points, rp.pivotA);
                    if ((ir & 0xFF) == INS_NONE) {
#pragma unroll
                        for (int i = 0; i < kMaxPts; ++i) piv[i] = lp[i].pivotB;
                        ir = insertion_point_index(piv, num_points, rp.pivotB);
                    }
                    if ((ir & 0xFF) != INS_NONE) { put4(lp, ir >> 8, rp); put4(ltype, ir >> 8, ir & 0xFF); }
                }
            }
#pragma unroll
            for (int i = 0; i < kMaxPts; ++i) {
                if (i < num_points) {
                    const int ty = ltype[i], lo = lold[i];
                    if (ty == INS_APPEND || (ty == INS_SIMILAR && lo < 0)) create |= 1u << i;
                    else if (ty == INS_SIMILAR) {
#pragma unroll
                        for (int j = 0; j < kMaxPts; ++j) sel(pts[j], j == lo, lp[i]);
                        changed |= 1u << lo;
                    } else if (ty == INS_REPLACE) {
                        if (lo >= 0) dead |= 1u << lo;
                        create |= 1u << i;
                    }
                }
            }
        }
        // new points go to the head of the list in creation order (the last created first); survivors keep their order
        // the survivors' remaining fields, read before anything is written (the source may be this very array)
        float4 ext_l[kMaxPts], ext_i[kMaxPts];
        float ext_f[kMaxPts];
        uint64_t ext_id[kMaxPts] = {0, 0, 0, 0};
        if (mf.pid) {   // contact events: ids of the old points; the ones that end here are reported
#pragma unroll
            for (int i = 0; i < kMaxPts; ++i) {
                if (i < n_old) {
                    ext_id[i] = src.pid[(size_t)i * src.cap + sidx];
                    if ((dead >> i) & 1u) emit_event(ev, EDYNHIP_EVENT_POINT_DESTROYED, ia, ib, ext_id[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxPts; ++i) {
            if (i < n_old && !((dead >> i) & 1u)) {
                const size_t s = pt_at(src.cap, (uint32_t)i, sidx);
                ext_l[i] = src.lnrm[s]; ext_i[i] = src.imp[s]; ext_f[i] = src.pB[s].w;
            } else {
                ext_l[i] = ext_i[i] = make_float4(0, 0, 0, 0); ext_f[i] = 0;
            }
        }
        int n_out = 0;
        if (create) {
            float friction = sqrtf(A.friction * B.friction);             // material_mixing.hpp:16-18
            float restitution = fminf(A.restitution, B.restitution);     // :12-14
            if (const float *e = mix_lookup(b, ia, ib)) { restitution = e[0]; friction = e[1]; }   // assign_material_properties, collision_util.cpp:293-299
#pragma unroll
            for (int i = kMaxPts - 1; i >= 0; --i) {
                if ((create >> i) & 1u) {
                    const CPoint &p = lp[i];
                    const size_t d = pt_at(mf.cap, (uint32_t)n_out, m);
                    mf.pA[d] = to4(p.pivotA, p.distance);
                    mf.pB[d] = to4(p.pivotB, friction);
                    mf.nrm[d] = to4(p.normal, __int_as_float(p.attachment));
                    mf.lnrm[d] = to4(local_normal(p, A, B), restitution);
                    mf.imp[d] = make_float4(0, 0, 0, __uint_as_float(0u));
                    if (mf.pid) {   // id = (step of creation + 1) << 32 | manifold index << 2 | local slot
                        const uint64_t id = ((uint64_t)(ev.step + 1u) << 32) | ((uint64_t)m << 2) | (uint64_t)i;
                        mf.pid[slot_at(mf.cap, (uint32_t)n_out, m)] = id;
                        emit_event(ev, EDYNHIP_EVENT_POINT_CREATED, ia, ib, id);
                    }
                    ++n_out;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxPts; ++i) {
            if (i < n_old && !((dead >> i) & 1u)) {
                const CPoint &p = pts[i];
                const size_t d = pt_at(mf.cap, (uint32_t)n_out, m);
                const f3 ln = ((changed >> i) & 1u) ? local_normal(p, A, B) : from4(ext_l[i]);
                mf.pA[d] = to4(p.pivotA, p.distance);
                mf.pB[d] = to4(p.pivotB, ext_f[i]);
                mf.nrm[d] = to4(p.normal, __int_as_float(p.attachment));
                mf.lnrm[d] = to4(ln, ext_l[i].w);
                mf.imp[d] = make_float4(ext_i[i].x, ext_i[i].y, ext_i[i].z, __uint_as_float(__float_as_uint(ext_i[i].w) + 1u));
                if (mf.pid) mf.pid[slot_at(mf.cap, (uint32_t)n_out, m)] = ext_id[i];
                ++n_out;
            }
        }
        if (mf.xmat) {
            // contact_extras: the survivors' mixed material and rolling / spinning impulses move to their new slots (read
            // first: the source may be this very array), created points get the materials mixed now and zero impulses
            // (assign_material_properties, collision_util.cpp:309-315; material_mixing.hpp:20-34)
            float4 xm[kMaxPts], xi[kMaxPts];
#pragma unroll
            for (int i = 0; i < kMaxPts; ++i) {
                if (i < n_old && !((dead >> i) & 1u)) { const size_t s = (size_t)i * src.cap + sidx; xm[i] = src.xmat[s]; xi[i] = src.ximp[s]; }
                else { xm[i] = xi[i] = make_float4(0, 0, 0, 0); }
            }
            const float4 ma = b.mat2[ia], mb = b.mat2[ib];
            float stiff = kLarge, damp = kLarge;
            if (ma.z < kLarge || mb.z < kLarge) { stiff = 1.0f / (1.0f / ma.z + 1.0f / mb.z); damp = 1.0f / (1.0f / ma.w + 1.0f / mb.w); }
            float4 fresh = make_float4(fmaxf(ma.y, mb.y), fmaxf(ma.x, mb.x), stiff, damp);
            if (const float *e = mix_lookup(b, ia, ib)) fresh = make_float4(e[3], e[2], e[4], e[5]);
            int slot = 0;
#pragma unroll
            for (int i = kMaxPts - 1; i >= 0; --i) if ((create >> i) & 1u) { const size_t d = (size_t)slot * mf.cap + m; mf.xmat[d] = fresh; mf.ximp[d] = make_float4(0, 0, 0, 0); ++slot; }
#pragma unroll
            for (int i = 0; i < kMaxPts; ++i) if (i < n_old && !((dead >> i) & 1u)) { const size_t d = (size_t)slot * mf.cap + m; mf.xmat[d] = xm[i]; mf.ximp[d] = xi[i]; ++slot; }
        }
        uint32_t colour = info >> 8;
        if (n_out == 0) colour = kNoColour;   // inactive pairs hold no solver colour
        mf.info[m] = (uint32_t)n_out | (colour << 8);
    }
}

// Contact point / active manifold census for edynhip_get_stats (not on the per-step path).
__global__ void k_count_points(uint32_t M, const uint32_t *__restrict__ info, Counters *cnt) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t pts = m < M ? (info[m] & 0xFF) : 0, act = pts ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pts += __shfl_xor(pts, off); act += __shfl_xor(act, off); }
    if ((threadIdx.x & 63) == 0 && pts) { atomicAdd(&cnt->num_points, pts); atomicAdd(&cnt->num_active, act); }
}
int count_points(edynhip_ctx *c) {
    EH_HIP(c, hipMemsetAsync(&c->cnt->num_points, 0, 2 * sizeof(uint32_t), c->stream));
    const uint32_t M = c->num_manifolds;
    if (M) hipLaunchKernelGGL(k_count_points, dim3((M + 255) / 256), dim3(256), 0, c->stream, M, c->m[c->cur].info, c->cnt);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

template <bool EXT>
__global__ void __launch_bounds__(64, EXT ? 1 : 2) k_debug_collide(uint32_t n, const int32_t *__restrict__ st, const float4 *__restrict__ sp, const float *__restrict__ pos,
                                const float4 *__restrict__ orn, float threshold, float *out, uint32_t *count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ctx ctx{mk3(pos[6 * i], pos[6 * i + 1], pos[6 * i + 2]), q_from4(orn[2 * i]), mk3(pos[6 * i + 3], pos[6 * i + 4], pos[6 * i + 5]),
            q_from4(orn[2 * i + 1]), threshold};
    CResult r;
    if (EXT) { r.num = 0; if (!collide_ext(st[2 * i], sp[2 * i], st[2 * i + 1], sp[2 * i + 1], ctx, r)) return; }   // (the plain kernel wrote this pair's zero count)
    else collide(st[2 * i], sp[2 * i], st[2 * i + 1], sp[2 * i + 1], ctx, r);
    count[i] = (uint32_t)r.num;
    for (int k = 0; k < r.num; ++k) {
        float *o = out + ((size_t)i * 4 + k) * 11;
        o[0] = r.pt[k].pivotA.x; o[1] = r.pt[k].pivotA.y; o[2] = r.pt[k].pivotA.z;
        o[3] = r.pt[k].pivotB.x; o[4] = r.pt[k].pivotB.y; o[5] = r.pt[k].pivotB.z;
        o[6] = r.pt[k].normal.x; o[7] = r.pt[k].normal.y; o[8] = r.pt[k].normal.z;
        o[9] = r.pt[k].distance; o[10] = (float)r.pt[k].attachment;
    }
}
// Pairs with a polyhedron: one workgroup per pair - its lanes rotate the pair's meshes into scratch (update_rotated_mesh), lane 0 collides.
__global__ void __launch_bounds__(64)
k_debug_collide_poly(uint32_t first, uint32_t n, const int32_t *__restrict__ st, const float4 *__restrict__ sp, const float *__restrict__ pos,
                     const float4 *__restrict__ orn, float threshold, float *out, uint32_t *count, Meshes meshes, float4 *scratch, uint32_t stride, int group) {
    const uint32_t i = first + blockIdx.x;
    if (blockIdx.x >= n) return;
    const int tA = st[2 * i], tB = st[2 * i + 1];
    if (tA != SHAPE_POLYHEDRON && tB != SHAPE_POLYHEDRON) return;
    float4 *rot[2] = {scratch + (size_t)(2 * blockIdx.x) * stride, scratch + (size_t)(2 * blockIdx.x + 1) * stride};
    for (int side = 0; side < 2; ++side) {
        if (st[2 * i + side] != SHAPE_POLYHEDRON) continue;
        const MeshDesc d = meshes.desc[(uint32_t)sp[2 * i + side].x];
        const q4 q = q_from4(orn[2 * i + side]) * q4{0, 0, 0, 1};
        for (uint32_t k = threadIdx.x; k < d.rot_size; k += 64) rotate_mesh_item(meshes, d, q, rot[side], k);
    }
    __syncthreads();
    Ctx ctx{mk3(pos[6 * i], pos[6 * i + 1], pos[6 * i + 2]), q_from4(orn[2 * i]), mk3(pos[6 * i + 3], pos[6 * i + 4], pos[6 * i + 5]),
            q_from4(orn[2 * i + 1]), threshold};
    CResult r;
    r.num = 0;
    if (group != 0 && tA == SHAPE_POLYHEDRON && tB == SHAPE_POLYHEDRON) {   // the first `group` lanes of the workgroup: what k_np_pp_axes + k_np_pp_contacts run
        __shared__ __align__(16) unsigned char lds_raw[sizeof(PPLds)];
        PPLds &lds = *reinterpret_cast<PPLds *>(lds_raw);
        if ((int)threadIdx.x >= group) return;
        PPProf<false> noprof;
        const PPSide A = pp_side(meshes, sp[2 * i], rot[0]), B = pp_side(meshes, sp[2 * i + 1], rot[1]);
        PPSeparation sep;
        if (group == 16) { if (pp_group_axes<16>(meshes, A, B, ctx.posB - ctx.posA, ctx.threshold, sep, noprof)) pp_group_contacts<16>(A, B, ctx, sep, lds, r, noprof); }
        else if (group == 4) { if (pp_group_axes<4>(meshes, A, B, ctx.posB - ctx.posA, ctx.threshold, sep, noprof)) pp_group_contacts<4>(A, B, ctx, sep, lds, r, noprof); }
        else if (pp_group_axes<8>(meshes, A, B, ctx.posB - ctx.posA, ctx.threshold, sep, noprof)) pp_group_contacts<8>(A, B, ctx, sep, lds, r, noprof);
        if (threadIdx.x != 0) return;
    } else {
        if (threadIdx.x != 0) return;
        collide_poly(meshes, tA, sp[2 * i], rot[0], tB, sp[2 * i + 1], rot[1], ctx, r);
    }
    count[i] = (uint32_t)r.num;
    for (int k = 0; k < r.num; ++k) {
        float *o = out + ((size_t)i * 4 + k) * 11;
        o[0] = r.pt[k].pivotA.x; o[1] = r.pt[k].pivotA.y; o[2] = r.pt[k].pivotA.z;
        o[3] = r.pt[k].pivotB.x; o[4] = r.pt[k].pivotB.y; o[5] = r.pt[k].pivotB.z;
        o[6] = r.pt[k].normal.x; o[7] = r.pt[k].normal.y; o[8] = r.pt[k].normal.z;
        o[9] = r.pt[k].distance; o[10] = (float)r.pt[k].attachment;
    }
}
int debug_collide(edynhip_ctx *c, uint32_t n, const int32_t *st, const float *sp, const float *pos, const float *orn, float threshold,
                  float *out, uint32_t *count) {
    if (n == 0) return EDYNHIP_OK;
    // one scratch allocation, carved up: [shape types | shape params | positions | orientations | points | counts]
    const size_t b_st = (size_t)n * 8, b_sp = (size_t)n * 32, b_pos = (size_t)n * 24, b_orn = (size_t)n * 32, b_out = (size_t)n * 44 * 4, b_cnt = (size_t)n * 4;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_sp = up16(b_st), o_pos = o_sp + up16(b_sp), o_orn = o_pos + up16(b_pos), o_out = o_orn + up16(b_orn), o_cnt = o_out + up16(b_out);
    char *d = nullptr;
    EH_HIP(c, hipMalloc((void **)&d, o_cnt + up16(b_cnt)));
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d, st, b_st, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_sp, sp, b_sp, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_pos, pos, b_pos, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_orn, orn, b_orn, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_out, 0, b_out, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_collide<false>, dim3((n + 63) / 64), dim3(64), 0, s, n, (const int32_t *)d, (const float4 *)(d + o_sp), (const float *)(d + o_pos),
                           (const float4 *)(d + o_orn), threshold, (float *)(d + o_out), (uint32_t *)(d + o_cnt));
        bool any_cyl = false;
        for (uint32_t i = 0; i < 2 * n && !any_cyl; ++i) any_cyl = st[i] == EDYNHIP_SHAPE_CYLINDER;
        if (any_cyl)
            hipLaunchKernelGGL(k_debug_collide<true>, dim3((n + 63) / 64), dim3(64), 0, s, n, (const int32_t *)d, (const float4 *)(d + o_sp), (const float *)(d + o_pos),
                               (const float4 *)(d + o_orn), threshold, (float *)(d + o_out), (uint32_t *)(d + o_cnt));
        bool any_poly = false;
        for (uint32_t i = 0; i < 2 * n; ++i)
            if (st[i] == EDYNHIP_SHAPE_POLYHEDRON) {
                any_poly = true;
                const float id = sp[4 * i];
                if (!(id >= 0) || id >= (float)c->host_meshes.desc.size()) { (void)hipStreamSynchronize(s); (void)hipFree(d); return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_debug_collide: unknown mesh id"); }
            }
        if (any_poly) {
            uint32_t stride = 0;
            for (const MeshDesc &md : c->host_meshes.desc) stride = std::max(stride, md.rot_size);
            const uint32_t chunk = 16384;
            const int group = getenv("EDYNHIP_POLY_GROUP") ? atoi(getenv("EDYNHIP_POLY_GROUP")) : 8;   // (as narrowphase() below)
            float4 *scratch = nullptr;
            e = hipMalloc((void **)&scratch, (size_t)chunk * 2 * stride * sizeof(float4));
            for (uint32_t first = 0; first < n && e == hipSuccess; first += chunk) {
                const uint32_t cnt = std::min(chunk, n - first);
                hipLaunchKernelGGL(k_debug_collide_poly, dim3(cnt), dim3(64), 0, s, first, cnt, (const int32_t *)d, (const float4 *)(d + o_sp), (const float *)(d + o_pos),
                                   (const float4 *)(d + o_orn), threshold, (float *)(d + o_out), (uint32_t *)(d + o_cnt), c->meshes, scratch, stride, group);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (scratch) (void)hipFree(scratch);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, d + o_out, b_out, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(count, d + o_cnt, b_cnt, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_debug_collide", e);
    return EDYNHIP_OK;
}

int narrowphase(edynhip_ctx *c) {
    const uint32_t M = c->num_manifolds;
    if (M == 0) return EDYNHIP_OK;
    const Staging st{c->np_ra, c->np_rb, c->np_rn, c->np_rnum};
    hipLaunchKernelGGL(k_np_detect, dim3((M + 63) / 64), dim3(64), 0, c->stream, M, c->m[c->cur], c->b, c->sleeping, st);
    if (c->has_cylinder) hipLaunchKernelGGL(k_np_detect_ext, dim3((M + 63) / 64), dim3(64), 0, c->stream, M, c->m[c->cur], c->b, c->sleeping, st);
    if (c->has_polyhedron) {
        EH_TRY(update_rotated(c));
        uint32_t *pw = c->poly_work;   // [count | start | cursor : kPolyBins each][total][key : cap][list : cap][hint : 2 x cap], allocated with the first polyhedron (mesh.hip)
        const size_t cap = c->m[0].cap;
        uint32_t *const hints = pw + 3 * kPolyBins + 1 + 2 * cap;   // [2][cap]: one hint array per manifold array
        const PolyBins pb{pw, pw + kPolyBins, pw + 2 * kPolyBins, pw + 3 * kPolyBins, pw + 3 * kPolyBins + 1, pw + 3 * kPolyBins + 1 + cap, hints + (size_t)c->cur * cap};
        static const bool use_hints = !(getenv("EDYNHIP_POLY_HINT") && getenv("EDYNHIP_POLY_HINT")[0] == '0');   // developer knob (A/B)
        hipLaunchKernelGGL(k_poly_count, dim3((M + 255) / 256), dim3(256), 0, c->stream, M, c->m[c->cur], c->b, c->sleeping, pb, c->meshes, hints + (size_t)(c->cur ^ 1) * cap,
                           !c->inplace_step, use_hints);
        hipLaunchKernelGGL(k_poly_scan, dim3(1), dim3(1024), 0, c->stream, pb);
        hipLaunchKernelGGL(k_poly_scatter, dim3((M + 255) / 256), dim3(256), 0, c->stream, M, pb);
        // developer knobs (A/B): EDYNHIP_POLY_GROUP=0 the round-4 form (k_np_detect_poly for polyhedron pairs too), =8 / 16 lanes per pair in the separating-axis kernel (default 8)
        static const int group = getenv("EDYNHIP_POLY_GROUP") ? atoi(getenv("EDYNHIP_POLY_GROUP")) : 8;
        if (group != 0) {
            const uint32_t waves = 5120;   // 20 per CU of an MI355X (the axes kernel runs five per SIMD): the grids stride over the pairs
            static const int group2 = getenv("EDYNHIP_POLY_GROUP2") ? atoi(getenv("EDYNHIP_POLY_GROUP2")) : 4;   // lanes per surviving pair: 4 (default: 16 pairs per wave), 8 or 16
            static const bool prof = getenv("EDYNHIP_PP_PROF") != nullptr;   // developer profile of the phases: printed every 100th step, per context (its counters live with the context: the shards of a multi-device world step on their own threads and devices)
            unsigned long long *&prof_dev = c->pp_prof_dev;
            int &prof_calls = c->pp_prof_calls;
            if (prof && !prof_dev) { void *q = nullptr; EH_HIP(c, hipMalloc(&q, 128)); c->allocs.push_back(q); prof_dev = (unsigned long long *)q; EH_HIP(c, hipMemsetAsync(prof_dev, 0, 128, c->stream)); }
            const Manifolds &mfc = c->m[c->cur];
            unsigned long long *const pd = prof ? prof_dev : nullptr;
            // (measured and dropped: the contact kernel compiled for three waves per SIMD - 168 VGPRs and 260 B of spills - 324 -> 304 steps/s on polyheap32k)
            if (prof) {
                if (group == 16) hipLaunchKernelGGL((k_np_pp_axes<16, true>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                else hipLaunchKernelGGL((k_np_pp_axes<8, true>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                if (group2 == 8) hipLaunchKernelGGL((k_np_pp_contacts<8, true>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                else hipLaunchKernelGGL((k_np_pp_contacts<4, true>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
            } else {
                if (group == 16) hipLaunchKernelGGL((k_np_pp_axes<16, false>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                else hipLaunchKernelGGL((k_np_pp_axes<8, false>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                if (group2 == 8) hipLaunchKernelGGL((k_np_pp_contacts<8, false>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                else if (group2 == 16) hipLaunchKernelGGL((k_np_pp_contacts<16, false>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
                else hipLaunchKernelGGL((k_np_pp_contacts<4, false>), dim3(waves), dim3(64), 0, c->stream, mfc, c->b, st, c->meshes, pb, pd);
            }
            if (prof && ++prof_calls % 100 == 0) {
                unsigned long long h[16];
                EH_HIP(c, hipMemcpyAsync(h, prof_dev, 128, hipMemcpyDeviceToHost, c->stream));
                EH_HIP(c, hipStreamSynchronize(c->stream));
                EH_HIP(c, hipMemsetAsync(prof_dev, 0, 128, c->stream));
                const double p = 100.0 * (double)std::max<unsigned long long>(h[0], 1), q = 100.0 * (double)std::max<unsigned long long>(h[4], 1);
                fprintf(stderr, "[pp prof, G=%d/%d] pairs/step %.0f, us per pair-iteration: setup+faces %.2f edges %.2f stores+wait %.2f; survivors/step %.0f: setup+polygons %.2f hulls %.2f contacts %.2f stores+wait %.2f\n",
                        group, group2, h[0] / 100.0, h[1] / p, h[2] / p, h[3] / p, h[4] / 100.0, h[5] / q, h[6] / q, h[7] / q, h[8] / q);
            }
        }
        hipLaunchKernelGGL(k_np_detect_poly, dim3((M + 63) / 64), dim3(64), 0, c->stream, M, c->m[c->cur], c->b, st, c->meshes, pb, group != 0);
    }
    hipLaunchKernelGGL(k_np_merge, dim3((M + 63) / 64), dim3(64), 0, c->stream, M, c->m[c->cur], c->b, c->cfg.fixed_dt, c->sleeping, c->m[c->cur ^ 1], c->points_in_prev, st, event_sink(c));
    c->points_in_prev = false;
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

}  // namespace eh
