// Broadphase on the GPU: the reference keeps three incrementally-updated dynamic AABB trees on the
// CPU (src/edyn/collision/broadphase.cpp:99-195, dynamic_tree.cpp); here the tree over the
// procedural bodies is a linear BVH (Morton sort + Karras radix-tree construction every 8th step,
// bottom-up refit every step), every procedural body queries it in parallel, and the shaped
// non-procedural bodies (static planes etc.) are tested by brute force. Because the reference's
// final predicates use the true AABBs (broadphase.cpp:119-155), the manifold set
//     S_t = { p in S_{t-1} : intersect(box0 grown 0.026, box1) }  U  { p : should_collide, intersect(query grown 0.02, other) }
// does not depend on the tree, and is reproduced bit-exactly including pair orientation.
// Every pair is reported by exactly one lane - its owner, the querying procedural body - so the sorted pair list
// needs no global sort: each lane sorts its few keys, a scan over the per-owner counts places them (k_bp_compact).
// The list becomes this step's manifold array; contact points persist through a lookup in the owner's segment of
// the previous step's sorted array.
#include "ctx.hpp"
#include "dmath.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace eh {
using namespace dm;

constexpr float kQueryGrow = 0.03f;              // conservative candidate margin (> 0.026)
constexpr float kBreaking = 0.02f;               // broadphase.hpp:15 (m_aabb_offset)
// candidate lists (see "candidate lists" below)
constexpr float kListSlack = 0.035f;     // base slack
constexpr float kListLookaheadMax = 6.0f, kListLookaheadMin = 1.5f;   // + this many steps of motion at the current velocity (LBVH::lookahead adapts between the two)
constexpr float kListTest = 0.026f;      // the widest margin any exact predicate uses (separation threshold 0.02 * 1.3)
static_assert(kListTest >= 0.02f * 1.3f - 1e-6f && kListTest <= kQueryGrow, "list margin must cover every exact predicate");
// broadphase.hpp:18: contact_breaking_threshold * scalar(1.3), evaluated in fp32 like the reference
__device__ __forceinline__ float separation_threshold() { return 0.02f * 1.3f; }

DI int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
DI float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void k_step_reset(Counters *cnt) {   // (inside edynhip_step the previous step's k_finish does this instead)
    int t = threadIdx.x;
    if (t == 0) {
        cnt->num_pairs = 0; cnt->pair_overflow = 0; cnt->num_points = 0; cnt->num_active = 0;
        cnt->uncoloured = 0; cnt->colour_overflow = 0; cnt->pairs_changed = 0; cnt->num_found = 0; cnt->num_new = 0; cnt->num_extra = 0; cnt->num_awake = 0; cnt->pairs_differ = 0;
        cnt->tree_found = 0; cnt->tree_marks = 0;
        cnt->unc_count = 0; cnt->df_abort = 0; cnt->bp_rebuild = 0;   // also the sticky ones: a stand-alone run follows set_* calls or a failed step
    }
    if (t < 3) { cnt->bounds_min[t] = 0x7FFFFFFF; cnt->bounds_max[t] = (int)0x80000000; }
    for (int k = t; k < 4 * (int)kMaxColours; k += 64) { cnt->colour_start[k] = 0; cnt->colour_end[k] = 0; }
}

__global__ void __launch_bounds__(256)
k_bp_bounds(const uint32_t *__restrict__ proc, uint32_t np, const float4 *__restrict__ amin,
            const float4 *__restrict__ amax, Counters *cnt) {
    __shared__ float slo[4][3], shi[4][3];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < np; t += gridDim.x * blockDim.x) {   // grid-stride
        uint32_t b = proc[t];
        float4 a = amin[b], c = amax[b];
        float cx = (a.x + c.x) * 0.5f, cy = (a.y + c.y) * 0.5f, cz = (a.z + c.z) * 0.5f;
        lo[0] = fminf(lo[0], cx); hi[0] = fmaxf(hi[0], cx);
        lo[1] = fminf(lo[1], cy); hi[1] = fmaxf(hi[1], cy);
        lo[2] = fminf(lo[2], cz); hi[2] = fmaxf(hi[2], cz);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; ++k) { slo[wave][k] = lo[k]; shi[wave][k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = fminf(fminf(slo[0][k], slo[1][k]), fminf(slo[2][k], slo[3][k]));
        float h = fmaxf(fmaxf(shi[0][k], shi[1][k]), fmaxf(shi[2][k], shi[3][k]));
        if (l <= h) { atomicMin(&cnt->bounds_min[k], f2ord(l)); atomicMax(&cnt->bounds_max[k], f2ord(h)); }
    }
}

DI uint32_t expand10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void k_bp_morton(const uint32_t *__restrict__ proc, uint32_t np, const float4 *__restrict__ amin,
                            const float4 *__restrict__ amax, const Counters *__restrict__ cnt, uint64_t *keys) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= np) return;
    uint32_t b = proc[t];
    float4 a = amin[b], c = amax[b];
    float ctr[3] = {(a.x + c.x) * 0.5f, (a.y + c.y) * 0.5f, (a.z + c.z) * 0.5f};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = ord2f(cnt->bounds_min[k]), hi = ord2f(cnt->bounds_max[k]);
        float ext = hi - lo;
        float u = ext > 0.0f ? (ctr[k] - lo) / ext : 0.0f;
        int qi = (int)(u * 1023.0f);
        q[k] = (uint32_t)min(max(qi, 0), 1023);
    }
    uint32_t code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    keys[t] = ((uint64_t)code << 32) | b;
}

// Karras 2012 radix tree over unique 64-bit keys. Node ids: internal i in [0,n-2], leaf k -> n-1+k.
DI int delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ void k_bp_build(const uint64_t *__restrict__ keys, int n, uint32_t *parent, uint32_t *left, uint32_t *right,
                           uint32_t *visit) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    visit[i] = 0;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    uint32_t lc = (min(i, j) == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    uint32_t rc = (max(i, j) == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    left[i] = lc; right[i] = rc;
    parent[lc] = (uint32_t)i; parent[rc] = (uint32_t)i;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
}

// Ropes ("escape links") for a stackless traversal: rope[x] = the node a depth-first walk visits after x's subtree - the right
// sibling of the first ancestor-or-self that is a left child, or kRopeEnd. Computed with the topology (every few steps).
constexpr uint32_t kRopeEnd = 0xFFFFFFFFu;
__global__ void k_bp_ropes(int n, const uint32_t *__restrict__ parent, const uint32_t *__restrict__ right, uint32_t *rope) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * n - 1) return;
    uint32_t c = (uint32_t)x, r = kRopeEnd;
    for (;;) {
        const uint32_t p = parent[c];
        if (p == 0xFFFFFFFFu) break;                     // reached the root: nothing follows
        if (right[p] != c) { r = right[p]; break; }      // c is a left child: its right sibling comes next
        c = p;
    }
    rope[x] = r;
}

// Node boxes are exchanged between workgroups inside this launch, so they travel as 8-byte
// agent-scope atomics on both sides (per-CU L1s and per-XCD L2s are not coherent for plain accesses).
using gu64 = __attribute__((address_space(1))) unsigned long long;
DI unsigned long long pack2(float a, float b) { return ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a); }
// node record: nmin = (min.xyz, bits(left child)), nmax = (max.xyz, bits(rope)); leaves carry 0xFFFFFFFF as left child
DI void store_box(float4 *nmin, float4 *nmax, uint32_t node, f3 mn, f3 mx, uint32_t lc, uint32_t rc) {
    unsigned long long *p0 = (unsigned long long *)&nmin[node];
    unsigned long long *p1 = (unsigned long long *)&nmax[node];
    __hip_atomic_store(p0, pack2(mn.x, mn.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p0 + 1, pack2(mn.z, __uint_as_float(lc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p1, pack2(mx.x, mx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p1 + 1, pack2(mx.z, __uint_as_float(rc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DI void load_box(const float4 *nmin, const float4 *nmax, uint32_t node, f3 &mn, f3 &mx) {
    unsigned long long *p0 = (unsigned long long *)&nmin[node];
    unsigned long long *p1 = (unsigned long long *)&nmax[node];
    unsigned long long a = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long b = __hip_atomic_load(p0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long c = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long d = __hip_atomic_load(p1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mn = {__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b)};
    mx = {__uint_as_float((uint32_t)c), __uint_as_float((uint32_t)(c >> 32)), __uint_as_float((uint32_t)d)};
}

__global__ void k_bp_refit(const uint64_t *__restrict__ keys, int n, const uint32_t *__restrict__ parent,
                           const uint32_t *__restrict__ left, const uint32_t *__restrict__ right, const uint32_t *__restrict__ rope,
                           const float4 *__restrict__ amin, const float4 *__restrict__ amax, float4 *nmin, float4 *nmax,
                           uint32_t *visit, const Counters *cnt, float4 *ref_min, float4 *ref_max, const float4 *__restrict__ linvel,
                           const float4 *__restrict__ angvel, float dt, uint32_t force, float gacc, float lookahead) {
    if (!(cnt->bp_rebuild | force)) return;   // the candidate lists are still valid: nobody walks the tree this step
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t body = (uint32_t)(keys[k] & 0xFFFFFFFFu);
    f3 mn = from4(amin[body]), mx = from4(amax[body]);
    {   // this body's ref box and slack (see "candidate lists"); the leaf holds the ref box grown by the slack
        const f3 v = from4(linvel[body]), w = from4(angvel[body]);
        const f3 ext = mx - mn;
        const float reach = 0.5f * sqrtf(ext.x * ext.x + ext.y * ext.y + ext.z * ext.z);   // no point of the body is further from its centre
        // the motion of the next `lookahead` steps at the current velocity, plus what gravity adds to it
        const float horizon = lookahead * dt;
        float slack = kListSlack + horizon * (sqrtf(length_sqr(v)) + sqrtf(length_sqr(w)) * reach) + 0.5f * gacc * horizon * horizon;
        if (!(slack < 4.0f)) slack = 4.0f;   // also catches NaN / inf velocities
        ref_min[body] = to4(mn, slack); ref_max[body] = to4(mx, 0.0f);
        mn = mn - mk3(slack, slack, slack); mx = mx + mk3(slack, slack, slack);
    }
    uint32_t node = (uint32_t)(n - 1 + k);
    store_box(nmin, nmax, node, mn, mx, 0xFFFFFFFFu, n > 1 ? rope[node] : kRopeEnd);
    if (n == 1) return;
    uint32_t p = parent[node];
    while (p != 0xFFFFFFFFu) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // box stores drained before the arrival ticket
        uint32_t old = atomicAdd(&visit[p], 1u);
        if (old == 0) return;                              // first arriver leaves; the second one continues
        uint32_t sib = left[p] == node ? right[p] : left[p];
        f3 smn, smx;
        load_box(nmin, nmax, sib, smn, smx);
        mn = {fminf(mn.x, smn.x), fminf(mn.y, smn.y), fminf(mn.z, smn.z)};
        mx = {fmaxf(mx.x, smx.x), fmaxf(mx.y, smx.y), fmaxf(mx.z, smx.z)};
        store_box(nmin, nmax, p, mn, mx, left[p], rope[p]);
        node = p;
        p = parent[p];
    }
}

DI box3 body_box(const float4 *amin, const float4 *amax, uint32_t b) { return {from4(amin[b]), from4(amax[b])}; }
// should_collide_default (should_collide.cpp:23-57): group/mask bits both ways, minus the explicit exclusion lists
// (collision_exclusion: <= 16 entities per body, comp/collision_exclusion.hpp:16-31). It gates the CREATION of a manifold
// only (broadphase.cpp:145,165); an existing manifold lives until its AABBs separate.
constexpr uint32_t kMaxExclusions = 16;
struct Filt { const uint64_t *group, *mask; const uint32_t *excl; bool bypass; };   // excl: [body][16], ~0u-terminated, or nullptr (no lists); bypass: the host's pair filter decides (edynhip_set_pair_filter)
DI bool excluded_one_way(const uint32_t *excl, uint32_t a, uint32_t b) {
    const uint32_t *l = excl + (size_t)a * kMaxExclusions;
    for (uint32_t k = 0; k < kMaxExclusions; ++k) {
        const uint32_t e = l[k];
        if (e == 0xFFFFFFFFu) break;
        if (e == b) return true;
    }
    return false;
}
DI bool filter_ok(const Filt &f, uint32_t a, uint32_t b) {
    if (f.bypass) return true;
    if ((f.group[a] & f.mask[b]) == 0 || (f.group[b] & f.mask[a]) == 0) return false;
    if (f.excl && (excluded_one_way(f.excl, a, b) || excluded_one_way(f.excl, b, a))) return false;
    return true;
}
// Previous-step manifold of the canonical pair (hi, lo), or ~0u. The previous array is sorted by (hi, lo), so
// the candidates are the short contiguous run of manifolds whose higher body is `hi`.
DI uint32_t find_prev(const Manifolds &prev, uint32_t pm, uint32_t hi, uint32_t lo) {
    if (pm == 0) return 0xFFFFFFFFu;
    // the owner's segment of the previous array is in ascending order of `other` (the array is sorted by canonical key): binary search
    uint32_t a = prev.seg_start[hi], b = prev.seg_end[hi];
    while (a < b) {
        const uint32_t mid = (a + b) >> 1, o = (uint32_t)(prev.skey[mid] >> 1);
        if (o == lo) return mid;
        if (o < lo) a = mid + 1; else b = mid;
    }
    return 0xFFFFFFFFu;
}
// Canonical pair key: (owner << 32 | other) << 1 | swapped. The OWNER is the procedural body whose query reports the
// pair - the higher index when both are procedural (the other side skips lower->higher hits) - so every pair is
// produced by exactly one lane, and the sorted pair list is simply "every owner's partners in ascending order,
// owners in ascending order": each lane sorts its few keys in LDS, a scan over the per-owner counts gives the
// offsets, a compaction kernel writes the list. No global sort (it used to be 7 radix passes, ~150 us per step).
// swapped = the manifold's body[0] is `other` (the querying body is body[0], broadphase.cpp:151,171).
constexpr int kBpBlock = 64;  // one wave per workgroup: LDS per block stays small, so many blocks share a CU
// The lane's own keys all start with the same owner: LDS keeps only the low half, (other << 1 | swapped), 4 bytes a key.
// kPairLanes lanes share an owner (each takes every kPairLanes-th candidate): they append to the owner's list through an LDS counter.
// (Sixteen lanes per owner, also only for scenes with many pairs per body, were measured in round 4 and are slower everywhere: pile32k 836
// against 888 steps/s, the polyhedron heap 299 against 303 - the kernel is not bound by a lane's chain of candidates.)
constexpr int kPairLanes = 4, kOwnersPerBlock = kBpBlock / kPairLanes;
struct Emit { uint32_t (*mine)[kOwnersPerBlock]; uint32_t *count; int ol; uint64_t *extra; uint32_t cap; Counters *cnt; uint32_t tree; };   // tree: kept forest-certificate manifolds
DI void emit_pair(uint64_t skey, Emit &e) {
    const uint32_t slot = atomicAdd(&e.count[e.ol], 1u);
    if (slot < (uint32_t)kOwnCap) { e.mine[slot][e.ol] = (uint32_t)skey; return; }
    const uint32_t g = atomicAdd(&e.cnt->num_extra, 1u);   // rare: an owner with more than kOwnCap partners
    if (g < e.cap) e.extra[g] = skey; else e.cnt->pair_overflow = 1;
}
// Decide whether the pair {i (procedural, the querying body = owner), j} is in this step's set.
DI void consider_pair(uint32_t i, uint32_t j, const box3 &bi, const float4 *amin, const float4 *amax, const Filt &f,
                      bool j_procedural, const Manifolds &prev, uint32_t pm, Emit &em) {
    const uint64_t key = ((uint64_t)i << 32) | j;
    const box3 bj = body_box(amin, amax, j);
    uint32_t pidx = find_prev(prev, pm, i, j);
    if (pidx != 0xFFFFFFFFu) {   // destroy_separated_manifolds, broadphase.cpp:119-134
        const uint64_t ps = prev.skey[pidx];
        const bool swapped = ps & 1;                      // body[0] == j
        const box3 &x0 = swapped ? bj : bi, &x1 = swapped ? bi : bj;
        if (intersect(inset(x0, -separation_threshold()), x1)) { emit_pair(ps, em); em.tree += prev.tree[pidx]; }
        return;
    }
    if (!filter_ok(f, i, j)) return;
    // collide_tree, broadphase.cpp:136-155: querying body's box grown by 0.02 vs the other's true box.
    // Bodies are visited in descending index order, so the higher index (= i here) gets to create the pair first.
    if (j_procedural) {
        if (intersect(inset(bi, -kBreaking), bj)) emit_pair(key << 1, em);
        else if (intersect(inset(bj, -kBreaking), bi)) emit_pair((key << 1) | 1, em);
    } else {
        if (intersect(inset(bi, -kBreaking), bj)) emit_pair(key << 1, em);
    }
}

// A sleeping body does not query (broadphase.cpp:101,183 views exclude sleeping_tag), so an awake body i that touches a
// sleeping body j with a HIGHER index has to report the pair itself although j is its owner. Such pairs are new contacts
// onto a sleeping island - rare events - and take the unsorted `extra` path (j re-emits its existing manifolds itself).
DI void consider_sleeping_owner(uint32_t i, uint32_t j, const box3 &bi, const float4 *amin, const float4 *amax, const Filt &f,
                                const Manifolds &prev, uint32_t pm, Emit &em) {
    if (find_prev(prev, pm, j, i) != 0xFFFFFFFFu) return;
    if (!filter_ok(f, i, j)) return;
    const box3 bj = body_box(amin, amax, j);
    if (!intersect(inset(bi, -kBreaking), bj)) return;
    const uint64_t skey = (((((uint64_t)j << 32) | i) << 1) | 1u);   // body[0] = i, the querying body
    const uint32_t g = atomicAdd(&em.cnt->num_extra, 1u);
    if (g < em.cap) em.extra[g] = skey; else em.cnt->pair_overflow = 1;
}

// ---- candidate lists (Verlet lists) -----------------------------------------------------------------------------------
// The tree walk is a chain of dependent loads (~150 node visits per body, a few hundred nanoseconds each) with only one lane
// per body to hide it: ~140 us per step on a 32k pile whose pair set barely changes. So the walk is NOT done every step:
// when it runs it queries with a FAT box and leaves every body a list of candidate partners plus a copy of
// the AABBs it was built from (ref boxes); the steps that follow only run the exact predicates over those lists - until
// some body has moved further than kListSlack from its ref box, which a body-parallel check detects on the device at the
// start of the step (no host round trip: the refit and the walk are enqueued every step and return at once when the
// lists are still valid).
// Every body carries its own slack s_i (how far any face of its AABB may stray from the ref box before the lists are void):
// a base value plus the distance it would cover in a few steps at its current speed, so that fast bodies do not force a walk
// every step. The leaves of the tree hold the ref boxes GROWN by the body's slack, and body i queries with its ref box grown
// by 0.026 + s_i.
// Why the pair set stays exact: every predicate below tests one box grown by at most 0.026 against the other. While no face
// of box_i has moved more than s_i and none of box_j more than s_j, grow(box_i(t), 0.026) lies inside grow(box_i(t0),
// 0.026 + s_i) and box_j(t) inside grow(box_j(t0), s_j): an overlap now implies that those two overlapped at t0, which is
// exactly the walk's leaf test - the pair is on the list. (The slack only has to be a number; how it is chosen decides
// how long the lists live, not whether they are right.)
constexpr uint32_t kHigherBit = 0x80000000u;      // list entry flag: the candidate has a HIGHER index (recorded when island
                                                  // sleeping is on: it matters only while that body sleeps, see below)
// The subtrees three levels below the root (k_bp_split, computed with the topology): a body whose candidate list overflowed walks the
// tree in k_bp_pairs every step, stackless (ropes), one of these subtrees per lane - each walk ends at its subtree's rope. (Until round
// 4 the list-building walk k_bp_walk ran the same way, kWalkSplit lanes per body; it is group-cooperative now, below.)
constexpr uint32_t kWalkSplit = 8;
__global__ void k_bp_split(int n, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right, uint32_t *split) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t cur[kWalkSplit], nxt[kWalkSplit];
    uint32_t nc = 0;
    if (n > 1) cur[nc++] = 0;   // the root (internal node 0)
    for (int level = 0; level < 3; ++level) {
        uint32_t nn = 0;
        for (uint32_t e = 0; e < nc; ++e) {
            const uint32_t x = cur[e];
            if (x < (uint32_t)(n - 1)) { nxt[nn++] = left[x]; nxt[nn++] = right[x]; } else nxt[nn++] = x;
        }
        nc = nn;
        for (uint32_t e = 0; e < nc; ++e) cur[e] = nxt[e];
    }
    for (uint32_t e = 0; e < kWalkSplit; ++e) split[e] = e < nc ? cur[e] : kRopeEnd;
}
// Round 4: a GROUP of kWalkLanes lanes walks the tree for one body together. The walk used to be a chain of dependent node loads per
// lane (~0.25 us each, 150-300 of them on a pile, far more for the fat boxes of a heap in motion: 0.55 ms per step on the polyhedron
// heap); now the group keeps a stack of pending nodes in LDS, pops up to kWalkLanes of them per round, tests them in parallel and pushes
// the children of the internal nodes that overlap (wave prefix sums place them): a walk of N node tests takes ~N / kWalkLanes + depth
// dependent rounds. A stack that would overflow marks the body's list as overflowed - the same safe fallback as a list with more than
// kListCap entries (the body then walks the tree in k_bp_pairs every step).
constexpr uint32_t kWalkLanes = 16, kWalkStack = 192;
__global__ void __launch_bounds__(256)
k_bp_walk(const uint64_t *__restrict__ keys, int n, const float4 *__restrict__ nmin, const float4 *__restrict__ nmax,
          const float4 *__restrict__ amin, const float4 *__restrict__ amax, CandLists cl, const Counters *cnt, uint32_t *visit, bool both_ways, uint32_t force,
          const uint32_t *__restrict__ right) {
    if (!(cnt->bp_rebuild | force)) return;
    __shared__ uint32_t stack_mem[256 / kWalkLanes][kWalkStack];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t / (int)kWalkLanes;
    const uint32_t s = threadIdx.x % kWalkLanes, g = threadIdx.x / kWalkLanes;
    volatile uint32_t *stack = stack_mem[g];
    if (s == 0 && k < n - 1) visit[k] = 0;   // (visit: arm the refit counters for the next refit)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t group_mask = ((1ull << kWalkLanes) - 1ull) << (lane & ~(kWalkLanes - 1u)), lower = (1ull << lane) - 1ull;
    const uint32_t first_leaf = (uint32_t)(n - 1);
    uint32_t i = 0, top = 0, found = 0;
    bool overflow = false;
    box3 q{mk3(0, 0, 0), mk3(0, 0, 0)};
    uint32_t *row = nullptr;
    if (k < n) {
        i = (uint32_t)(keys[k] & 0xFFFFFFFFu);
        const float4 a4 = cl.ref_min[i], c4 = cl.ref_max[i];          // written by this step's refit: the current AABB and the slack
        q = inset(box3{from4(a4), from4(c4)}, -(kListTest + a4.w));
        row = cl.list + (size_t)i * kListCap;
        if (n > 1) { if (s == 0) stack[0] = 0u; top = 1; }           // the root (internal node 0)
    }
    while (__any(top > 0)) {   // (every lane of the wave stays in the loop: the ballots below need them; a finished group idles)
        const uint32_t take = min(top, kWalkLanes);
        const bool has = s < take;
        const uint32_t node = has ? stack[top - 1 - s] : 0u;
        top -= take;
        const bool leaf = node >= first_leaf;
        bool hit = false;
        uint32_t lc = 0, rc = 0;
        if (has) {
            const float4 lo4 = nmin[node], hi4 = nmax[node];
            if (!leaf) rc = right[node];
            hit = intersect(box3{from4(lo4), from4(hi4)}, q);
            lc = __float_as_uint(lo4.w);
        }
        bool cand = false;
        uint32_t j = 0;
        if (hit && leaf) {
            j = (uint32_t)(keys[node - first_leaf] & 0xFFFFFFFFu);
            cand = j < i || (both_ways && j > i);
        }
        const bool inner = hit && !leaf;
        const uint64_t bc = __ballot(cand) & group_mask, bi = __ballot(inner) & group_mask;
        if (cand) {
            const uint32_t at = found + (uint32_t)__popcll(bc & lower);
            if (at < kListCap) row[at] = j < i ? j : (j | kHigherBit);
        }
        found += (uint32_t)__popcll(bc);
        const uint32_t pushes = 2u * (uint32_t)__popcll(bi);
        if (top + pushes > kWalkStack) { overflow = true; top = 0; }   // (uniform within the group)
        else {
            if (inner) {
                const uint32_t at = top + 2u * (uint32_t)__popcll(bi & lower);
                stack[at] = lc; stack[at + 1] = rc;
            }
            top += pushes;
        }
    }
    if (s == 0 && k < n) cl.count[i] = (overflow || found > kListCap) ? kListOverflow : found;
}
__global__ void __launch_bounds__(kBpBlock)
k_bp_pairs(const uint64_t *__restrict__ keys, int n, const float4 *__restrict__ nmin, const float4 *__restrict__ nmax,
           const float4 *__restrict__ amin, const float4 *__restrict__ amax, Filt f,
           const uint32_t *__restrict__ np_list, uint32_t num_np, Manifolds prev, uint32_t pm,
           uint64_t *own_keys, uint32_t *own_count, uint64_t *extra, uint32_t cap, Counters *cnt, CandLists cl,
           const uint32_t *__restrict__ flags, bool sleeping, const uint32_t *__restrict__ split, const uint32_t *__restrict__ rope) {
    __shared__ uint32_t mine[kOwnCap][kOwnersPerBlock];   // each owner's pair keys, low halves
    __shared__ uint32_t mine_n[kOwnersPerBlock];          // ... and how many its lanes have emitted (may exceed kOwnCap: the surplus went to `extra`)
    __shared__ uint32_t tree_kept;                        // forest-certificate manifolds this block's owners keep (Counters::tree_found)
    const int tx = threadIdx.x, ol = tx / kPairLanes;
    const uint32_t s = (uint32_t)(tx % kPairLanes);
    if (tx < kOwnersPerBlock) mine_n[tx] = 0;
    if (tx == 0) tree_kept = 0;
    __syncthreads();
    const int k = blockIdx.x * kOwnersPerBlock + ol;
    const bool valid = k < n;
    const uint32_t i = valid ? (uint32_t)(keys[k] & 0xFFFFFFFFu) : 0u;
    const bool asleep_owner = valid && sleeping && (flags[i] & BF_ASLEEP);
    if (asleep_owner) {
        // a sleeping owner keeps its manifolds as they are (destroy_separated_manifolds excludes them): its previous
        // segment is already in ascending order
        if (s == 0) {
            uint32_t c = 0, kept = 0;
            if (pm) for (uint32_t p0 = prev.seg_start[i], e = prev.seg_end[i]; p0 < e; ++p0, ++c) {
                if (c < (uint32_t)kOwnCap) own_keys[(size_t)i * kOwnCap + c] = prev.skey[p0];
                else { const uint32_t g = atomicAdd(&cnt->num_extra, 1u); if (g < cap) extra[g] = prev.skey[p0]; else cnt->pair_overflow = 1; }
                kept += prev.tree[p0];
            }
            own_count[i] = min(c, (uint32_t)kOwnCap);
            if (kept) atomicAdd(&cnt->tree_found, kept);   // (sleeping worlds: rare path, no block reduction)
        }
    } else if (valid) {
        Emit em{mine, mine_n, ol, extra, cap, cnt, 0u};
        const box3 bi = body_box(amin, amax, i);
        const uint32_t lc = cl.count[i];
        if (lc != kListOverflow) {
            const uint32_t *row = cl.list + (size_t)i * kListCap;
            for (uint32_t t = s; t < lc; t += kPairLanes) {
                const uint32_t e = row[t], j = e & ~kHigherBit;
                if (!(e & kHigherBit)) consider_pair(i, j, bi, amin, amax, f, true, prev, pm, em);
                else if (sleeping && (flags[j] & BF_ASLEEP)) consider_sleeping_owner(i, j, bi, amin, amax, f, prev, pm, em);
            }
        } else if (n > 1) {   // more than kListCap bodies around this one: walk the (freshly refitted) tree, a subtree per lane
            const box3 q = inset(bi, -kQueryGrow);
            const uint32_t first_leaf = (uint32_t)(n - 1);
            for (uint32_t sub = s; sub < kWalkSplit; sub += kPairLanes) {
                const uint32_t start = split[sub];
                if (start == kRopeEnd) continue;
                const uint32_t stop = rope[start];
                uint32_t node = start;
                while (node != stop) {
                    const float4 lo4 = nmin[node], hi4 = nmax[node];
                    const bool hit = intersect(box3{from4(lo4), from4(hi4)}, q);
                    if (hit && node >= first_leaf) {
                        const uint32_t j = (uint32_t)(keys[node - first_leaf] & 0xFFFFFFFFu);
                        if (j < i) consider_pair(i, j, bi, amin, amax, f, true, prev, pm, em);
                        else if (sleeping && j > i && (flags[j] & BF_ASLEEP)) consider_sleeping_owner(i, j, bi, amin, amax, f, prev, pm, em);
                    }
                    node = (hit && node < first_leaf) ? __float_as_uint(lo4.w) : __float_as_uint(hi4.w);
                }
            }
        }
        {
            const box3 q = inset(bi, -kQueryGrow);
            for (uint32_t t = s; t < num_np; t += kPairLanes) {
                uint32_t j = np_list[t];
                box3 bj = body_box(amin, amax, j);
                if (intersect(bj, q)) consider_pair(i, j, bi, amin, amax, f, false, prev, pm, em);
            }
        }
        if (em.tree) atomicAdd(&tree_kept, em.tree);
    }
    __syncthreads();
    if (valid && !asleep_owner && s == 0) {
        // ascending by `other` (insertion sort: a handful of keys), then out to this owner's slot block
        const int cnt_i = (int)min(mine_n[ol], (uint32_t)kOwnCap);
        for (int a = 1; a < cnt_i; ++a) {
            const uint32_t v = mine[a][ol];
            int b = a - 1;
            while (b >= 0 && mine[b][ol] > v) { mine[b + 1][ol] = mine[b][ol]; --b; }
            mine[b + 1][ol] = v;
        }
        for (int a = 0; a < cnt_i; ++a) own_keys[(size_t)i * kOwnCap + a] = ((uint64_t)i << 33) | mine[a][ol];
        own_count[i] = (uint32_t)cnt_i;
    }
    if (tx == 0 && tree_kept) atomicAdd(&cnt->tree_found, tree_kept);   // one atomic per block for the kept certificate manifolds
}
// After the scan of own_count: total pair count for the host, and the per-owner blocks copied to their final places.
// Also compares the new key list with the previous step's manifold array (prev_skey[pm], sorted the same way): when nothing differs
// the step keeps that array and works in place (Counters::pairs_differ, read by the host with the pair count).
// SCAN (round 5): scenes of up to kCompactScanBodies bodies need no library scan in front of this kernel (two launches
// less per step): every block of 256 owners sums the counts of the owners before its own (<= 256 KB, coalesced, eight loads in flight per
// lane) and scans its own 256 in LDS; `own_offset` is not read then. Same cost as the library scan + the old kernel (15.6 against 16.3 us on
// the headline pile) without the library on the hot path. Measured and dropped: per-block totals accumulated by k_bp_pairs with one
// atomic per owner (256 owners per address: k_bp_pairs 38 -> 68 us).
constexpr uint32_t kCompactScanBodies = 65536;
template <bool SCAN>
__global__ void __launch_bounds__(256)
k_bp_compact(uint32_t nbodies, const uint64_t *__restrict__ own_keys, const uint32_t *__restrict__ own_count,
                             const uint32_t *__restrict__ own_offset, const uint64_t *__restrict__ extra, uint64_t *out, uint32_t cap,
                             Counters *cnt, const uint64_t *__restrict__ prev_skey, uint32_t pm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total, offset = 0;
    if (SCAN) {
        __shared__ uint32_t red[2][4], wsum[4];
        const uint32_t first = blockIdx.x * 256u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        uint32_t before = 0, all = 0;
        for (uint32_t j0 = threadIdx.x; j0 < nbodies; j0 += 8u * 256u) {   // eight independent (coalesced) loads in flight per lane
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t j = j0 + 256u * u; v[u] = j < nbodies ? own_count[j] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { all += v[u]; before += j0 + 256u * u < first ? v[u] : 0u; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off); all += __shfl_xor(all, off); }
        const uint32_t mine = i < nbodies ? own_count[i] : 0u;
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
        __syncthreads();
        total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        offset = red[0][0] + red[0][1] + red[0][2] + red[0][3] + inc - mine;
        for (uint32_t w = 0; w < wave; ++w) offset += wsum[w];
    } else total = own_offset[nbodies];
    const uint32_t nextra = min(cnt->num_extra, cap);
    if (i == 0) {
        cnt->num_pairs = total + nextra; if (total + nextra > cap) cnt->pair_overflow = 1;
        cnt->bp_rebuilt = cnt->bp_rebuild;
        cnt->bp_rebuild = 0;   // consumed by k_bp_refit / k_bp_walk above; this step's k_finish decides for the next step
        if (total + nextra != pm || nextra != 0) cnt->pairs_differ = 1;   // (surplus keys arrive unsorted: no comparison)
    }
    if (i < nbodies) {
        const uint32_t c = own_count[i], o = SCAN ? offset : own_offset[i];
        bool differ = false;
        for (uint32_t a = 0; a < c; ++a)
            if (o + a < cap) {
                const uint64_t key = own_keys[(size_t)i * kOwnCap + a];
                out[o + a] = key;
                differ = differ || o + a >= pm || prev_skey[o + a] != key;
            }
        if (differ) cnt->pairs_differ = 1;
    }
    for (uint32_t e = i; e < nextra; e += gridDim.x * blockDim.x) if (total + e < cap) out[total + e] = extra[e];
}

// ---- the host's pair filter (edynhip_set_pair_filter: settings.should_collide_func, broadphase.cpp:143-153) ---------------------------
// k_bp_list_new: the positions, in the step's sorted pair list, of the candidates that have no manifold yet (order arbitrary).
// k_bp_drop_rejected: the list without the pairs the host rejected (`rejected`: their positions, ascending).
__global__ void k_bp_list_new(const uint64_t *__restrict__ skeys, uint32_t M, Manifolds prev, uint32_t pm, uint32_t *list, uint32_t *count) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const uint64_t key = skeys[m] >> 1;
    if (find_prev(prev, pm, (uint32_t)(key >> 32), (uint32_t)key) == 0xFFFFFFFFu) list[atomicAdd(count, 1u)] = m;
}
__global__ void k_bp_drop_rejected(const uint64_t *__restrict__ in, uint64_t *out, uint32_t M, const uint32_t *__restrict__ rejected, uint32_t nrej) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    uint32_t a = 0, b = nrej;   // number of rejected positions below m
    while (a < b) { const uint32_t mid = (a + b) >> 1; if (rejected[mid] < m) a = mid + 1; else b = mid; }
    if (a < nrej && rejected[a] == m) return;
    out[m - a] = in[m];
}

// New manifold array from the sorted pair keys; contact points persist from the previous array.
// `speculative`: the launch was enqueued before the host read the pair count (broadphase(), below) - M is then taken from the device
// counters, and the launch does nothing when the host is going to take another path (capacity error, an unsorted surplus that needs the
// extra sort first, an unchanged pair set that keeps last step's array); `first` = the first manifold of this launch (the host covers
// what a speculative grid was too small for with a second launch).
__global__ void k_bp_build_manifolds(const uint64_t *__restrict__ skeys, uint32_t M, Manifolds cur, Manifolds prev, uint32_t pm,
                                     Counters *cnt, uint2 *new_edges, uint32_t *new_edge_m, bool copy_points, EventSink ev, uint8_t *prev_matched,
                                     uint32_t first, bool speculative, bool inplace_allowed) {
    if (speculative) {
        M = cnt->num_pairs;
        if (cnt->pair_overflow || cnt->num_extra || M > cur.cap || (inplace_allowed && M > 0 && M == pm && !cnt->pairs_differ)) return;
    }
    uint32_t m = first + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t found = 0;
    bool is_new = false;
    uint32_t new_hi = 0, new_lo = 0;
    if (m < M) {
    if (m == 0 && M != pm) cnt->pairs_changed = 1;
    const uint64_t sk = skeys[m];
    const uint64_t key = sk >> 1;
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const bool swapped = sk & 1;
    cur.skey[m] = sk;
    cur.bodyA[m] = swapped ? lo : hi;
    cur.bodyB[m] = swapped ? hi : lo;
    {   // segments of the new array, for next step's lookups
        const uint32_t ph = m > 0 ? (uint32_t)(skeys[m - 1] >> 33) : 0xFFFFFFFFu;
        const uint32_t nh = m + 1 < M ? (uint32_t)(skeys[m + 1] >> 33) : 0xFFFFFFFFu;
        if (ph != hi) cur.seg_start[hi] = m;
        if (nh != hi) cur.seg_end[hi] = m + 1;
    }
    uint32_t p = find_prev(prev, pm, hi, lo);
    uint32_t info = kNoColour << 8;
    if (p == 0xFFFFFFFFu) {
        is_new = true; new_hi = hi; new_lo = lo;   // listed below, one counter atomic per workgroup
        if (ev.buf) emit_event(ev, EDYNHIP_EVENT_MANIFOLD_CREATED, swapped ? lo : hi, swapped ? hi : lo, 0);
    }
    cur.prev_idx[m] = p;
    cur.tree[m] = p != 0xFFFFFFFFu ? prev.tree[p] : (uint8_t)0;
    if (p != 0xFFFFFFFFu) {
        found = 1;
        info = prev.info[p];
        if (ev.buf) prev_matched[p] = 1;
        // inside a full step the narrowphase reads the old points straight from the previous array (no copy here)
        const uint32_t np = copy_points ? (info & 0xFF) : 0u;
        for (uint32_t k = 0; k < np; ++k) {
            const size_t ts = pt_at(prev.cap, k, p), td = pt_at(cur.cap, k, m);
            cur.pA[td] = prev.pA[ts]; cur.pB[td] = prev.pB[ts]; cur.nrm[td] = prev.nrm[ts];
            cur.lnrm[td] = prev.lnrm[ts]; cur.imp[td] = prev.imp[ts];
            const size_t s = slot_at(prev.cap, k, p), d = slot_at(cur.cap, k, m);
            if (cur.pid) cur.pid[d] = prev.pid[s];
            if (cur.xmat) { cur.xmat[d] = prev.xmat[s]; cur.ximp[d] = prev.ximp[s]; }
        }
    }
    cur.info[m] = info;
    }
    // the new manifolds' list slots and the found count: one atomic each per WORKGROUP (a scene in motion creates thousands of
    // manifolds per step; one atomic per manifold on one address made this kernel 10x slower there than on the settled pile)
    __shared__ uint32_t w_new[16], w_found[16], base_new;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t mask = __ballot(is_new);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) found += __shfl_xor(found, off);
    if (lane == 0) { w_new[wave] = (uint32_t)__popcll(mask); w_found[wave] = found; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tn = 0, tf = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { tn += w_new[w]; tf += w_found[w]; }
        base_new = tn ? atomicAdd(&cnt->num_new, tn) : 0u;
        if (tn) cnt->pairs_changed = 1;
        if (tf) atomicAdd(&cnt->num_found, tf);
    }
    __syncthreads();
    if (is_new) {
        uint32_t slot = base_new + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) slot += w_new[w];
        new_edges[slot] = make_uint2(new_hi, new_lo); new_edge_m[slot] = m;
    }
}

static inline uint32_t blocks(uint32_t n, uint32_t bs) { return (n + bs - 1) / bs; }
static bool bp_lists_enabled() {   // development knob: EDYNHIP_BP_LISTS=0 walks the tree every step
    static const bool on = !(getenv("EDYNHIP_BP_LISTS") && getenv("EDYNHIP_BP_LISTS")[0] == '0');
    return on;
}

// Contact events: the previous step's manifolds that no pair of this step continues (destroy_separated_manifolds,
// broadphase.cpp:99-134; or a body was removed) - their points end with them.
__global__ void k_ev_destroyed(uint32_t pm, Manifolds prev, uint8_t *matched, EventSink ev) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pm) return;
    if (!matched[p]) {
        const uint32_t a = prev.bodyA[p], b = prev.bodyB[p], np = prev.info[p] & 0xFF;
        for (uint32_t k = 0; k < np; ++k) emit_event(ev, EDYNHIP_EVENT_POINT_DESTROYED, a, b, prev.pid[(size_t)k * prev.cap + p]);
        emit_event(ev, EDYNHIP_EVENT_MANIFOLD_DESTROYED, a, b, 0);
    }
    matched[p] = 0;   // armed for the step in which this array is `prev` again
}

int broadphase(edynhip_ctx *c) {
    hipStream_t s = c->stream;
    const uint32_t np = c->bvh.num_proc;
    Manifolds &prev = c->m[c->cur], &cur = c->m[c->cur ^ 1];
    const uint32_t pm = c->num_manifolds;
    c->inplace_step = false;   // set below when this step keeps last step's manifold array; read (and cleared) by islands()
    if (!c->full_step) hipLaunchKernelGGL(k_step_reset, dim3(1), dim3(64), 0, s, c->cnt);
    uint32_t M = 0;
    if (np > 0) {
        // The tree TOPOLOGY (Morton order + Karras hierarchy) is rebuilt every kRebuildPeriod steps; in between only the
        // boxes are refitted. A refitted tree is still a valid BVH (boxes stay conservative) and the exact predicates
        // run at the leaves, so the pair set is unaffected - only traversal efficiency degrades slowly as bodies move.
        constexpr uint32_t kRebuildPeriod = 8;
        const bool rebuild = c->bvh.age == 0;
        c->bvh.age = (c->bvh.age + 1) % kRebuildPeriod;
        if (rebuild) {
        hipLaunchKernelGGL(k_bp_bounds, dim3(std::min(blocks(np, 256), 32u)), dim3(256), 0, s, c->bvh.np_list + c->bvh.num_np, np, c->b.amin, c->b.amax, c->cnt);
        hipLaunchKernelGGL(k_bp_morton, dim3(blocks(np, 256)), dim3(256), 0, s, c->bvh.np_list + c->bvh.num_np, np, c->b.amin, c->b.amax, c->cnt, c->bvh.keys);
        EH_TRY(sort_u64(c, c->bvh.keys, c->bvh.keys_sorted, np, 32, 62));   // stable: equal codes keep ascending body order
        if (np > 1)
            hipLaunchKernelGGL(k_bp_build, dim3(blocks(np - 1, 256)), dim3(256), 0, s, c->bvh.keys_sorted, (int)np, c->bvh.parent, c->bvh.left, c->bvh.right, c->bvh.visit);
        if (np > 1)
            hipLaunchKernelGGL(k_bp_ropes, dim3(blocks(2 * np - 1, 256)), dim3(256), 0, s, (int)np, c->bvh.parent, c->bvh.right, c->bvh.rope);
        hipLaunchKernelGGL(k_bp_split, dim3(1), dim3(64), 0, s, (int)np, c->bvh.left, c->bvh.right, c->bvh.split);
        }
        // candidate lists: (refit -> walk, both no-ops while the lists are valid) -> exact predicates over the lists.
        // A new topology re-sorts the leaves but the lists are indexed by body, so they survive it; they are rebuilt when a
        // body has moved too far (checked on the device by the previous step's k_finish: Counters::bp_rebuild) or when the
        // host touched the bodies in between (lists_dirty), and always in a stand-alone stage run.
        const CandLists cl{c->bvh.cand_list, c->bvh.cand_count, c->bvh.ref_min, c->bvh.ref_max};
        // A new topology also needs its boxes: the node records (box + left child + rope) are only written by the refit, and a body
        // whose list overflowed walks the tree in k_bp_pairs EVERY step - with last refit's records under this step's leaf order it
        // would test the wrong bodies (found by the settled C3 scene: fast spheres, lists of > 64 candidates, pairs lost in the step
        // after a topology rebuild). So a rebuild step refits and re-walks.
        const uint32_t force = (c->bvh.lists_dirty || !c->full_step || !bp_lists_enabled() || rebuild) ? 1u : 0u;
        c->bvh.lists_dirty = false;
        hipLaunchKernelGGL(k_bp_refit, dim3(blocks(np, 256)), dim3(256), 0, s, c->bvh.keys_sorted, (int)np, c->bvh.parent, c->bvh.left, c->bvh.right, c->bvh.rope, c->b.amin, c->b.amax, c->bvh.nmin, c->bvh.nmax, c->bvh.visit, c->cnt, c->bvh.ref_min, c->bvh.ref_max, c->b.linvel, c->b.angvel, c->cfg.fixed_dt, force,
                           sqrtf(c->cfg.gravity[0] * c->cfg.gravity[0] + c->cfg.gravity[1] * c->cfg.gravity[1] + c->cfg.gravity[2] * c->cfg.gravity[2]), c->bvh.lookahead);
        hipLaunchKernelGGL(k_bp_walk, dim3(blocks(np * kWalkLanes, 256)), dim3(256), 0, s, c->bvh.keys_sorted, (int)np, c->bvh.nmin, c->bvh.nmax, c->b.amin, c->b.amax, cl, c->cnt, c->bvh.visit, c->sleeping, force, c->bvh.right);
        hipLaunchKernelGGL(k_bp_pairs, dim3(blocks(np, kOwnersPerBlock)), dim3(kBpBlock), 0, s, c->bvh.keys_sorted, (int)np, c->bvh.nmin, c->bvh.nmax, c->b.amin, c->b.amax, Filt{c->b.group, c->b.mask, c->excl, c->pair_filter != nullptr}, c->bvh.np_list, c->bvh.num_np, prev, pm, c->own_keys, c->own_count, c->pair_keys, cur.cap, c->cnt, cl, c->b.flags, c->sleeping, c->bvh.split, c->bvh.rope);
        {   // developer knob EDYNHIP_BP_STATS=1: what k_bp_pairs had to do, every 100th step (candidate-list lengths, owners that walked the tree, pairs kept)
            static const bool bp_stats = getenv("EDYNHIP_BP_STATS") != nullptr;
            if (bp_stats && c->step_index % 100 == 50) {
                std::vector<uint32_t> cnt(c->b.n), own(c->b.n);
                EH_HIP(c, hipMemcpyAsync(cnt.data(), c->bvh.cand_count, (size_t)c->b.n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                EH_HIP(c, hipMemcpyAsync(own.data(), c->own_count, (size_t)c->b.n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                EH_HIP(c, hipStreamSynchronize(s));
                uint64_t sum = 0, pairs = 0; uint32_t mx = 0, over = 0, listed = 0;
                uint32_t hist[9] = {0};
                for (uint32_t i = 0; i < c->b.n; ++i) {
                    pairs += own[i];
                    if (cnt[i] == kListOverflow) { ++over; continue; }
                    sum += cnt[i]; mx = std::max(mx, cnt[i]); ++listed;
                    ++hist[std::min<uint32_t>(cnt[i] / 16, 8)];
                }
                fprintf(stderr, "[bp stats] step %u: bodies %u, candidates per body mean %.1f max %u, owners walking the tree %u, pairs kept %llu (%.2f per body); candidates/16 histogram:",
                        c->step_index, c->b.n, listed ? (double)sum / listed : 0.0, mx, over, (unsigned long long)pairs, (double)pairs / std::max(c->b.n, 1u));
                for (int k = 0; k < 9; ++k) fprintf(stderr, " %u", hist[k]);
                fprintf(stderr, "; look-ahead %.2f steps, rebuild history %04x\n", c->bvh.lookahead, c->bvh.rebuild_hist & 0xFFFFu);
            }
        }
        // owners in index order: offsets = exclusive scan of the per-owner counts (own_count[n] = 0 -> own_offset[n] = total)
        static const bool direct_env = !(getenv("EDYNHIP_DIRECT_COMPACT") && getenv("EDYNHIP_DIRECT_COMPACT")[0] == '0');   // developer knob (A/B)
        if (c->b.n <= kCompactScanBodies && direct_env) {
            hipLaunchKernelGGL(k_bp_compact<true>, dim3(blocks(c->b.n, 256)), dim3(256), 0, s, c->b.n, c->own_keys, c->own_count, c->own_offset, c->pair_keys, c->pair_keys_sorted, cur.cap, c->cnt, prev.skey, pm);
        } else {
            EH_TRY(scan_u32(c, c->own_count, c->own_offset, c->b.n + 1));
            hipLaunchKernelGGL(k_bp_compact<false>, dim3(blocks(c->b.n, 256)), dim3(256), 0, s, c->b.n, c->own_keys, c->own_count, c->own_offset, c->pair_keys, c->pair_keys_sorted, cur.cap, c->cnt, prev.skey, pm);
        }
        // The pair count is needed on the host to size the manifold kernels - but the first of them need not wait for it: inside a full
        // step the build is enqueued right behind the counter publish, over a grid sized from last step's count, reads the count on the
        // device and stands down by itself in the cases decided below (speculative launch; what its grid did not cover is launched
        // after the fetch). The GPU builds while the answer travels to the host (11 us of idle GPU per step on the headline pile).
        static const bool spec_env = !(getenv("EDYNHIP_SPECULATE") && getenv("EDYNHIP_SPECULATE")[0] == '0');
        static const bool inplace_env = !(getenv("EDYNHIP_INPLACE") && getenv("EDYNHIP_INPLACE")[0] == '0');
        const bool inplace_allowed = inplace_env && c->full_step && !c->events && !c->force_islands;
        const EventSink ev = event_sink(c);
        uint32_t covered = 0;
        uint32_t ticket = 0;
        EH_TRY(publish_counters(c, sizeof(Counters) - sizeof(uint32_t) * 8 * kMaxColours, &ticket));
        const uint32_t spec_grid = std::min<uint32_t>(cur.cap / 512u, blocks(pm + pm / 16 + 2048, 512));
        if (spec_env && c->full_step && pm > 0 && spec_grid > 0 && !c->pair_filter) {   // (a host pair filter may shorten the list first)
            covered = spec_grid * 512u;
            hipLaunchKernelGGL(k_bp_build_manifolds, dim3(spec_grid), dim3(512), 0, s, c->pair_keys_sorted, 0u, cur, prev, pm, c->cnt, c->new_edges, c->new_edge_m, false, ev, c->prev_matched,
                               0u, true, inplace_allowed);
        }
        EH_TRY(wait_counters(c, ticket));
        if (c->cnt_host->pair_overflow) return set_error(c, EDYNHIP_ERR_CAPACITY, c->cnt_host->pair_overflow == 2 ? "broadphase: BVH traversal stack exhausted" : "broadphase: pair capacity (max_manifolds) exceeded");
        if (c->cnt_host->df_abort) return set_error(c, EDYNHIP_ERR_INTERNAL, "dataflow solve: a hand-off never arrived in the previous step (workgroups not co-resident?)");
        M = c->cnt_host->num_pairs;
        {   // adapt the lists' look-ahead (ctx.hpp LBVH::lookahead): rebuilt in each of the last 4 steps -> halve; at most twice in the last 8 -> double
            static const bool adapt = !(getenv("EDYNHIP_BP_ADAPT") && getenv("EDYNHIP_BP_ADAPT")[0] == '0');   // developer knob (A/B)
            const bool rebuilt = force != 0 || c->cnt_host->bp_rebuilt != 0;
            c->bvh.rebuild_hist = (c->bvh.rebuild_hist << 1) | (rebuilt ? 1u : 0u);
            if (adapt && c->full_step) {
                if ((c->bvh.rebuild_hist & 0xFu) == 0xFu) c->bvh.lookahead = std::max(kListLookaheadMin, 0.5f * c->bvh.lookahead);
                else if (__builtin_popcount(c->bvh.rebuild_hist & 0xFFu) <= 2) c->bvh.lookahead = std::min(kListLookaheadMax, 2.0f * c->bvh.lookahead);
            }
        }
        if (c->cnt_host->num_extra) {   // some owner had more than kOwnCap partners: its surplus sits unsorted at the end
            int hb = 1; while ((1u << hb) < c->b.n && hb < 31) ++hb;
            EH_HIP(c, hipMemcpyAsync(c->pair_keys, c->pair_keys_sorted, (size_t)M * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
            EH_TRY(sort_u64(c, c->pair_keys, c->pair_keys_sorted, M, 0, 33 + hb));
        }
        // Unchanged pair set (the common case once a scene has settled, and all that a sleeping world ever sees): the previous
        // step's manifold array IS this step's - no rebuild, no copy of the contact points into the other array; the
        // narrowphase works in place and sleeping manifolds cost nothing. (With contact events the build also keeps the
        // created / destroyed bookkeeping, so it runs.)
        if (inplace_allowed && M > 0 && M == pm && !c->cnt_host->pairs_differ) {
            c->points_in_prev = false;
            c->inplace_step = true;
            c->prev_num_manifolds = pm;
            c->num_manifolds = M;
            EH_HIP(c, hipGetLastError());
            return EDYNHIP_OK;
        }
        // The host's pair filter: the step's new candidates go to the host, the rejected ones leave the list (slow path, see edynhip.h).
        if (c->pair_filter && M > 0) {
            uint32_t *count = &c->cnt->num_new;   // (free until the build counts the new manifolds; k_step_reset zeroed it)
            EH_HIP(c, hipMemsetAsync(count, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_bp_list_new, dim3(blocks(M, 256)), dim3(256), 0, s, c->pair_keys_sorted, M, prev, pm, c->filter_new_idx, count);
            uint32_t nnew = 0;
            EH_HIP(c, hipMemcpyAsync(&nnew, count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            EH_HIP(c, hipStreamSynchronize(s));
            if (nnew > 0) {
                std::vector<uint32_t> idx(nnew);
                EH_HIP(c, hipMemcpy(idx.data(), c->filter_new_idx, (size_t)nnew * sizeof(uint32_t), hipMemcpyDeviceToHost));
                std::sort(idx.begin(), idx.end());
                std::vector<uint64_t> keys(M);   // (the whole list: one contiguous copy instead of nnew gathers)
                EH_HIP(c, hipMemcpy(keys.data(), c->pair_keys_sorted, (size_t)M * sizeof(uint64_t), hipMemcpyDeviceToHost));
                std::vector<uint32_t> rejected;
                for (uint32_t m : idx) {
                    const uint64_t sk = keys[m], key = sk >> 1;
                    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
                    const bool swapped = sk & 1;   // body[0] - the querying body - is `lo`
                    if (!c->pair_filter(c->pair_filter_user, swapped ? lo : hi, swapped ? hi : lo)) rejected.push_back(m);
                }
                if (!rejected.empty()) {
                    EH_HIP(c, hipMemcpyAsync(c->filter_new_idx, rejected.data(), rejected.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(k_bp_drop_rejected, dim3(blocks(M, 256)), dim3(256), 0, s, c->pair_keys_sorted, c->pair_keys, M, c->filter_new_idx, (uint32_t)rejected.size());
                    M -= (uint32_t)rejected.size();
                    EH_HIP(c, hipMemcpyAsync(c->pair_keys_sorted, c->pair_keys, (size_t)M * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
                    EH_HIP(c, hipMemcpyAsync(&c->cnt->num_pairs, &M, sizeof(uint32_t), hipMemcpyHostToDevice, s));
                    EH_HIP(c, hipStreamSynchronize(s));   // (`rejected` and M are host memory that must outlive the copies)
                }
            }
            EH_HIP(c, hipMemsetAsync(count, 0, sizeof(uint32_t), s));
        }
        if (!c->full_step) {   // inside edynhip_step the previous step's k_finish already cleared these
            EH_HIP(c, hipMemsetAsync(cur.seg_start, 0, (size_t)c->b.cap * sizeof(uint32_t), s));
            EH_HIP(c, hipMemsetAsync(cur.seg_end, 0, (size_t)c->b.cap * sizeof(uint32_t), s));
        }
        if (c->cnt_host->num_extra) covered = 0;   // the speculative launch stood down: the keys were not sorted yet
        if (M > covered)
            hipLaunchKernelGGL(k_bp_build_manifolds, dim3(blocks(M - covered, 512)), dim3(512), 0, s, c->pair_keys_sorted, M, cur, prev, pm, c->cnt, c->new_edges, c->new_edge_m, !c->full_step, ev, c->prev_matched,
                               covered, false, false);
        if (M == 0 && pm != 0) c->force_islands = true;
        if (ev.buf && pm > 0) hipLaunchKernelGGL(k_ev_destroyed, dim3(blocks(pm, 256)), dim3(256), 0, s, pm, prev, c->prev_matched, ev);
    }
    c->points_in_prev = c->full_step && M > 0 && np > 0;
    c->cur ^= 1;
    c->prev_num_manifolds = pm;
    c->num_manifolds = M;
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

}  // namespace eh
