// Islands, graph colouring, constraint-row preparation, the sequential-impulse (PGS) velocity solver,
// integration, the position solver and the post-step derived state, as gfx950 kernels.
//
// Reference functions reproduced (arithmetic order kept):
//   island labelling            src/edyn/simulation/island_manager.cpp:117-350 (connected components; static nodes do not connect)
//   apply_gravity               include/edyn/sys/apply_gravity.hpp:12-17
//   contact rows                src/edyn/constraints/contact_constraint.cpp:15-56
//   point / hinge rows          src/edyn/constraints/point_constraint.cpp:9-46, hinge_constraint.cpp:26-67
//   prepare_row                 src/edyn/constraints/constraint_row.cpp:6-22
//   warm start / solve / apply  src/edyn/constraints/constraint_row.cpp:24-57, island_solver.cpp:76-111
//   friction circle             src/edyn/constraints/constraint_row_friction.cpp:11-66
//   integrate_velocities        src/edyn/dynamics/island_solver.cpp:357-376, src/edyn/math/quaternion.cpp:7-22
//   position solver             include/edyn/dynamics/position_solver.hpp:16-51, contact_constraint.cpp:58-90,
//                               hinge_constraint.cpp:180-213, island_solver.cpp:350-353,538-542
//   update_aabbs / inertias     src/edyn/util/aabb_util.cpp:42-70, src/edyn/sys/update_inertias.cpp:12-24
//
//   sleeping                    src/edyn/simulation/island_manager.cpp:524-623 (k_sleep_*)
//
// What is NOT in the reference: within an island the reference sweeps rows strictly sequentially
// (Gauss-Seidel). Here the contact graph is edge-coloured so that the manifolds of one colour share no
// procedural body, and every body meets its manifolds in colour order. Per iteration: joints by colour, then
// contacts by colour; a manifold sweeps its normal rows and then its friction rows (the friction circle uses
// the normal impulse just updated, as in the reference); the reference's global "all rows, then all friction
// rows" split is kept per manifold. Two schedules produce that order, with bit-identical results:
//   * dataflow (contact-only scenes): ONE launch per velocity solve (k_contact_solve_df2 / _df) and per position
//     iteration (k_pos_contacts_df); a manifold waits for the tagged hand-off of its two bodies' state from each
//     body's previous manifold and hands it on to the next - no kernel boundary or barrier per colour;
//   * per colour (scenes with joints, or when the resident-grid launch is refused): one launch per colour with one
//     manifold per lane (k_contact_solve, k_joint_solve, k_pos_contacts, k_pos_joints).
#include <mutex>
#include "ctx.hpp"
#include "dpolyhedron.hpp"

namespace eh {
using namespace dm;

static inline uint32_t blocks(uint32_t n, uint32_t bs) { return (n + bs - 1) / bs; }

DI bool is_dynamic(uint32_t flags) { return (flags & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC; }
constexpr uint32_t kCsBlock = 1024, kCsKeys = 256, kCsDirectBlocks = 256, kCsSuper = 16;   // the colour sort (k_cs_*)
// Edge priority for the colouring rounds: lower edge index wins, which reproduces sequential first-fit
// colouring in canonical pair order (near-optimal colour counts on stacked scenes: max degree or +1) at the
// price of more rounds when a whole scene is coloured from scratch; steady state only colours new edges.
DI uint64_t edge_prio(uint32_t e) { return (uint64_t)(0xFFFFFFFFu - e); }

// ------------------------------------------------------------------ islands (lock-free union-find)
// Links always point to a SMALLER body index and only ever move towards the root, so any value ever stored in
// parent[x] is an ancestor of x (or x itself) for the rest of the launch. Plain, possibly stale (per-CU L1 / per-XCD
// L2) loads are therefore safe for the walks: a stale value is merely a longer path. Only the hook itself must be
// exact - it is a device-scope compare-and-swap on the true memory value, and on failure the walk continues from the
// fresh value it returned. (Agent-scope atomic loads here were measured ~8x slower: every step went to the fabric.)
DI uint32_t cc_find(uint32_t *parent, uint32_t x) {
    uint32_t p = parent[x];
    while (p != x) {
        const uint32_t gp = parent[p];
        if (gp != p) parent[x] = gp;   // path halving; racing writers all store ancestors
        x = p; p = gp;
    }
    return x;
}
DI bool cc_union(uint32_t *parent, uint32_t a, uint32_t b) {   // true: this call joined two trees (the edge certifies the union)
    uint32_t ra = cc_find(parent, a), rb = cc_find(parent, b);
    while (ra != rb) {
        if (ra < rb) { const uint32_t t = ra; ra = rb; rb = t; }   // hook the larger root under the smaller
        const uint32_t seen = atomicCAS(&parent[ra], ra, rb);
        if (seen == ra) return true;
        ra = cc_find(parent, seen);   // ra was no longer a root: continue from what it points to now
        rb = cc_find(parent, rb);
    }
    return false;
}
// Island labels are maintained incrementally, and the HOST picks the mode from counters it fetched with the pair count:
//   CC_SKIP         unchanged pair set (in-place step): nothing is launched;
//   CC_INCREMENTAL  no island can have split: start from last step's labels (roots = a depth-1 forest), hook the new edges;
//   CC_FULL         recompute over all joints and manifolds (scene edits, or a certificate manifold disappeared).
// "No island can have split" is decided with a CERTIFICATE: every union the forest ever performed was made on a particular
// edge - a joint, or a manifold, which is then marked (Manifolds::tree). The marked edges form a spanning forest of the
// contact graph, so as long as every marked manifold of the previous array is still in this step's pair set
// (Counters::tree_found == tree_total, counted by k_bp_pairs while it re-tests the existing pairs) the components are intact,
// whatever other pairs went away. Manifolds that carry contact points are hooked first and separated AABBs almost always
// belong to manifolds without points, so on a settled pile the full recompute (~100 us: its cost is the depth of the
// initial forest) went from most steps to almost none. (Round 2 recomputed whenever ANY pair disappeared.)
enum { CC_SKIP = 0, CC_INCREMENTAL = 1, CC_FULL = 2 };
DI void cc_count_marks(uint32_t marks, Counters *cnt) {   // every lane of the wave calls this
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) marks += __shfl_xor(marks, off);
    if ((threadIdx.x & 63) == 0 && marks) atomicAdd(&cnt->tree_marks, marks);
}
// CC_FULL only: every body starts at its smallest dynamic lower-index neighbour it has CONTACT POINTS with (its manifolds
// with lower-index partners are the contiguous segment [seg_start, seg_end) of the sorted array). Links point to smaller
// indices, so this is a valid forest and most unions below find their roots already merged. Clears the segment's marks.
// (Round 6, built, measured and withdrawn - profiles/r06_tree_repair_experiment/, the patch is kept there: LOCAL REPAIR of the certificate. k_bp_pairs
//  listed the marked manifolds the new pair set drops, an extra workgroup of k_bp_compact looked for a replacement path a - c - b over manifolds
//  that exist in both arrays (c among a's lower-index partners) and marked it, the host then kept the incremental mode. Bit-exact, but a step
//  drops SEVERAL certificate manifolds and every one needs its path: 8 of 135 relabelling steps repaired on the headline pile, 20 of 384 on
//  mixed32k, 13 of 129 on islands256k, none on the polyhedron heap, which paid 4 % for the listing. EDYNHIP_TREE_STATS=1 prints the counts.)
// (Round 6, measured and dropped: offering the edges in classes of decreasing STABILITY - manifolds whose oldest point has lived for 32 steps,
//  then the other manifolds with points, then the pointless ones - so that the certificate consists of long-lived contacts. The number of
//  steps that relabel in full did not move (mixed32k 372 against 373 of 440, pile32k 127 / 128, islands256k 277 / 277, the polyhedron heap
//  every step either way) and the two kernels got slower (k_cc_hook_bodies 55 -> 70 us): what breaks a certificate on these scenes is not
//  a young contact flickering but some long-lived pair of 32 768 bodies separating, in nearly every step. scripts/runs/r6k.sh.)
__global__ void k_cc_init(uint32_t n, uint32_t *forest, Counters *cnt, Manifolds mf, uint32_t M, const uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cnt->num_islands = 0;
    uint32_t marks = 0;
    if (i < n) {
        uint32_t parent = i;
        if (M && is_dynamic(flags[i])) {
            for (uint32_t s = mf.seg_start[i], e = mf.seg_end[i]; s < e; ++s) {
                const uint32_t lo = (uint32_t)(mf.skey[s] >> 1);
                uint8_t mark = 0;
                if (parent == i && (mf.info[s] & 0xFF) != 0 && is_dynamic(flags[lo])) { parent = lo; mark = 1; marks = 1; }
                mf.tree[s] = mark;
            }
        }
        forest[i] = parent;
    }
    cc_count_marks(marks, cnt);
}
// CC_FULL, between the initial forest and the unions (round 5): every body's link goes straight to its root. The initial forest of a pile
// is a set of chains ~100 links deep (each body under its lowest lower-index partner); without this pass every union of k_cc_hook_bodies
// walks such a chain twice. The walks halve the paths they pass, so a second pass costs little where the first has been.
__global__ void k_cc_compress(uint32_t n, uint32_t *forest, const uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && is_dynamic(flags[i])) forest[i] = cc_find(forest, i);   // (a racing hook cannot exist here: only finds run in this launch)
}
__global__ void k_cc_hook(uint32_t M, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                          const uint32_t *__restrict__ flags, uint32_t *island) {   // joints (CC_FULL): edges that only an edit removes
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M) return;
    uint32_t a = bA[e], b = bB[e];
    if (is_dynamic(flags[a]) && is_dynamic(flags[b])) (void)cc_union(island, a, b);
}
// Vertex-centric hooking for the full recompute: one lane per body walks the contiguous run of manifolds in which it
// is the higher-index partner. All unions of one body are issued by one lane in sequence, so lanes do not fight over
// the same root the way one-lane-per-edge does when a body has 6-12 partners. Manifolds with contact points first.
__global__ void k_cc_hook_bodies(uint32_t n, Manifolds mf, uint32_t M, const uint32_t *__restrict__ flags, uint32_t *island, Counters *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t marks = 0;
    if (i < n && M != 0 && is_dynamic(flags[i])) {
        const uint32_t s0 = mf.seg_start[i], s1 = mf.seg_end[i];
        auto hook = [&](uint32_t s) {
            const uint32_t lo = (uint32_t)(mf.skey[s] >> 1);
            if (!is_dynamic(flags[lo])) return;
            if (island[i] == island[lo]) return;   // both under the same node: one tree already (two loads instead of two walks; most edges of a pile end here)
            if (cc_union(island, i, lo)) { mf.tree[s] = 1; ++marks; }
        };
        // the manifolds with contact points first; the others are remembered (a bit each: an owner keeps at most kOwnCap = 64 in its segment,
        // longer segments take the plain second pass) and visited afterwards without reading the point counts again
        uint64_t later = 0;
        const bool fits = s1 - s0 <= 64u;
        for (uint32_t s = s0; s < s1; ++s) {
            if ((mf.info[s] & 0xFF) != 0) hook(s);
            else if (fits) later |= 1ull << (s - s0);
        }
        if (fits) for (; later; later &= later - 1) hook(s0 + (uint32_t)__ffsll((long long)later) - 1u);
        else for (uint32_t s = s0; s < s1; ++s) if ((mf.info[s] & 0xFF) == 0) hook(s);
    }
    cc_count_marks(marks, cnt);
}
__global__ void k_cc_hook_new(const uint2 *__restrict__ edges, const uint32_t *__restrict__ edge_m, uint8_t *tree, const uint32_t *__restrict__ flags,
                              uint32_t *island, Counters *cnt) {   // CC_INCREMENTAL: `island` holds last step's labels
    const uint32_t n = cnt->num_new;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->num_islands = 0;
    uint32_t marks = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        uint2 ed = edges[e];
        if (is_dynamic(flags[ed.x]) && is_dynamic(flags[ed.y]) && cc_union(island, ed.x, ed.y)) { tree[edge_m[e]] = 1; ++marks; }
    }
    cc_count_marks(marks, cnt);
}
// The per-body start of the solve (k_solve_begin: gravity, zeroed deltas, hand-off chain heads) - also folded into k_cc_flatten, the
// per-body kernel that precedes it, when nothing that runs in between reads velocities or sleep flags (no island sleeping, no restitution).
DI void solve_begin_body(uint32_t i, Bodies &b, float dt, uint32_t *first_slot) {
    first_slot[i] = 0xFFFFFFFFu;
    uint32_t fl = b.flags[i];
    float inv_m = 0;
    if (is_dynamic(fl)) {
        inv_m = B_POS(b, i).w;
        f3 g = from4(b.grav[i]);
        if (!(g.x == 0 && g.y == 0 && g.z == 0) && !(fl & BF_ASLEEP)) {   // apply_gravity.hpp:13 excludes sleeping bodies
            f3 v = from4(b.linvel[i]);
            v += g * dt;
            b.linvel[i] = to4(v, 0);
        }
    }
    B_DV(b, i) = make_float4(0, 0, 0, inv_m);
    B_DW(b, i) = make_float4(0, 0, 0, 0);
}
enum { SL_FAST = 1, SL_DISABLED = 2, SL_HAS_ASLEEP = 4, SL_HAS_AWAKE = 8, SL_WAKE = 16, SL_SPLIT = 32 };   // island state bits (k_sleep_*)
enum { SLA_KEEP = 0, SLA_AWAKE = 1, SLA_SLEEP = 2 };
// split_state (island sleeping, full relabel only): a body whose root differs from the root of last step's root of its island is a
// part of an island that fell apart - both parts are marked, k_sleep_decide starts their timers again (split_islands,
// island_manager.cpp:411-447: every part of a split island ends up with an empty sleep_timestamp).
template <bool BEGIN>
__global__ void k_cc_flatten(uint32_t n, const uint32_t *__restrict__ flags, uint32_t *island, uint32_t *label, Counters *cnt, int mode, Bodies b, float dt, uint32_t *first_slot,
                             uint32_t *split_state) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cnt->tree_total = (mode == CC_FULL ? 0u : cnt->tree_total) + cnt->tree_marks;   // the hooks are done (kernel boundary)
    uint32_t root = 0;
    if (i < n) {
        uint32_t r = cc_find(island, i);
        if (split_state && is_dynamic(flags[i]) && !(flags[i] & BF_REMOVED)) {
            const uint32_t o = label[i];   // last step's root of this body's island (a full relabel works on a scratch forest: `label` is still last step's here)
            if (o < n && is_dynamic(flags[o]) && !(flags[o] & BF_REMOVED)) {
                const uint32_t ro = cc_find(island, o);
                if (ro != r) { atomicOr(&split_state[r], (uint32_t)SL_SPLIT); atomicOr(&split_state[ro], (uint32_t)SL_SPLIT); }
            }
        }
        label[i] = r;
        root = (r == i && is_dynamic(flags[i])) ? 1u : 0u;
        if (BEGIN) solve_begin_body(i, b, dt, first_slot);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) root += __shfl_xor(root, off);
    if ((threadIdx.x & 63) == 0 && root) atomicAdd(&cnt->num_islands, root);
}

// ------------------------------------------------------------------ island sleeping (island_manager.cpp:524-623)
// Islands are identified by their label (lowest body index). Per step, after the labels: (1) reduce every island's
// bodies into state bits, (2) mark the islands that received a manifold created this step, (3) one lane per island
// decides - wake (new edge, or sleeping and awake bodies merged), keep sleeping, run / restart the timer, go to sleep
// once the timer has run for more than island_time_to_sleep (measured on the step time stamps, ctx.hpp sim_clock) - (4) every body applies its island's decision (put_to_sleep zeroes velocities).
// merge_islands (island_manager.cpp:297-350): the BIGGEST of the islands that merge - nodes + edges - survives with its sleep timer.
// Labels are lowest body indices, so the surviving timer is carried to the merged island's label: per new island, the timer of the biggest
// of last step's islands it consists of; size = its procedural bodies + the edges it had before this step (manifolds that persist from
// the previous array, joints), ties: the lowest old label (the checker's coloured order counts the same; pinned to the engine by
// tests/test_reference_engine.py::test_island_merge_keeps_the_bigger_islands_sleep_timer_like_the_real_engine). Three small passes
// in the steps of a world with island sleeping that relabel: k_sleep_sizes (sizes of last step's islands, keyed by last step's labels -
// a copy taken before the hooks, the union-find halves paths in place), the candidate maximum in k_sleep_scan, k_sleep_carry.
struct SleepMerge { const uint32_t *old_label; uint32_t prev_n; uint32_t *size; unsigned long long *best; double *carried; };
DI void add_by_label(uint32_t label, uint32_t amount, uint32_t *dst) {   // every lane of the wave calls this (amount 0 = nothing): one atomic per distinct label and wave
    uint64_t todo = __ballot(amount != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t l = (uint32_t)__shfl((int)label, leader);
        const bool mine = amount != 0 && label == l;
        uint32_t sum = mine ? amount : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&dst[l], sum);
        todo &= ~__ballot(mine);
    }
}
DI void or_by_label(uint32_t label, uint32_t bits, uint32_t *dst) {   // every lane of the wave calls this (bits 0 = nothing): one atomic per distinct label and wave
    // (one lane per body OR-ing into its island's word serialises on that word: a 32k-body pile - one island - spent 0.4 ms per step here)
    uint64_t todo = __ballot(bits != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t l = (uint32_t)__shfl((int)label, leader);
        const bool mine = bits != 0 && label == l;
        uint32_t all = mine ? bits : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) all |= (uint32_t)__shfl_xor((int)all, off);
        // the word only gains bits while this kernel runs (k_sleep_decide zeroed it): a stale read can cost a redundant atomic, never a lost bit
        if ((int)(threadIdx.x & 63) == leader && (__hip_atomic_load(&dst[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & all) != all) atomicOr(&dst[l], all);
        todo &= ~__ballot(mine);
    }
}
__global__ void k_sleep_sizes(uint32_t n, Bodies b, Manifolds mf, uint32_t M, Joints j, SleepMerge sm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t label = 0, amount = 0;
    if (i < n && i < sm.prev_n && is_dynamic(b.flags[i]) && !(b.flags[i] & BF_REMOVED)) {
        label = sm.old_label[i];
        amount = 1;
        if (M) for (uint32_t s = mf.seg_start[i], e = mf.seg_end[i]; s < e; ++s) amount += mf.prev_idx[s] != 0xFFFFFFFFu ? 1u : 0u;   // this body's manifolds (it is their owner) that existed before this step
        if (label >= n) amount = 0;
    }
    add_by_label(label, amount, sm.size);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < j.n; e += gridDim.x * blockDim.x) {   // joints (few): plain atomics
        const uint32_t a = j.bodyA[e], bb = j.bodyB[e], x = is_dynamic(b.flags[a]) ? a : bb;
        if (x < sm.prev_n && is_dynamic(b.flags[x]) && sm.old_label[x] < n) atomicAdd(&sm.size[sm.old_label[x]], 1u);
    }
}
__global__ void k_sleep_carry(uint32_t n, Bodies b, SleepMerge sm, const double *__restrict__ since) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = sm.best[i];
    sm.best[i] = 0ull; sm.size[i] = 0u;   // armed for the next relabelling step
    sm.carried[i] = key ? since[~(uint32_t)key] : -1.0;
}
__global__ void k_sleep_scan(uint32_t n, Bodies b, uint32_t *state, SleepMerge sm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t label = 0, bits = 0;
    if (i < n) {
        const uint32_t fl = b.flags[i];
        if (is_dynamic(fl)) {
            label = b.island[i];
            // one of last step's island roots: a candidate for the timer of the island it is in now
            if (sm.best && i < sm.prev_n && !(fl & BF_REMOVED) && sm.old_label[i] == i)
                atomicMax(&sm.best[label], ((unsigned long long)sm.size[i] << 32) | (unsigned long long)(~i));
            const f3 v = from4(b.linvel[i]), w = from4(b.angvel[i]);
            const float lin = 0.005f, ang = 3.1415926535897932384626433832795029f / 48.0f;   // config/constants.hpp:41-42
            bits = (fl & BF_ASLEEP) ? SL_HAS_ASLEEP : SL_HAS_AWAKE;
            if (length_sqr(v) > lin * lin || length_sqr(w) > ang * ang) bits |= SL_FAST;
            if (fl & BF_NOSLEEP) bits |= SL_DISABLED;
        }
    }
    or_by_label(label, bits, state);
}
__global__ void k_sleep_edges(const uint2 *__restrict__ edges, const Counters *cnt, const uint32_t *__restrict__ label, uint32_t *state) {
    const uint32_t n = cnt->num_new;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x)
        atomicOr(&state[label[edges[e].x]], (uint32_t)SL_WAKE);   // .x = the pair's owner: always procedural
}
__global__ void k_sleep_decide(uint32_t n, Bodies b, uint32_t *state, uint32_t *action, double *since, double now, const double *__restrict__ carried) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = state[i];
    state[i] = 0;
    if (!is_dynamic(b.flags[i]) || b.island[i] != i) { since[i] = -1.0; return; }
    if (carried) since[i] = carried[i];   // a relabelling step: the timer of the biggest island this one is made of (k_sleep_carry)
    if (s & SL_SPLIT) since[i] = -1.0;   // a part of an island that split: the timer starts again
    const bool wake = (s & SL_WAKE) || ((s & SL_HAS_ASLEEP) && (s & SL_HAS_AWAKE));
    if ((s & SL_HAS_ASLEEP) && !(s & SL_HAS_AWAKE) && !wake) { action[i] = SLA_KEEP; return; }
    uint32_t a = SLA_AWAKE;
    if (!(s & SL_DISABLED) && !(s & SL_FAST)) {
        const double t0 = since[i];
        if (!(t0 >= 0.0)) since[i] = now;                                    // not running (a negative value or the all-ones fill)
        else if (now - t0 > 2.0) { a = SLA_SLEEP; since[i] = -1.0; }         // island_time_to_sleep, constants.hpp:48
    } else since[i] = -1.0;
    action[i] = a;
}
__global__ void k_sleep_apply(uint32_t n, Bodies b, const uint32_t *__restrict__ action, Counters *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t awake = 0;
    if (i < n) {
        uint32_t fl = b.flags[i];
        if (is_dynamic(fl)) {
            const uint32_t a = action[b.island[i]];
            if (a == SLA_AWAKE) { if (fl & BF_ASLEEP) { fl &= ~BF_ASLEEP; b.flags[i] = fl; } }
            else if (a == SLA_SLEEP) {
                fl |= BF_ASLEEP; b.flags[i] = fl;
                b.linvel[i] = make_float4(0, 0, 0, 0); b.angvel[i] = make_float4(0, 0, 0, 0);
            }
            awake = (fl & BF_ASLEEP) ? 0u : 1u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) awake += __shfl_xor(awake, off);
    if ((threadIdx.x & 63) == 0 && awake) atomicAdd(&cnt->num_awake, awake);
}

// ------------------------------------------------------------------ colouring
// In every island that has an edge to colour in this step (a new or re-activated contact) the top colour carried over from the last
// step is released and first-fit again, so colour classes freed by vanished contacts are reclaimed when the island next changes and
// the colour count (= dependent hops per sweep) does not drift upwards - while an island in which nothing happened keeps its
// colouring untouched (4096 settled mini-piles: nothing to recolour). Per ISLAND, not per world: an island is coloured - and
// therefore solved - the same way whatever else the world holds, so a shard of the world (edyn_amd/parallel.py) steps exactly like
// the whole. k_col_tops finds each island's top colour (+ 1; .x) and whether it has an uncoloured active edge (.y), k_col_prepare releases.
__device__ __forceinline__ uint32_t manifold_label(uint32_t a, uint32_t b, uint32_t fa, const uint32_t *__restrict__ island) { return island[is_dynamic(fa) ? a : b]; }
__global__ void __launch_bounds__(1024) k_col_tops(uint32_t M, const uint32_t *__restrict__ info, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                                                   const uint32_t *__restrict__ flags, const uint32_t *__restrict__ island, uint2 *isl_top, uint32_t *cs_sup) {
    __shared__ uint32_t s_label[16], s_top[16];
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    // (first kernel of the colouring chain: also clears the colour sort's super-block key counts, k_cs_hist<true>)
    for (uint32_t g = m; g < (kCsDirectBlocks / kCsSuper) * kCsKeys; g += gridDim.x * blockDim.x) cs_sup[g] = 0u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t label = 0xFFFFFFFFu, top = 0;
    if (m < M) {
        const uint32_t in = info[m], np = in & 0xFF, col = in >> 8;
        const uint32_t a = bA[m], b = bB[m], fa = flags[a], fb = flags[b];
        if (np > 0 && !edge_asleep(fa, fb)) {
            const uint32_t l = manifold_label(a, b, fa, island);
            if (col != kNoColour) { label = l; top = col + 1; }
            else isl_top[l].y = 1u;   // the island has something to colour in this step (identical plain stores)
        }
    }
    // Neighbours in the canonical order mostly share their island. A wave of one island hands its maximum to the workgroup, which
    // issues one atomic per run of equal labels (a big island: one per 1024 manifolds instead of thousands on one address); a wave
    // that spans islands issues one atomic per island it touches. A plain look comes first: once an island's top colour has landed
    // the others have nothing to add.
    auto post = [&](uint32_t l, uint32_t t) { if (__atomic_load_n(&isl_top[l].x, __ATOMIC_RELAXED) < t) atomicMax(&isl_top[l].x, t); };
    uint64_t todo = __ballot(top != 0);
    uint32_t w_label = 0xFFFFFFFFu, w_top = 0;
    if (todo) {
        const uint32_t l0 = __shfl(label, __ffsll((long long)todo) - 1);
        if (__ballot(top != 0 && label == l0) == todo) {   // one island
            uint32_t t = top;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t = max(t, (uint32_t)__shfl_xor(t, off));
            w_label = l0; w_top = t;
        } else {
            while (todo) {
                const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1, l = __shfl(label, leader);
                const bool mine = top != 0 && label == l;
                uint32_t t = mine ? top : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t = max(t, (uint32_t)__shfl_xor(t, off));
                if (lane == leader) post(l, t);
                todo &= ~__ballot(mine);
            }
        }
    }
    if (lane == 0) { s_label[wave] = w_label; s_top[wave] = w_top; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t l = 0xFFFFFFFFu, t = 0;
        for (int w = 0; w < 16; ++w) {
            if (s_top[w] == 0) continue;
            if (s_label[w] != l) { if (t) post(l, t); l = s_label[w]; t = 0; }
            t = max(t, s_top[w]);
        }
        if (t) post(l, t);
    }
}
__global__ void __launch_bounds__(1024) k_col_prepare(uint32_t M, uint32_t *__restrict__ info, const uint32_t *__restrict__ bA,
                              const uint32_t *__restrict__ bB, const uint32_t *__restrict__ flags, uint64_t *used,
                              uint64_t *best0, uint64_t *best1, Counters *cnt, const uint32_t *__restrict__ island, const uint2 *__restrict__ isl_top,
                              uint32_t *unc_list) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t unc = 0;
    if (m < M) {
        uint32_t in = info[m];
        uint32_t np = in & 0xFF, col = in >> 8;
        const uint32_t a = bA[m], b = bB[m];
        const uint32_t fa = flags[a], fb = flags[b];
        // a sleeping manifold is out of the solve but keeps (and blocks) its colour for when its island wakes
        const bool asleep = edge_asleep(fa, fb);
        if (np > 0 && col != kNoColour && !asleep) {
            const uint2 top = isl_top[manifold_label(a, b, fa, island)];
            if (top.y && top.x >= 2 && col + 1 == top.x) { col = kNoColour; info[m] = np | (kNoColour << 8); }
        }
        if (np > 0) {
            bool da = is_dynamic(fa), db = is_dynamic(fb);
            if (col != kNoColour) {
                if (da) atomicOr((unsigned long long *)&used[a], 1ull << col);
                if (db) atomicOr((unsigned long long *)&used[b], 1ull << col);
            } else if (!asleep) {
                unc = 1;
                if (da) { best0[a] = 0; best1[a] = 0; }
                if (db) { best0[b] = 0; best1[b] = 0; }
            }
        }
    }
    // list the uncoloured edges for k_col_rounds (beyond its capacity only the count matters): one pair of atomics per WORKGROUP - a
    // restless scene has an uncoloured edge in almost every wave, and thousands of atomics on one cache line cost more than the kernel
    __shared__ uint32_t wcount[16], wbase;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t mask = __ballot(unc != 0);
    if (lane == 0) wcount[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) total += wcount[w];
        wbase = total ? atomicAdd(&cnt->unc_count, total) : 0u;
        if (total) atomicAdd(&cnt->uncoloured, total);
    }
    __syncthreads();
    if (unc) {
        uint32_t at = wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) at += wcount[w];
        if (at < kColUncCap) unc_list[at] = m;
    }
}
// The rounds for a list of uncoloured edges that one workgroup can hold - the steady state: some hundred (a restless heap: some
// thousand) new contacts per step - run in one workgroup with workgroup barriers between the phases instead of two launches per
// round; same rule, same result as k_col_best / k_col_assign. The edges' endpoints sit in LDS (index | dynamic << 31), the list is
// compacted as edges take their colour (a round only visits what is still uncoloured). The endpoint marks carry the round in their upper
// half - (round + 1) << 32 | priority - so a mark of an earlier round loses against any mark of this one and nothing has to be zeroed
// between rounds (round 5: the workgroup is bound by the memory requests one CU can issue - a restless heap of polyhedra lists 15 000
// edges per step and still has two thirds of them after three rounds; the zeroing stores were two of six requests per edge and round).
// Longer lists, and what is left after max_rounds, go to the multi-block rounds (the host sees cnt->uncoloured != 0).
__global__ void __launch_bounds__(1024) k_col_rounds(uint32_t *info, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                                                     const uint32_t *__restrict__ flags, uint64_t *best0, uint64_t *best1, uint64_t *used, Counters *cnt,
                                                     uint32_t *list, uint32_t cap, uint32_t max_rounds) {
    extern __shared__ uint32_t col_lds[];   // ea[cap], eb[cap]
    uint32_t *ea = col_lds, *eb = col_lds + cap;
    __shared__ uint32_t live, wr;
    const uint32_t n0 = cnt->unc_count;
    if (n0 == 0 || n0 > cap) { if (threadIdx.x == 0) cnt->col_wg_rounds = 0; return; }
    for (uint32_t e = threadIdx.x; e < n0; e += blockDim.x) {
        const uint32_t m = list[e], a = bA[m], b = bB[m];
        ea[e] = a | (is_dynamic(flags[a]) ? 0x80000000u : 0u);
        eb[e] = b | (is_dynamic(flags[b]) ? 0x80000000u : 0u);
    }
    if (threadIdx.x == 0) live = n0;
    __syncthreads();
    auto ld = [](const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };   // other waves' stores: not through a stale L1 line
    auto ld32 = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    uint32_t round = 0;
    for (; round < max_rounds; ++round) {
        const uint32_t n = live;
        if (n == 0) break;
        uint64_t *cur = best0;
        const uint64_t stamp = (uint64_t)(round + 1u) << 32;   // (edge_prio fits the lower half)
        for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {           // phase 1: every endpoint learns its best uncoloured edge
            const uint64_t pr = stamp | edge_prio(list[e]);
            const uint32_t a = ea[e], b = eb[e];
            if (a >> 31) atomicMax((unsigned long long *)&cur[a & 0x7FFFFFFFu], pr);
            if (b >> 31) atomicMax((unsigned long long *)&cur[b & 0x7FFFFFFFu], pr);
        }
        if (threadIdx.x == 0) wr = 0;
        __threadfence_block(); __syncthreads();   // one workgroup, one CU: its stores only have to reach L2 before the other waves' (atomic) loads
        for (uint32_t base = 0; base < n; base += blockDim.x) {           // phase 2: edges that are best at both ends take a colour, the others stay listed
            const uint32_t e = base + threadIdx.x;
            bool keep = false;
            uint32_t m = 0, a = 0, b = 0;
            if (e < n) {
                m = list[e]; a = ea[e]; b = eb[e];
                const bool da = a >> 31, db = b >> 31;
                const uint32_t ia = a & 0x7FFFFFFFu, ib = b & 0x7FFFFFFFu;
                const uint64_t pr = stamp | edge_prio(m);
                if ((da && ld(&cur[ia]) != pr) || (db && ld(&cur[ib]) != pr)) keep = true;
                else {
                    const uint64_t busy = (da ? ld(&used[ia]) : 0ull) | (db ? ld(&used[ib]) : 0ull);
                    const uint64_t avail = ~busy & ((1ull << kSerialColour) - 1ull);
                    const uint32_t c = avail ? (uint32_t)__ffsll((long long)avail) - 1 : kSerialColour;   // nothing free: the serial bucket
                    info[m] = (ld32(&info[m]) & 0xFF) | (c << 8);
                    if (da) atomicOr((unsigned long long *)&used[ia], 1ull << c);
                    if (db) atomicOr((unsigned long long *)&used[ib], 1ull << c);
                }
            }
            __syncthreads();   // this block of the list has been read: its survivors may now be written over the front of it
            const uint64_t mask = __ballot(keep);
            if (mask) {
                const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)mask) - 1;
                uint32_t at = 0;
                if (lane == leader) at = atomicAdd(&wr, (uint32_t)__popcll(mask));
                at = __shfl(at, leader) + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (keep) { list[at] = m; ea[at] = a; eb[at] = b; }
            }
        }
        __threadfence_block(); __syncthreads();
        if (threadIdx.x == 0) live = wr;
        __syncthreads();
    }
    const uint32_t left = live;
    if (left) {   // handed to the multi-block rounds, which start from clean marks in both arrays
        for (uint32_t e = threadIdx.x; e < left; e += blockDim.x) {
            const uint32_t a = ea[e], b = eb[e];
            if (a >> 31) { best0[a & 0x7FFFFFFFu] = 0; best1[a & 0x7FFFFFFFu] = 0; }
            if (b >> 31) { best0[b & 0x7FFFFFFFu] = 0; best1[b & 0x7FFFFFFFu] = 0; }
        }
    }
    if (threadIdx.x == 0) { cnt->uncoloured = left; cnt->col_wg_rounds = round; }
}
// Round 6, an experiment kept behind EDYNHIP_COL_LDS=1 (measured: 3 % faster on the polyhedron heap, 1-2 us SLOWER per step on every pile - the
// 159 KB of LDS it asks for and its clearing - so k_col_rounds above stays the default; the two are compared by
// test_colouring_rounds_in_lds_and_in_global_memory_colour_alike): the same rounds with the endpoint marks in LDS. The kernel above spends its time on the memory requests one CU can issue - eight
// scattered 8-byte requests per listed edge and round (two marks set, two read, the list read twice and rewritten), 15 000 edges and ~32
// rounds on a restless heap. Here the list, the edges' endpoints and the marks live in LDS; global memory is touched once per edge, when it
// takes its colour. The marks are a HASHED table (slot = hash(body), 4 bytes: the priority): two bodies may share a slot, and then an edge is
// "best" only if it beats the uncoloured edges of every body in its two slots - a stricter test than the rule's, so an edge may take its
// colour a round later, never earlier: it still takes it after all its higher-priority neighbours and before all lower ones, and the result
// is the greedy colouring in priority order whatever the rounds were (winners of one round are never adjacent). Every thread owns the edges
// t, t + 1024, ... and keeps its survivors packed at the front of that column: no list compaction across threads, three barriers per round.
__global__ void __launch_bounds__(1024) k_col_rounds_lds(uint32_t *info, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                                                     const uint32_t *__restrict__ flags, uint64_t *used, Counters *cnt,
                                                     const uint32_t *__restrict__ list, uint32_t cap, uint32_t lds_words, uint32_t max_rounds) {
    extern __shared__ uint32_t col_lds[];   // em[n0] manifold, es[n0] endpoint slots (slot A | dynamic A << 15 | slot B << 16 | dynamic B << 31), mark[T]
    __shared__ uint32_t alive[2], left_sum;
    const uint32_t n0 = cnt->unc_count;
    if (n0 == 0 || n0 > cap) { if (threadIdx.x == 0) cnt->col_wg_rounds = 0; return; }
    uint32_t *em = col_lds, *es = col_lds + n0, *mark = col_lds + 2 * n0;
    uint32_t T = 1024, bits = 10;   // table size: what the workgroup's LDS leaves, at most 2^15 slots, no more than ~16 per edge (it is cleared once)
    while (T < 32768u && 2 * T <= lds_words - 2 * n0 && T < 16u * n0) { T *= 2; ++bits; }
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < T; i += blockDim.x) mark[i] = 0;
    uint32_t mine = 0;   // live edges of this thread: em[tid + k * 1024], k < mine (its share of the list, packed at the front of its column)
    for (uint32_t e = tid; e < n0; e += blockDim.x) {
        const uint32_t m = list[e], a = bA[m], b = bB[m];
        const uint32_t sa = (a * 0x9E3779B1u) >> (32u - bits), sb = (b * 0x9E3779B1u) >> (32u - bits);
        const uint32_t to = tid + mine++ * blockDim.x;
        em[to] = m;
        es[to] = sa | (is_dynamic(flags[a]) ? 0x8000u : 0u) | (sb << 16) | (is_dynamic(flags[b]) ? 0x80000000u : 0u);
    }
    if (tid < 2) alive[tid] = 0;
    if (tid == 0) left_sum = 0;
    __syncthreads();
    auto ld = [](const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };   // other waves' atomics: not through a stale L1 line
    auto ld32 = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    uint32_t round = 0;
    for (; round < max_rounds; ++round) {
        for (uint32_t k = 0; k < mine; ++k) {                                // phase 1: every slot learns the best uncoloured edge of its bodies
            const uint32_t e = tid + k * blockDim.x, s = es[e], pr = (uint32_t)edge_prio(em[e]);
            if (s & 0x8000u) atomicMax(&mark[s & 0x7FFFu], pr);
            if (s >> 31) atomicMax(&mark[(s >> 16) & 0x7FFFu], pr);
        }
        __syncthreads();
        if (tid == 0) alive[(round + 1) & 1] = 0;
        // phase 2: an edge that is best in both its slots takes its colour. The winners of a thread are taken four at a time with their loads
        // issued together - bodies and point count, then the bodies' colour masks: two dependent round trips per batch, where one winner after
        // the other paid them per edge (a thread owns up to 16 listed edges; winners of a round are never adjacent, so their masks are independent)
        for (uint32_t k = 0;;) {
            uint32_t nw = 0, we[4];
            for (; k < mine && nw < 4u; ++k) {
                const uint32_t e = tid + k * blockDim.x, s = es[e], pr = (uint32_t)edge_prio(em[e]);
                if (((s & 0x8000u) && mark[s & 0x7FFFu] != pr) || ((s >> 31) && mark[(s >> 16) & 0x7FFFu] != pr)) continue;
                we[nw++] = e;
            }
            if (nw == 0) break;
            uint32_t wm[4], ws[4], ia[4], ib[4], in[4];
            uint64_t busy[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (j < nw) { wm[j] = em[we[j]]; ws[j] = es[we[j]]; ia[j] = bA[wm[j]]; ib[j] = bB[wm[j]]; in[j] = ld32(&info[wm[j]]); }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (j < nw) busy[j] = ((ws[j] & 0x8000u) ? ld(&used[ia[j]]) : 0ull) | ((ws[j] >> 31) ? ld(&used[ib[j]]) : 0ull);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (j < nw) {
                    const uint64_t avail = ~busy[j] & ((1ull << kSerialColour) - 1ull);
                    const uint32_t c = avail ? (uint32_t)__ffsll((long long)avail) - 1 : kSerialColour;   // nothing free: the serial bucket
                    info[wm[j]] = (in[j] & 0xFF) | (c << 8);
                    // its marks go at once (nobody else can equal them: the others in these slots read either this priority or zero and stay)
                    if (ws[j] & 0x8000u) { atomicOr((unsigned long long *)&used[ia[j]], 1ull << c); mark[ws[j] & 0x7FFFu] = 0; }
                    if (ws[j] >> 31) { atomicOr((unsigned long long *)&used[ib[j]], 1ull << c); mark[(ws[j] >> 16) & 0x7FFFu] = 0; }
                    em[we[j]] = 0xFFFFFFFFu;
                }
        }
        __threadfence_block(); __syncthreads();   // the marks have been read; the colours' atomics are at the L2 before the next round reads `used`
        uint32_t kept = 0;
        for (uint32_t k = 0; k < mine; ++k) {                                // the survivors clear their slots and move to the front of the column
            const uint32_t e = tid + k * blockDim.x, m = em[e];
            if (m == 0xFFFFFFFFu) continue;
            const uint32_t s = es[e];
            if (s & 0x8000u) mark[s & 0x7FFFu] = 0;
            if (s >> 31) mark[(s >> 16) & 0x7FFFu] = 0;
            const uint32_t to = tid + kept * blockDim.x;
            if (to != e) { em[to] = m; es[to] = s; }
            ++kept;
        }
        mine = kept;
        if (__ballot(mine != 0) && (tid & 63u) == 0) alive[round & 1] = 1;
        __syncthreads();
        if (!alive[round & 1]) { ++round; break; }
    }
    // what max_rounds left over goes to the multi-block rounds (their marks were cleared by k_col_prepare and never touched here)
    uint32_t left = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) left += __shfl_xor(left, off);
    if ((tid & 63u) == 0 && left) atomicAdd(&left_sum, left);
    __syncthreads();
    if (tid == 0) { cnt->uncoloured = left_sum; cnt->col_wg_rounds = round; }
}
__global__ void k_col_best(uint32_t M, const uint32_t *__restrict__ info, const uint32_t *__restrict__ bA,
                           const uint32_t *__restrict__ bB, const uint32_t *__restrict__ flags, uint64_t *best_cur,
                           uint64_t *best_next, const Counters *cnt) {
    if (cnt->uncoloured == 0) return;
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    uint32_t in = info[m];
    if ((in & 0xFF) == 0 || (in >> 8) != kNoColour) return;
    uint32_t a = bA[m], b = bB[m];
    if (edge_asleep(flags[a], flags[b])) return;
    uint64_t pr = edge_prio(m);
    if (is_dynamic(flags[a])) { atomicMax((unsigned long long *)&best_cur[a], pr); best_next[a] = 0; }
    if (is_dynamic(flags[b])) { atomicMax((unsigned long long *)&best_cur[b], pr); best_next[b] = 0; }
}
__global__ void k_col_assign(uint32_t M, uint32_t *info, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                             const uint32_t *__restrict__ flags, const uint64_t *__restrict__ best_cur, uint64_t *used,
                             Counters *cnt) {
    if (cnt->uncoloured == 0) return;
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t done = 0;
    if (m < M) {
        uint32_t in = info[m];
        if ((in & 0xFF) != 0 && (in >> 8) == kNoColour) {
            uint32_t a = bA[m], b = bB[m];
            bool da = is_dynamic(flags[a]), db = is_dynamic(flags[b]);
            uint64_t pr = edge_prio(m);
            if (!edge_asleep(flags[a], flags[b]) && !(da && best_cur[a] != pr) && !(db && best_cur[b] != pr)) {
                uint64_t busy = (da ? used[a] : 0ull) | (db ? used[b] : 0ull);
                const uint64_t avail = ~busy & ((1ull << kSerialColour) - 1ull);
                const uint32_t c = avail ? (uint32_t)__ffsll((long long)avail) - 1 : kSerialColour;   // nothing free: the serial bucket
                info[m] = (in & 0xFF) | (c << 8);
                if (da) used[a] |= 1ull << c;
                if (db) used[b] |= 1ull << c;
                done = 1;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicSub(&cnt->uncoloured, done);
}
DI uint32_t colour_key(uint32_t m, const uint32_t *__restrict__ info, const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                       const uint32_t *__restrict__ flags, bool sleeping) {
    // within a colour, manifolds are grouped by point count (4 first): the solve kernels then know a lane's point
    // count from its position alone (no dependent load) and waves are uniform in it
    uint32_t in = info[m];
    uint32_t np = in & 0xFF;
    if (sleeping && np && edge_asleep(flags[bA[m]], flags[bB[m]])) np = 0;   // not part of this step's solve
    return np ? (((in >> 8) << 2) | (4u - np)) : 0xFFu;
}
// Stable counting sort of the manifolds by (colour, point count) key - at most 256 distinct keys. (rocPRIM's radix sort
// falls back to a merge sort for an 8-bit key range: 1 block-sort + 16 merge launches, ~90 us per step.)
//  k_cs_hist:    per 1024-element block, the count of every key          -> hist[key * nblocks + block]
//  scan_u32:     exclusive scan of that key-major table                   -> where each (key, block) run starts
//  k_cs_scatter: every element's slot = its run's start + its rank among the block's earlier elements with its key
//  Scenes of up to kCsDirectBlocks blocks (the headline pile: 174) take two launches instead of five (round 5): the
//  histogram table is block-major there and k_cs_hist<true> also accumulates the key counts of every kCsSuper blocks (`sup`, integer
//  sums: the same values whatever the order); k_cs_scatter<true> gets the run starts of its block from at most 16 + 15 table rows,
//  builds the key starts from the totals, and block 0 publishes the colour ranges (k_col_offsets' work). No library scan on the hot path.
template <bool DIRECT>
__global__ void __launch_bounds__(kCsBlock) k_cs_hist(uint32_t M, uint32_t *keys, uint32_t *hist, uint32_t nblocks, const uint32_t *__restrict__ info,
                                                      const uint32_t *__restrict__ bA, const uint32_t *__restrict__ bB,
                                                      const uint32_t *__restrict__ flags, bool sleeping, uint32_t *sup) {
    __shared__ uint32_t h[kCsKeys];
    if (threadIdx.x < kCsKeys) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t m = blockIdx.x * kCsBlock + threadIdx.x;
    if (m < M) {
        const uint32_t key = colour_key(m, info, bA, bB, flags, sleeping);
        keys[m] = key;
        atomicAdd(&h[key & 0xFFu], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kCsKeys) {
        hist[DIRECT ? blockIdx.x * kCsKeys + threadIdx.x : threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
        if (DIRECT && h[threadIdx.x]) atomicAdd(&sup[(blockIdx.x / kCsSuper) * kCsKeys + threadIdx.x], h[threadIdx.x]);
    }
}
template <bool DIRECT>
__global__ void __launch_bounds__(kCsBlock) k_cs_scatter(uint32_t M, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ start,
                                                         uint32_t nblocks, uint32_t *keys_sorted, uint32_t *order, Counters *cnt, const uint32_t *__restrict__ sup) {
    __shared__ uint32_t base[kCsKeys];
    if (DIRECT) {   // `start` is the block-major histogram table
        __shared__ uint32_t wsum[4], part[kCsKeys];
        uint32_t before = 0, total = 0;
        {   // lanes 0..255: the super-block rows (key totals; what lies before this block's super-block); lanes 256..511: the rows of
            // the blocks before this one inside its super-block - at most 16 + 15 independent loads per key
            const uint32_t k = threadIdx.x & (kCsKeys - 1), q = threadIdx.x >> 8, sb = blockIdx.x / kCsSuper, nsup = (nblocks + kCsSuper - 1) / kCsSuper;
            if (q == 0) {
                uint32_t v[kCsDirectBlocks / kCsSuper];
#pragma unroll
                for (uint32_t u = 0; u < kCsDirectBlocks / kCsSuper; ++u) v[u] = u < nsup ? sup[u * kCsKeys + k] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < kCsDirectBlocks / kCsSuper; ++u) { total += v[u]; before += u < sb ? v[u] : 0u; }
            } else if (q == 1) {
                uint32_t v[kCsSuper], inner = 0;
#pragma unroll
                for (uint32_t u = 0; u < kCsSuper; ++u) { const uint32_t b = sb * kCsSuper + u; v[u] = b < blockIdx.x ? start[b * kCsKeys + k] : 0u; }
#pragma unroll
                for (uint32_t u = 0; u < kCsSuper; ++u) inner += v[u];
                part[k] = inner;
            }
        }
        __syncthreads();
        if (threadIdx.x < kCsKeys) before += part[threadIdx.x];
        // exclusive scan of the 256 key totals (lanes 0..255 = waves 0..3)
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        uint32_t inc = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
        if (wave < 4 && lane == 63) wsum[wave] = inc;
        __syncthreads();
        if (threadIdx.x < kCsKeys) {
            uint32_t key_start = inc - total;
            for (uint32_t w = 0; w < wave; ++w) key_start += wsum[w];
            base[threadIdx.x] = key_start + before;
            if (blockIdx.x == 0 && threadIdx.x < 4 * kMaxContactColours && total) {   // what k_col_offsets reads off the sorted keys
                cnt->colour_start[threadIdx.x] = key_start; cnt->colour_end[threadIdx.x] = key_start + total;
            }
        }
    } else if (threadIdx.x < kCsKeys) base[threadIdx.x] = start[threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    const uint32_t m = blockIdx.x * kCsBlock + threadIdx.x;
    const bool valid = m < M;
    const uint32_t key = valid ? (keys[m] & 0xFFu) : 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // rank among the lower lanes of this wave with the same key, and the wave's count of that key (at its first lane)
    uint32_t rank = 0, count = 0;
    bool first = false;
    uint64_t todo = __ballot(valid);
    while (todo) {
        const uint32_t k = __shfl(key, __ffsll((long long)todo) - 1);
        const uint64_t same = __ballot(valid && key == k);
        if (valid && key == k) {
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            count = (uint32_t)__popcll(same);
            first = rank == 0;
        }
        todo &= ~same;
    }
    for (uint32_t w = 0; w < kCsBlock / 64; ++w) {   // waves take their turns in order: keeps the sort stable
        if (wave == w && valid) {
            const uint32_t pos = base[key] + rank;
            keys_sorted[pos] = key; order[pos] = m;
        }
        __syncthreads();
        if (wave == w && valid && first) base[key] += count;
        __syncthreads();
    }
}
__global__ void k_col_offsets(uint32_t M, const uint32_t *__restrict__ keys_sorted, Counters *cnt) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    uint32_t c = keys_sorted[p];
    if (c >= 4 * kMaxContactColours) return;
    if (p == 0 || keys_sorted[p - 1] != c) cnt->colour_start[c] = p;
    if (p == M - 1 || keys_sorted[p + 1] != c) cnt->colour_end[c] = p + 1;
}

// ------------------------------------------------------------------ row math shared by prep / solve
struct BRef { f3 pos, org; q4 orn; f3 v, w; float inv_m; m3 inv_I; };   // org: constraint_body::origin - the frame of every pivot (= pos without a centre-of-mass offset)
DI BRef load_bref(const Bodies &b, uint32_t i) {
    BRef r;
    float4 p = B_POS(b, i);
    uint32_t fl = b.flags[i];
    r.pos = from4(p); r.org = B_ORG(b, i); r.orn = q_from4(B_ORN(b, i));
    if (is_dynamic(fl)) {
        r.inv_m = p.w;
        r.inv_I = {from4(B_IW(b, i, 0)), from4(B_IW(b, i, 1)), from4(B_IW(b, i, 2))};
    } else { r.inv_m = 0; r.inv_I = m3_zero(); }
    if ((fl & BF_KIND_MASK) == EDYNHIP_KIND_STATIC) { r.v = mk3(0, 0, 0); r.w = mk3(0, 0, 0); }
    else { r.v = from4(b.linvel[i]); r.w = from4(b.angvel[i]); }
    return r;
}
DI float eff_mass(f3 J0, f3 J1, f3 J2, f3 J3, float imA, const m3 &iA, float imB, const m3 &iB) {
    float s = dot(J0, J0) * imA + dot(mul(iA, J1), J1) + dot(J2, J2) * imB + dot(mul(iB, J3), J3);
    return 1.0f / s;
}
DI float rel_speed(f3 J0, f3 J1, f3 J2, f3 J3, f3 vA, f3 wA, f3 vB, f3 wB) {
    return dot(J0, vA) + dot(J1, wA) + dot(J2, vB) + dot(J3, wB);
}

__global__ void k_solve_begin(uint32_t n, Bodies b, float dt, uint32_t *first_slot) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) solve_begin_body(i, b, dt, first_slot);
}

DI void store_row(float4 *rw, size_t base, size_t cap, f3 Jl, f3 JaA, f3 JaB, float eff, float rhs, float imp, float mu,
                  const BRef &A, const BRef &B) {
    rw[base] = to4(Jl, eff);
    rw[base + cap] = to4(JaA, rhs);
    rw[base + 2 * cap] = to4(JaB, imp);
    rw[base + 3 * cap] = to4(mul(A.inv_I, JaA), mu);
    rw[base + 4 * cap] = to4(mul(B.inv_I, JaB), kLarge);   // .w of a normal row: its upper limit (soft contacts lower it)
}
// EXTRAS: the world has contact_extras materials (soft normal rows, rolling / spinning rows); the plain instantiation is the
// headline path and carries none of that code.
template <bool EXTRAS>
__global__ void k_prep_contacts(uint32_t n_active, Rows rows, uint32_t rcap, Manifolds mf, Bodies b, float dt,
                                const uint32_t *__restrict__ keys_sorted, bool push, bool by_key, bool write_pw) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_active) return;
    // speculative launch (solve(): enqueued before the host has read the colouring's counters): n_active is the number of ALL
    // manifolds; the active ones are the prefix of the sorted order whose keys name a colour
    if (by_key && keys_sorted[p] >= 4u * kMaxContactColours) return;
    const uint32_t m = rows.order[p];
    const uint32_t np = mf.info[m] & 0xFF;
    const uint32_t ia = mf.bodyA[m], ib = mf.bodyB[m];
    rows.bA[p] = ia; rows.bB[p] = ib; rows.np[p] = np;
    rows.label[p] = b.island[is_dynamic(b.flags[ia]) ? ia : ib];
    if (push) {   // hand-off slot of (body, colour): k_push_links turns these into each body's chain
        const uint32_t col = keys_sorted[p] >> 2;
        if (is_dynamic(b.flags[ia])) rows.slot_of[(size_t)ia * kMaxColours + col] = 2 * p;
        if (is_dynamic(b.flags[ib])) rows.slot_of[(size_t)ib * kMaxColours + col] = 2 * p + 1;
    }
    // (One lane per contact POINT instead of per manifold was measured - the points of a manifold are independent here - and is slower,
    //  103 against 71 us on the headline pile: every lane then loads both bodies for a single point.)
    const BRef A = load_bref(b, ia), B = load_bref(b, ib);
    for (uint32_t k = 0; k < np; ++k) {
        const size_t s = slot_at(mf.cap, k, m), t = pt_at(mf.cap, k, m);   // (s: the slot-major extras arrays; t: the manifold's point record)
        const float4 a4 = mf.pA[t], b4 = mf.pB[t], n4 = mf.nrm[t], im = mf.imp[t];
        const f3 n = from4(n4);
        const float distance = a4.w, mu = b4.w;
        const f3 pAw = to_world(from4(a4), A.org, A.orn), pBw = to_world(from4(b4), B.org, B.orn);
        const f3 rA = pAw - A.pos, rB = pBw - B.pos;
        // normal row
        const f3 J0 = n, J1 = cross(rA, n), J2 = -n, J3 = -cross(rB, n);
        const float effn = eff_mass(J0, J1, J2, J3, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
        const float relvel = rel_speed(J0, J1, J2, J3, A.v, A.w, B.v, B.w);
        float error = distance > 0 ? distance / dt : 0.0f;
        float upper = kLarge;
        float4 xm = make_float4(0, 0, kLarge, kLarge), xi = make_float4(0, 0, 0, 0);
        if (EXTRAS) { xm = mf.xmat[s]; xi = mf.ximp[s]; }
        if (EXTRAS && distance < 0 && xm.z < kLarge) {   // soft contact (contact_extras_constraint.cpp:16-35): force-limited normal row
            const f3 vA = A.v + cross(A.w, rA), vB = B.v + cross(B.w, rB);
            const float normal_relvel = dot(vA - vB, n);
            const float spring_force = -distance * xm.z / (float)np;
            const float damper_force = -normal_relvel * xm.w / (float)np;
            upper = fmaxf(spring_force + damper_force, 0.0f) * dt;
            error = -kLarge;
        }
        const float rhsn = -(error * 0.2f + relvel * (1 + 0.0f));   // erp 0.2, zero restitution (restitution solver path)
        // friction rows
        f3 t0, t1;
        plane_space(n, t0, t1);
        const f3 K1 = cross(rA, t0), K3 = -cross(rB, t0);
        const f3 L1 = cross(rA, t1), L3 = -cross(rB, t1);
        const float eff0 = eff_mass(t0, K1, -t0, K3, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
        const float eff1 = eff_mass(t1, L1, -t1, L3, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
        const float rhs0 = -rel_speed(t0, K1, -t0, K3, A.v, A.w, B.v, B.w);
        const float rhs1 = -rel_speed(t1, L1, -t1, L3, A.v, A.w, B.v, B.w);
        const size_t base = (size_t)(k * kRowsPerPoint) * kRowF * rcap + p, rstride = (size_t)kRowF * rcap;
        store_row(rows.rw, base, rcap, n, J1, J3, effn, rhsn, im.x, mu, A, B);
        if (!EXTRAS && push && write_pw) {   // the dataflow position solve's copy of the point, indexed by the lane (Rows::pw)
            const size_t pb = (size_t)(k * kPosF) * rcap + p;
            rows.pw[pb] = a4; rows.pw[pb + rcap] = b4; rows.pw[pb + 2 * (size_t)rcap] = mf.lnrm[t]; rows.pw[pb + 3 * (size_t)rcap] = n4;
        }
        if (EXTRAS && upper != kLarge) rows.rw[base + 4 * (size_t)rcap].w = upper;
        if (EXTRAS) {   // rolling pair and spinning row (:37-78); roll_direction components do not exist on this path
            const size_t xb = (size_t)(k * kXPoint) * rcap + p;
            rows.rwx[xb + 9 * (size_t)rcap] = make_float4(xm.x, xm.y, 0, 0);
            auto axial = [&](int r, f3 ax, float imp, bool guarded) {
                const f3 ia = mul(A.inv_I, ax), ib = mul(B.inv_I, -ax);
                const float ssum = dot(ia, ax) + dot(ib, -ax);
                const float eff = guarded ? (ssum > kEps ? 1.0f / ssum : 0.0f) : 1.0f / ssum;
                const float rhs = guarded ? -rel_speed(mk3(0, 0, 0), ax, mk3(0, 0, 0), -ax, A.v, A.w, B.v, B.w) : -(dot(ax, A.w) + dot(-ax, B.w));
                rows.rwx[xb + (size_t)(3 * r) * rcap] = to4(ax, eff);
                rows.rwx[xb + (size_t)(3 * r + 1) * rcap] = to4(ia, rhs);
                rows.rwx[xb + (size_t)(3 * r + 2) * rcap] = to4(ib, imp);
            };
            if (xm.x > 0) {
                // a dynamic capsule rolls about its axis (roll_direction, shapes.hpp:136-139): the tangent axes are scaled by
                // the projection of that direction - both bodies' directions rotated by A's orientation, as the reference does
                // (contact_extras_constraint.cpp:44-55)
                f3 ax0 = t0, ax1 = t1;
                const uint32_t fa = b.flags[ia], fb = b.flags[ib];
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const uint32_t fl = side ? fb : fa;
                    if (is_dynamic(fl) && (((fl & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT) == (uint32_t)dc::SHAPE_CAPSULE || ((fl & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT) == (uint32_t)dc::SHAPE_CYLINDER)) {   // shape_rolling_direction, shapes.hpp:127-139
                        const f3 rdw = rotate(A.orn, dc::axis_vector(b.shape[side ? ib : ia].z));
                        ax0 *= dot(rdw, ax0); ax1 *= dot(rdw, ax1);
                    }
                }
                axial(0, ax0, xi.x, true); axial(1, ax1, xi.y, true);
            }
            if (xm.y > 0) axial(2, n, xi.z, false);
        }
        store_row(rows.rw, base + rstride, rcap, t0, K1, K3, eff0, rhs0, im.y, 0.0f, A, B);
        store_row(rows.rw, base + 2 * rstride, rcap, t1, L1, L3, eff1, rhs1, im.z, 0.0f, A, B);
    }
}

// ------------------------------------------------------------------ velocity solve: contacts
struct Delta { f3 dvA, dwA, dvB, dwB; float imA, imB; m3 iA, iB; };
DI void load_delta(const Bodies &b, uint32_t ia, uint32_t ib, Delta &d) {
    float4 va = B_DV(b, ia), vb = B_DV(b, ib);
    d.dvA = from4(va); d.imA = va.w; d.dwA = from4(B_DW(b, ia));
    d.dvB = from4(vb); d.imB = vb.w; d.dwB = from4(B_DW(b, ib));
    d.iA = {from4(B_IW(b, ia, 0)), from4(B_IW(b, ia, 1)), from4(B_IW(b, ia, 2))};
    d.iB = {from4(B_IW(b, ib, 0)), from4(B_IW(b, ib, 1)), from4(B_IW(b, ib, 2))};
}
DI void store_delta(const Bodies &b, uint32_t ia, uint32_t ib, const Delta &d) {
    if (d.imA != 0) { B_DV(b, ia) = to4(d.dvA, d.imA); B_DW(b, ia) = to4(d.dwA, 0); }   // non-procedural bodies keep zero deltas
    if (d.imB != 0) { B_DV(b, ib) = to4(d.dvB, d.imB); B_DW(b, ib) = to4(d.dwB, 0); }
}
DI void apply_impulse(Delta &d, f3 J0, f3 J1, f3 J2, f3 J3, float imp) {   // apply_row_impulse
    d.dvA += d.imA * J0 * imp;
    d.dvB += d.imB * J2 * imp;
    d.dwA += mul(d.iA, J1) * imp;
    d.dwB += mul(d.iB, J3) * imp;
}

// One launch per colour: each lane owns one manifold and sweeps its normal rows, then its friction rows
// (solve(constraint_row&) + apply_row_impulse, then solve_friction, per point in list order).
// Loads are all issued up front: indices -> {body deltas, every row of every point} -> arithmetic -> stores.
struct RowReg { float4 f[kRowF]; };
// ---- the arithmetic of a contact's normal and friction rows: two forms, selected per context (ctx.hpp Arith) ------------------------
// FUSED = false (the default): the reference's operations in the reference's order - solve(constraint_row&) + apply_row_impulse
// (constraint_row.cpp:24-57), solve_friction (constraint_row_friction.cpp:11-54): the coloured order then differs from the reference
// in the Gauss-Seidel visiting order ONLY (checker: ORDER_COLOURED with ARITH_REFERENCE, bit for bit).
// FUSED = true (EDYNHIP_FLAG_FUSED_VELOCITY_ROWS, opt-in): the velocity solve is a dependency chain - what bounds a step is the number
// of instructions between "a body's deltas arrived" and "its deltas are handed on"  - so this form writes the same
// row equations with fused multiply-adds and fewer operations, 385 instead of 619 instructions for a four-point manifold in the
// two-lane kernel; an fp-level deviation (1e-7 m/s per step, tests/test_arithmetic_fork.py), specified by the checker's
// ARITH_FUSED_VELOCITY and computed bit for bit by every velocity-solve kernel of this file:
//   dot(a, b)      = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x))
//   relative speed = (lin_A + ang_A) + (lin_B + ang_B)
//   delta          = fma(-relative speed, eff, rhs * eff)
//   normal row     : new = min(max(impulse + delta, 0), upper), applied = new - impulse
//   friction pair  : both impulses times max_len / len (ONE correctly rounded division) when outside the circle, applied = new - old
//   apply          : delta_v = fma(M^-1 J, applied, delta_v) per component
// Joint rows, contact_extras rows and the restitution solver keep the reference's operation order.
DI float dot3_fma(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
DI f3 fma3(f3 w, float s, f3 acc) { return mk3(__builtin_fmaf(w.x, s, acc.x), __builtin_fmaf(w.y, s, acc.y), __builtin_fmaf(w.z, s, acc.z)); }
DI float fused_delta(float rel, float eff, float rhs_eff) { return __builtin_fmaf(-rel, eff, rhs_eff); }
DI float fused_normal(float &cur, float dimp, float upper) {   // returns the applied impulse
    const float nw = __builtin_fminf(__builtin_fmaxf(cur + dimp, 0.0f), upper);
    const float applied = nw - cur;
    cur = nw;
    return applied;
}
DI void fused_circle(float &i0, float &i1, float max_len) {
    const float len2 = __builtin_fmaf(i1, i1, i0 * i0);
    if (len2 > max_len * max_len) {
        const float len = sqrtf(len2);
        const float sc = len > kEps ? max_len / len : 0.0f;
        i0 *= sc; i1 *= sc;
    }
}
template <bool FUSED>
DI void row_apply(Delta &d, const RowReg &r, float imp) {   // apply_row_impulse with precomputed I^-1 J^T
    const f3 Jl = from4(r.f[0]);
    if (FUSED) {
        d.dvA = fma3(d.imA * Jl, imp, d.dvA);
        d.dvB = fma3(d.imB * (-Jl), imp, d.dvB);
        d.dwA = fma3(from4(r.f[3]), imp, d.dwA);
        d.dwB = fma3(from4(r.f[4]), imp, d.dwB);
    } else {
        d.dvA += d.imA * Jl * imp;
        d.dvB += d.imB * (-Jl) * imp;
        d.dwA += from4(r.f[3]) * imp;
        d.dwB += from4(r.f[4]) * imp;
    }
}
template <bool FUSED>
DI float row_relspeed(const Delta &d, const RowReg &r) {
    const f3 Jl = from4(r.f[0]);
    if (FUSED) return (dot3_fma(Jl, d.dvA) + dot3_fma(from4(r.f[1]), d.dwA)) + (dot3_fma(-Jl, d.dvB) + dot3_fma(from4(r.f[2]), d.dwB));
    return rel_speed(Jl, from4(r.f[1]), -Jl, from4(r.f[2]), d.dvA, d.dwA, d.dvB, d.dwB);
}
// solve_friction's clamp to the friction circle (constraint_row_friction.cpp:26-42), shared by every velocity-solve kernel.
// (Round 4, measured and dropped: a branch that spares the square root and the two divisions - a third of a point's instructions -
// for points without normal impulse, where the result is a signed zero. Bit-identical, but a wave only skips the slow path when
// none of its 32 manifolds slides in that point slot, and on the settled bench piles 17-28 % of the loaded points sit ON the
// friction circle: ~90 % of the wave-tasks have a slider in every slot (scripts/sliding_fraction.py). 1.69 vs 1.66 us per task.)
DI void friction_circle(float &i0, float &i1, float &di0, float &di1, float c0, float c1, float max_len) {
    const float len2 = i0 * i0 + i1 * i1;
    if (len2 > square(max_len)) {
        const float len = sqrtf(len2);
        if (len > kEps) { i0 = i0 / len * max_len; i1 = i1 / len * max_len; }
        else { i0 = 0; i1 = 0; }
        di0 = i0 - c0; di1 = i1 - c1;
    }
}
// solve(constraint_row&)'s clamp (constraint_row.cpp:38-50) as selects: dimp keeps its computed value unless a limit cuts in
DI void normal_clamp(float &cur, float &dimp, float upper) {
    const float imp = cur + dimp;
    const bool lt = imp < 0.0f, gt = !lt && imp > upper;
    const float nw = lt ? 0.0f : (gt ? upper : imp);
    dimp = (lt || gt) ? nw - cur : dimp;
    cur = nw;
}
// One normal row (solve(constraint_row&) + apply_row_impulse) and one friction pair (solve_friction) in either arithmetic.
template <bool FUSED>
DI void row_solve_normal(Delta &d, RowReg &r, float upper) {
    float cur = r.f[2].w;
    if (FUSED) {
        const float applied = fused_normal(cur, fused_delta(row_relspeed<true>(d, r), r.f[0].w, r.f[1].w * r.f[0].w), upper);
        r.f[2].w = cur;
        row_apply<true>(d, r, applied);
    } else {
        const float drel = row_relspeed<false>(d, r);
        float dimp = (r.f[1].w - drel) * r.f[0].w;
        normal_clamp(cur, dimp, upper);
        r.f[2].w = cur;
        row_apply<false>(d, r, dimp);
    }
}
template <bool FUSED>
DI void row_solve_friction(Delta &d, const RowReg &rn, RowReg &ra, RowReg &rb) {
    const float max_len = rn.f[3].w * rn.f[2].w;   // mu * current normal impulse
    if (FUSED) {
        const float c0 = ra.f[2].w, c1 = rb.f[2].w;
        float i0 = c0 + fused_delta(row_relspeed<true>(d, ra), ra.f[0].w, ra.f[1].w * ra.f[0].w);
        float i1 = c1 + fused_delta(row_relspeed<true>(d, rb), rb.f[0].w, rb.f[1].w * rb.f[0].w);
        fused_circle(i0, i1, max_len);
        ra.f[2].w = i0; rb.f[2].w = i1;
        row_apply<true>(d, ra, i0 - c0);
        row_apply<true>(d, rb, i1 - c1);
    } else {
        float di0 = (ra.f[1].w - row_relspeed<false>(d, ra)) * ra.f[0].w;
        float i0 = ra.f[2].w + di0;
        float di1 = (rb.f[1].w - row_relspeed<false>(d, rb)) * rb.f[0].w;
        float i1 = rb.f[2].w + di1;
        friction_circle(i0, i1, di0, di1, ra.f[2].w, rb.f[2].w, max_len);
        ra.f[2].w = i0; rb.f[2].w = i1;
        row_apply<false>(d, ra, di0);
        row_apply<false>(d, rb, di1);
    }
}
// NP (points of the manifold) is a template parameter: lanes are grouped by point count inside a colour, so a wave
// runs one instantiation, every loop is fully unrolled without predication and the compiler can issue all
// 15*NP row loads plus the body loads back to back before the first use (one memory round trip after the indices).
// PUSH = false: body deltas are gathered from / scattered to the body records (bdvw indexed by rbA/rbB).
// PUSH = true : rbA = nullptr-free variant - deltas arrive in this lane's own slots and leave towards the slots of
//               each body's next manifold (`rbA` then carries Rows::next, `bdvw` carries Rows::dslot).
template <int NP>
DI void rows_load(RowReg (&R)[NP][kRowsPerPoint], const float4 *__restrict__ rw, uint32_t rcap, uint32_t p) {
    // slots [0, NP) are loaded unconditionally (slots >= np hold stale but finite rows that are never used or stored);
    // NP is 4 for 3-4 point manifolds and 2 for 1-2 point ones, so at most one slot is fetched in vain
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int r = 0; r < kRowsPerPoint; ++r)
#pragma unroll
            for (int f = 0; f < kRowF; ++f) R[k][r].f[f] = rw[(size_t)((k * kRowsPerPoint + r) * kRowF + f) * rcap + p];
}
// One manifold's share of a sweep: its normal rows, then its friction pairs, in contact-list order.
template <bool WARM, int NP, bool FUSED>
DI void rows_solve(Delta &d, RowReg (&R)[NP][kRowsPerPoint], uint32_t np) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
        RowReg &r = R[k][0];
        if (WARM) row_apply<FUSED>(d, r, r.f[2].w);
        else row_solve_normal<FUSED>(d, r, r.f[4].w);   // upper = large_scalar, or a soft contact's force limit
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
        RowReg &ra = R[k][1], &rb = R[k][2];
        if (WARM) {   // warm_start(constraint_row_friction&)
            row_apply<FUSED>(d, ra, ra.f[2].w);
            row_apply<FUSED>(d, rb, rb.f[2].w);
        } else row_solve_friction<FUSED>(d, R[k][0], ra, rb);
    }
}
template <int NP>
DI void rows_store_impulses(const RowReg (&R)[NP][kRowsPerPoint], float4 *__restrict__ rw, uint32_t rcap, uint32_t p, uint32_t np) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
#pragma unroll
        for (int r = 0; r < kRowsPerPoint; ++r) rw[(size_t)((k * kRowsPerPoint + r) * kRowF + 2) * rcap + p] = R[k][r].f[2];
    }
}
// The same sweep with the rows held as separate normal / friction sets, so that a caller can fetch the friction rows
// of the later points while the normal rows are already being solved (k_contact_solve_df). Identical arithmetic.
DI void row_load(RowReg &r, const float4 *__restrict__ rw, uint32_t rcap, uint32_t p, int k, int row) {
#pragma unroll
    for (int f = 0; f < kRowF; ++f) r.f[f] = rw[(size_t)((k * kRowsPerPoint + row) * kRowF + f) * rcap + p];
}
template <bool WARM, int NP, bool FUSED>
DI void rows_solve_normals(Delta &d, RowReg (&Rn)[NP], uint32_t np) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
        if (WARM) row_apply<FUSED>(d, Rn[k], Rn[k].f[2].w);
        else row_solve_normal<FUSED>(d, Rn[k], kLarge);
    }
}
template <bool WARM, bool FUSED>
DI void rows_solve_friction(Delta &d, const RowReg &rn, RowReg &ra, RowReg &rb) {
    if (WARM) {   // warm_start(constraint_row_friction&)
        row_apply<FUSED>(d, ra, ra.f[2].w);
        row_apply<FUSED>(d, rb, rb.f[2].w);
    } else row_solve_friction<FUSED>(d, rn, ra, rb);
}
// contact_extras rows of one manifold after its normal and friction rows: the rolling pairs of all points, then the
// spinning rows (island_solver.cpp:76-111 keeps the row kinds in this order). Rolling is solve_friction with the roll
// coefficient (constraint_row_friction.cpp:11-66), spinning is constraint_row_spin_friction.cpp:5-36; both angular only.
struct XRow { float4 f[kXRowF]; };
DI void xrow_apply(Delta &d, const XRow &r, float imp) { d.dwA += from4(r.f[1]) * imp; d.dwB += from4(r.f[2]) * imp; }
DI float xrow_relspeed(const Delta &d, const XRow &r) { const f3 ax = from4(r.f[0]); return dot(ax, d.dwA) + dot(-ax, d.dwB); }
template <bool WARM, int NP>
DI void extras_solve(Delta &d, const RowReg (&R)[NP][kRowsPerPoint], uint32_t np, float4 *__restrict__ rwx, uint32_t rcap, uint32_t p) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
        const size_t xb = (size_t)(k * kXPoint) * rcap + p;
        const float mu = rwx[xb + 9 * (size_t)rcap].x;
        if (!(mu > 0)) continue;
        XRow ra, rb;
#pragma unroll
        for (int f = 0; f < kXRowF; ++f) { ra.f[f] = rwx[xb + (size_t)f * rcap]; rb.f[f] = rwx[xb + (size_t)(3 + f) * rcap]; }
        if (WARM) { xrow_apply(d, ra, ra.f[2].w); xrow_apply(d, rb, rb.f[2].w); continue; }
        float di0 = (ra.f[1].w - xrow_relspeed(d, ra)) * ra.f[0].w;
        float i0 = ra.f[2].w + di0;
        float di1 = (rb.f[1].w - xrow_relspeed(d, rb)) * rb.f[0].w;
        float i1 = rb.f[2].w + di1;
        const float len2 = i0 * i0 + i1 * i1;
        const float max_len = mu * R[k][0].f[2].w;   // roll coefficient * current normal impulse
        if (len2 > square(max_len)) {
            const float len = sqrtf(len2);
            if (len > kEps) { i0 = i0 / len * max_len; i1 = i1 / len * max_len; }
            else { i0 = 0; i1 = 0; }
            di0 = i0 - ra.f[2].w; di1 = i1 - rb.f[2].w;
        }
        rwx[xb + 2 * (size_t)rcap].w = i0; rwx[xb + 5 * (size_t)rcap].w = i1;
        xrow_apply(d, ra, di0);
        xrow_apply(d, rb, di1);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((uint32_t)k >= np) continue;
        const size_t xb = (size_t)(k * kXPoint) * rcap + p;
        const float mu = rwx[xb + 9 * (size_t)rcap].y;
        if (!(mu > 0)) continue;
        XRow r;
#pragma unroll
        for (int f = 0; f < kXRowF; ++f) r.f[f] = rwx[xb + (size_t)(6 + f) * rcap];
        if (WARM) { xrow_apply(d, r, r.f[2].w); continue; }
        const float max_len = mu * R[k][0].f[2].w;
        float dimp = (r.f[1].w - xrow_relspeed(d, r)) * r.f[0].w;
        const float cur = r.f[2].w, imp = cur + dimp, lo = -max_len, hi = max_len;
        float out;
        if (imp < lo) { dimp = lo - cur; out = lo; }
        else if (imp > hi) { dimp = hi - cur; out = hi; }
        else out = imp;
        rwx[xb + 8 * (size_t)rcap].w = out;
        xrow_apply(d, r, dimp);
    }
}
template <bool WARM, int NP, bool PUSH, bool FUSED>
DI void contact_solve_np(uint32_t p, uint32_t np, const uint32_t *__restrict__ rbA, const uint32_t *__restrict__ rbB,
                         float4 *__restrict__ rw, uint32_t rcap, float4 *__restrict__ bdvw, const float *__restrict__ im,
                         float4 *__restrict__ rwx = nullptr) {
    uint32_t ia, ib;   // PUSH: destination slots; else body indices
    if (PUSH) { ia = rbA[2 * (size_t)p] & kSlotMask; ib = rbA[2 * (size_t)p + 1] & kSlotMask; }
    else { ia = rbA[p]; ib = rbB[p]; }
    RowReg R[NP][kRowsPerPoint];
    rows_load<NP>(R, rw, rcap, p);
    Delta d;
    {
        float4 va, wa, vb, wb;
        if (PUSH) { va = bdvw[dslot_at(2 * p, 0)]; wa = bdvw[dslot_at(2 * p, 1)]; vb = bdvw[dslot_at(2 * p + 1, 0)]; wb = bdvw[dslot_at(2 * p + 1, 1)]; }
        else { va = bdvw[2 * (size_t)ia]; wa = bdvw[2 * (size_t)ia + 1]; vb = bdvw[2 * (size_t)ib]; wb = bdvw[2 * (size_t)ib + 1]; }
        d.dvA = from4(va); d.dwA = from4(wa);
        d.dvB = from4(vb); d.dwB = from4(wb);
        if (PUSH) { d.imA = im[2 * (size_t)p]; d.imB = im[2 * (size_t)p + 1]; }   // slot .w lanes carry hand-off tags
        else { d.imA = va.w; d.imB = vb.w; }
    }
    rows_solve<WARM, NP, FUSED>(d, R, np);
    if (rwx) extras_solve<WARM, NP>(d, R, np, rwx, rcap, p);
    if (!WARM) rows_store_impulses<NP>(R, rw, rcap, p, np);
    const float wA = PUSH ? 0.0f : d.imA, wB = PUSH ? 0.0f : d.imB;
    const size_t oa0 = PUSH ? dslot_at(ia, 0) : 2 * (size_t)ia, oa1 = PUSH ? dslot_at(ia, 1) : 2 * (size_t)ia + 1;
    const size_t ob0 = PUSH ? dslot_at(ib, 0) : 2 * (size_t)ib, ob1 = PUSH ? dslot_at(ib, 1) : 2 * (size_t)ib + 1;
    if (d.imA != 0) { bdvw[oa0] = to4(d.dvA, wA); bdvw[oa1] = to4(d.dwA, 0); }   // non-procedural bodies keep zero deltas
    if (d.imB != 0) { bdvw[ob0] = to4(d.dvB, wB); bdvw[ob1] = to4(d.dwB, 0); }
}
template <bool WARM, bool PUSH, bool FUSED>
DI void contact_solve_lane(uint32_t p, uint32_t np, const uint32_t *__restrict__ rbA, const uint32_t *__restrict__ rbB,
                           float4 *__restrict__ rw, uint32_t rcap, float4 *__restrict__ bdvw, const float *__restrict__ im,
                           float4 *__restrict__ rwx = nullptr) {
    if (np > 2) contact_solve_np<WARM, 4, PUSH, FUSED>(p, np, rbA, rbB, rw, rcap, bdvw, im, rwx);
    else contact_solve_np<WARM, 2, PUSH, FUSED>(p, np, rbA, rbB, rw, rcap, bdvw, im, rwx);
}
struct Split { uint32_t e4, e3, e2; };   // ends of the 4-, 3-, 2-point groups of a colour's sorted range
DI uint32_t np_of(uint32_t p, const Split &sp) { return p < sp.e4 ? 4u : (p < sp.e3 ? 3u : (p < sp.e2 ? 2u : 1u)); }
template <bool WARM, bool PUSH, bool FUSED>
__global__ void __launch_bounds__(64)
k_contact_solve(uint32_t start, uint32_t end, Split sp, const uint32_t *__restrict__ rbA, const uint32_t *__restrict__ rbB,
                float4 *__restrict__ rw, uint32_t rcap, float4 *__restrict__ bdvw, const float *__restrict__ im, float4 *__restrict__ rwx) {
    const uint32_t p = start + blockIdx.x * blockDim.x + threadIdx.x;
    if (p < end) contact_solve_lane<WARM, PUSH, FUSED>(p, np_of(p, sp), rbA, rbB, rw, rcap, bdvw, im, rwx);
}
// Tail colours are tiny (tens to hundreds of manifolds) yet would each cost a full dependent launch; ONE
// workgroup sweeps them in colour order instead, separated by workgroup barriers (same CU, same L1).
struct TailRanges { uint32_t n; uint32_t start[kMaxColours]; uint32_t end[kMaxColours]; Split split[kMaxColours]; };
constexpr uint32_t kTailThreads = 256, kTailMax = 512;   // one wave per SIMD keeps the full register budget
template <bool WARM, bool PUSH, bool FUSED>
__global__ void __launch_bounds__(256)
k_contact_solve_tail(TailRanges tr, const uint32_t *rbA, const uint32_t *rbB, float4 *rw, uint32_t rcap, float4 *bdvw, const float *im, float4 *rwx) {
    for (uint32_t c = 0; c < tr.n; ++c) {
        for (uint32_t p = tr.start[c] + threadIdx.x; p < tr.end[c]; p += kTailThreads)
            contact_solve_lane<WARM, PUSH, FUSED>(p, np_of(p, tr.split[c]), rbA, rbB, rw, rcap, bdvw, im, rwx);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- dataflow sweep: ONE launch runs the warm start and every iteration over every colour ------------------------
// The per-colour launches above serialise a step into ~colours x (iterations+1) dependent kernels (~6.5 us each on
// MI355X, of which the arithmetic is ~1.5 us). The dependencies are much finer than that: manifold p only needs the
// deltas of ITS two bodies as left by each body's previous manifold in colour order - exactly the push hand-off
// slots. Here every resident lane owns the manifolds p = t, t+G, t+2G, ... (G = resident lanes) and walks them sweep
// after sweep; before solving one it polls its own two slots until both carry the tag of the hand-off it is waiting
// for, and afterwards it stores the updated deltas, tagged, into the slots of each body's next manifold. A hand-off
// between two waves through device-coherent (sc1) 16-byte accesses costs ~0.5 us (scripts/ubench/pingpong.hip), so a
// colour step costs ~2 us instead of a kernel boundary, and different bodies advance through their chains
// independently (no barrier of any kind). The arithmetic and the order in which each body sees its manifolds are
// exactly those of the per-colour sweeps, so the results are bit-identical.
//
// Progress: tasks are visited in (sweep, p) order by every lane and every dependency points to a smaller (sweep, p),
// so with all G lanes resident the smallest unfinished task can always run. Lanes of one wave that depend on each
// other (a colour boundary inside the wave) are handled by solving only the wave's lowest pending colour at a time.
// Tags: a slot handed over during sweep s carries s+1; a chain head (its body's first manifold) consumes the value its
// body's last manifold left in the previous sweep, i.e. tag s, every other slot tag s+1; k_push_links zeroes all tags.
typedef float v4f __attribute__((ext_vector_type(4)));
struct DfArgs {
    uint32_t na, stride, sweeps;      // active manifolds, resident lanes, iterations + 1
    const uint32_t *keys_sorted;      // [p] colour*4 + (4 - num_points)
    const uint32_t *next;             // Rows::next
    const float *im;                  // Rows::im
    float4 *rw; uint32_t rcap;
    float4 *dslot;
    Counters *cnt;
    uint64_t *trace;                  // developer aid (EDYNHIP_DF_TRACE): 4 timestamps per (sweep, round, wave), else nullptr
    const uint8_t *skip;              // mixed schedule: [p] != 0 = the manifold's island has joints and is solved by k_island_velocity; else nullptr
    uint32_t nap;                     // pause between two polls of a wave that found nothing: 0 none, 1 s_sleep 1, else s_sleep 4 (EDYNHIP_DF_NAP, developer knob)
    uint32_t xcd_lists;               // two-lane kernel: tasks are handed out per XCD (XcdLists below) instead of by p alone
};
// ---- XCD-local task lists (round 5) ---------------------------------------------------------------------------------------------------
// A hand-off between two waves of the SAME XCD is noticed ~0.2 us sooner than one that crosses XCDs (scripts/ubench/pingpong.hip: 280 against
// 460-530 ns per hop on an idle chip, same sc1 stores and polls: the poll is served by the XCD's own L2), and workgroup b of a launch runs
// on XCD (b + const) mod 8. The sorted order p is (colour, point count)-major and, inside such a class, in manifold order - which follows
// the bodies' indices, i.e. where they sit in the scene. So every class is cut into EIGHTHS, and the waves with blockIdx = g (mod 8) take
// the g-th eighth of every class: a body's manifolds - in whatever colour - mostly lie in the same eighth of their classes, so most of
// its hand-offs stay inside one XCD. Nothing but speed depends on the placement (every hand-off stays an sc1 store and an sc1 poll),
// and nothing but the ORDER IN WHICH WAVES PICK TASKS changes: every array stays indexed by p. List g = the g-th eighths of all classes
// in class (= colour) order, so a wave still meets its tasks in colour order and every dependency points to an earlier task.
constexpr uint32_t kXcds = 8, kCls = 4 * kMaxColours;
// The eighths are cut at multiples of 32 in p and every class starts a new wave-task in its list (the unused lanes of a class's last
// task idle): a wave's 32 manifolds are then 32-aligned in p exactly like in the plain assignment - one contiguous piece of the hand-off
// slot layout per poll (dslot_at / pslot_at) - and never straddle two classes (first version, unpadded: 576 instead of 72 class
// boundaries fell inside wave-tasks, which then run colour after colour and in the predicated form: velocity solve 0.55 -> 0.64 ms).
struct XcdLists { uint32_t pre[kCls + 1]; uint32_t first[kCls], lo[kCls], hi[kCls]; };   // list g: pre[c] = task slots before class c; first[c] = p of slot 0 of the class (32-aligned); [lo, hi) = the class's g-th eighth
DI void xcd_lists_build(XcdLists &L, const Counters *cnt, uint32_t g) {   // one wave (64 lanes); ends with a barrier
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n[4], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t c = 4 * lane + k, a = cnt->colour_start[c], e = cnt->colour_end[c], size = e > a && c < 4 * kMaxContactColours ? e - a : 0u;
        auto cut = [&](uint32_t i) {   // boundary i of the class's eighths: 0 = start, 8 = end, between: the nearest multiple of 32 inside the class
            if (i == 0) return a;
            if (i >= kXcds) return a + size;
            const uint32_t b = (a + (uint32_t)(((uint64_t)size * i) / kXcds) + 16u) & ~31u;
            return b < a ? a : (b > a + size ? a + size : b);
        };
        const uint32_t lo = cut(g), hi = cut(g + 1), v = lo & ~31u;
        L.first[c] = v; L.lo[c] = lo; L.hi[c] = hi;
        n[k] = hi > lo ? ((hi - v) + 31u) & ~31u : 0u;
        sum += n[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
    uint32_t at = inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) { L.pre[4 * lane + k] = at; at += n[k]; }
    if (lane == 63) L.pre[kCls] = at;
    __syncthreads();
}
DI uint32_t xcd_lists_p(const XcdLists &L, uint32_t q, uint32_t none) {   // q < L.pre[kCls]: the manifold in task slot q of the list, or `none` for an idle slot
    uint32_t lo = 0, hi = kCls;
#pragma unroll 1
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.pre[mid] <= q) lo = mid; else hi = mid; }
    const uint32_t p = L.first[lo] + (q - L.pre[lo]);
    return p >= L.lo[lo] && p < L.hi[lo] ? p : none;
}
DI void df_nap(uint32_t nap) { if (nap == 0) return; if (nap == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4); }
DI void df_poll(const float4 *slot, v4f &a0, v4f &a1, v4f &b0, v4f &b1) {   // both sides' (dv|tag, dw|tag): pieces 1 KiB apart (dslot_at)
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(slot) : "memory");
}
DI void df_publish(float4 *slot, f3 dv, f3 dw, uint32_t tag) {   // slot = &dslot[dslot_at(s, 0)]; the dw piece is 1 KiB further
    const float t = __uint_as_float(tag);
    const v4f v = {dv.x, dv.y, dv.z, t}, w = {dw.x, dw.y, dw.z, t};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:1024 sc1" : : "v"(slot), "v"(v), "v"(w) : "memory");
}
constexpr uint32_t kDfSpinLimit = 1u << 22;   // ~seconds; a hand-off normally arrives within microseconds
template <bool WARM, int NP, bool FUSED>
DI void df_task(const DfArgs &a, uint32_t p, bool valid, uint32_t np, uint32_t col, uint32_t sweep, uint64_t *trace_slot) {
    // Rows fetched before the wait: every normal row and the friction rows of the first two points. The friction rows
    // of points 2 and 3 are fetched when the hand-offs have arrived and land while the normal rows are being solved;
    // holding all 15*NP float4 across the wait would push the kernel into AGPR copies.
    constexpr int kEarly = NP > 2 ? 2 : NP;
    RowReg Rn[NP], Rf[NP][2];
    const uint64_t w0 = a.trace ? wall_clock64() : 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) row_load(Rn[k], a.rw, a.rcap, p, k, 0);
#pragma unroll
    for (int k = 0; k < kEarly; ++k) { row_load(Rf[k][0], a.rw, a.rcap, p, k, 1); row_load(Rf[k][1], a.rw, a.rcap, p, k, 2); }
    uint64_t w1 = 0, w2 = 0;
    const uint32_t nA = a.next[2 * (size_t)p], nB = a.next[2 * (size_t)p + 1];
    Delta d;
    d.imA = a.im[2 * (size_t)p]; d.imB = a.im[2 * (size_t)p + 1];
    d.dvA = d.dwA = d.dvB = d.dwB = mk3(0, 0, 0);
    const uint32_t wantA = (nA & kHeadBit) ? sweep : sweep + 1, wantB = (nB & kHeadBit) ? sweep : sweep + 1;
    bool gotA = d.imA == 0, gotB = d.imB == 0;   // read-only bodies hand nothing over: their deltas stay zero
    bool done = !valid;
    const float4 *mine = a.dslot + dslot_at(2 * p, 0);
    for (uint32_t spin = 0;; ++spin) {
        if (!done && !(gotA && gotB)) {
            v4f a0, a1, b0, b1;
            df_poll(mine, a0, a1, b0, b1);
            if (a.trace && w1 == 0) w1 = wall_clock64();
            if (!gotA && __float_as_uint(a0.w) == wantA && __float_as_uint(a1.w) == wantA) {
                d.dvA = mk3(a0.x, a0.y, a0.z); d.dwA = mk3(a1.x, a1.y, a1.z); gotA = true;
            }
            if (!gotB && __float_as_uint(b0.w) == wantB && __float_as_uint(b1.w) == wantB) {
                d.dvB = mk3(b0.x, b0.y, b0.z); d.dwB = mk3(b1.x, b1.y, b1.z); gotB = true;
            }
        }
        const uint64_t pending = __ballot(!done);
        if (pending == 0) {
            if (trace_slot && (threadIdx.x & 63) == 0) { trace_slot[0] = w0; trace_slot[1] = w1; trace_slot[2] = w2; trace_slot[3] = wall_clock64(); }
            break;
        }
        const uint32_t minc = __shfl(col, __ffsll((long long)pending) - 1);   // lanes are in colour order
        const bool mine_now = !done && col == minc;
        if (__ballot(mine_now && !(gotA && gotB)) == 0) {
            if (a.trace && w2 == 0) w2 = wall_clock64();
            if (mine_now) {
#pragma unroll
                for (int k = kEarly; k < NP; ++k) { row_load(Rf[k][0], a.rw, a.rcap, p, k, 1); row_load(Rf[k][1], a.rw, a.rcap, p, k, 2); }
                rows_solve_normals<WARM, NP, FUSED>(d, Rn, np);
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if ((uint32_t)k < np) rows_solve_friction<WARM, FUSED>(d, Rn[k], Rf[k][0], Rf[k][1]);
                // hand the deltas over first (the next manifolds are waiting for them), then store the impulses
                if (d.imA != 0) df_publish(a.dslot + dslot_at(nA & kSlotMask, 0), d.dvA, d.dwA, sweep + 1);
                if (d.imB != 0) df_publish(a.dslot + dslot_at(nB & kSlotMask, 0), d.dvB, d.dwB, sweep + 1);
                if (!WARM) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if ((uint32_t)k >= np) continue;
                        a.rw[(size_t)((k * kRowsPerPoint + 0) * kRowF + 2) * a.rcap + p] = Rn[k].f[2];
                        a.rw[(size_t)((k * kRowsPerPoint + 1) * kRowF + 2) * a.rcap + p] = Rf[k][0].f[2];
                        a.rw[(size_t)((k * kRowsPerPoint + 2) * kRowF + 2) * a.rcap + p] = Rf[k][1].f[2];
                    }
                }
                done = true;
            }
        } else {
            if (spin > kDfSpinLimit || ((spin & 1023u) == 1023u && __hip_atomic_load(&a.cnt->df_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (spin > kDfSpinLimit) atomicExch(&a.cnt->df_abort, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);   // ~0.1 us between polls
        }
    }
}
constexpr uint32_t kDfBlock = 64;   // one wave per workgroup: the dispatcher spreads the waves over all CUs
template <bool FUSED>
__global__ void __launch_bounds__(kDfBlock) k_contact_solve_df(DfArgs a) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    const uint32_t rounds = (a.na + a.stride - 1) / a.stride, nwaves = a.stride >> 6;
    for (uint32_t sweep = 0; sweep < a.sweeps; ++sweep)
        for (uint32_t base = 0, round = 0; base < a.na; base += a.stride, ++round) {
            const uint32_t pt = base + t;
            const bool valid = pt < a.na && !(a.skip && a.skip[pt]);
            if (!__any(valid)) continue;              // whole wave beyond the end (wave-uniform)
            const uint32_t p = valid ? pt : a.na - 1;
            const uint32_t key = a.keys_sorted[p];
            const uint32_t np = 4u - (key & 3u), col = key >> 2;
            const bool big = __any(valid && np > 2);  // lanes are grouped by point count: uniform except at a group boundary
            uint64_t *tr = a.trace ? a.trace + 4 * ((size_t)(sweep * rounds + round) * nwaves + blockIdx.x) : nullptr;
            if (sweep == 0) { if (big) df_task<true, 4, FUSED>(a, p, valid, np, col, sweep, tr); else df_task<true, 2, FUSED>(a, p, valid, np, col, sweep, tr); }
            else { if (big) df_task<false, 4, FUSED>(a, p, valid, np, col, sweep, tr); else df_task<false, 2, FUSED>(a, p, valid, np, col, sweep, tr); }
        }
}

// ---- push hand-off: link every (lane, side) to the same body's next manifold in colour order (cyclic) ----
// `isl_joint` (mixed schedule, else nullptr): islands with joints take no part in the hand-off chains - their manifolds are marked in
// Rows::skip and their bodies keep first_slot = none (k_island_velocity / k_island_position solve them on the body records).
// (Round 5, measured and dropped: the links computed inside k_prep_contacts, the slot table written by the colour sort's scatter - one
//  launch less, and no gain: the scatter grows by what the table costs (+7 us), and the velocity solve that then follows the row
//  preparation directly starts 10-25 us slower - it meets the row stores of the preparation still on their way out of the L2s, which this
//  small kernel otherwise absorbs. A/B on one box: 769 against 783 steps/s.)
__global__ void k_push_links(uint32_t n_active, Rows rows, const uint32_t *__restrict__ keys_sorted, Bodies b, const uint64_t *__restrict__ used,
                             const uint32_t *__restrict__ isl_joint) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_active) return;
    if (isl_joint) {
        const bool fused = isl_joint[rows.label[p]] != 0;
        rows.skip[p] = fused ? 1 : 0;
        if (fused) return;
    }
    const uint32_t col = keys_sorted[p] >> 2;
#pragma unroll
    for (uint32_t side = 0; side < 2; ++side) {
        const uint32_t body = side ? rows.bB[p] : rows.bA[p];
        const uint32_t slot = 2 * p + side;
        // every slot starts a step as (0,0,0 | tag 0): the chain head's seed, and "nothing handed over yet" elsewhere
        rows.dslot[dslot_at(slot, 0)] = make_float4(0, 0, 0, 0); rows.dslot[dslot_at(slot, 1)] = make_float4(0, 0, 0, 0);
        if (!is_dynamic(b.flags[body])) {   // read-only partner: permanent zero deltas, never written
            rows.next[slot] = slot;
            rows.im[slot] = 0.0f;
            continue;
        }
        rows.im[slot] = B_POS(b, body).w;
        const uint64_t mask = used[body];                         // colours of this body's active manifolds
        const uint64_t above = col >= 63 ? 0ull : mask & ~((2ull << col) - 1ull);
        const uint32_t nextc = (uint32_t)__ffsll((long long)(above ? above : mask)) - 1;
        const bool head = col == (uint32_t)__ffsll((long long)mask) - 1;   // the body's first manifold of a sweep
        rows.next[slot] = rows.slot_of[(size_t)body * kMaxColours + nextc] | (head ? kHeadBit : 0u);
        if (head) rows.first_slot[body] = slot;
    }
}

DI void store_impulses_of(uint32_t p, const Rows &rows, uint32_t rcap, const Manifolds &mf) {
    const uint32_t m = rows.order[p], np = rows.np[p];
    for (uint32_t k = 0; k < np; ++k) {
        const size_t d = slot_at(mf.cap, k, m), td = pt_at(mf.cap, k, m);
        float4 im = mf.imp[td];
        im.x = rows.rw[(size_t)((k * kRowsPerPoint + 0) * kRowF + 2) * rcap + p].w;
        im.y = rows.rw[(size_t)((k * kRowsPerPoint + 1) * kRowF + 2) * rcap + p].w;
        im.z = rows.rw[(size_t)((k * kRowsPerPoint + 2) * kRowF + 2) * rcap + p].w;
        mf.imp[td] = im;
        if (rows.rwx && mf.ximp) {   // contact_extras_constraint::store_applied_impulses (contact_extras_constraint.cpp:88-107)
            const size_t xb = (size_t)(k * kXPoint) * rcap + p;
            const float4 mu = rows.rwx[xb + 9 * (size_t)rcap];
            float4 xi = mf.ximp[d];
            if (mu.x > 0) { xi.x = rows.rwx[xb + 2 * (size_t)rcap].w; xi.y = rows.rwx[xb + 5 * (size_t)rcap].w; }
            if (mu.y > 0) xi.z = rows.rwx[xb + 8 * (size_t)rcap].w;
            mf.ximp[d] = xi;
        }
    }
}
__global__ void k_store_impulses(uint32_t n_active, Rows rows, uint32_t rcap, Manifolds mf) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_active) store_impulses_of(p, rows, rcap, mf);
}

// ------------------------------------------------------------------ joints (point, hinge)
// Slot r of a joint (ctx.hpp Joints): 0..2 the three linear rows, hinge 3/4 the rows along p and q, every other slot an
// axial row {0, ax, 0, -ax} along `wax` (hinge axis for slots 5..8 of a hinge, relative spin for slot 3 of a point joint).
DI void joint_rowJ(int type, int r, f3 rA, f3 rB, f3 wp, f3 wq, f3 wax, f3 wbx, f3 &J0, f3 &J1, f3 &J2, f3 &J3) {
    const bool hinge = type == EDYNHIP_JOINT_HINGE;
    if (type == EDYNHIP_JOINT_GRAVITY) {   // {dn, 0, -dn, -0}: no lever arms
        J0 = wp; J1 = mk3(0, 0, 0); J2 = -wp; J3 = -mk3(0, 0, 0);
    } else if (type == EDYNHIP_JOINT_DISTANCE || type == EDYNHIP_JOINT_SOFT_DISTANCE || type == EDYNHIP_JOINT_CONE) {
        // every row of the distance constraints runs along the pivot separation, of the cone along the cone normal
        // (kept in wp; wq, wax = the two lever arms crossed with it)
        J0 = wp; J1 = wq; J2 = -wp; J3 = -wax;
    } else if (type == EDYNHIP_JOINT_CVJOINT && r >= 3) {
        // twist rows 3..6: {0, twist axis of A (wp), 0, -twist axis of B (wq)}; 7: bend friction axis (wax); 8: bend spring axis (wbx)
        J0 = mk3(0, 0, 0); J2 = mk3(0, 0, 0);
        if (r < 7) { J1 = wp; J3 = -wq; }
        else { const f3 ax = r == 7 ? wax : wbx; J1 = ax; J3 = -ax; }
    } else if (r < 3) {
        // J = {I.row[i], -skew(rA).row[i], -I.row[i], skew(rB).row[i]}
        f3 e = r == 0 ? mk3(1, 0, 0) : (r == 1 ? mk3(0, 1, 0) : mk3(0, 0, 1));
        f3 sa = r == 0 ? mk3(0, -rA.z, rA.y) : (r == 1 ? mk3(rA.z, 0, -rA.x) : mk3(-rA.y, rA.x, 0));
        f3 sb = r == 0 ? mk3(0, -rB.z, rB.y) : (r == 1 ? mk3(rB.z, 0, -rB.x) : mk3(-rB.y, rB.x, 0));
        J0 = e; J1 = -sa; J2 = -e; J3 = sb;
    } else {
        f3 ax = (hinge && r == 3) ? wp : ((hinge && r == 4) ? wq : wax);
        J0 = mk3(0, 0, 0); J1 = ax; J2 = mk3(0, 0, 0); J3 = -ax;
    }
}
// atan2 evaluated in double and rounded once: correctly rounded fp32, a value that does not depend on a math library
// (the reference's std::atan2(float) depends on the C library's last bit - the same convention as integrate()'s sin/cos).
DI float atan2_cr(float y, float x) { return (float)atan2((double)y, (double)x); }
DI float normalize_angle(float a) {   // math.hpp:53-63
    a = fmodf(a, kPi2);
    if (a < -kPi) return a + kPi2;
    if (a > kPi) return a - kPi2;
    return a;
}
// cvjoint_constraint.cpp:25-37 relative twist angle; quaternion.cpp:25-38 shortest_arc
DI q4 shortest_arc(f3 v0, f3 v1) {
    const f3 c = cross(v0, v1);
    const float d = dot(v0, v1);
    if (d <= -1 + kEps) {
        f3 n, m;
        plane_space(v0, n, m);
        return q4{n.x, n.y, n.z, 0};
    }
    const float s = sqrtf((1 + d) * 2);
    const float rs = 1 / s;
    return normalize(q4{c.x * rs, c.y * rs, c.z * rs, s * 0.5f});
}
DI float cvjoint_relative_angle(q4 ornA, q4 ornB, f3 tA, f3 tB, f3 colA1, f3 colA2, f3 colB1) {
    const q4 arc = shortest_arc(tB, tA);
    const f3 angle_axisB = rotate(conjugate(ornA) * arc * ornB, colB1);
    return atan2_cr(dot(angle_axisB, colA2), dot(angle_axisB, colA1));
}
DI float track_angle(float tracked, float new_angle) {   // update_angle (hinge_constraint.cpp:80-89, cvjoint_constraint.cpp:39-47)
    const float previous = normalize_angle(tracked);
    const float d0 = new_angle - previous;
    const float d1 = d0 + kPi2 * (d0 < 0 ? 1.0f : -1.0f);
    return tracked + (fabsf(d0) < fabsf(d1) ? d0 : d1);
}
// (also marks the island of every awake joint in `isl_joint`, cleared again by k_finish: what the schedules in solve() key on)
__global__ void k_prep_joints(Joints j, Bodies b, float dt, uint32_t *__restrict__ isl_joint) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n) return;
    const uint32_t ia = j.bodyA[i], ib = j.bodyB[i];
    if (edge_asleep(b.flags[ia], b.flags[ib])) return;   // sleeping island: its joints are not prepared or solved
    if (is_dynamic(b.flags[ia])) isl_joint[b.island[ia]] = 1u;
    else if (is_dynamic(b.flags[ib])) isl_joint[b.island[ib]] = 1u;
    if (j.type[i] == EDYNHIP_JOINT_NULL) { j.rmask[i] = 0; return; }   // null_constraint: an island-graph edge without rows
    const BRef A = load_bref(b, ia), B = load_bref(b, ib);
    const f3 pA = to_world(from4(j.pivA[i]), A.org, A.orn), pB = to_world(from4(j.pivB[i]), B.org, B.orn);
    const f3 rA = pA - A.pos, rB = pB - B.pos;
    const int type = j.type[i];
    const bool hinge = type == EDYNHIP_JOINT_HINGE;
    f3 wp = mk3(0, 0, 0), wq = mk3(0, 0, 0), wax = mk3(0, 0, 0), wbx = mk3(0, 0, 0);
    f3 rAx = rA, rBx = rB;   // lever arms stored for the solve kernel (the cone's A arm ends on the cone, not at its pivot)
    if (hinge) { wp = rotate(A.orn, from4(j.pA[i])); wq = rotate(A.orn, from4(j.qA[i])); }
    auto P = [&](int k) { return j.params[(size_t)k * j.cap + i]; };
    uint32_t mask = hinge ? 0x1Fu : 0x7u;
    // optional rows: per-slot error / restitution / limits (hinge_constraint.cpp:69-178, point_constraint.cpp:33-46)
    if (type == EDYNHIP_JOINT_GENERIC) return;   // k_prep_generic
    float err[kJointBaseSlots], rest[kJointBaseSlots], lo[kJointBaseSlots], hi[kJointBaseSlots];
#pragma unroll
    for (int r = 0; r < kJointBaseSlots; ++r) { err[r] = 0; rest[r] = 0; lo[r] = -kScalarMax; hi[r] = kScalarMax; }
    if (type == EDYNHIP_JOINT_GRAVITY) {   // gravity_constraint.cpp:6-28
        const f3 d = A.pos - B.pos;
        const float l2 = fmaxf(length_sqr(d), kEps);
        const float l = sqrtf(l2);
        wp = d / l;
        const float F = 6.674e-11f / (l2 * A.inv_m * B.inv_m);   // gravitational_constant, math/constants.hpp
        const float Pg = F * dt;
        lo[0] = -Pg; hi[0] = Pg; err[0] = kLarge;
        mask = 0x1u;
    } else if (type == EDYNHIP_JOINT_CONE) {   // cone_constraint.cpp:12-95
        const f3 fx = from4(j.axA[i]), fy = from4(j.pA[i]), fz = from4(j.qA[i]);   // columns of the frame in A
        const m3 frame = m3_columns(fx, fy, fz);
        const f3 pivA = from4(j.pivA[i]);
        const f3 pivotB_in_A = to_object(pB, A.org, A.orn);
        const f3 pf = to_object(pivotB_in_A, pivA, frame);
        const float scaling_y = 1.0f / P(0), scaling_z = 1.0f / P(1);
        const f3 ps = pf * mk3(1, scaling_y, scaling_z);
        const float proj = ps.y * ps.y + ps.z * ps.z;
        f3 normal_scaled, tangent_scaled;
        if (proj > kEps) {
            normal_scaled = normalize(mk3(-sqrtf(proj), ps.y, ps.z));
            tangent_scaled = normalize(mk3(0, -ps.z, ps.y));
        } else {
            normal_scaled = normalize(mk3(-1, 1, 0));
            tangent_scaled = normalize(mk3(0, 0, 1));
        }
        const float error = dot(ps, normal_scaled);
        const f3 dir_on_cone = mk3(-normal_scaled.x, normal_scaled.y, normal_scaled.z);
        const float cone_proj = dot(ps, dir_on_cone);
        const f3 descale = mk3(1, 1 / scaling_y, 1 / scaling_z);
        const f3 point_on_cone = (dir_on_cone * cone_proj) * descale;
        const f3 pivotA_world = to_world(pivA + mul(frame, point_on_cone), A.org, A.orn);
        const f3 tangent = normalize(tangent_scaled * descale);
        const f3 normal = normalize(cross(tangent, point_on_cone));
        const f3 nw = rotate(A.orn, mul(frame, normal));
        rAx = pivotA_world - A.pos;
        wp = nw; wq = cross(rAx, nw); wax = cross(rB, nw);
        err[0] = -error / dt; rest[0] = P(2); lo[0] = 0; hi[0] = kLarge;
        mask = 0x1u;
        if (P(3) > 0 && P(4) > 0) {
            const float deflection = P(4) + error;
            const float spring_impulse = P(3) * deflection * dt;
            lo[1] = 0; hi[1] = fmaxf(0.0f, spring_impulse);
            err[1] = -deflection / dt;
            mask = 0x3u;
        }
    } else if (type == EDYNHIP_JOINT_CVJOINT) {   // cvjoint_constraint.cpp:49-222
        const float twist_min = P(0), twist_max = P(1), twist_restitution = P(2), bump_angle = P(3), bump_stiffness = P(4),
                    twist_friction_torque = P(5), twist_rest_angle = P(6), twist_stiffness = P(7), twist_damping = P(8),
                    bend_stiffness = P(12), bend_friction_torque = P(13), bend_damping = P(14);
        const f3 tA = rotate(A.orn, from4(j.axA[i])), tB = rotate(B.orn, from4(j.axB[i]));
        wp = tA; wq = tB;
        lo[0] = lo[1] = lo[2] = -kLarge; hi[0] = hi[1] = hi[2] = kLarge;
        mask = 0xFu;   // three pivot rows (no error term: cvjoint leaves the pivots to its position solve) + the twist row
        const bool has_limit = twist_min < twist_max;
        float angle = j.angle[i];
        lo[3] = -kLarge; hi[3] = kLarge;
        if (has_limit) {
            const float current = cvjoint_relative_angle(A.orn, B.orn, tA, tB, from4(j.pA[i]), from4(j.qA[i]), from4(j.pB[i]));
            angle = track_angle(angle, current);
            j.angle[i] = angle;
            float limit_error;
            const float mid = (twist_min + twist_max) / 2.0f;
            if (angle < mid) { limit_error = twist_min - angle; lo[3] = -kLarge; hi[3] = 0; }
            else { limit_error = twist_max - angle; lo[3] = 0; hi[3] = kLarge; }
            if (angle > twist_min && angle < twist_max) err[3] = limit_error / dt;
            rest[3] = twist_restitution;
        }
        if (has_limit && bump_stiffness > 0 && bump_angle > 0) {
            float defl = 0;
            const float bmin = twist_min + bump_angle, bmax = twist_max - bump_angle;
            if (angle < bmin) defl = angle - bmin;
            else if (angle > bmax) defl = angle - bmax;
            const float imp = bump_stiffness * defl * dt;
            lo[4] = fminf(imp, 0.0f); hi[4] = fmaxf(0.0f, imp);
            err[4] = -defl / dt;
            mask |= 1u << 4;
        }
        if (has_limit && twist_stiffness > 0) {
            const float defl = angle - twist_rest_angle;
            const float imp = twist_stiffness * defl * dt;
            lo[5] = fminf(imp, 0.0f); hi[5] = fmaxf(0.0f, imp);
            err[5] = -defl / dt;
            mask |= 1u << 5;
        }
        if (has_limit && (twist_friction_torque > 0 || twist_damping > 0)) {
            float fi = twist_friction_torque * dt;
            if (twist_damping > 0) {
                const float relvel = dot(A.w, tA) - dot(B.w, tB);
                fi += fabsf(relvel) * twist_damping * dt;
            }
            lo[6] = -fi; hi[6] = fi;
            mask |= 1u << 6;
        }
        if (bend_friction_torque > 0 || bend_damping > 0) {
            const f3 twA = dot(A.w, tA) * tA, twB = dot(B.w, tB) * tB;
            const f3 angvel_rel = (A.w - twA) - (B.w - twB);
            const float angspd_rel = sqrtf(length_sqr(angvel_rel));
            wax = angspd_rel > kEps ? angvel_rel / angspd_rel : rotate(A.orn, from4(j.pA[i]));
            float fi = bend_friction_torque * dt;
            if (twist_damping > 0) fi += fabsf(angspd_rel) * bend_damping * dt;   // (the reference tests twist_damping here, :199)
            lo[7] = -fi; hi[7] = fi;
            mask |= 1u << 7;
        }
        if (bend_stiffness > 0) {
            f3 bend_axis = cross(rotate(A.orn, mk3(P(9), P(10), P(11))), tB);
            const float len = sqrtf(length_sqr(bend_axis));
            const float bend_angle = (float)asin((double)len);   // correctly rounded, like atan2_cr
            if (len > kEps) bend_axis = div_recip(bend_axis, len);
            else bend_axis = rotate(A.orn, from4(j.pA[i]));
            wbx = bend_axis;
            const float imp = bend_stiffness * bend_angle * dt;
            lo[8] = fminf(imp, 0.0f); hi[8] = fmaxf(0.0f, imp);
            err[8] = -bend_angle / dt;
            mask |= 1u << 8;
        }
    } else if (type == EDYNHIP_JOINT_DISTANCE) {   // distance_constraint.cpp:7-31
        f3 d = pA - pB;
        const float dist_sqr = length_sqr(d);
        if (!(dist_sqr > kEps)) d = mk3(1, 0, 0);
        wp = d; wq = cross(rA, d); wax = cross(rB, d);
        const float distance = P(0);
        err[0] = 0.5f * (dist_sqr - distance * distance) / dt;
        lo[0] = -kLarge; hi[0] = kLarge;
        mask = 0x1u;
    } else if (type == EDYNHIP_JOINT_SOFT_DISTANCE) {   // soft_distance_constraint.cpp:8-62
        const f3 d = pA - pB;
        const float dist_sqr = length_sqr(d), dist = sqrtf(dist_sqr);
        const f3 dn = dist_sqr > kEps ? d / dist : mk3(1, 0, 0);
        wp = dn; wq = cross(rA, dn); wax = cross(rB, dn);
        const float spring_impulse = P(1) * (P(0) - dist) * dt;
        lo[0] = fminf(spring_impulse, 0.0f); hi[0] = fmaxf(0.0f, spring_impulse);
        err[0] = spring_impulse > 0 ? -kLarge : kLarge;
        const float relspd = rel_speed(dn, wq, -dn, -wax, A.v, A.w, B.v, B.w);
        const float damping_impulse = P(2) * relspd * dt;
        lo[1] = -fabsf(damping_impulse); hi[1] = fabsf(damping_impulse);
        mask = 0x3u;
    } else if (!hinge) {
#pragma unroll
        for (int r = 0; r < 3; ++r) err[r] = (comp(pA, r) - comp(pB, r)) / dt;
        const float friction_torque = P(0);
        if (friction_torque > 0) {
            f3 spin = A.w - B.w;
            const float lsqr = length_sqr(spin);
            if ((double)lsqr > 1e-18) {   // try_normalize, vector3.hpp:239-248
                wax = div_recip(spin, sqrtf(lsqr));   // vector3 operator/= multiplies by the reciprocal
                const float fi = friction_torque * dt;
                lo[3] = -fi; hi[3] = fi;
                mask |= 1u << 3;
            }
        }
    } else {
        const float angle_min = P(0), angle_max = P(1), limit_restitution = P(2), bump_stop_angle = P(3), bump_stop_stiffness = P(4),
                    torque = P(5), speed = P(6), rest_angle = P(7), stiffness = P(8), damping = P(9);
        const bool has_limit = angle_min < angle_max, has_spring = stiffness > 0, has_torque = torque > 0 || damping > 0;
        if (has_limit || has_spring || has_torque) wax = rotate(A.orn, from4(j.axA[i]));
        float angle = j.angle[i];
        if (has_limit || has_spring) {
            const f3 angle_axisB = rotate(B.orn, from4(j.pB[i]));
            const float current = atan2_cr(dot(angle_axisB, wq), dot(angle_axisB, wp));
            const float previous = normalize_angle(angle);
            const float d0 = current - previous;
            const float d1 = d0 + kPi2 * (d0 < 0 ? 1.0f : -1.0f);
            angle += fabsf(d0) < fabsf(d1) ? d0 : d1;
            j.angle[i] = angle;
        }
        if (has_limit) {
            const float halfway = (angle_min + angle_max) / 2.0f;
            float limit_error;
            if (angle < halfway) { limit_error = angle_min - angle; lo[5] = -kLarge; hi[5] = 0; }
            else { limit_error = angle_max - angle; lo[5] = 0; hi[5] = kLarge; }
            err[5] = limit_error / dt; rest[5] = limit_restitution;
            mask |= 1u << 5;
            if (bump_stop_stiffness > 0 && bump_stop_angle > 0) {
                float defl = 0;
                const float bmin = angle_min + bump_stop_angle, bmax = angle_max - bump_stop_angle;
                if (angle < bmin) defl = angle - bmin;
                else if (angle > bmax) defl = angle - bmax;
                if (defl != 0) {
                    const float imp = bump_stop_stiffness * defl * dt;
                    lo[6] = fminf(imp, 0.0f); hi[6] = fmaxf(0.0f, imp);
                    err[6] = -defl / dt;
                    mask |= 1u << 6;
                }
            }
        }
        if (has_spring) {
            const float defl = angle - rest_angle;
            const float imp = stiffness * defl * dt;
            lo[7] = fminf(imp, 0.0f); hi[7] = fmaxf(0.0f, imp);
            err[7] = -defl / dt;
            mask |= 1u << 7;
        }
        if (has_torque) {
            float ti = torque * dt;
            if (damping > 0) {
                const float relvel = dot(A.w, wax) - dot(B.w, wax);
                ti += fabsf(relvel) * damping * dt;
            }
            lo[8] = -ti; hi[8] = ti;
            err[8] = -speed;
            mask |= 1u << 8;
        }
    }
    j.rA[i] = to4(rAx, 0); j.rB[i] = to4(rBx, 0); j.wp[i] = to4(wp, 0); j.wq[i] = to4(wq, 0); j.wax[i] = to4(wax, 0); j.wbx[i] = to4(wbx, 0);
    j.rmask[i] = mask;
#pragma unroll
    for (int r = 0; r < kJointBaseSlots; ++r) {
        if (!((mask >> r) & 1u)) continue;
        f3 J0, J1, J2, J3;
        joint_rowJ(type, r, rAx, rBx, wp, wq, wax, wbx, J0, J1, J2, J3);
        const float em = eff_mass(J0, J1, J2, J3, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
        const float relvel = rel_speed(J0, J1, J2, J3, A.v, A.w, B.v, B.w);
        const size_t s = (size_t)r * j.cap + i;
        j.eff[s] = em;
        j.rhs[s] = -(err[r] * 0.2f + relvel * (1 + rest[r]));
        j.lo[s] = lo[r]; j.hi[s] = hi[r];
    }
}
// generic_constraint.cpp:10-258. One lane per joint; degree of freedom d = 0..2 linear along frame[0]'s columns (rotated by A),
// 3..5 angular (twist about the x axes via shortest_arc; the other two from the angle between B's x axis and A's z / y axis);
// rows of d in slots 4 d + {0 limit, 1 bump stop, 2 spring, 3 friction/damping}. The Jacobian vectors of each degree of
// freedom are kept for the solve kernel (Joints::gJ).
DI void generic_rowJ(const Joints &j, uint32_t i, int d, f3 &J0, f3 &J1, f3 &J2, f3 &J3) {
    const f3 v0 = from4(j.gJ[(size_t)(3 * d) * j.cap + i]), v1 = from4(j.gJ[(size_t)(3 * d + 1) * j.cap + i]);
    if (d < 3) { J0 = v0; J1 = v1; J2 = -v0; J3 = -from4(j.gJ[(size_t)(3 * d + 2) * j.cap + i]); }
    else { J0 = mk3(0, 0, 0); J1 = v0; J2 = mk3(0, 0, 0); J3 = -v1; }
}
__global__ void k_prep_generic(Joints j, Bodies b, float dt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n || j.type[i] != EDYNHIP_JOINT_GENERIC) return;
    const uint32_t ia = j.bodyA[i], ib = j.bodyB[i];
    if (edge_asleep(b.flags[ia], b.flags[ib])) return;
    const BRef A = load_bref(b, ia), B = load_bref(b, ib);
    const f3 pA = to_world(from4(j.pivA[i]), A.org, A.orn), pB = to_world(from4(j.pivB[i]), B.org, B.orn);
    const f3 rA = pA - A.pos, rB = pB - B.pos;
    const f3 pivot_offset = pB - pA;
    const f3 colA[3] = {from4(j.axA[i]), from4(j.pA[i]), from4(j.qA[i])};
    const f3 axisA_x = rotate(A.orn, colA[0]), axisB_x = rotate(B.orn, from4(j.axB[i]));
    uint32_t mask = 0;
    for (int d = 0; d < 6; ++d) {
        auto P = [&](int k) { return j.params[(size_t)(10 * d + k) * j.cap + i]; };
        const bool limit_enabled = P(0) != 0, angular = d >= 3;
        const float vmin = P(1), vmax = P(2), limit_restitution = P(3), bump_len = P(4), bump_stiffness = P(5), friction = P(6),
                    rest = P(7), spring_stiffness = P(8), damping = P(9);
        const bool non_zero_limit = vmin < vmax;
        f3 J0, J1, J2, J3, axA = mk3(0, 0, 0), axB = mk3(0, 0, 0);
        float current;
        if (!angular) {
            const f3 axisA = rotate(A.orn, d == 0 ? colA[0] : (d == 1 ? colA[1] : colA[2]));
            J0 = axisA; J1 = cross(rA, axisA); J2 = -axisA; J3 = -cross(rB, axisA);
            current = dot(pivot_offset, axisA);
            j.gJ[(size_t)(3 * d) * j.cap + i] = to4(axisA, 0); j.gJ[(size_t)(3 * d + 1) * j.cap + i] = to4(J1, 0);
            j.gJ[(size_t)(3 * d + 2) * j.cap + i] = to4(cross(rB, axisA), 0);
        } else {
            const int k = d - 3;
            if (k == 0) {
                current = cvjoint_relative_angle(A.orn, B.orn, axisA_x, axisB_x, colA[1], colA[2], from4(j.pB[i]));
                axA = axisA_x; axB = axisB_x;
            } else {
                const f3 other = rotate(A.orn, k == 1 ? colA[2] : colA[1]);
                const float cos_angle = fminf(fmaxf(dot(axisB_x, other), -1.0f), 1.0f);
                current = kPi * 0.5f - (float)acos((double)cos_angle);   // correctly rounded, like atan2_cr
                f3 axis = cross(other, axisB_x);
                if (!try_normalize(axis)) axis = k == 1 ? mk3(0, 0, 1) : mk3(0, 1, 0);
                axA = axB = -axis;
            }
            J0 = mk3(0, 0, 0); J1 = axA; J2 = mk3(0, 0, 0); J3 = -axB;
            j.gJ[(size_t)(3 * d) * j.cap + i] = to4(axA, 0); j.gJ[(size_t)(3 * d + 1) * j.cap + i] = to4(axB, 0);
        }
        const float em = eff_mass(J0, J1, J2, J3, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
        const float relvel = rel_speed(J0, J1, J2, J3, A.v, A.w, B.v, B.w);
        auto put = [&](int kind, float error, float erp, float restitution, float lo, float hi) {
            const size_t s = (size_t)(4 * d + kind) * j.cap + i;
            j.eff[s] = em; j.rhs[s] = -(error * erp + relvel * (1 + restitution)); j.lo[s] = lo; j.hi[s] = hi;
            mask |= 1u << (4 * d + kind);
        };
        if (limit_enabled) {
            float error = 0, erp = 0.2f, restitution = 0, lo = -kLarge, hi = kLarge;
            if (non_zero_limit) {
                float limit_error;
                const float mid = (vmin + vmax) / 2.0f;
                if (current < mid) { limit_error = vmin - current; lo = -kLarge; hi = 0; }
                else { limit_error = vmax - current; lo = 0; hi = kLarge; }
                if (angular) error = limit_error / dt;
                else { if (current > vmin && current < vmax) error = limit_error / dt; erp = 0.9f; }
                restitution = limit_restitution;
            } else if (angular) {
                error = -current / dt;
            }
            put(0, error, erp, restitution, lo, hi);
        }
        if (limit_enabled && non_zero_limit && bump_stiffness > 0 && bump_len > 0) {
            float defl = 0;
            const float bmin = vmin + bump_len, bmax = vmax - bump_len;
            if (current < bmin) defl = current - bmin;
            else if (current > bmax) defl = current - bmax;
            const float imp = bump_stiffness * defl * dt;
            put(1, -defl / dt, 0.2f, 0.0f, fminf(imp, 0.0f), fmaxf(0.0f, imp));
        }
        if (spring_stiffness > 0) {
            const float defl = current - rest;
            const float imp = spring_stiffness * defl * dt;
            put(2, -defl / dt, 0.2f, 0.0f, fminf(imp, 0.0f), fmaxf(0.0f, imp));
        }
        if (friction > 0 || damping > 0) {
            float fi = friction * dt;
            if (damping > 0) {
                const float rel = angular ? dot(A.w, axA) - dot(B.w, axB) : relvel;
                fi += fabsf(rel) * damping * dt;
            }
            put(3, 0.0f, 0.2f, 0.0f, -fi, fi);
        }
    }
    j.rmask[i] = mask;
}
// The same joint solve with everything that is constant during a step's sweeps held in registers (the island-fused fast
// path, k_island_velocity): load once, solve per sweep against the caller's Delta, store the impulses once at the end.
// A lane of that kernel owns either a joint or a manifold, so the joint's registers are a VIEW of the manifold's row
// registers RowReg[4][3] (every index is a compile-time constant after unrolling):
//   slot r -> R[r/3][r%3]: f[0] = (impulse, lo, hi, rhs), f[1].x = eff        R[3][0].f[0..4] = rA, rB, wp, wq, wax
//   R[3][1].f[0] = wbx, f[1..3] = rows of I_A^-1      R[3][2].f[0..2] = rows of I_B^-1, f[3] = (type, mask) as bits
// Not for generic constraints (24 slots). Identical arithmetic to joint_solve_lane.
typedef RowReg JRegs[4][kRowsPerPoint];
DI void jlane_load(JRegs &R, const Joints &j, const Bodies &b, uint32_t i, uint32_t ia, uint32_t ib) {
    const uint32_t mask = j.rmask[i];
    R[3][2].f[3] = make_float4(__int_as_float((int)j.type[i]), __uint_as_float(mask), 0, 0);
    R[3][0].f[0] = j.rA[i]; R[3][0].f[1] = j.rB[i]; R[3][0].f[2] = j.wp[i]; R[3][0].f[3] = j.wq[i]; R[3][0].f[4] = j.wax[i];
    R[3][1].f[0] = j.wbx[i];
#pragma unroll
    for (int r = 0; r < 3; ++r) { R[3][1].f[1 + r] = B_IW(b, ia, r); R[3][2].f[r] = B_IW(b, ib, r); }
#pragma unroll
    for (int r = 0; r < kJointBaseSlots; ++r) {
        const size_t s = (size_t)r * j.cap + i;
        const bool on = (mask >> r) & 1u;
        R[r / 3][r % 3].f[0] = on ? make_float4(j.impulse[s], j.lo[s], j.hi[s], j.rhs[s]) : make_float4(0, 0, 0, 0);
        R[r / 3][r % 3].f[1].x = on ? j.eff[s] : 0.0f;
    }
}
template <bool WARM>
DI void jlane_solve(JRegs &R, Delta &d) {
    const int type = __float_as_int(R[3][2].f[3].x);
    const uint32_t mask = __float_as_uint(R[3][2].f[3].y);
    const f3 rA = from4(R[3][0].f[0]), rB = from4(R[3][0].f[1]), wp = from4(R[3][0].f[2]), wq = from4(R[3][0].f[3]), wax = from4(R[3][0].f[4]), wbx = from4(R[3][1].f[0]);
    d.iA = {from4(R[3][1].f[1]), from4(R[3][1].f[2]), from4(R[3][1].f[3])};
    d.iB = {from4(R[3][2].f[0]), from4(R[3][2].f[1]), from4(R[3][2].f[2])};
#pragma unroll
    for (int r = 0; r < kJointBaseSlots; ++r) {
        if (!((mask >> r) & 1u)) continue;
        float4 &q = R[r / 3][r % 3].f[0];   // (impulse, lo, hi, rhs)
        f3 J0, J1, J2, J3;
        joint_rowJ(type, r, rA, rB, wp, wq, wax, wbx, J0, J1, J2, J3);
        if (WARM) {
            apply_impulse(d, J0, J1, J2, J3, q.x);
        } else {
            const float imp = q.x;
            float drel = rel_speed(J0, J1, J2, J3, d.dvA, d.dwA, d.dvB, d.dwB);
            float dimp = (q.w - drel) * R[r / 3][r % 3].f[1].x;
            float ni = imp + dimp;
            if (ni < q.y) { dimp = q.y - imp; ni = q.y; }
            else if (ni > q.z) { dimp = q.z - imp; ni = q.z; }
            q.x = ni;
            apply_impulse(d, J0, J1, J2, J3, dimp);
        }
    }
}
DI void jlane_store(const JRegs &R, const Joints &j, uint32_t i) {
    const uint32_t mask = __float_as_uint(R[3][2].f[3].y);
#pragma unroll
    for (int r = 0; r < kJointBaseSlots; ++r)
        if ((mask >> r) & 1u) j.impulse[(size_t)r * j.cap + i] = R[r / 3][r % 3].f[0].x;
}
template <bool WARM>
DI void joint_solve_lane(uint32_t i, const Joints &j, const Bodies &b) {
    const uint32_t ia = j.bodyA[i], ib = j.bodyB[i];
    if (edge_asleep(b.flags[ia], b.flags[ib])) return;
    Delta d;
    load_delta(b, ia, ib, d);
    const f3 rA = from4(j.rA[i]), rB = from4(j.rB[i]), wp = from4(j.wp[i]), wq = from4(j.wq[i]), wax = from4(j.wax[i]), wbx = from4(j.wbx[i]);
    const int type = j.type[i];
    const uint32_t mask = j.rmask[i];
    auto row = [&](int r, f3 J0, f3 J1, f3 J2, f3 J3) {
        const size_t s = (size_t)r * j.cap + i;
        float imp = j.impulse[s];
        if (WARM) {
            apply_impulse(d, J0, J1, J2, J3, imp);
        } else {
            const float lo = j.lo[s], hi = j.hi[s];
            float drel = rel_speed(J0, J1, J2, J3, d.dvA, d.dwA, d.dvB, d.dwB);
            float dimp = (j.rhs[s] - drel) * j.eff[s];
            float ni = imp + dimp;
            if (ni < lo) { dimp = lo - imp; ni = lo; }
            else if (ni > hi) { dimp = hi - imp; ni = hi; }
            j.impulse[s] = ni;
            apply_impulse(d, J0, J1, J2, J3, dimp);
        }
    };
    if (type == EDYNHIP_JOINT_GENERIC) {   // up to 24 rows: visit the set slots in ascending order
        uint32_t todo = mask;
        while (todo) {
            const int r = __ffs((int)todo) - 1;
            todo &= todo - 1u;
            f3 J0, J1, J2, J3;
            generic_rowJ(j, i, r >> 2, J0, J1, J2, J3);
            row(r, J0, J1, J2, J3);
        }
    } else {
#pragma unroll
        for (int r = 0; r < kJointBaseSlots; ++r) {   // unrolled: the loads of all rows issue together
            if (!((mask >> r) & 1u)) continue;
            f3 J0, J1, J2, J3;
            joint_rowJ(type, r, rA, rB, wp, wq, wax, wbx, J0, J1, J2, J3);
            row(r, J0, J1, J2, J3);
        }
    }
    store_delta(b, ia, ib, d);
}
template <bool WARM>
__global__ void k_joint_solve(uint32_t start, uint32_t end, Joints j, Bodies b) {
    const uint32_t i = start + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < end) joint_solve_lane<WARM>(i, j, b);
}
// hinge_constraint::reset_angle (hinge_constraint.cpp:19-24) for the joints whose definition was just (re)written
__global__ void k_joint_reset_angle(Joints j, Bodies b, const uint8_t *__restrict__ which) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n || !which[i] || (j.type[i] != EDYNHIP_JOINT_HINGE && j.type[i] != EDYNHIP_JOINT_CVJOINT)) return;
    const q4 ornA = q_from4(B_ORN(b, j.bodyA[i])), ornB = q_from4(B_ORN(b, j.bodyB[i]));
    if (j.type[i] == EDYNHIP_JOINT_CVJOINT) {   // cvjoint_constraint::reset_angle, cvjoint_constraint.cpp:12-23
        j.angle[i] = cvjoint_relative_angle(ornA, ornB, rotate(ornA, from4(j.axA[i])), rotate(ornB, from4(j.axB[i])),
                                            from4(j.pA[i]), from4(j.qA[i]), from4(j.pB[i]));
        return;
    }
    const f3 p = rotate(ornA, from4(j.pA[i])), q = rotate(ornA, from4(j.qA[i]));
    const f3 angle_axisB = rotate(ornB, from4(j.pB[i]));
    j.angle[i] = atan2_cr(dot(angle_axisB, q), dot(angle_axisB, p));
}
int joint_reset_angles(edynhip_ctx *c, const uint8_t *which_dev) {
    if (c->j.n == 0) return EDYNHIP_OK;
    hipLaunchKernelGGL(k_joint_reset_angle, dim3(blocks(c->j.n, 128)), dim3(128), 0, c->stream, c->j, c->b, which_dev);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

// ------------------------------------------------------------------ integration
__global__ void k_integrate(uint32_t n, Bodies b, float dt, float *isl_err, uint32_t *isl_done, const float4 *__restrict__ dslot,
                            const uint32_t *__restrict__ first_slot, float *pos_err, uint32_t pos_iters) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    isl_err[i] = 0; isl_done[i] = 0;   // per-island position-solver state (indexed by island label = a body index)
    for (uint32_t it = 0; it < pos_iters; ++it) pos_err[(size_t)it * b.cap + i] = 0;
    if (!is_dynamic(b.flags[i]) || (b.flags[i] & BF_ASLEEP)) return;   // sleeping islands are not solved or integrated (solver.cpp:408)
    float4 p4 = B_POS(b, i);
    f3 v = from4(b.linvel[i]), w = from4(b.angvel[i]);
    f3 dv, dw;
    if (dslot) {   // push hand-off: a sweep leaves each body's deltas in the slots of its first manifold
        const uint32_t fs = first_slot[i];
        // (no slot: no contact touched it - zero deltas - or, mixed schedule, its island was solved on the body records)
        dv = fs != 0xFFFFFFFFu ? from4(dslot[dslot_at(fs, 0)]) : from4(B_DV(b, i));
        dw = fs != 0xFFFFFFFFu ? from4(dslot[dslot_at(fs, 1)]) : from4(B_DW(b, i));
    } else { dv = from4(B_DV(b, i)); dw = from4(B_DW(b, i)); }
    v += dv;
    w += dw;
    f3 pos = from4(p4);
    pos += v * dt;
    q4 orn = integrate(q_from4(B_ORN(b, i)), w, dt);
    b.linvel[i] = to4(v, 0); b.angvel[i] = to4(w, 0);
    B_POS(b, i) = to4(pos, p4.w); B_ORN(b, i) = to4(orn);
}

// ------------------------------------------------------------------ position solver
struct PBody { f3 pos; q4 orn; float inv_m; m3 iw, il; bool proc; f3 org, com; bool has_com; };   // org: the frame of the pivots (position_solver.hpp:53-59)
DI PBody load_pbody(const Bodies &b, uint32_t i) {
    PBody r;
    float4 p = B_POS(b, i);
    r.pos = from4(p); r.orn = q_from4(B_ORN(b, i));
    r.has_com = b.origin && b.com[i].w != 0.0f;
    r.com = r.has_com ? from4(b.com[i]) : mk3(0, 0, 0);
    r.org = r.has_com ? from4(b.origin[i]) : r.pos;   // (the stored origin: after the integration it is the previous one until a correction or the end of the step refreshes it)
    r.proc = is_dynamic(b.flags[i]);
    if (r.proc) {
        r.inv_m = p.w;
        r.iw = {from4(B_IW(b, i, 0)), from4(B_IW(b, i, 1)), from4(B_IW(b, i, 2))};
        r.il = {from4(B_IL(b, i, 0)), from4(B_IL(b, i, 1)), from4(B_IL(b, i, 2))};
    } else { r.inv_m = 0; r.iw = m3_zero(); r.il = m3_zero(); }
    return r;
}
DI void store_pbody(const Bodies &b, uint32_t i, const PBody &r) {
    if (r.has_com) b.origin[i] = to4(r.org, 0);
    if (!r.proc) return;
    B_POS(b, i) = to4(r.pos, r.inv_m); B_ORN(b, i) = to4(r.orn);
    B_IW(b, i, 0) = to4(r.iw.r0, 0); B_IW(b, i, 1) = to4(r.iw.r1, 0); B_IW(b, i, 2) = to4(r.iw.r2, 0);
}
DI void pos_apply(PBody &x, f3 Jl, f3 Ja, float corr) {
    if (!x.proc) return;
    x.pos += x.inv_m * Jl * corr;
    f3 ang = mul(x.iw, Ja) * corr;
    x.orn = x.orn + quaternion_derivative(x.orn, ang);
    x.orn = normalize(x.orn);
    m3 basis = to_m3(x.orn);
    x.iw = mul(mul(basis, x.il), transpose(basis));
}
DI float xchg1(float v) {   // value held by the partner lane (lane ^ 1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
}
DI f3 xchg1(f3 v) { return {xchg1(v.x), xchg1(v.y), xchg1(v.z)}; }
// ---- position correction of one contact point, one body per lane: the coloured order's own arithmetic ("fused", round 4) -----------
// The position solve walks the same dependency chains as the velocity solve with 2.2x the work per task - per point a quaternion
// rotation of the pivot and the normal, the product I_w Ja through a freshly rebuilt world inertia (two 3x3 products), five correctly
// rounded divisions. Like the velocity rows it therefore has its own arithmetic in the coloured order (specified by the checker's
// coloured order, reproduced here bit for bit, within SURVEY 8(d)'s tolerances of the reference's): the same correction
// (contact_constraint.cpp:58-90, position_solver.hpp:16-51) written with
//   * R = the rotation matrix of the (unit) orientation, built without the renormalising division, used for the pivot, the normal
//     and the inertia product I_w Ja = R (I_l (R^T Ja)) - three matrix-vector products instead of two matrix-matrix ones;
//   * every dot product as an fma chain; 1 / ((t1 + t2) + (o1 + o2)) for the effective mass;
//   * the re-normalisation of the orientation as ONE division, q * (1 / |q|).
// The body record's world inertia is not read by contacts any more; it is rebuilt (reference arithmetic) from the final orientation
// by whoever stores the body, for the joints' position solve and the next step.
DI m3 basis_unit(q4 q) {
    const float xs = q.x + q.x, ys = q.y + q.y, zs = q.z + q.z;
    const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const float xx = q.x * xs, yy = q.y * ys, zz = q.z * zs;
    return {{1.0f - (yy + zz), __builtin_fmaf(q.x, ys, -wz), __builtin_fmaf(q.x, zs, wy)},
            {__builtin_fmaf(q.x, ys, wz), 1.0f - (xx + zz), __builtin_fmaf(q.y, zs, -wx)},
            {__builtin_fmaf(q.x, zs, -wy), __builtin_fmaf(q.y, zs, wx), 1.0f - (xx + yy)}};
}
DI f3 mv_fma(const m3 &m, f3 v) { return mk3(dot3_fma(m.r0, v), dot3_fma(m.r1, v), dot3_fma(m.r2, v)); }
DI f3 mtv_fma(const m3 &m, f3 v) {   // transpose(m) * v
    return mk3(__builtin_fmaf(m.r2.x, v.z, __builtin_fmaf(m.r1.x, v.y, m.r0.x * v.x)), __builtin_fmaf(m.r2.y, v.z, __builtin_fmaf(m.r1.y, v.y, m.r0.y * v.x)),
               __builtin_fmaf(m.r2.z, v.z, __builtin_fmaf(m.r1.z, v.y, m.r0.z * v.x)));
}
DI f3 cross_fma(f3 a, f3 b) { return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))); }
// The unit of the coloured order's position solve is the MANIFOLD (block correction, round 4; specified by the checker's
// coloured order, contact_solve_position_block): the corrections of its <= 4 points are evaluated from the transforms the manifold was entered with,
// each body's translation / rotation vector is the sum, in list order, of the rounded products (inv_m Jl) corr_i / (I_w Ja_i) corr_i,
// and the body is moved once - one rotation matrix, one quaternion update, one square root and one division per body and manifold
// instead of one per point (a four-point task: ~1 200 -> ~520 instructions between "transforms arrived" and "transforms handed on").
// This lane's body is X (sideB: X is body[1]); piv[k] = (pivot of X, .w: distance on side A), l4 = local normal, n4 = (normal, attachment);
// live[k]: point k takes part (uniform within the lane pair). Both lanes of the pair execute it together (DPP exchanges).
// Returns true when X was corrected.
template <int NP>
DI bool pos_manifold_block(PBody &X, bool sideB, float4 (&piv)[NP], const float4 (&l4)[NP], float4 (&n4)[NP], const bool (&live)[NP], float &max_err) {
    const m3 R = basis_unit(X.orn);
    f3 t = mk3(0, 0, 0), rot = mk3(0, 0, 0);
    bool any = false;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (!live[k]) continue;
        const int attach = __float_as_int(n4[k].w);
        const f3 pXw = mv_fma(R, from4(piv[k])) + X.org;
        const f3 pOw = xchg1(pXw);
        const f3 pAw = sideB ? pOw : pXw, pBw = sideB ? pXw : pOw;
        // the normal rotates with the body it is attached to; that body's lane computes it and shares it
        const f3 nrot = mv_fma(R, from4(l4[k]));
        const f3 nother = xchg1(nrot);
        f3 n = from4(n4[k]);
        if (attach == dc::NA_ON_A) n = sideB ? nother : nrot;
        else if (attach == dc::NA_ON_B) n = sideB ? nrot : nother;
        const float distance = dot3_fma(pAw - pBw, n);
        n4[k] = to4(n, n4[k].w);
        piv[k].w = distance;   // meaningful on side A only (pA.w = distance, pB.w = friction)
        if (distance > -kEps) continue;   // (the same bits in both lanes of the pair)
        // J = {n, rA x n, -n, -(rB x n)}
        const f3 rX = pXw - X.pos;
        const f3 Jl = sideB ? -n : n;
        const f3 cx = cross_fma(rX, n);
        const f3 Ja = sideB ? -cx : cx;
        const f3 w = mv_fma(R, mv_fma(X.il, mtv_fma(R, Ja)));   // I_w Ja
        const float mine = dot3_fma(Jl, Jl) * X.inv_m + dot3_fma(w, Ja);
        const float em = 1.0f / (mine + xchg1(mine));
        const float corr = (-distance * 0.2f) * em;
        max_err = fmaxf(fabsf(distance), max_err);
        t = t + (X.inv_m * Jl) * corr;
        rot = rot + w * corr;
        any = true;
    }
    if (!any || !X.proc) return false;
    X.pos = X.pos + t;
    const q4 q = X.orn + quaternion_derivative(X.orn, rot);
    const float l2 = __builtin_fmaf(q.w, q.w, __builtin_fmaf(q.z, q.z, __builtin_fmaf(q.y, q.y, q.x * q.x)));
    const float rl = 1.0f / sqrtf(l2);
    X.orn = q4{q.x * rl, q.y * rl, q.z * rl, q.w * rl};
    if (X.has_com) X.org = to_world(-X.com, X.pos, X.orn); else X.org = X.pos;   // position_solver.hpp:34-41
    return true;
}
DI void pos_rebuild_inertia(PBody &X) { const m3 basis = to_m3(X.orn); X.iw = mul(mul(basis, X.il), transpose(basis)); }   // update_inertia, reference arithmetic
// position_solver.hpp:34-41: after a correction the origins follow the new transforms (a body without an offset has none: its pivots use pos)
DI void pos_origin(PBody &x) { if (x.has_com) x.org = to_world(-x.com, x.pos, x.orn); else x.org = x.pos; }
// The contacts of a manifold corrected the reference's way (the default; ctx.hpp Arith): contact_constraint::solve_position
// (contact_constraint.cpp:58-90) point after point in list order, each through position_solver::solve (position_solver.hpp:16-51) - a
// point sees the transforms the previous point left, every correction re-normalises the orientation and is followed by a rebuilt
// world inertia. One body per lane (X; sideB: X is body[1]), the partner's terms cross over with DPP: every quantity is computed by
// exactly the reference's fp32 operations in the reference's order. The world inertia follows the orientation lazily: `iw_stale` says
// that X.iw has to be rebuilt (same operations, same value as the reference's rebuild right after the correction) before it is read.
// live[k]: point k takes part (uniform within the lane pair). Returns true when X was corrected.
template <int NP>
DI bool pos_manifold_points(PBody &X, bool sideB, float4 (&piv)[NP], const float4 (&l4)[NP], float4 (&n4)[NP], const bool (&live)[NP], float &max_err, bool &iw_stale) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (!live[k]) continue;
        const int attach = __float_as_int(n4[k].w);
        const f3 pXw = to_world(from4(piv[k]), X.org, X.orn);
        const f3 pOw = xchg1(pXw);
        const f3 pAw = sideB ? pOw : pXw, pBw = sideB ? pXw : pOw;
        // the normal rotates with the body it is attached to; that body's lane computes it and shares it
        const f3 nrot = rotate(X.orn, from4(l4[k]));
        const f3 nother = xchg1(nrot);
        f3 n = from4(n4[k]);
        if (attach == dc::NA_ON_A) n = sideB ? nother : nrot;
        else if (attach == dc::NA_ON_B) n = sideB ? nrot : nother;
        const float distance = dot(pAw - pBw, n);
        const f3 rX = pXw - X.pos;
        n4[k] = to4(n, n4[k].w);
        piv[k].w = distance;   // meaningful on side A only (pA.w = distance, pB.w = friction)
        if (distance > -kEps) continue;   // (the same bits in both lanes of the pair)
        if (iw_stale) { const m3 basis = to_m3(X.orn); X.iw = mul(mul(basis, X.il), transpose(basis)); iw_stale = false; }
        // J = {n, rA x n, -n, -(rB x n)}
        const f3 Jl = sideB ? -n : n;
        const f3 cx = cross(rX, n);
        const f3 Ja = sideB ? -cx : cx;
        const f3 iwJa = mul(X.iw, Ja);
        const float t1 = dot(Jl, Jl) * X.inv_m, t2 = dot(iwJa, Ja);
        const float o1 = xchg1(t1), o2 = xchg1(t2);
        const float a1 = sideB ? o1 : t1, a2 = sideB ? o2 : t2, b1 = sideB ? t1 : o1, b2 = sideB ? t2 : o2;
        const float em = 1.0f / (a1 + a2 + b1 + b2);   // get_effective_mass's order
        const float error = -distance;
        const float corr = error * 0.2f * em;           // contact_position_correction_rate
        max_err = fmaxf(fabsf(error), max_err);
        if (!X.proc) continue;
        X.pos += X.inv_m * Jl * corr;
        X.orn = normalize(X.orn + quaternion_derivative(X.orn, iwJa * corr));
        iw_stale = true;
        if (X.has_com) X.org = to_world(-X.com, X.pos, X.orn); else X.org = X.pos;   // position_solver.hpp:34-41
        any = true;
    }
    return any;
}
DI void pos_solve(PBody &A, PBody &B, f3 J0, f3 J1, f3 J2, f3 J3, float error, float &max_err) {
    float em = eff_mass(J0, J1, J2, J3, A.inv_m, A.iw, B.inv_m, B.iw);
    float corr = error * 0.2f * em;   // contact_position_correction_rate / error_correction_rate
    pos_apply(A, J0, J1, corr);
    pos_apply(B, J2, J3, corr);
    pos_origin(A); pos_origin(B);
    max_err = fmaxf(fabsf(error), max_err);
}
// One atomic per wave when all its active lanes belong to one island (the common case: a pile is one
// island). Every lane of the wave must call this (inactive lanes pass active = false).
DI void publish_error(bool active, float max_err, uint32_t label, float *isl_err) {
    uint32_t rep = active ? label : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rep = min(rep, (uint32_t)__shfl_xor((int)rep, off));
    float m = active ? max_err : 0.0f;
    if (__all(!active || label == rep)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0 && m > 0 && rep != 0xFFFFFFFFu) atomicMax((unsigned int *)&isl_err[rep], __float_as_uint(m));
    } else if (m > 0) {
        atomicMax((unsigned int *)&isl_err[label], __float_as_uint(m));
    }
}

// Position solve of one colour with TWO lanes per manifold: even lane = body A side, odd lane = body B side.
// The kernel is instruction-bound (each correction re-normalises the quaternion and rebuilds I_w = R I_l R^T, as
// position_solver::solve does), and the two bodies' updates are independent, so each lane owns one body. The few
// shared scalars (world pivots, normal, effective-mass terms) are swapped with the partner lane through DPP.
// Every quantity is computed by exactly the same fp32 operations in the same order as the one-lane formulation.

// ---- dataflow velocity sweep, two lanes per manifold (even lane = body A's side, odd lane = body B's side) ----
// The per-hop cost of k_contact_solve_df is "notice the hand-off" + the manifold's arithmetic, and the arithmetic is a
// serial chain on one lane. Here each lane owns ONE body of the manifold: it holds that body's deltas, loads only its
// side's row pieces (J_lin, own J_ang, own I^-1 J_ang: 3 of the 5 float4 per row), computes its two terms of the
// relative speed and applies the impulse to its own body; the partner's two terms and the scalars that live in the
// other side's pieces (rhs / accumulated impulse, mu) cross over with DPP quad_perm[1,0,3,2]. The sums are formed in
// the same order as rel_speed() (((JlA.dvA + JaA.dwA) + JlB.dvB) + JaB.dwB), so results stay bit-identical.
// What a lane keeps per row is prepared while the hand-offs are still in flight, so that the section between "inputs
// arrived" and "deltas published" - the part every later manifold of the two bodies waits for - is as short as
// possible: jl = +-J_lin (body B's linear Jacobian is -J_lin), ijl = inv_mass * jl, ja = own J_ang, ija = own
// I^-1 J_ang, the row's scalars (eff, rhs) and the accumulated impulse in BOTH lanes (each lane computes the same
// impulse; no exchange and no select is needed for it). A relative speed is four terms summed in rel_speed()'s order,
// ((JlA.dvA + JaA.dwA) + JlB.dvB) + JaB.dwB; each lane computes its two terms and reads the pair's four with
// quad_perm broadcasts ([0,0,2,2] = the A lane's value, [1,1,3,3] = the B lane's), which fold into the adds as DPP
// operands. Results are bit-identical to the one-lane formulation (same operations in the same order).
struct Row2 { f3 jl, ijl, ja, ija; float eff, rhs, imp; };
DI float dppA(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0 /* quad_perm:[0,0,2,2] */, 0xF, 0xF, true)); }
DI float dppB(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true)); }
DI void df2_poll(const float4 *slot, v4f &h0, v4f &h1) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:1024 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(h0), "=&v"(h1) : "v"(slot) : "memory");
}
struct Side { f3 dv, dw; };
// (Row2::rhs: the row's rhs in the reference arithmetic, rhs * eff - prepared before the wait - in the fused one)
template <bool FUSED>
DI float df2_relspeed(const Side &x, const Row2 &r) {
    if (FUSED) {
        const float p = dot3_fma(r.jl, x.dv) + dot3_fma(r.ja, x.dw);   // this side's half
        return p + xchg1(p);                                            // (A's half) + (B's half): the same bits in both lanes
    }
    const float lin = dot(r.jl, x.dv), ang = dot(r.ja, x.dw);          // rel_speed()'s order: ((JlA.dvA + JaA.dwA) + JlB.dvB) + JaB.dwB
    return dppA(lin) + dppA(ang) + dppB(lin) + dppB(ang);
}
template <bool FUSED>
DI void df2_apply(Side &x, const Row2 &r, float imp) {
    if (FUSED) { x.dv = fma3(r.ijl, imp, x.dv); x.dw = fma3(r.ija, imp, x.dw); }
    else { x.dv += r.ijl * imp; x.dw += r.ija * imp; }
}
template <bool WARM, bool FUSED>
DI void df2_point(Side &x, Row2 (&R)[kRowsPerPoint], float mu) {
    Row2 &rn = R[0];
    if (WARM) {   // warm_start: the normal row, then the friction pair
        df2_apply<FUSED>(x, rn, rn.imp);
        return;
    }
    if (FUSED) {
        const float applied = fused_normal(rn.imp, fused_delta(df2_relspeed<true>(x, rn), rn.eff, rn.rhs), kLarge);
        df2_apply<true>(x, rn, applied);
    } else {
        float dimp = (rn.rhs - df2_relspeed<false>(x, rn)) * rn.eff;
        normal_clamp(rn.imp, dimp, kLarge);
        df2_apply<false>(x, rn, dimp);
    }
    (void)mu;
}
template <bool WARM, bool FUSED>
DI void df2_friction(Side &x, Row2 (&R)[kRowsPerPoint], float mu) {
    Row2 &ra = R[1], &rb = R[2];
    if (WARM) {   // warm_start(constraint_row_friction&)
        df2_apply<FUSED>(x, ra, ra.imp);
        df2_apply<FUSED>(x, rb, rb.imp);
        return;
    }
    const float c0 = ra.imp, c1 = rb.imp;
    if (FUSED) {
        float i0 = c0 + fused_delta(df2_relspeed<true>(x, ra), ra.eff, ra.rhs);
        float i1 = c1 + fused_delta(df2_relspeed<true>(x, rb), rb.eff, rb.rhs);
        fused_circle(i0, i1, mu * R[0].imp);   // mu * current normal impulse
        ra.imp = i0; rb.imp = i1;
        df2_apply<true>(x, ra, i0 - c0);
        df2_apply<true>(x, rb, i1 - c1);
    } else {
        float di0 = (ra.rhs - df2_relspeed<false>(x, ra)) * ra.eff;
        float i0 = c0 + di0;
        float di1 = (rb.rhs - df2_relspeed<false>(x, rb)) * rb.eff;
        float i1 = c1 + di1;
        friction_circle(i0, i1, di0, di1, c0, c1, mu * R[0].imp);   // mu * current normal impulse
        ra.imp = i0; rb.imp = i1;
        df2_apply<false>(x, ra, di0);
        df2_apply<false>(x, rb, di1);
    }
}
// EXACT: every live lane of the wave has exactly NP points (the common case: lanes are grouped by point count), so the
// rows need no per-lane point-count predicate.
template <bool WARM, int NP, bool EXACT, bool FUSED>
DI void df2_task(const DfArgs &a, uint32_t p, bool valid, bool sideB, uint32_t np, uint32_t col, uint32_t sweep, uint64_t *trace_slot) {
    Row2 R[NP][kRowsPerPoint];
    float mu[NP];
    const uint64_t w0 = a.trace ? wall_clock64() : 0;
    const uint32_t slot = 2 * p + (sideB ? 1u : 0u);
    const uint32_t nx = a.next[slot];
    const float im = a.im[slot];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int r = 0; r < kRowsPerPoint; ++r) {
            const size_t base = (size_t)((k * kRowsPerPoint + r) * kRowF) * a.rcap + p;
            const float4 f0 = a.rw[base];                                           // (J_lin, eff)
            const float4 fa = a.rw[base + (size_t)(sideB ? 2 : 1) * a.rcap];        // A: (J_angA, rhs)   B: (J_angB, impulse)
            const float4 fi = a.rw[base + (size_t)(sideB ? 4 : 3) * a.rcap];        // A: (I_A^-1 J_angA, mu)   B: (I_B^-1 J_angB, -)
            Row2 &q = R[k][r];
            const f3 Jl = from4(f0);
            q.jl = sideB ? -Jl : Jl;
            q.ijl = im * q.jl;
            q.ja = from4(fa); q.ija = from4(fi);
            q.eff = f0.w;
            q.rhs = FUSED ? dppA(fa.w) * f0.w : dppA(fa.w); q.imp = dppB(fa.w);
            if (r == 0) mu[k] = dppA(fi.w);
        }
    }
    Side x;
    x.dv = x.dw = mk3(0, 0, 0);
    const uint32_t want = (nx & kHeadBit) ? sweep : sweep + 1;
    bool got = im == 0;   // read-only bodies hand nothing over: their deltas stay zero
    bool done = !valid;
    const float4 *mine = a.dslot + dslot_at(slot, 0);
    uint64_t w1 = 0, w2 = 0;
    for (uint32_t spin = 0;; ++spin) {
        if (!done && !got) {
            v4f h0, h1;
            df2_poll(mine, h0, h1);
            if (__float_as_uint(h0.w) == want && __float_as_uint(h1.w) == want) {
                x.dv = mk3(h0.x, h0.y, h0.z); x.dw = mk3(h1.x, h1.y, h1.z); got = true;
            }
        }
        if (a.trace && w1 == 0) w1 = wall_clock64();
        const uint64_t pending = __ballot(!done);
        if (pending == 0) {
            if (trace_slot && (threadIdx.x & 63) == 0) { trace_slot[0] = w0; trace_slot[1] = w1; trace_slot[2] = w2; trace_slot[3] = wall_clock64(); }
            break;
        }
        const uint32_t minc = __shfl(col, __ffsll((long long)pending) - 1);   // lanes are in colour order
        const bool mine_now = !done && col == minc;                           // both lanes of a pair share p, hence colour
        if (__ballot(mine_now && !got) == 0) {
            if (a.trace && w2 == 0) w2 = wall_clock64();
            // the DPP reads need both lanes of a pair: `mine_now` is uniform within a pair
            if (mine_now) {
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (EXACT || (uint32_t)k < np) df2_point<WARM, FUSED>(x, R[k], mu[k]);
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (EXACT || (uint32_t)k < np) df2_friction<WARM, FUSED>(x, R[k], mu[k]);
                // hand the deltas over first (the next manifold of this body is waiting for them), then store the impulses
                if (im != 0) df_publish(a.dslot + dslot_at(nx & kSlotMask, 0), x.dv, x.dw, sweep + 1);
                if (!WARM && sideB) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if (!EXACT && (uint32_t)k >= np) continue;
#pragma unroll
                        for (int r = 0; r < kRowsPerPoint; ++r)
                            a.rw[(size_t)((k * kRowsPerPoint + r) * kRowF + 2) * a.rcap + p] = to4(R[k][r].ja, R[k][r].imp);
                    }
                }
                done = true;
            }
        } else {
            if (spin > kDfSpinLimit || ((spin & 1023u) == 1023u && __hip_atomic_load(&a.cnt->df_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (spin > kDfSpinLimit) atomicExch(&a.cnt->df_abort, 1u);
                break;
            }
            df_nap(a.nap);   // ~0.1 us between polls by default
        }
    }
}
template <bool WARM, bool FUSED>
DI void df2_dispatch(const DfArgs &a, uint32_t p, bool valid, bool sideB, uint32_t np, uint32_t col, uint32_t sweep, uint64_t *tr) {
    // lanes are grouped by point count: np is uniform over a wave except at a group boundary. The warm-start sweep runs
    // once and keeps the predicated form only (fewer instantiations: the kernel has to stay within the instruction cache).
    // Waves of one-point manifolds (sphere contacts: the majority of a mixed scene) get their own instantiation: the NP = 2 form
    // fetched a second point slot in vain for each of them (round 3: 1.30x the algorithmic traffic on mixed32k) and ran its code.
    if (!WARM && __all(!valid || np == 4u)) df2_task<WARM, 4, true, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
    else if (!WARM && __all(!valid || np == 2u)) df2_task<WARM, 2, true, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
    else if (!WARM && __all(!valid || np == 1u)) df2_task<WARM, 1, true, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
    else if (__any(np > 2u)) df2_task<WARM, 4, false, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
    else df2_task<WARM, 2, false, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
}
template <bool FUSED>
__global__ void __launch_bounds__(64) k_contact_solve_df2(DfArgs a) {
    const uint32_t t = blockIdx.x * 32u + (threadIdx.x >> 1);   // 32 manifolds per wave, two lanes each
    const bool sideB = threadIdx.x & 1u;
    if (a.xcd_lists) {   // (gridDim.x is a multiple of 8: solve())
        __shared__ XcdLists L;
        xcd_lists_build(L, a.cnt, blockIdx.x % kXcds);
        const uint32_t count = L.pre[kCls], step = (gridDim.x / kXcds) * 32u, mine = (blockIdx.x / kXcds) * 32u + (threadIdx.x >> 1);
        for (uint32_t sweep = 0; sweep < a.sweeps; ++sweep)
            for (uint32_t base = 0; base < count; base += step) {
                const uint32_t q = base + mine;
                const uint32_t pt = q < count ? xcd_lists_p(L, q, a.na) : a.na;
                const bool valid = pt < a.na && !(a.skip && a.skip[pt]);
                if (!__any(valid)) continue;
                const uint32_t p = pt < a.na ? pt : a.na - 1;
                const uint32_t key = a.keys_sorted[p];
                const uint32_t np = valid ? 4u - (key & 3u) : 0u, col = key >> 2;
                if (sweep == 0) df2_dispatch<true, FUSED>(a, p, valid, sideB, np, col, sweep, nullptr);
                else df2_dispatch<false, FUSED>(a, p, valid, sideB, np, col, sweep, nullptr);
            }
        return;
    }
    const uint32_t rounds = (a.na + a.stride - 1) / a.stride, nwaves = a.stride >> 5;
    for (uint32_t sweep = 0; sweep < a.sweeps; ++sweep)
        for (uint32_t base = 0, round = 0; base < a.na; base += a.stride, ++round) {
            const uint32_t pt = base + t;
            const bool valid = pt < a.na && !(a.skip && a.skip[pt]);
            if (!__any(valid)) continue;              // whole wave beyond the end (wave-uniform)
            const uint32_t p = valid ? pt : a.na - 1;
            const uint32_t key = a.keys_sorted[p];
            const uint32_t np = valid ? 4u - (key & 3u) : 0u, col = key >> 2;
            uint64_t *tr = a.trace ? a.trace + 4 * ((size_t)(sweep * rounds + round) * nwaves + blockIdx.x) : nullptr;
            if (sweep == 0) df2_dispatch<true, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
            else df2_dispatch<false, FUSED>(a, p, valid, sideB, np, col, sweep, tr);
        }
}

// ---- dataflow velocity sweep, FOUR lanes per manifold ---------------------------------------------------------------
// Measured on the two-lane kernel (profiles/r02_dftrace_pile32k.txt and its ISA): the section between "inputs arrived" and
// "deltas published" is ~630 VALU instructions for a four-point manifold and takes 1.65 us - 5.5 cycles per instruction with one
// wave per SIMD: it is ISSUE-bound (a wave64 instruction occupies its SIMD for 4 cycles however few lanes are live), not
// latency-bound. Every hop of every body's chain pays it, 18 colours x 11 sweeps deep. The work of one row is four 3-vector
// dot products (J_linA.dvA, J_angA.dwA, J_linB.dvB, J_angB.dwB) and four 3-vector updates: here each of FOUR lanes owns one
// of the four 3-vectors of a manifold (role = lane & 3: 0 = A linear, 1 = A angular, 2 = B linear, 3 = B angular): its piece of
// the body deltas, its J and its I^-1 J^T (or inv_mass * J) per row. A row then costs one dot product, a quad reduction through
// DPP broadcasts in rel_speed()'s order ((d0 + d1) + d2) + d3, the scalar impulse update (computed redundantly by the four
// lanes: no exchange) and one 3-vector update - about 40 % of the two-lane kernel's instructions - and the wave holds 16
// manifolds. Same operations in the same order on the same values: bit-identical to the one- and two-lane kernels
// (tests/test_gpu_parity.py::test_per_colour_and_dataflow_schedules_are_bit_identical). The hand-off slots are already laid out
// in 16-byte pieces (side, half) with their own tags: each lane polls and publishes exactly its own piece.
struct Row4 { f3 j, ij; float eff, rhs, imp; };
DI float qb0(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, true)); }
DI float qb1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55 /* quad_perm:[1,1,1,1] */, 0xF, 0xF, true)); }
DI float qb2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xAA /* quad_perm:[2,2,2,2] */, 0xF, 0xF, true)); }
DI float qb3(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF /* quad_perm:[3,3,3,3] */, 0xF, 0xF, true)); }
DI void df4_poll(const float4 *piece, v4f &h) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(h) : "v"(piece) : "memory");
}
DI void df4_publish(float4 *piece, f3 d, uint32_t tag) {
    const v4f v = {d.x, d.y, d.z, __uint_as_float(tag)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(piece), "v"(v) : "memory");
}
DI float xchg2(float v) {   // value held by the lane two away inside the quad (lane ^ 2)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true));
}
template <bool FUSED>
DI float df4_relspeed(const f3 &x, const Row4 &r) {   // roles 0..3 = A linear, A angular, B linear, B angular
    if (FUSED) {   // (lin_A + ang_A) + (lin_B + ang_B)
        const float t = dot3_fma(r.j, x);
        const float side = t + xchg1(t);
        return side + xchg2(side);
    }
    const float t = dot(r.j, x);   // rel_speed()'s order: ((d0 + d1) + d2) + d3
    return qb0(t) + qb1(t) + qb2(t) + qb3(t);
}
template <bool FUSED>
DI void df4_apply(f3 &x, const Row4 &r, float imp) { if (FUSED) x = fma3(r.ij, imp, x); else x += r.ij * imp; }
template <bool WARM, bool FUSED>
DI void df4_normal(f3 &x, Row4 &rn) {   // Row4::rhs: the row's rhs (reference arithmetic) or rhs * eff (fused)
    if (WARM) { df4_apply<FUSED>(x, rn, rn.imp); return; }
    if (FUSED) {
        const float applied = fused_normal(rn.imp, fused_delta(df4_relspeed<true>(x, rn), rn.eff, rn.rhs), kLarge);
        df4_apply<true>(x, rn, applied);
    } else {
        float dimp = (rn.rhs - df4_relspeed<false>(x, rn)) * rn.eff;
        normal_clamp(rn.imp, dimp, kLarge);
        df4_apply<false>(x, rn, dimp);
    }
}
template <bool WARM, bool FUSED>
DI void df4_friction(f3 &x, Row4 (&R)[kRowsPerPoint], float mu) {
    Row4 &ra = R[1], &rb = R[2];
    if (WARM) {   // warm_start(constraint_row_friction&)
        df4_apply<FUSED>(x, ra, ra.imp);
        df4_apply<FUSED>(x, rb, rb.imp);
        return;
    }
    const float c0 = ra.imp, c1 = rb.imp;
    if (FUSED) {
        float i0 = c0 + fused_delta(df4_relspeed<true>(x, ra), ra.eff, ra.rhs);
        float i1 = c1 + fused_delta(df4_relspeed<true>(x, rb), rb.eff, rb.rhs);
        fused_circle(i0, i1, mu * R[0].imp);   // mu * current normal impulse
        ra.imp = i0; rb.imp = i1;
        df4_apply<true>(x, ra, i0 - c0);
        df4_apply<true>(x, rb, i1 - c1);
    } else {
        float di0 = (ra.rhs - df4_relspeed<false>(x, ra)) * ra.eff;
        float i0 = c0 + di0;
        float di1 = (rb.rhs - df4_relspeed<false>(x, rb)) * rb.eff;
        float i1 = c1 + di1;
        friction_circle(i0, i1, di0, di1, c0, c1, mu * R[0].imp);
        ra.imp = i0; rb.imp = i1;
        df4_apply<false>(x, ra, di0);
        df4_apply<false>(x, rb, di1);
    }
}
template <bool WARM, int NP, bool EXACT, bool FUSED>
DI void df4_task(const DfArgs &a, uint32_t p, bool valid, uint32_t role, uint32_t np, uint32_t col, uint32_t sweep, uint64_t *trace_slot) {
    Row4 R[NP][kRowsPerPoint];
    float mu[NP];
    const uint64_t w0 = a.trace ? wall_clock64() : 0;
    const uint32_t side = role >> 1;
    const bool ang = role & 1u;
    const uint32_t slot = 2 * p + side;
    const uint32_t nx = a.next[slot];
    const float im = a.im[slot];
    const size_t off_a = (size_t)(ang ? 1u + side : 0u) * a.rcap, off_i = (size_t)(3u + side) * a.rcap;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int r = 0; r < kRowsPerPoint; ++r) {
            const size_t base = (size_t)((k * kRowsPerPoint + r) * kRowF) * a.rcap + p;
            const float4 fa = a.rw[base + off_a];     // role 0, 2: (J_lin, eff)   1: (J_angA, rhs)   3: (J_angB, impulse)
            float4 fi = fa;
            if (ang) fi = a.rw[base + off_i];         // role 1: (I_A^-1 J_angA, mu)   3: (I_B^-1 J_angB, -)
            Row4 &q = R[k][r];
            const f3 Ja = from4(fa);
            q.j = role == 2u ? -Ja : Ja;              // body B's linear Jacobian is -J_lin
            const f3 lin = im * q.j;
            q.ij = ang ? from4(fi) : lin;
            q.eff = qb0(fa.w); q.rhs = FUSED ? qb1(fa.w) * q.eff : qb1(fa.w); q.imp = qb3(fa.w);   // rhs * eff, as fused_delta takes it
            if (r == 0) mu[k] = qb1(fi.w);
        }
    }
    f3 x = mk3(0, 0, 0);
    const uint32_t want = (nx & kHeadBit) ? sweep : sweep + 1;
    bool got = im == 0;   // read-only bodies hand nothing over: their deltas stay zero
    bool done = !valid;
    const float4 *mine = a.dslot + dslot_at(slot, ang ? 1u : 0u);
    uint64_t w1 = 0, w2 = 0;
    for (uint32_t spin = 0;; ++spin) {
        if (!done && !got) {
            v4f h;
            df4_poll(mine, h);
            if (__float_as_uint(h.w) == want) { x = mk3(h.x, h.y, h.z); got = true; }
        }
        if (a.trace && w1 == 0) w1 = wall_clock64();
        const uint64_t pending = __ballot(!done);
        if (pending == 0) {
            if (trace_slot && (threadIdx.x & 63) == 0) { trace_slot[0] = w0; trace_slot[1] = w1; trace_slot[2] = w2; trace_slot[3] = wall_clock64(); }
            break;
        }
        const uint32_t minc = __shfl(col, __ffsll((long long)pending) - 1);   // lanes are in colour order
        const bool mine_now = !done && col == minc;                           // the four lanes of a manifold share p, hence colour
        if (__ballot(mine_now && !got) == 0) {
            if (a.trace && w2 == 0) w2 = wall_clock64();
            // the DPP reads need all four lanes of a manifold: `mine_now` is uniform within a quad
            if (mine_now) {
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (EXACT || (uint32_t)k < np) df4_normal<WARM, FUSED>(x, R[k][0]);
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (EXACT || (uint32_t)k < np) df4_friction<WARM, FUSED>(x, R[k], mu[k]);
                // hand the deltas over first (the next manifold of this body is waiting for them), then store the impulses
                if (im != 0) df4_publish(a.dslot + dslot_at(nx & kSlotMask, ang ? 1u : 0u), x, sweep + 1);
                if (!WARM && role == 3u) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if (!EXACT && (uint32_t)k >= np) continue;
#pragma unroll
                        for (int r = 0; r < kRowsPerPoint; ++r)
                            a.rw[(size_t)((k * kRowsPerPoint + r) * kRowF + 2) * a.rcap + p] = to4(R[k][r].j, R[k][r].imp);
                    }
                }
                done = true;
            }
        } else {
            if (spin > kDfSpinLimit || ((spin & 1023u) == 1023u && __hip_atomic_load(&a.cnt->df_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (spin > kDfSpinLimit) atomicExch(&a.cnt->df_abort, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}
template <bool WARM, bool FUSED>
DI void df4_dispatch(const DfArgs &a, uint32_t p, bool valid, uint32_t role, uint32_t np, uint32_t col, uint32_t sweep, uint64_t *tr) {
    if (!WARM && __all(!valid || np == 4u)) df4_task<WARM, 4, true, FUSED>(a, p, valid, role, np, col, sweep, tr);
    else if (!WARM && __all(!valid || np == 2u)) df4_task<WARM, 2, true, FUSED>(a, p, valid, role, np, col, sweep, tr);
    else if (__any(np > 2u)) df4_task<WARM, 4, false, FUSED>(a, p, valid, role, np, col, sweep, tr);
    else df4_task<WARM, 2, false, FUSED>(a, p, valid, role, np, col, sweep, tr);
}
template <bool FUSED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_contact_solve_df4(DfArgs a) {
    const uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 2);   // 16 manifolds per wave, four lanes each
    const uint32_t role = threadIdx.x & 3u;
    const uint32_t rounds = (a.na + a.stride - 1) / a.stride, nwaves = a.stride >> 4;
    for (uint32_t sweep = 0; sweep < a.sweeps; ++sweep)
        for (uint32_t base = 0, round = 0; base < a.na; base += a.stride, ++round) {
            const uint32_t pt = base + t;
            const bool valid = pt < a.na && !(a.skip && a.skip[pt]);
            if (!__any(valid)) continue;              // whole wave beyond the end (wave-uniform)
            const uint32_t p = valid ? pt : a.na - 1;
            const uint32_t key = a.keys_sorted[p];
            const uint32_t np = valid ? 4u - (key & 3u) : 0u, col = key >> 2;
            uint64_t *tr = a.trace ? a.trace + 4 * ((size_t)(sweep * rounds + round) * nwaves + blockIdx.x) : nullptr;
            if (sweep == 0) df4_dispatch<true, FUSED>(a, p, valid, role, np, col, sweep, tr);
            else df4_dispatch<false, FUSED>(a, p, valid, role, np, col, sweep, tr);
        }
}
template <int NP, bool BLOCK>
DI void pos_contacts_np(bool in_range, uint32_t pc, bool sideB, uint32_t np, const Rows &rows, const Manifolds &mf, const Bodies &b,
                        float *isl_err, const uint32_t *isl_done, uint32_t m, uint32_t ia, uint32_t ib, uint32_t label) {
    const uint32_t ix = sideB ? ib : ia;
    const float4 *__restrict__ pvsrc = sideB ? mf.pB : mf.pA;
    float4 piv[NP], l4[NP], n4[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {   // unconditional: slots >= np hold finite stale data that is never used or stored
        const size_t s = pt_at(mf.cap, (uint32_t)k, m);
        piv[k] = pvsrc[s]; l4[k] = mf.lnrm[s]; n4[k] = mf.nrm[s];
    }
    PBody X = load_pbody(b, ix);
    // the island's early-out flag is a third-level dependent load (p -> label -> flag): it is consumed only by the
    // stores, so the arithmetic does not wait for it (a finished island merely computes into registers it drops)
    const uint32_t done = isl_done[label];
    float max_err = 0;
    bool soft[NP];   // soft contacts take no position correction (contact_extras_constraint.cpp:81-86)
#pragma unroll
    for (int k = 0; k < NP; ++k) soft[k] = mf.xmat != nullptr && mf.xmat[(size_t)k * mf.cap + m].z < kLarge;
    bool live[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) live[k] = (uint32_t)k < np && in_range && !soft[k];   // uniform within a lane pair
    bool iw_stale = false;
    const bool corrected = BLOCK ? pos_manifold_block<NP>(X, sideB, piv, l4, n4, live, max_err) : pos_manifold_points<NP>(X, sideB, piv, l4, n4, live, max_err, iw_stale);
    if (BLOCK ? corrected : iw_stale) pos_rebuild_inertia(X);   // what store_pbody leaves for the joints' position solve and the next step
    const bool active = in_range && done == 0;
    if (active) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const size_t s = pt_at(mf.cap, (uint32_t)k, m);
            if ((uint32_t)k < np && !soft[k]) {
                if (!sideB) mf.pA[s] = piv[k];
                else mf.nrm[s] = n4[k];
            }
        }
        store_pbody(b, ix, X);
    }
    publish_error(active && !sideB, max_err, label, isl_err);
}
// one (manifold, side) lane of the position solve; `pc` must be a valid sorted position even when !in_range
// all lanes stay in the code below (no early exit): DPP exchanges and wave reductions need every lane executing
template <bool BLOCK>
DI void pos_contacts_item(bool in_range, uint32_t pc, bool sideB, const Rows &rows, const Manifolds &mf, const Bodies &b,
                          float *isl_err, const uint32_t *isl_done) {
    const uint32_t m = rows.order[pc];
    const uint32_t ia = rows.bA[pc], ib = rows.bB[pc], np = in_range ? rows.np[pc] : 0;
    const uint32_t label = rows.label[pc];
    // lanes are grouped by point count inside a colour, so a wave takes one branch (except at a group boundary)
    if (__any(np > 2)) pos_contacts_np<4, BLOCK>(in_range, pc, sideB, np, rows, mf, b, isl_err, isl_done, m, ia, ib, label);
    else pos_contacts_np<2, BLOCK>(in_range, pc, sideB, np, rows, mf, b, isl_err, isl_done, m, ia, ib, label);
}
template <bool BLOCK>
DI void pos_contacts_lane(uint32_t start, uint32_t end, uint32_t t, const Rows &rows, const Manifolds &mf, const Bodies &b,
                          float *isl_err, const uint32_t *isl_done) {
    const uint32_t p = start + (t >> 1);
    pos_contacts_item<BLOCK>(p < end, p < end ? p : start, t & 1, rows, mf, b, isl_err, isl_done);
}
template <bool BLOCK>
__global__ void __launch_bounds__(128)
k_pos_contacts(uint32_t start, uint32_t end, Rows rows, Manifolds mf, Bodies b, float *isl_err, const uint32_t *__restrict__ isl_done) {
    pos_contacts_lane<BLOCK>(start, end, blockIdx.x * blockDim.x + threadIdx.x, rows, mf, b, isl_err, isl_done);
}
// Tail colours of the position solve in ONE workgroup (see k_contact_solve_tail); uniform trip counts so that every
// lane reaches the wave-level reductions.
template <bool BLOCK>
__global__ void __launch_bounds__(256)
k_pos_contacts_tail(TailRanges tr, Rows rows, Manifolds mf, Bodies b, float *isl_err, const uint32_t *isl_done) {
    for (uint32_t c = 0; c < tr.n; ++c) {
        for (uint32_t base = tr.start[c]; base < tr.end[c]; base += 128) {
            pos_contacts_lane<BLOCK>(base, tr.end[c], threadIdx.x, rows, mf, b, isl_err, isl_done);
        }
        __threadfence_block();
        __syncthreads();
    }
}
// The serial bucket (ctx.hpp kSerialColour): its manifolds may share bodies, so one lane (velocity) / one lane pair (position)
// takes them one after the other, in sorted order; the fence makes each manifold's stores visible to the next one's loads.
template <bool WARM, bool FUSED>
__global__ void __launch_bounds__(64)
k_contact_solve_serial(uint32_t start, uint32_t end, Split sp, const uint32_t *rbA, const uint32_t *rbB, float4 *rw, uint32_t rcap, float4 *bdvw, float4 *rwx) {
    if (threadIdx.x != 0) return;
    for (uint32_t p = start; p < end; ++p) {
        contact_solve_lane<WARM, false, FUSED>(p, np_of(p, sp), rbA, rbB, rw, rcap, bdvw, nullptr, rwx);
        __threadfence();
    }
}
template <bool BLOCK>
__global__ void __launch_bounds__(64)
k_pos_contacts_serial(uint32_t start, uint32_t end, Rows rows, Manifolds mf, Bodies b, float *isl_err, const uint32_t *isl_done) {
    for (uint32_t p = start; p < end; ++p) {
        pos_contacts_lane<BLOCK>(p, p + 1, threadIdx.x, rows, mf, b, isl_err, isl_done);
        __threadfence();
    }
}
// one joint of the position solve; every lane of the wave calls it (publish_error reduces over the wave), `in_range` says
// whether `i` is a joint of this pass
DI void pos_joints_lane(uint32_t i, bool in_range, const Joints &j, const Bodies &b, float *isl_err, const uint32_t *__restrict__ isl_done) {
    // point, distance and cone constraints have no solve_position (island_solver.cpp:252-260)
    bool active = in_range && (j.type[i] == EDYNHIP_JOINT_HINGE || j.type[i] == EDYNHIP_JOINT_CVJOINT || j.type[i] == EDYNHIP_JOINT_GENERIC);
    if (active && edge_asleep(b.flags[j.bodyA[i]], b.flags[j.bodyB[i]])) active = false;
    uint32_t label = 0;
    float max_err = 0;
    if (active) {
    const uint32_t ia = j.bodyA[i], ib = j.bodyB[i];
    PBody A = load_pbody(b, ia), B = load_pbody(b, ib);
    label = b.island[A.proc ? ia : ib];
    active = isl_done[label] == 0;
    if (active) {
    if (j.type[i] == EDYNHIP_JOINT_GENERIC) {   // generic_constraint.cpp:260-290: the limited linear degrees of freedom
        for (int k = 0; k < 3; ++k) {
            if (j.params[(size_t)(10 * k) * j.cap + i] == 0) continue;
            const f3 pA = to_world(from4(j.pivA[i]), A.org, A.orn), pB = to_world(from4(j.pivB[i]), B.org, B.orn);
            const f3 off = pB - pA, rA = pA - A.pos, rB = pB - B.pos;
            const f3 axisA = rotate(A.orn, from4(k == 0 ? j.axA[i] : (k == 1 ? j.pA[i] : j.qA[i])));
            const float proj = dot(off, axisA), vmin = j.params[(size_t)(10 * k + 1) * j.cap + i], vmax = j.params[(size_t)(10 * k + 2) * j.cap + i];
            float error = 0;
            if (proj < vmin) error = proj - vmin;
            else if (proj > vmax) error = proj - vmax;
            pos_solve(A, B, axisA, cross(rA, axisA), -axisA, -cross(rB, axisA), error, max_err);
        }
        store_pbody(b, ia, A);
        store_pbody(b, ib, B);
    } else {
    if (j.type[i] == EDYNHIP_JOINT_CVJOINT) {   // cvjoint_constraint.cpp:224-246: angular correction along the twist axes
        const f3 tA = rotate(A.orn, from4(j.axA[i])), tB = rotate(B.orn, from4(j.axB[i]));
        const float current = cvjoint_relative_angle(A.orn, B.orn, tA, tB, from4(j.pA[i]), from4(j.qA[i]), from4(j.pB[i]));
        const float twist_min = j.params[i], twist_max = j.params[(size_t)j.cap + i];
        float twist_error = 0;
        if (twist_min < twist_max) {
            const float angle = track_angle(j.angle[i], current);
            j.angle[i] = angle;
            if (angle < twist_min) twist_error = angle - twist_min;
            else if (angle > twist_max) twist_error = angle - twist_max;
        } else {
            twist_error = current;
        }
        pos_solve(A, B, mk3(0, 0, 0), tA, mk3(0, 0, 0), -tB, twist_error, max_err);
    } else {
    const f3 axisA = rotate(A.orn, from4(j.axA[i])), axisB = rotate(B.orn, from4(j.axB[i]));
    f3 pp, qq;
    plane_space(axisA, pp, qq);
    const f3 u = cross(axisA, axisB);
    float e = dot(u, pp);
    if (fabsf(e) > kEps) pos_solve(A, B, mk3(0, 0, 0), pp, mk3(0, 0, 0), -pp, e, max_err);
    e = dot(u, qq);
    if (fabsf(e) > kEps) pos_solve(A, B, mk3(0, 0, 0), qq, mk3(0, 0, 0), -qq, e, max_err);
    }
    const f3 pA = to_world(from4(j.pivA[i]), A.org, A.orn), pB = to_world(from4(j.pivB[i]), B.org, B.orn);
    f3 dir = pA - pB;
    const float err = length(dir);
    if (err > kEps) {
        dir = div_recip(dir, err);
        const f3 rA = pA - A.pos, rB = pB - B.pos;
        pos_solve(A, B, dir, cross(rA, dir), -dir, -cross(rB, dir), -err, max_err);
    }
    store_pbody(b, ia, A);
    store_pbody(b, ib, B);
    }
    }
    }
    publish_error(active, max_err, label, isl_err);
}
__global__ void k_pos_joints(uint32_t start, uint32_t end, Joints j, Bodies b, float *isl_err, const uint32_t *__restrict__ isl_done) {
    const uint32_t i = start + blockIdx.x * blockDim.x + threadIdx.x;
    pos_joints_lane(i, i < end, j, b, isl_err, isl_done);
}
// ---- dataflow position solve: one launch per position iteration (the island early-out between iterations stays a
// separate tiny kernel). Same protocol as k_contact_solve_df with lane = (manifold, side): the hand-off is the body's
// position, orientation and a "corrected in this solve" bit (48 bytes, three tagged 16-byte pieces); a corrected body's
// world inertia is rebuilt from the orientation with the operations of position_solver::solve (bit-identical), an
// uncorrected body still carries the inertia of its record (which active lanes only ever overwrite with equal values).
struct DfPosArgs {
    uint32_t na, stride, iter;      // active manifolds, resident manifolds per round, position iteration (= hand-off sweep)
    const uint32_t *keys_sorted;
    const uint32_t *next;
    float4 *pslot;                  // [3 * (2p + side) + {0,1,2}] = (pos|tag) (orn.xyz|tag) (orn.w, corrected, 0 | tag)
    Rows rows; Manifolds mf; Bodies b;
    float *err_out;                 // [island label] max |error| of THIS iteration (atomicMax on the bits)
    const float *err_prev;          // the previous iteration's, nullptr in the first: an island whose previous iteration stayed
                                    // below the threshold is finished (island_solver.cpp:350-353); a finished island publishes
                                    // no error, so its entry stays 0 and it stays finished - no separate flag pass is needed
    Counters *cnt;
    const uint8_t *skip;            // mixed schedule: manifolds of islands with joints (k_island_position solves those), else nullptr
    uint64_t *trace;                // developer aid (EDYNHIP_DFP_TRACE): 4 timestamps per (round, wave) of this iteration, else nullptr
    uint32_t xcd_lists;             // tasks are handed out per XCD (XcdLists, see the velocity kernel)
};
constexpr float kPosErrorThreshold = 0.005f;
DI void dfp_poll(const float4 *slot, v4f &h0, v4f &h1, v4f &h2) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:2048 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(h0), "=&v"(h1), "=&v"(h2) : "v"(slot) : "memory");
}
DI void dfp_publish(float4 *slot, f3 pos, q4 orn, bool corrected, uint32_t tag) {
    const float t = __uint_as_float(tag);
    const v4f h0 = {pos.x, pos.y, pos.z, t}, h1 = {orn.x, orn.y, orn.z, t}, h2 = {orn.w, corrected ? 1.0f : 0.0f, 0.0f, t};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:1024 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:2048 sc1" : : "v"(slot), "v"(h0), "v"(h1), "v"(h2) : "memory");
}
// Seeds the position hand-off chains from the integrated transforms; the side-A lane also copies its manifold's solved
// impulses back to the contact points (what k_store_impulses does when the position solve runs per colour).
__global__ void k_pos_seed(uint32_t n_active, Rows rows, Bodies b, float4 *pslot, uint32_t rcap, Manifolds mf, const uint8_t *__restrict__ skip) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= 2 * n_active) return;
    const uint32_t p = slot >> 1, body = (slot & 1u) ? rows.bB[p] : rows.bA[p];
    if (!(slot & 1u)) store_impulses_of(p, rows, rcap, mf);
    if (skip && skip[p]) return;   // an island with joints: not on the hand-off chains
    float4 h0 = make_float4(0, 0, 0, 0), h1 = h0, h2 = h0;
    // the chain head starts from the integrated transform; the slot of a read-only body (static, kinematic) holds that body's transform
    // for the whole solve - its lane takes it from there without looking at the tag (no gather through the body index in the solve)
    if (((rows.next[slot] & kHeadBit) && is_dynamic(b.flags[body])) || !is_dynamic(b.flags[body])) {
        const float4 ps = B_POS(b, body), q = B_ORN(b, body);
        h0 = make_float4(ps.x, ps.y, ps.z, 0); h1 = make_float4(q.x, q.y, q.z, 0); h2 = make_float4(q.w, 0, 0, 0);
    }
    pslot[pslot_at(slot, 0)] = h0; pslot[pslot_at(slot, 1)] = h1; pslot[pslot_at(slot, 2)] = h2;
}
DI void pos_writeback(Bodies &b, uint32_t i, const float4 *__restrict__ pslot, const uint32_t *__restrict__ first_slot) {
    const uint32_t fs = first_slot[i];
    if (fs == 0xFFFFFFFFu) return;                       // no contacts: nothing moved it
    const float4 h0 = pslot[pslot_at(fs, 0)], h1 = pslot[pslot_at(fs, 1)], h2 = pslot[pslot_at(fs, 2)];
    if (h2.y == 0.0f) return;                            // never corrected: the record already holds this transform
    PBody X = load_pbody(b, i);
    X.pos = mk3(h0.x, h0.y, h0.z); X.orn = q4{h1.x, h1.y, h1.z, h2.x};
    const m3 basis = to_m3(X.orn);
    X.iw = mul(mul(basis, X.il), transpose(basis));
    store_pbody(b, i, X);
}
__global__ void k_pos_writeback(uint32_t n, Bodies b, const float4 *__restrict__ pslot, const uint32_t *__restrict__ first_slot) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && is_dynamic(b.flags[i])) pos_writeback(b, i, pslot, first_slot);
}
// A position task's inputs, in the two dependent load levels they take: everything whose address follows from the lane's sorted
// position p (key, manifold and body index, island label, next slot, inverse mass, the points' lane-indexed copies Rows::pw, and the
// first look at the own hand-off slot), then what needs one of those (the body's local inverse inertia through the body index, the
// island's error of the previous iteration through its label). Round 3's kernel took four levels (key -> order / bodies / next -> the
// manifold's points and the body record -> poll): its trace showed 3.5 us from "task begins" to "first poll back" against 1.5 us of
// arithmetic and 1.3 us of waiting per task. Measured and dropped (round 4, scripts/runs/c5-c9.sh): requesting the wave's NEXT task's
// inputs while the current one runs - before its polls, or at the moment its hand-offs have arrived, with the island's error looked at
// only after the arithmetic (256 VGPRs): the time to the first poll falls from 2.3 to 1.3 us and the arithmetic section grows by as
// much; leaving out the scattered point stores (experiment): no difference - a launch is one sweep over 16-18 colours that advances
// round by round at the pace of each round's slowest wave, ~4 us per colour whatever a single task saves; the inertia
// rows as a lane-indexed copy too (costs k_push_links what it saves here); gathering the points through the manifold index one level
// earlier instead of copying them (position kernel 84 instead of 71 us per iteration).
struct DfpIn { uint32_t key, m, ix, label, nx; float im; float4 pw0[3]; };   // pw0 = the first point's (own pivot, local normal, (normal, attachment))
struct DfpIn2 { float4 il[3]; float4 iw[3]; float err; };   // iw: the record's world inertia - what a body not yet corrected in this solve carries (point-by-point form only)
DI void dfp_load1(const DfPosArgs &a, uint32_t p, bool sideB, DfpIn &in) {
    const uint32_t slot = 2 * p + (sideB ? 1u : 0u);
    const size_t rcap = a.mf.cap;
    in.key = a.keys_sorted[p];
    in.m = a.rows.order[p]; in.ix = sideB ? a.rows.bB[p] : a.rows.bA[p]; in.label = a.rows.label[p]; in.nx = a.next[slot]; in.im = a.rows.im[slot];
    // the first point (every task has one); the others follow with the second level, once the key has told how many there are
    in.pw0[0] = a.rows.pw[p + (sideB ? rcap : 0)]; in.pw0[1] = a.rows.pw[p + 2 * rcap]; in.pw0[2] = a.rows.pw[p + 3 * rcap];
}
template <bool BLOCK>
DI void dfp_load2(const DfPosArgs &a, const DfpIn &in, DfpIn2 &in2) {
    const bool proc = in.im != 0.0f;
    const float4 z = make_float4(0, 0, 0, 0);
    in2.il[0] = proc ? B_IL(a.b, in.ix, 0) : z; in2.il[1] = proc ? B_IL(a.b, in.ix, 1) : z; in2.il[2] = proc ? B_IL(a.b, in.ix, 2) : z;
    if (!BLOCK) { in2.iw[0] = proc ? B_IW(a.b, in.ix, 0) : z; in2.iw[1] = proc ? B_IW(a.b, in.ix, 1) : z; in2.iw[2] = proc ? B_IW(a.b, in.ix, 2) : z; }   // the same 128-byte record
    in2.err = a.err_prev ? a.err_prev[in.label] : 1.0f;
}
template <int NP, bool BLOCK>
DI void dfp_task(const DfPosArgs &a, uint32_t p, bool valid, bool sideB, uint32_t np, uint32_t col, const DfpIn &in, const DfpIn2 &in2,
                 const v4f &f0, const v4f &f1, const v4f &f2, uint64_t w0, uint64_t *trace_slot) {
    const Manifolds &mf = a.mf;
    uint64_t w1 = 0, w2 = 0;
    const uint32_t m = in.m, label = in.label;
    const uint32_t slot = 2 * p + (sideB ? 1u : 0u);
    const uint32_t nx = in.nx;
    float4 piv[NP], l4[NP], n4[NP];
    piv[0] = in.pw0[0]; l4[0] = in.pw0[1]; n4[0] = in.pw0[2];
#pragma unroll
    for (int k = 1; k < NP; ++k) {
        const size_t rcap = mf.cap, pb = (size_t)(k * kPosF) * rcap + p;
        piv[k] = a.rows.pw[pb + (sideB ? rcap : 0)]; l4[k] = a.rows.pw[pb + 2 * rcap]; n4[k] = a.rows.pw[pb + 3 * rcap];
    }
    PBody X;   // the transform comes with the hand-off (a read-only body's: from its seeded slot); pivots are anchored at the position
    X.inv_m = in.im; X.proc = in.im != 0.0f;
    X.il = {from4(in2.il[0]), from4(in2.il[1]), from4(in2.il[2])};
    if (!BLOCK) X.iw = {from4(in2.iw[0]), from4(in2.iw[1]), from4(in2.iw[2])};
    X.has_com = false; X.com = mk3(0, 0, 0);   // (worlds with centre-of-mass offsets do not take the dataflow position solve)
    X.pos = X.org = mk3(0, 0, 0); X.orn = q4{0, 0, 0, 1};
    const uint32_t done_isl = in2.err < kPosErrorThreshold ? 1u : 0u;
    const uint32_t want = (nx & kHeadBit) ? a.iter : a.iter + 1;
    bool got = false;
    bool corrected = false;
    // The world inertia is not part of the hand-off. Block form: contacts do not read it (pos_manifold_block). Point-by-point form:
    // position_solver::solve rebuilds it after every correction - here it is rebuilt from the handed-over orientation, same operations,
    // same value, just before the first correction that reads it (`iw_stale`: the body was corrected earlier in this solve; an
    // uncorrected body still carries the inertia of its record); the rebuild after a task's last correction would be thrown away.
    // k_pos_writeback rebuilds it from the final orientation.
    // An island that met the error threshold in an earlier iteration takes no part in this one (island_solver.cpp:350-353):
    // its lanes neither wait for nor publish hand-offs - every consumer of its bodies is in the same island, equally
    // finished - and each body's chain-head slot keeps the transform of the island's last iteration for k_pos_writeback.
    bool done = !valid || done_isl != 0;
    const float4 *mine = a.pslot + pslot_at(slot, 0);
    float max_err = 0;
    bool act = false;
    auto accept = [&](const v4f &h0, const v4f &h1, const v4f &h2) {
        if ((__float_as_uint(h0.w) == want && __float_as_uint(h1.w) == want && __float_as_uint(h2.w) == want) || !X.proc) {
            X.pos = mk3(h0.x, h0.y, h0.z); X.orn = q4{h1.x, h1.y, h1.z, h2.x};
            X.org = X.pos;
            corrected = h2.y != 0.0f;
            got = true;
        }
    };
    if (!done) accept(f0, f1, f2);   // the look the kernel loop took before it requested the next task's second level
    for (uint32_t spin = 0;; ++spin) {
        if (!done && !got && spin > 0) {
            v4f h0, h1, h2;
            dfp_poll(mine, h0, h1, h2);
            accept(h0, h1, h2);
        }
        if (a.trace && w1 == 0) w1 = wall_clock64();
        const uint64_t pending = __ballot(!done);
        if (pending == 0) {
            if (trace_slot && (threadIdx.x & 63) == 0) { trace_slot[0] = w0; trace_slot[1] = w1; trace_slot[2] = w2; trace_slot[3] = wall_clock64(); }
            break;
        }
        const uint32_t minc = __shfl(col, __ffsll((long long)pending) - 1);
        const bool mine_now = !done && col == minc;            // both lanes of a pair share p, hence colour
        if (__ballot(mine_now && !got) == 0) {
            if (a.trace && w2 == 0) w2 = wall_clock64();
            // the arithmetic below is pos_contacts_np's; `act` is uniform within a lane pair (DPP exchanges)
            act = mine_now && done_isl == 0;
            bool live[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) live[k] = (uint32_t)k < np && act;
            bool iw_stale = corrected;
            const bool applied = BLOCK ? pos_manifold_block<NP>(X, sideB, piv, l4, n4, live, max_err) : pos_manifold_points<NP>(X, sideB, piv, l4, n4, live, max_err, iw_stale);
            // hand the transform on first (the body's next manifold is waiting for it), then store the points' distances / normals
            if (mine_now) {
                if (X.proc) dfp_publish(a.pslot + pslot_at(nx & kSlotMask, 0), X.pos, X.orn, corrected || applied, a.iter + 1);
                done = true;
            }
            if (act) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const size_t s = pt_at(mf.cap, (uint32_t)k, m);
                    if ((uint32_t)k < np) {
                        if (!sideB) mf.pA[s] = piv[k];
                        else mf.nrm[s] = n4[k];
                    }
                }
                // the body record is NOT written here: two waves on different XCDs would leave the same line dirty in
                // two L2s; k_pos_writeback stores each body's final transform once, from its chain head's slot
            }
        } else {
            if (spin > kDfSpinLimit || ((spin & 1023u) == 1023u && __hip_atomic_load(&a.cnt->df_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (spin > kDfSpinLimit) atomicExch(&a.cnt->df_abort, 1u);
                break;
            }
            if (spin > 0) __builtin_amdgcn_s_sleep(4);
        }
    }
    publish_error(valid && done_isl == 0 && !sideB, max_err, label, a.err_out);
}
template <bool BLOCK>
__global__ void __launch_bounds__(64) k_pos_contacts_df(DfPosArgs a) {
    const uint32_t t = blockIdx.x * 32u + (threadIdx.x >> 1);   // 32 manifolds per wave, two lanes each
    const bool sideB = threadIdx.x & 1u;
    const uint32_t nwaves = a.stride >> 5;
    __shared__ XcdLists L;
    uint32_t count = a.na, step = a.stride, mine = t;
    if (a.xcd_lists) {
        xcd_lists_build(L, a.cnt, blockIdx.x % kXcds);
        count = L.pre[kCls]; step = (gridDim.x / kXcds) * 32u; mine = (blockIdx.x / kXcds) * 32u + (threadIdx.x >> 1);
    }
    for (uint32_t base = 0, round = 0; base < count; base += step, ++round) {
        const uint32_t q = base + mine;
        const uint32_t pt = a.xcd_lists ? (q < count ? xcd_lists_p(L, q, a.na) : a.na) : q;
        const bool valid = pt < a.na && !(a.skip && a.skip[pt]);
        if (!__any(valid)) continue;
        const uint64_t w0 = a.trace ? wall_clock64() : 0;
        const uint32_t p = pt < a.na ? pt : a.na - 1;
        DfpIn in; DfpIn2 in2;
        dfp_load1(a, p, sideB, in);
        v4f f0, f1, f2;
        dfp_poll(a.pslot + pslot_at(2 * p + (sideB ? 1u : 0u), 0), f0, f1, f2);   // the first look at the own slot travels with level 1
        dfp_load2<BLOCK>(a, in, in2);
        const uint32_t np = valid ? 4u - (in.key & 3u) : 0u, col = in.key >> 2;
        uint64_t *tr = a.trace ? a.trace + 4 * ((size_t)round * nwaves + blockIdx.x) : nullptr;
        if (__any(np > 2)) dfp_task<4, BLOCK>(a, p, valid, sideB, np, col, in, in2, f0, f1, f2, w0, tr);
        else if (__any(np > 1)) dfp_task<2, BLOCK>(a, p, valid, sideB, np, col, in, in2, f0, f1, f2, w0, tr);
        else dfp_task<1, BLOCK>(a, p, valid, sideB, np, col, in, in2, f0, f1, f2, w0, tr);
    }
}

// ---- island-fused schedule -----------------------------------------------------------------------------------------
// Scenes whose velocity solve cannot take the dataflow launch (joints, contact_extras rows) used to cost one launch per
// colour and sweep - ~180 dependent 5-10 us launches a step for a field of rag dolls. Islands are independent, so ONE
// wave per island runs its whole solve instead: the island's constraints (joints and active manifolds) are bucketed by
// island label, sorted by phase (joint colours, then contact colours) in LDS, and the wave walks the phases of every
// sweep in the order of the per-colour schedule with a workgroup barrier between them (same CU, same L1: what
// k_contact_solve_tail does for one colour suffix). The arithmetic is the per-colour kernels' lane functions on the same
// global arrays, so results are bit-identical with the per-colour schedule; only islands too large for one wave to be
// worth it (host decision from the previous step's largest island, kIslFusedLimit) keep the launches.
struct JointColours { uint32_t n; uint32_t start[kMaxColours + 1]; };
struct IslLists {
    uint32_t *cnt;      // [bodies + 1] items per island label; zero between steps (the fill counts it back down)
    uint32_t *off;      // [bodies + 1] exclusive scan of cnt
    uint32_t *list;     // labels of the islands that have items, in no particular order (islands are independent)
    uint32_t *items;    // phase << 24 | index (joint: position in colour order; manifold: sorted position p), by island
    uint32_t *sorted;   // the same, sorted by phase: only islands beyond the LDS list use it
    uint32_t *joint;    // [label] != 0: the island has (awake) joints - marked per step by k_prep_joints, cleared by k_finish
};
constexpr uint32_t kIslPhaseShift = 24, kIslIdMask = 0xFFFFFFu, kIslContactPhase = 64, kIslPhases = 128;
constexpr uint32_t kIslLdsItems = 1024;    // an island's phase-sorted list lives in LDS up to this size
constexpr uint32_t kIslFusedLimit = 4096;  // largest island (items) for which one wave per island beats the launches
DI uint32_t isl_item_label(uint32_t t, uint32_t nj, uint32_t na, const Joints &j, const Rows &rows, const Bodies &b, bool &is_joint, uint32_t &id) {
    if (t < nj) {
        is_joint = true; id = t;
        const uint32_t ia = j.bodyA[t], ib = j.bodyB[t], fa = b.flags[ia], fb = b.flags[ib];
        if (edge_asleep(fa, fb)) return 0xFFFFFFFFu;
        if (is_dynamic(fa)) return b.island[ia];
        if (is_dynamic(fb)) return b.island[ib];
        return 0xFFFFFFFFu;
    }
    is_joint = false; id = t - nj;
    return id < na ? rows.label[id] : 0xFFFFFFFFu;
}
// Lanes of a wave that bucket into the same island share ONE atomic (a pile with a few joints is a single island: without this
// every constraint of the scene would hit the same counter). Returns what `op(label, lanes)` gave the group's first lane; `rank`
// is this lane's position within its group.
template <typename Op>
DI uint32_t isl_wave_reserve(uint32_t label, Op op, uint32_t &rank) {
    uint32_t result = 0;
    rank = 0;
    uint64_t todo = __ballot(label != 0xFFFFFFFFu);
    const uint64_t below = (1ull << (threadIdx.x & 63u)) - 1ull;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t l = (uint32_t)__shfl((int)label, lead);
        const uint64_t same = __ballot(label == l);
        uint32_t got = 0;
        if ((int)(threadIdx.x & 63u) == lead) got = op(l, (uint32_t)__popcll(same));
        got = (uint32_t)__shfl((int)got, lead);
        if (label == l) { result = got; rank = (uint32_t)__popcll(same & below); }
        todo &= ~same;
    }
    return result;
}
__global__ void k_isl_count(uint32_t nj, uint32_t na, Joints j, Rows rows, Bodies b, IslLists L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_joint; uint32_t id, rank;
    const uint32_t label = t < nj + na ? isl_item_label(t, nj, na, j, rows, b, is_joint, id) : 0xFFFFFFFFu;
    isl_wave_reserve(label, [&](uint32_t l, uint32_t count) { return atomicAdd(&L.cnt[l], count); }, rank);
}
__global__ void k_isl_fill(uint32_t nj, uint32_t na, Joints j, JointColours jc, Rows rows, const uint32_t *__restrict__ keys_sorted, Bodies b,
                           IslLists L, Counters *cnt) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_joint = false; uint32_t id = 0, rank;
    const uint32_t label = t < nj + na ? isl_item_label(t, nj, na, j, rows, b, is_joint, id) : 0xFFFFFFFFu;
    // the counter counts back down to zero: no clearing pass between steps
    const uint32_t before = isl_wave_reserve(label, [&](uint32_t l, uint32_t count) { return atomicSub(&L.cnt[l], count); }, rank);
    const bool jointed = label != 0xFFFFFFFFu && L.joint[label] != 0;
    const uint64_t free_lanes = __ballot(label != 0xFFFFFFFFu && !is_joint && !jointed);   // manifolds of islands without joints
    if (free_lanes && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)free_lanes) - 1)) atomicAdd(&cnt->isl_free, (uint32_t)__popcll(free_lanes));
    if (label == 0xFFFFFFFFu) return;
    uint32_t phase;
    if (is_joint) { phase = 0; while (phase + 1 < jc.n && id >= jc.start[phase + 1]) ++phase; }
    else phase = kIslContactPhase + (keys_sorted[id] >> 2);
    const uint32_t left = before - rank;   // this lane's share of the group's reservation
    const uint32_t o = L.off[label];
    L.items[o + left - 1] = (phase << kIslPhaseShift) | id;
    if (left == 1) {
        L.list[atomicAdd(&cnt->isl_num, 1u)] = label;
        const uint32_t size = L.off[label + 1] - o;
        atomicMax(&cnt->isl_max_items, size);
        if (jointed) atomicMax(&cnt->isl_max_jitems, size);
    }
}
// Sorts one island's items by phase (counting sort through LDS) and lists the phases that occur. Returns the list to
// walk (LDS, or the global scratch for an island beyond kIslLdsItems) - all lanes of the (single-wave) workgroup call it.
struct IslShared { uint32_t items[kIslLdsItems]; uint32_t start[kIslPhases + 1]; uint32_t cursor[kIslPhases]; uint32_t phases[kIslPhases]; uint32_t nph; };
DI const uint32_t *isl_sort(IslShared &S, const IslLists &L, uint32_t o, uint32_t size) {
    const uint32_t t = threadIdx.x;
    __syncthreads();   // the previous island's walk is over
    S.cursor[t] = 0; S.cursor[t + 64] = 0;
    __syncthreads();
    for (uint32_t q = t; q < size; q += 64) atomicAdd(&S.cursor[L.items[o + q] >> kIslPhaseShift], 1u);
    __syncthreads();
    // exclusive scan of the 128 counts: lane t owns phases t and t + 64
    const uint32_t c0 = S.cursor[t], c1 = S.cursor[t + 64];
    uint32_t s0 = c0, s1 = c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u0 = __shfl_up(s0, d), u1 = __shfl_up(s1, d);
        if ((int)t >= d) { s0 += u0; s1 += u1; }
    }
    const uint32_t total0 = __shfl(s0, 63);
    S.start[t] = s0 - c0; S.start[t + 64] = total0 + s1 - c1;
    if (t == 63) S.start[kIslPhases] = total0 + s1;
    const uint64_t m0 = __ballot(c0 != 0), m1 = __ballot(c1 != 0);
    const uint64_t below = (1ull << t) - 1ull;
    if (c0) S.phases[__popcll(m0 & below)] = t;
    if (c1) S.phases[__popcll(m0) + __popcll(m1 & below)] = t + 64;
    if (t == 0) S.nph = (uint32_t)(__popcll(m0) + __popcll(m1));
    __syncthreads();
    S.cursor[t] = S.start[t]; S.cursor[t + 64] = S.start[t + 64];
    __syncthreads();
    uint32_t *dst = size <= kIslLdsItems ? S.items : L.sorted + o;
    for (uint32_t q = t; q < size; q += 64) {
        const uint32_t it = L.items[o + q];
        dst[atomicAdd(&S.cursor[it >> kIslPhaseShift], 1u)] = it;
    }
    __threadfence_block();
    __syncthreads();
    return dst;
}
struct IslSolveArgs {
    IslLists L; const Counters *cnt;
    Joints j; Bodies b; Rows rows; Manifolds mf; uint32_t rcap; float4 *rwx;
    uint32_t iters;                 // velocity: iterations after the warm start; position: position iterations
    float *isl_err; uint32_t *isl_done;
    uint32_t only_jointed;          // mixed schedule: islands without joints belong to the dataflow launch - skip them here
};
template <bool WARM, bool FUSED>
DI void isl_velocity_sweep(const IslSolveArgs &a, const IslShared &S, const uint32_t *lst) {
    const uint32_t nph = S.nph;
    for (uint32_t k = 0; k < nph; ++k) {
        const uint32_t ph = S.phases[k], q0 = S.start[ph], q1 = S.start[ph + 1];
        if (ph < kIslContactPhase) {
            for (uint32_t q = q0 + threadIdx.x; q < q1; q += 64) joint_solve_lane<WARM>(lst[q] & kIslIdMask, a.j, a.b);
        } else {
            for (uint32_t q = q0 + threadIdx.x; q < q1; q += 64) {
                const uint32_t p = lst[q] & kIslIdMask, np = a.rows.np[p];
                if (__any(np > 2)) contact_solve_np<WARM, 4, false, FUSED>(p, np, a.rows.bA, a.rows.bB, a.rows.rw, a.rcap, a.b.dvw, nullptr, a.rwx);
                else contact_solve_np<WARM, 2, false, FUSED>(p, np, a.rows.bA, a.rows.bB, a.rows.rw, a.rcap, a.b.dvw, nullptr, a.rwx);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}
// Fast path for an island of at most 64 constraints whose dynamic bodies lie within kIslBodySlots indices of each other (a
// rag doll, a chain, a small heap): each lane owns ONE constraint and keeps its rows in registers for the whole solve, the
// bodies' velocity deltas live in LDS, and a phase costs an LDS round trip plus the row arithmetic instead of two
// dependent trips to memory. Same operations in the same order as the lane functions above.
constexpr uint32_t kIslBodySlots = 256, kIslNoSlot = 0xFFFFFFFFu;
struct IslFast { float4 dv[kIslBodySlots], dw[kIslBodySlots]; };
template <bool FUSED>
DI void isl_velocity_fast(const IslSolveArgs &a, const IslShared &S, IslFast &F, bool has, uint32_t item, uint32_t ia, uint32_t ib, uint32_t base) {
    const uint32_t ph = item >> kIslPhaseShift, id = item & kIslIdMask;
    const bool is_joint = ph < kIslContactPhase;
    Delta d;
    d.dvA = d.dwA = d.dvB = d.dwB = mk3(0, 0, 0); d.imA = d.imB = 0;
    d.iA = d.iB = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
    uint32_t sa = kIslNoSlot, sb = kIslNoSlot, np = 0;
    RowReg R[4][kRowsPerPoint];
    if (has) {
        const float4 va = B_DV(a.b, ia), vb = B_DV(a.b, ib);
        d.imA = va.w; d.imB = vb.w;
        if (va.w != 0) { sa = ia - base; F.dv[sa] = va; F.dw[sa] = B_DW(a.b, ia); }
        if (vb.w != 0) { sb = ib - base; F.dv[sb] = vb; F.dw[sb] = B_DW(a.b, ib); }
        if (is_joint) {
            jlane_load(R, a.j, a.b, id, ia, ib);
        } else {
            np = a.rows.np[id];
            rows_load<4>(R, a.rows.rw, a.rcap, id);
        }
    }
    __syncthreads();
    const uint32_t nph = S.nph;
    for (uint32_t sweep = 0; sweep <= a.iters; ++sweep) {
        for (uint32_t k = 0; k < nph; ++k) {
            if (has && S.phases[k] == ph) {
                if (sa != kIslNoSlot) { d.dvA = from4(F.dv[sa]); d.dwA = from4(F.dw[sa]); } else { d.dvA = d.dwA = mk3(0, 0, 0); }
                if (sb != kIslNoSlot) { d.dvB = from4(F.dv[sb]); d.dwB = from4(F.dw[sb]); } else { d.dvB = d.dwB = mk3(0, 0, 0); }
                if (is_joint) { if (sweep == 0) jlane_solve<true>(R, d); else jlane_solve<false>(R, d); }
                else { if (sweep == 0) rows_solve<true, 4, FUSED>(d, R, np); else rows_solve<false, 4, FUSED>(d, R, np); }
                if (sa != kIslNoSlot) { F.dv[sa] = to4(d.dvA, d.imA); F.dw[sa] = to4(d.dwA, 0); }
                if (sb != kIslNoSlot) { F.dv[sb] = to4(d.dvB, d.imB); F.dw[sb] = to4(d.dwB, 0); }
            }
            __syncthreads();
        }
    }
    if (has) {
        if (sa != kIslNoSlot) { B_DV(a.b, ia) = F.dv[sa]; B_DW(a.b, ia) = F.dw[sa]; }
        if (sb != kIslNoSlot) { B_DV(a.b, ib) = F.dv[sb]; B_DW(a.b, ib) = F.dw[sb]; }
        if (is_joint) jlane_store(R, a.j, id);
        else if (a.iters) rows_store_impulses<4>(R, a.rows.rw, a.rcap, id, np);
    }
}
template <bool FUSED>
__global__ void __launch_bounds__(64) k_island_velocity(IslSolveArgs a) {
    __shared__ IslShared S;
    __shared__ IslFast F;
    const uint32_t num = a.cnt->isl_num, t = threadIdx.x;
    for (uint32_t k = blockIdx.x; k < num; k += gridDim.x) {
        const uint32_t label = a.L.list[k], o = a.L.off[label], size = a.L.off[label + 1] - o;
        if (a.only_jointed && !a.L.joint[label]) continue;
        const uint32_t *lst = isl_sort(S, a.L, o, size);
        bool fast = false;
        uint32_t item = 0, ia = 0, ib = 0, base = 0;
        const bool has = t < size;
        if (size <= 64 && a.rwx == nullptr) {
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            bool generic = false;
            if (has) {
                item = lst[t];
                const uint32_t id = item & kIslIdMask;
                if ((item >> kIslPhaseShift) < kIslContactPhase) { ia = a.j.bodyA[id]; ib = a.j.bodyB[id]; generic = a.j.type[id] == EDYNHIP_JOINT_GENERIC; }
                else { ia = a.rows.bA[id]; ib = a.rows.bB[id]; }
                if (B_DV(a.b, ia).w != 0) { lo = min(lo, ia); hi = max(hi, ia); }
                if (B_DV(a.b, ib).w != 0) { lo = min(lo, ib); hi = max(hi, ib); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, off)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, off)); }
            fast = !__any(generic) && lo <= hi && hi - lo < kIslBodySlots;
            base = lo;
        }
        if (fast) {
            isl_velocity_fast<FUSED>(a, S, F, has, item, ia, ib, base);
        } else {
            isl_velocity_sweep<true, FUSED>(a, S, lst);
            for (uint32_t it = 0; it < a.iters; ++it) isl_velocity_sweep<false, FUSED>(a, S, lst);
        }
    }
}
template <bool BLOCK>
__global__ void __launch_bounds__(64) k_island_position(IslSolveArgs a) {
    __shared__ IslShared S;
    __shared__ uint32_t s_done;
    const uint32_t num = a.cnt->isl_num, t = threadIdx.x;
    for (uint32_t k = blockIdx.x; k < num; k += gridDim.x) {
        const uint32_t label = a.L.list[k], o = a.L.off[label], size = a.L.off[label + 1] - o;
        if (a.only_jointed && !a.L.joint[label]) continue;
        const uint32_t *lst = isl_sort(S, a.L, o, size);
        const uint32_t nph = S.nph;
        for (uint32_t it = 0; it < a.iters; ++it) {
            for (uint32_t kk = 0; kk < nph; ++kk) {
                const uint32_t ph = S.phases[kk], q0 = S.start[ph], q1 = S.start[ph + 1];
                // uniform trip counts: every lane reaches the wave-level exchanges and reductions of the lane functions
                if (ph < kIslContactPhase) {
                    for (uint32_t base = q0; base < q1; base += 64) {
                        const bool in = base + t < q1;
                        pos_joints_lane(lst[in ? base + t : q0] & kIslIdMask, in, a.j, a.b, a.isl_err, a.isl_done);
                    }
                } else {
                    for (uint32_t base = q0; base < q1; base += 32) {
                        const uint32_t q = base + (t >> 1);
                        const bool in = q < q1;
                        pos_contacts_item<BLOCK>(in, lst[in ? q : q0] & kIslIdMask, t & 1u, a.rows, a.mf, a.b, a.isl_err, a.isl_done);
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
            // k_pos_flags for this island: below the threshold it takes no part in further iterations (island_solver.cpp:350-353)
            if (t == 0) {
                const float e = __uint_as_float(__hip_atomic_load((const uint32_t *)&a.isl_err[label], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const uint32_t done = e < kPosErrorThreshold ? 1u : 0u;
                if (done) a.isl_done[label] = 1;
                __hip_atomic_store((uint32_t *)&a.isl_err[label], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_done = done;
            }
            __threadfence_block();
            __syncthreads();
            if (s_done) break;
        }
    }
}

__global__ void k_pos_flags(uint32_t n, float *isl_err, uint32_t *isl_done) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (isl_err[i] < kPosErrorThreshold) isl_done[i] = 1;
    isl_err[i] = 0;
}

// ------------------------------------------------------------------ derived state
// update_aabbs (update_aabbs.cpp:53-78 -> aabb_util.cpp:42-70) and update_inertias (update_inertias.cpp:12-24) of one body.
__device__ __forceinline__ void derive_body(Bodies &b, uint32_t i, const dc::Meshes &meshes) {
    const uint32_t fl = b.flags[i];
    const uint32_t kind = fl & BF_KIND_MASK;
    if (kind == EDYNHIP_KIND_STATIC || (fl & BF_ASLEEP)) return;   // update_aabbs / update_inertias exclude sleeping bodies
    const int st = (int)((fl & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
    const q4 orn = q_from4(B_ORN(b, i));
    f3 pos = from4(B_POS(b, i));   // below: where the SHAPE sits
    if (b.origin && b.com[i].w != 0.0f) {   // update_origins (update_origins.cpp:13-19, solver.cpp:453: before the AABBs)
        pos = to_world(-from4(b.com[i]), pos, orn);
        b.origin[i] = to4(pos, 0);
    }
    const m3 basis = to_m3(orn);
    if (st == dc::SHAPE_BOX) {   // aabb_util.cpp:42-63
        const f3 h = from4(b.shape[i]);
        float mn[3] = {pos.x, pos.y, pos.z}, mx[3] = {pos.x, pos.y, pos.z};
        const f3 rws[3] = {basis.r0, basis.r1, basis.r2};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx) {
                float e = comp(rws[r], cidx) * -comp(h, cidx);
                float f = -e;
                if (e < f) { mn[r] += e; mx[r] += f; } else { mn[r] += f; mx[r] += e; }
            }
        b.amin[i] = make_float4(mn[0], mn[1], mn[2], 0);
        b.amax[i] = make_float4(mx[0], mx[1], mx[2], 0);
    } else if (st == dc::SHAPE_SPHERE) {   // aabb_util.cpp:65-70
        const float r = b.shape[i].x;
        b.amin[i] = make_float4(pos.x - r, pos.y - r, pos.z - r, 0);
        b.amax[i] = make_float4(pos.x + r, pos.y + r, pos.z + r, 0);
    } else if (st == dc::SHAPE_CAPSULE) {   // aabb_util.cpp:81-88
        const float4 sh = b.shape[i];
        const f3 v = rotate(orn, dc::axis_vector(sh.z)) * sh.y;
        const f3 p0 = pos - v, p1 = pos + v;
        b.amin[i] = make_float4(fminf(p0.x, p1.x) - sh.x, fminf(p0.y, p1.y) - sh.x, fminf(p0.z, p1.z) - sh.x, 0);
        b.amax[i] = make_float4(fmaxf(p0.x, p1.x) + sh.x, fmaxf(p0.y, p1.y) + sh.x, fmaxf(p0.z, p1.z) + sh.x, 0);
    } else if (st == dc::SHAPE_CYLINDER) {   // aabb_util.cpp:72-79
        const box3 bb = dc::cylinder_aabb(dc::cyl_of(b.shape[i]), pos, orn);
        b.amin[i] = to4(bb.mn, 0); b.amax[i] = to4(bb.mx, 0);
    } else if (st == dc::SHAPE_POLYHEDRON) {   // update_aabbs.cpp:22-32: the point cloud of the rotated mesh, moved to the position
        const box3 bb = dc::polyhedron_aabb(meshes, b.shape[i], pos, orn);
        b.amin[i] = to4(bb.mn, 0); b.amax[i] = to4(bb.mx, 0);
    }
    if (kind == EDYNHIP_KIND_DYNAMIC) {   // update_inertias.cpp:12-24
        const m3 il = {from4(B_IL(b, i, 0)), from4(B_IL(b, i, 1)), from4(B_IL(b, i, 2))};
        const m3 iw = mul(mul(basis, il), transpose(basis));
        B_IW(b, i, 0) = to4(iw.r0, 0); B_IW(b, i, 1) = to4(iw.r1, 0); B_IW(b, i, 2) = to4(iw.r2, 0);
    }
}
// End of the step. Per body: the position solve's final transform (dataflow mode: it lives in the body's chain-head slot
// until now), the derived state (AABB, world inertia), the next step's scratch, and the broadphase's question for the next
// step - has this body left the slack box its candidate list was built for? (Counters::bp_rebuild, see broadphase.hip.)
__global__ void k_finish(uint32_t n, Bodies b, uint64_t *used, uint32_t *next_seg_start, uint32_t *next_seg_end, Counters *cnt,
                         const float4 *__restrict__ pslot, const uint32_t *__restrict__ first_slot, CandLists cl, uint32_t *__restrict__ isl_joint, uint2 *__restrict__ isl_top, dc::Meshes meshes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {   // the next step's counters (what k_step_reset does for a stand-alone stage run)
        const int t = threadIdx.x;
        if (t == 0) {
            cnt->num_pairs = 0; cnt->pair_overflow = 0; cnt->num_points = 0; cnt->num_active = 0;
            cnt->uncoloured = 0; cnt->colour_overflow = 0; cnt->pairs_changed = 0; cnt->num_found = 0; cnt->num_new = 0; cnt->num_extra = 0; cnt->num_awake = 0; cnt->unc_count = 0; cnt->pairs_differ = 0; cnt->tree_found = 0; cnt->tree_marks = 0;
        }
        if (t < 3) { cnt->bounds_min[t] = 0x7FFFFFFF; cnt->bounds_max[t] = (int)0x80000000; }
        for (int k = t; k < 4 * (int)kMaxColours; k += (int)blockDim.x) { cnt->colour_start[k] = 0; cnt->colour_end[k] = 0; }
    }
    // pre-clear the next step's per-body scratch (colour masks, segment index of the manifold buffer it will fill)
    bool moved = false;
    if (i < n) {
        used[i] = 0; next_seg_start[i] = 0; next_seg_end[i] = 0; isl_joint[i] = 0; isl_top[i] = make_uint2(0u, 0u);
        const uint32_t fl = b.flags[i];
        if (pslot && is_dynamic(fl)) pos_writeback(b, i, pslot, first_slot);
        derive_body(b, i, meshes);
        if (cl.count && is_dynamic(fl) && !(fl & BF_REMOVED)) {
            const float4 a = b.amin[i], c = b.amax[i], ra = cl.ref_min[i], rc = cl.ref_max[i];
            const float d = fmaxf(fmaxf(fmaxf(fabsf(a.x - ra.x), fabsf(a.y - ra.y)), fabsf(a.z - ra.z)),
                                  fmaxf(fmaxf(fabsf(c.x - rc.x), fabsf(c.y - rc.y)), fabsf(c.z - rc.z)));
            moved = !(d <= ra.w);   // ra.w = this body's slack (a NaN box counts as moved); a body whose list overflowed walks the
                                    // (still valid) tree in k_bp_pairs instead - no reason to rebuild everybody's list
        }
    }
    if (__any(moved) && (threadIdx.x & 63) == 0) cnt->bp_rebuild = 1u;
}
__global__ void k_refresh_derived(uint32_t n, Bodies b, dc::Meshes meshes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) derive_body(b, i, meshes);
}
int refresh_derived(edynhip_ctx *c) {
    if (c->b.n == 0) return EDYNHIP_OK;
    hipLaunchKernelGGL(k_refresh_derived, dim3((c->b.n + 255) / 256), dim3(256), 0, c->stream, c->b.n, c->b, c->meshes);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

// ------------------------------------------------------------------ host orchestration
// The dataflow kernels need all their workgroups resident together. hipLaunchCooperativeKernel guarantees that, but on
// this runtime every cooperative launch is bracketed by ~12 us of idle GPU on each side (4 launches per step = ~0.1 ms,
// scripts/prof_timeline.py). A caller that owns the device (EDYNHIP_FLAG_EXCLUSIVE_DEVICE: no other context, stream or
// process launches work on it while a step runs - the benchmark, one process per GPU) gets plain launches instead: the
// grid is sized to what the occupancy query says fits, nothing else competes for the slots, so it is resident as a whole;
// the spin limit still turns a violated assumption into an error instead of a hang.
static hipError_t launch_resident(edynhip_ctx *c, const void *kernel, uint32_t grid, uint32_t block, void **params) {
    if (c->cfg.flags & EDYNHIP_FLAG_EXCLUSIVE_DEVICE) return hipLaunchKernel(kernel, dim3(grid), dim3(block), params, 0, c->stream);
    // One cooperative launch at a time per device: contexts of several host threads on ONE device (the multi-GPU world on a single GPU,
    // edynhip_world_create with a device listed more than once) otherwise enter the runtime's cooperative-launch path concurrently, and the
    // process then dies in the runtime's exit handler (ROCm 7.2: hsa_shut_down, `scripts/exit_probe.py multi2`). Uncontended on a node with one context per GPU.
    // A context's FIRST cooperative launch (whatever the runtime sets up lazily for it) is alone in the process.
    static std::mutex coop_launch[64], coop_first;
    std::unique_lock<std::mutex> first(coop_first, std::defer_lock);
    if (!c->coop_launched) first.lock();
    std::lock_guard<std::mutex> guard(coop_launch[(uint32_t)c->device & 63u]);
    const hipError_t e = hipLaunchCooperativeKernel(kernel, dim3(grid), dim3(block), params, 0, c->stream);
    c->coop_launched = true;
    return e;
}
static void rec(edynhip_ctx *c, int idx) {
    if (c->timer.e && ((c->timer.mask >> idx) & 1u)) (void)hipEventRecord(c->timer.e[idx], c->stream);
}

int islands(edynhip_ctx *c) {
    hipStream_t s = c->stream;
    const uint32_t n = c->b.n, M = c->num_manifolds;
    if (n == 0) return EDYNHIP_OK;
    const Manifolds &mf = c->m[c->cur];
    const bool sleeping = c->sleep_active();   // (a world in which no body can sleep runs no sleep kernels: ctx.hpp num_sleepable)
    c->island_labels_valid = true;
    // union-find forest lives in isl_done (scratch until the position solver) to keep b.island stable for readers
    uint32_t *forest = c->isl_done;
    c->solve_begin_done = false;
    const uint32_t force = c->force_islands ? 1u : 0u;
    const uint32_t pm = c->prev_num_manifolds;
    c->force_islands = false;
    // No manifold now or in the previous step, nothing edited, no sleep decisions to take: the labels stand and every kernel below would
    // return at once - not launched at all (a world of joints only: 4 of its ~20 launches per step)
    if (!force && M == 0 && pm == 0 && !sleeping && c->full_step) return EDYNHIP_OK;
    // An in-place step (broadphase.hip: the pair set is last step's) has nothing to relabel. Otherwise the counters fetched with
    // the pair count say whether every certificate manifold is still there (see CC_INCREMENTAL above).
    const bool inplace = c->inplace_step;
    c->inplace_step = false;
    const int mode = force ? CC_FULL : inplace ? CC_SKIP
                     : (c->full_step && c->cnt_host->tree_found == c->cnt_host->tree_total) ? CC_INCREMENTAL : CC_FULL;
    (void)pm;
    if (mode == CC_FULL) ++c->cc_full_steps; else if (mode == CC_INCREMENTAL) ++c->cc_incremental_steps;
    // island sleeping: last step's labels, before the hooks rewrite them (the merge rule of k_sleep_sizes / k_sleep_carry reads them)
    if (sleeping && mode != CC_SKIP && c->sleep_old_label)
        EH_HIP(c, hipMemcpyAsync(c->sleep_old_label, c->b.island, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    // the solve's per-body start rides on the flatten kernel when nothing in between looks at velocities or sleep flags
    const bool begin = c->full_step && !sleeping && !c->has_restitution;
    auto flatten = [&](uint32_t *forest_or_labels) {
        uint32_t *split = (sleeping && mode == CC_FULL) ? c->sleep_state : nullptr;
        if (begin) hipLaunchKernelGGL(k_cc_flatten<true>, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b.flags, forest_or_labels, c->b.island, c->cnt, mode, c->b, c->cfg.fixed_dt, c->rows.first_slot, split);
        else hipLaunchKernelGGL(k_cc_flatten<false>, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b.flags, forest_or_labels, c->b.island, c->cnt, mode, c->b, c->cfg.fixed_dt, c->rows.first_slot, split);
        c->solve_begin_done = begin;
    };
    if (mode == CC_FULL) {
        hipLaunchKernelGGL(k_cc_init, dim3(blocks(n, 256)), dim3(256), 0, s, n, forest, c->cnt, mf, M, c->b.flags);
        if (c->j.n) hipLaunchKernelGGL(k_cc_hook, dim3(blocks(c->j.n, 256)), dim3(256), 0, s, c->j.n, c->j.bodyA, c->j.bodyB, c->b.flags, forest);
        static const int compress_env = getenv("EDYNHIP_CC_COMPRESS") ? atoi(getenv("EDYNHIP_CC_COMPRESS")) : 1;   // developer knob: passes of k_cc_compress
        if (M) for (int pass = 0; pass < compress_env; ++pass) hipLaunchKernelGGL(k_cc_compress, dim3(blocks(n, 256)), dim3(256), 0, s, n, forest, c->b.flags);
        // (round 5, measured and dropped: one lane per EDGE on the compressed forest instead of the per-body walks - 2 x 67 us against 57:
        //  what costs is not the depth of the finds any more but the unions themselves, thousands of trees hooking into one root)
        if (M) hipLaunchKernelGGL(k_cc_hook_bodies, dim3(blocks(n, 256)), dim3(256), 0, s, n, mf, M, c->b.flags, forest, c->cnt);
        flatten(forest);
    } else if (mode == CC_INCREMENTAL) {   // the labels themselves are the forest (roots = lowest index: depth 1)
        hipLaunchKernelGGL(k_cc_hook_new, dim3(32), dim3(256), 0, s, c->new_edges, c->new_edge_m, mf.tree, c->b.flags, c->b.island, c->cnt);
        flatten(c->b.island);
    }
    if (sleeping) {
        const bool relabelled = mode != CC_SKIP && c->sleep_old_label != nullptr;
        SleepMerge sm{c->sleep_old_label, c->sleep_prev_n, c->sleep_size, c->sleep_best, c->sleep_carried};
        if (!relabelled) sm.best = nullptr;
        if (relabelled) hipLaunchKernelGGL(k_sleep_sizes, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, mf, M, c->j, sm);
        hipLaunchKernelGGL(k_sleep_scan, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, c->sleep_state, sm);
        if (relabelled) hipLaunchKernelGGL(k_sleep_carry, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, sm, c->sleep_since);
        c->sleep_prev_n = n;
        hipLaunchKernelGGL(k_sleep_edges, dim3(32), dim3(256), 0, s, c->new_edges, c->cnt, c->b.island, c->sleep_state);
        hipLaunchKernelGGL(k_sleep_decide, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, c->sleep_state, c->sleep_action, c->sleep_since, c->sim_clock,
                           relabelled ? c->sleep_carried : (const double *)nullptr);
        hipLaunchKernelGGL(k_sleep_apply, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, c->sleep_action, c->cnt);
    }
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

// `between`: called once, after the steady-state colouring and its counter publish are enqueued and before the host waits for the
// counters - what it enqueues runs while the answer travels (it must not depend on the answer). *first_final tells the caller whether
// the counters of that first publish were the final ones (no multi-block colouring rounds, no second sort).
template <typename Between>
static int colour_contacts(edynhip_ctx *c, Between between, bool *first_final) {
    *first_final = false;
    hipStream_t s = c->stream;
    const uint32_t M = c->num_manifolds, n = c->b.n;
    Manifolds &mf = c->m[c->cur];
    c->num_active = 0;
    if (M == 0) { c->num_colours = 0; return EDYNHIP_OK; }
    if (!c->full_step) {   // inside edynhip_step: `used` was cleared by the previous k_finish, the counters by k_step_reset
        EH_HIP(c, hipMemsetAsync(c->used, 0, (size_t)n * sizeof(uint64_t), s));
        EH_HIP(c, hipMemsetAsync(c->isl_top, 0, (size_t)n * sizeof(uint2), s));
        EH_HIP(c, hipMemsetAsync(&c->cnt->uncoloured, 0, 2 * sizeof(uint32_t), s));   // uncoloured, colour_overflow
        EH_HIP(c, hipMemsetAsync(c->cnt->colour_start, 0, 8 * kMaxColours * sizeof(uint32_t), s));
    }
    hipLaunchKernelGGL(k_col_tops, dim3(blocks(M, 1024)), dim3(1024), 0, s, M, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->b.island, c->isl_top, c->cs_sup);
    hipLaunchKernelGGL(k_col_prepare, dim3(blocks(M, 1024)), dim3(1024), 0, s, M, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->used, c->best[0], c->best[1], c->cnt, c->b.island, c->isl_top, c->col_unc);
    uint32_t round = 0, total_rounds = 0;
    auto run_rounds = [&](uint32_t count) {
        for (uint32_t r = 0; r < count; ++r, ++round) {
            uint64_t *bc = c->best[round & 1], *bn = c->best[(round + 1) & 1];
            hipLaunchKernelGGL(k_col_best, dim3(blocks(M, 256)), dim3(256), 0, s, M, mf.info, mf.bodyA, mf.bodyB, c->b.flags, bc, bn, c->cnt);
            hipLaunchKernelGGL(k_col_assign, dim3(blocks(M, 256)), dim3(256), 0, s, M, mf.info, mf.bodyA, mf.bodyB, c->b.flags, bc, c->used, c->cnt);
        }
        total_rounds += count;
    };
    bool first = true;
    auto sort_and_fetch = [&]() -> int {
        {
            const uint32_t nb = blocks(M, kCsBlock);
            static const bool direct_env = !(getenv("EDYNHIP_DIRECT_SORT") && getenv("EDYNHIP_DIRECT_SORT")[0] == '0');   // developer knob (A/B)
            if (nb <= kCsDirectBlocks && direct_env) {
                if (!first) EH_HIP(c, hipMemsetAsync(c->cs_sup, 0, (size_t)(kCsDirectBlocks / kCsSuper) * kCsKeys * sizeof(uint32_t), s));   // (the step's first sort: cleared by k_col_tops)
                hipLaunchKernelGGL(k_cs_hist<true>, dim3(nb), dim3(kCsBlock), 0, s, M, c->col_keys, c->cs_hist, nb, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->sleeping, c->cs_sup);
                hipLaunchKernelGGL(k_cs_scatter<true>, dim3(nb), dim3(kCsBlock), 0, s, M, c->col_keys, c->cs_hist, nb, c->col_keys_sorted, c->rows.order, c->cnt, c->cs_sup);
            } else {
                hipLaunchKernelGGL(k_cs_hist<false>, dim3(nb), dim3(kCsBlock), 0, s, M, c->col_keys, c->cs_hist, nb, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->sleeping, (uint32_t *)nullptr);
                EH_TRY(scan_u32(c, c->cs_hist, c->cs_start, kCsKeys * nb));
                hipLaunchKernelGGL(k_cs_scatter<false>, dim3(nb), dim3(kCsBlock), 0, s, M, c->col_keys, c->cs_start, nb, c->col_keys_sorted, c->rows.order, c->cnt, (const uint32_t *)nullptr);
                hipLaunchKernelGGL(k_col_offsets, dim3(blocks(M, 256)), dim3(256), 0, s, M, c->col_keys_sorted, c->cnt);
            }
        }
        uint32_t ticket = 0;
        EH_TRY(publish_counters(c, sizeof(Counters), &ticket));
        if (first) between();
        first = false;
        EH_TRY(wait_counters(c, ticket));
        return EDYNHIP_OK;
    };
    // Steady state: the few new edges are coloured by one workgroup (k_col_rounds) and ONE fetch brings the offsets; what it
    // could not finish (a long list, or more rounds than it runs) is left to the multi-block rounds below.
    static const bool col_lds_env = getenv("EDYNHIP_COL_LDS") && getenv("EDYNHIP_COL_LDS")[0] == '1';   // developer knob (A/B): the rounds with their marks in a hashed LDS table (k_col_rounds)
    constexpr uint32_t kColMinTable = 8192;   // mark slots the longest list still leaves room for
    if (c->col_lds_edges == 0) {   // the listed edges and the mark table live in LDS: as much as one workgroup may have (8 bytes per edge + 4 per slot)
        int max_lds = 0;
        (void)hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device);
        uint32_t bytes = max_lds > 65536 ? (uint32_t)max_lds - 1024u : 48u * 1024u;
        const void *fn = col_lds_env ? (const void *)k_col_rounds_lds : (const void *)k_col_rounds;
        if (bytes > 48u * 1024u && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            bytes = 48u * 1024u;
        }
        c->col_lds_bytes = bytes;
        c->col_lds_edges = std::min<uint32_t>(kColUncCap, (bytes - (col_lds_env ? 4u * kColMinTable : 0u)) / 8u);
    }
    if (col_lds_env)
        hipLaunchKernelGGL(k_col_rounds_lds, dim3(1), dim3(1024), (size_t)c->col_lds_bytes, s, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->used, c->cnt,
                           c->col_unc, c->col_lds_edges, c->col_lds_bytes / 4u, 256u);
    else
        hipLaunchKernelGGL(k_col_rounds, dim3(1), dim3(1024), (size_t)c->col_lds_edges * 8u, s, mf.info, mf.bodyA, mf.bodyB, c->b.flags, c->best[0], c->best[1], c->used, c->cnt,
                           c->col_unc, c->col_lds_edges, 256u);
    EH_TRY(sort_and_fetch());
    *first_final = c->cnt_host->uncoloured == 0;
    if (c->cnt_host->uncoloured != 0) {
        uint32_t batch = 4;
        while (c->cnt_host->uncoloured != 0) {
            if (c->cnt_host->colour_overflow) break;
            run_rounds(batch);
            if (batch < 64) batch *= 2;   // a scene coloured from scratch needs hundreds of rounds; steady state needs none
            EH_TRY(fetch_counters(c, 8 * sizeof(uint32_t)));
            if (total_rounds > 65536) return set_error(c, EDYNHIP_ERR_COLOURS, "colouring did not converge");
        }
        EH_HIP(c, hipMemsetAsync(c->cnt->colour_start, 0, 8 * kMaxColours * sizeof(uint32_t), s));
        EH_TRY(sort_and_fetch());
    }
    // (no contact count is an error any more: what does not fit the 62 parallel colours goes to the serial bucket, ctx.hpp)
    c->stats.colour_rounds = total_rounds + c->cnt_host->col_wg_rounds;
    uint32_t nc = 0, na = 0;
    for (uint32_t k = 0; k < kMaxColours; ++k) {
        uint32_t begin = 0xFFFFFFFFu, pos = 0, cnt4[4];
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t a = c->cnt_host->colour_start[4 * k + g], e = c->cnt_host->colour_end[4 * k + g];
            cnt4[g] = e - a;
            if (e > a && a < begin) begin = a;
        }
        if (begin == 0xFFFFFFFFu) { c->colour_start[k] = c->colour_end[k] = 0; continue; }
        c->colour_start[k] = begin;
        pos = begin;
        for (uint32_t g = 0; g < 3; ++g) { pos += cnt4[g]; c->colour_split[k][g] = pos; }
        c->colour_end[k] = pos + cnt4[3];
        nc = k + 1;
        na = c->colour_end[k] > na ? c->colour_end[k] : na;
    }
    c->num_colours = nc;
    c->num_active = na;
    return EDYNHIP_OK;
}

// Kernel instantiations by (warm start, push hand-offs) and the context's contact arithmetic (ctx.hpp Arith: the reference's operations
// by default, EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / EDYNHIP_FLAG_BLOCK_POSITION opt in to the coloured order's own forms).
using ContactSolveFn = void (*)(uint32_t, uint32_t, Split, const uint32_t *, const uint32_t *, float4 *, uint32_t, float4 *, const float *, float4 *);
using ContactTailFn = void (*)(TailRanges, const uint32_t *, const uint32_t *, float4 *, uint32_t, float4 *, const float *, float4 *);
using ContactSerialFn = void (*)(uint32_t, uint32_t, Split, const uint32_t *, const uint32_t *, float4 *, uint32_t, float4 *, float4 *);
static ContactSolveFn contact_solve_fn(bool warm, bool push, bool fused) {
    static const ContactSolveFn t[2][2][2] = {{{k_contact_solve<false, false, false>, k_contact_solve<false, false, true>}, {k_contact_solve<false, true, false>, k_contact_solve<false, true, true>}},
                                              {{k_contact_solve<true, false, false>, k_contact_solve<true, false, true>}, {k_contact_solve<true, true, false>, k_contact_solve<true, true, true>}}};
    return t[warm][push][fused];
}
static ContactTailFn contact_tail_fn(bool warm, bool push, bool fused) {
    static const ContactTailFn t[2][2][2] = {{{k_contact_solve_tail<false, false, false>, k_contact_solve_tail<false, false, true>}, {k_contact_solve_tail<false, true, false>, k_contact_solve_tail<false, true, true>}},
                                             {{k_contact_solve_tail<true, false, false>, k_contact_solve_tail<true, false, true>}, {k_contact_solve_tail<true, true, false>, k_contact_solve_tail<true, true, true>}}};
    return t[warm][push][fused];
}
static ContactSerialFn contact_serial_fn(bool warm, bool fused) {
    static const ContactSerialFn t[2][2] = {{k_contact_solve_serial<false, false>, k_contact_solve_serial<false, true>}, {k_contact_solve_serial<true, false>, k_contact_solve_serial<true, true>}};
    return t[warm][fused];
}
static const void *df_velocity_fn(uint32_t lanes, bool fused) {
    if (lanes == 4u) return fused ? (const void *)k_contact_solve_df4<true> : (const void *)k_contact_solve_df4<false>;
    if (lanes == 2u) return fused ? (const void *)k_contact_solve_df2<true> : (const void *)k_contact_solve_df2<false>;
    return fused ? (const void *)k_contact_solve_df<true> : (const void *)k_contact_solve_df<false>;
}
static const void *df_position_fn(bool block) { return block ? (const void *)k_pos_contacts_df<true> : (const void *)k_pos_contacts_df<false>; }

int solve(edynhip_ctx *c) {
    hipStream_t s = c->stream;
    const bool fused_rows = (c->cfg.flags & EDYNHIP_FLAG_FUSED_VELOCITY_ROWS) != 0, block_pos = (c->cfg.flags & EDYNHIP_FLAG_BLOCK_POSITION) != 0;
    const uint32_t n = c->b.n;
    if (n == 0) return EDYNHIP_OK;
    Manifolds &mf = c->m[c->cur];
    const float dt = c->cfg.fixed_dt;
    const uint32_t rcap = mf.cap;
    rec(c, 3);
    EH_TRY(restitution(c));   // solve_restitution comes first in solver::update (solver.cpp:397); a no-op without bouncy materials
    // gravity / zeroed deltas do not depend on the colouring: enqueued first, they run while the host waits for the counters
    if (!c->solve_begin_done) hipLaunchKernelGGL(k_solve_begin, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, dt, c->rows.first_slot);
    c->solve_begin_done = false;
    // Contact-only scenes: the whole velocity solve as one dataflow launch (see k_contact_solve_df).
    if (c->df_mode < 0) {
        c->df_mode = 0;
        const char *env = getenv("EDYNHIP_DATAFLOW");
        int per_cu = 0, ncu = 0, coop = 0;
        if (!(env && env[0] == '0') &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, df_velocity_fn(1u, fused_rows), kDfBlock, 0) == hipSuccess &&
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess &&
            hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device) == hipSuccess && per_cu > 0 && ncu > 0 && coop) {
            c->df_lanes = (uint32_t)per_cu * (uint32_t)ncu;   // resident waves (one per workgroup)
            int per_cu2 = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, df_velocity_fn(2u, fused_rows), 64, 0) == hipSuccess && per_cu2 > 0)
                c->df2_waves = (uint32_t)per_cu2 * (uint32_t)ncu;
            int per_cu4 = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, df_velocity_fn(4u, fused_rows), 64, 0) == hipSuccess && per_cu4 > 0)
                c->df4_waves = (uint32_t)per_cu4 * (uint32_t)ncu;
            int per_cu_p = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_p, df_position_fn(block_pos), 64, 0) == hipSuccess && per_cu_p > 0)
                c->dfp_waves = (uint32_t)per_cu_p * (uint32_t)ncu;
            c->df_mode = 1;
        }
        (void)hipGetLastError();
    }
    const Joints &j = c->j;
    // Contact-only worlds on the dataflow schedule: the row preparation does not wait for the host to read the colouring's counters.
    // It is enqueued right behind the counter publish over ALL manifolds - the active ones are the prefix of the sorted order whose
    // keys name a colour (k_prep_contacts by_key) - and runs while the answer travels to the host and the next launches travel back
    // (14 us of idle GPU per step on the headline pile, profiles/r04_timeline_pile32k.txt). If the colouring turns out unfinished
    // (uncoloured edges left for the multi-block rounds: a scene coloured from scratch) the rows are prepared again after the second sort;
    // a non-empty serial bucket only drops `push` - the slot table written for it is read through the bodies' colour masks alone.
    static const bool spec_env = !(getenv("EDYNHIP_SPECULATE") && getenv("EDYNHIP_SPECULATE")[0] == '0');
    const bool kPwInPrep = true;   // k_prep_contacts also writes the position solve's lane-indexed copy of the points (Rows::pw) on the push schedules
    bool spec_prep = false;
    auto speculative_prep = [&]() {
        rec(c, 4);
        if (!(spec_env && j.n == 0 && !c->extras && c->df_mode == 1 && c->full_step && c->num_manifolds > 0)) return;
        const uint32_t M = c->num_manifolds;
        hipLaunchKernelGGL(k_prep_contacts<false>, dim3(blocks(M, 128)), dim3(128), 0, s, M, c->rows, rcap, mf, c->b, dt, c->col_keys_sorted, true, true, kPwInPrep);
        spec_prep = true;
    };
    bool first_final = false;
    EH_TRY(colour_contacts(c, speculative_prep, &first_final));
    if (!first_final) spec_prep = false;   // the sorted order changed after the speculative launch: prepare again
    // colour_contacts fetched the counters: with island sleeping, remember whether anything is still awake
    c->all_asleep = c->sleep_active() && c->full_step && c->num_manifolds > 0 && c->cnt_host->num_awake == 0;
    if (c->num_manifolds == 0) rec(c, 4);
    const uint32_t na = c->num_active, nc = c->num_colours;
    if (j.n) hipLaunchKernelGGL(k_prep_joints, dim3(blocks(j.n, 128)), dim3(128), 0, s, j, c->b, dt, c->isl_joint);
    if (j.n && c->has_generic) hipLaunchKernelGGL(k_prep_generic, dim3(blocks(j.n, 64)), dim3(64), 0, s, j, c->b, dt);
    // without joints every delta hand-off stays inside the contact sweeps; contact_extras rows exist on the per-colour schedule only
    // a non-empty serial bucket (a body with more than 62 coloured contacts) also needs the per-colour schedule
    const bool serial = nc == kSerialColour + 1 && c->colour_end[kSerialColour] > c->colour_start[kSerialColour];
    const uint32_t nc_par = serial ? kSerialColour : nc;   // colours solved in parallel
    // Schedules. Contact-only scene: one dataflow launch (push hand-offs). With joints (or contact_extras rows): the constraints are
    // bucketed by island every step, and the statistics of the PREVIOUS step's bucketing (they arrive with the counters this step
    // fetched anyway) choose: "mixed" - islands with joints go to the island-fused kernels, the (many) manifolds of islands without
    // joints stay on the dataflow launch: a pile next to a rag doll keeps its fast path; "fused" - every island is small: one wave per
    // island; else one launch per colour.
    static const bool isl_env = !(getenv("EDYNHIP_ISLAND_FUSED") && getenv("EDYNHIP_ISLAND_FUSED")[0] == '0');
    static const bool mixed_env = !(getenv("EDYNHIP_MIXED") && getenv("EDYNHIP_MIXED")[0] == '0');
    const IslLists isl{c->isl_cnt, c->isl_off, c->isl_list, c->isl_items, c->isl_sorted, c->isl_joint};
    const bool contacts_only = j.n == 0 && !c->extras;
    const bool isl_candidate = isl_env && !contacts_only && !serial && (na + j.n) > 0 && (size_t)na + j.n < kIslIdMask;
    uint32_t largest = 0xFFFFFFFFu, largest_jointed = 0xFFFFFFFFu, free_manifolds = 0;
    if (isl_candidate) {
        const bool fetched = c->last_fetch_step == c->step_index;   // this step's counters were read after the previous step's fill
        if (fetched && c->isl_prep_step + 1 == c->step_index) {
            largest = c->cnt_host->isl_max_items; largest_jointed = c->cnt_host->isl_max_jitems; free_manifolds = c->cnt_host->isl_free;
        } else if (!fetched && c->isl_cache_epoch == c->topology_epoch) {
            largest = c->isl_cache_max; largest_jointed = c->isl_cache_jmax; free_manifolds = c->isl_cache_free;
        }
    }
    constexpr uint32_t kMixedMinFree = 1024;   // manifolds outside jointed islands that make the dataflow launch worth its fixed cost
    bool mixed = isl_candidate && mixed_env && j.n > 0 && !c->extras && na > 0 && c->df_mode == 1 && c->cfg.num_position_iterations <= kMaxDfPosIters &&
                       free_manifolds >= kMixedMinFree && largest_jointed <= kIslFusedLimit && !c->b.com;
    bool push = na > 0 && !serial && (contacts_only || mixed);
    if (na) {
        if (c->extras) hipLaunchKernelGGL(k_prep_contacts<true>, dim3(blocks(na, 128)), dim3(128), 0, s, na, c->rows, rcap, mf, c->b, dt, c->col_keys_sorted, push, false, false);
        else if (!spec_prep) hipLaunchKernelGGL(k_prep_contacts<false>, dim3(blocks(na, 128)), dim3(128), 0, s, na, c->rows, rcap, mf, c->b, dt, c->col_keys_sorted, push, false, kPwInPrep);
    }
    bool isl_fused = false;
    if (isl_candidate) {
        JointColours jc{};
        jc.n = j.num_colours;
        for (uint32_t k = 0; k <= j.num_colours && k <= kMaxColours; ++k) jc.start[k] = j.colour_start[k];
        // A world of joints only (no contact this step, no sleeping) whose bodies and joints have not been edited since the lists were
        // built has the same islands and the same lists: they are kept (chains16k: 4 launches of 28 per step)
        const bool keep_lists = na == 0 && !c->sleep_active() && c->isl_lists_epoch == c->topology_epoch;
        if (!keep_lists) {
        EH_HIP(c, hipMemsetAsync(&c->cnt->isl_num, 0, 4 * sizeof(uint32_t), s));   // isl_num, isl_max_items, isl_max_jitems, isl_free
        hipLaunchKernelGGL(k_isl_count, dim3(blocks(j.n + na, 256)), dim3(256), 0, s, j.n, na, j, c->rows, c->b, isl);
        EH_TRY(scan_u32(c, c->isl_cnt, c->isl_off, n + 1));
        hipLaunchKernelGGL(k_isl_fill, dim3(blocks(j.n + na, 256)), dim3(256), 0, s, j.n, na, j, jc, c->rows, c->col_keys_sorted, c->b, isl, c->cnt);
        c->isl_lists_epoch = (na == 0 && !c->sleep_active()) ? c->topology_epoch : 0xFFFFFFFFu;
        }
        if (largest == 0xFFFFFFFFu && c->last_fetch_step != c->step_index) {
            // no contacts and nothing with a shape: the step reads no counters at all and the islands are those of the
            // joints, fixed until the scene is edited - read this step's values once and keep them
            EH_TRY(fetch_counters(c, sizeof(Counters) - sizeof(uint32_t) * 8 * kMaxColours));
            largest = c->isl_cache_max = c->cnt_host->isl_max_items;
            c->isl_cache_jmax = c->cnt_host->isl_max_jitems; c->isl_cache_free = c->cnt_host->isl_free;
            c->isl_cache_epoch = c->topology_epoch;
        }
        c->isl_prep_step = c->step_index;
        isl_fused = !mixed && largest <= kIslFusedLimit;
        // The decision above came from the PREVIOUS step's islands. Islands can merge in one step (a rag doll falls onto the pile), and
        // a wave that finds itself with a 100 000-constraint island would need a large fraction of a second for it: confirm with this
        // step's sizes (one more counter fetch, ~15 us, only on the fused / mixed schedules and only when the lists were rebuilt)
        if ((isl_fused || mixed) && !keep_lists) {
            EH_TRY(fetch_counters(c, sizeof(Counters) - sizeof(uint32_t) * 8 * kMaxColours));
            if (isl_fused && c->cnt_host->isl_max_items > kIslFusedLimit) isl_fused = false;
            if (mixed && c->cnt_host->isl_max_jitems > kIslFusedLimit) {   // the jointed islands outgrew the fused kernels: the whole step per colour
                mixed = false; push = false;
                if (na) hipLaunchKernelGGL(k_prep_contacts<false>, dim3(blocks(na, 128)), dim3(128), 0, s, na, c->rows, rcap, mf, c->b, dt, c->col_keys_sorted, false, false, false);
            }
        }
    }
    if (push) {
        hipLaunchKernelGGL(k_push_links, dim3(blocks(na, 256)), dim3(256), 0, s, na, c->rows, c->col_keys_sorted, c->b, c->used, mixed ? c->isl_joint : nullptr);
    }
    const uint8_t *df_skip = mixed ? c->rows.skip : nullptr;
    IslSolveArgs isl_args{isl, c->cnt, j, c->b, c->rows, mf, rcap, c->extras ? c->rows.rwx : nullptr, 0u, c->isl_err, c->isl_done, mixed ? 1u : 0u};
    constexpr uint32_t kIslGrid = 4096;   // one wave each; a block takes islands blockIdx.x, + kIslGrid, ...
    rec(c, 5);
    uint32_t launches = 0;
    auto joints_pass = [&](bool warm) {
        for (uint32_t k = 0; k < j.num_colours; ++k) {
            uint32_t a = j.colour_start[k], e = j.colour_start[k + 1];
            if (e <= a) continue;
            if (warm) hipLaunchKernelGGL(k_joint_solve<true>, dim3(blocks(e - a, 128)), dim3(128), 0, s, a, e, j, c->b);
            else hipLaunchKernelGGL(k_joint_solve<false>, dim3(blocks(e - a, 128)), dim3(128), 0, s, a, e, j, c->b);
            ++launches;
        }
    };
    // maximal suffix of colours that each fit one workgroup -> one launch for all of them
    TailRanges tail{};
    uint32_t first_tail = nc_par;
    while (first_tail > 0 && c->colour_end[first_tail - 1] - c->colour_start[first_tail - 1] <= kTailMax) --first_tail;
    if (nc_par - first_tail >= 2) {
        for (uint32_t k = first_tail; k < nc_par; ++k)
            if (c->colour_end[k] > c->colour_start[k]) {
                tail.start[tail.n] = c->colour_start[k]; tail.end[tail.n] = c->colour_end[k];
                tail.split[tail.n] = Split{c->colour_split[k][0], c->colour_split[k][1], c->colour_split[k][2]};
                ++tail.n;
            }
    } else first_tail = nc_par;
    auto contacts_pass = [&](bool warm) {
        for (uint32_t k = 0; k < first_tail; ++k) {
            uint32_t a = c->colour_start[k], e = c->colour_end[k];
            if (e <= a) continue;
            const Rows &r = c->rows;
            const Split sp{c->colour_split[k][0], c->colour_split[k][1], c->colour_split[k][2]};
            const dim3 g(blocks(e - a, 64)), bl(64);
            if (push) hipLaunchKernelGGL(contact_solve_fn(warm, true, fused_rows), g, bl, 0, s, a, e, sp, (const uint32_t *)r.next, (const uint32_t *)nullptr, r.rw, rcap, r.dslot, (const float *)r.im, (float4 *)nullptr);
            else hipLaunchKernelGGL(contact_solve_fn(warm, false, fused_rows), g, bl, 0, s, a, e, sp, (const uint32_t *)r.bA, (const uint32_t *)r.bB, r.rw, rcap, c->b.dvw, (const float *)nullptr, c->extras ? r.rwx : (float4 *)nullptr);
            ++launches;
        }
        if (tail.n) {
            const Rows &r = c->rows;
            const dim3 g(1), bl(kTailThreads);
            if (push) hipLaunchKernelGGL(contact_tail_fn(warm, true, fused_rows), g, bl, 0, s, tail, (const uint32_t *)r.next, (const uint32_t *)nullptr, r.rw, rcap, r.dslot, (const float *)r.im, (float4 *)nullptr);
            else hipLaunchKernelGGL(contact_tail_fn(warm, false, fused_rows), g, bl, 0, s, tail, (const uint32_t *)r.bA, (const uint32_t *)r.bB, r.rw, rcap, c->b.dvw, (const float *)nullptr, c->extras ? r.rwx : (float4 *)nullptr);
            ++launches;
        }
        if (serial) {
            const Rows &r = c->rows;
            const uint32_t a = c->colour_start[kSerialColour], e = c->colour_end[kSerialColour];
            const Split sp{c->colour_split[kSerialColour][0], c->colour_split[kSerialColour][1], c->colour_split[kSerialColour][2]};
            hipLaunchKernelGGL(contact_serial_fn(warm, fused_rows), dim3(1), dim3(64), 0, s, a, e, sp, (const uint32_t *)r.bA, (const uint32_t *)r.bB, r.rw, rcap, c->b.dvw, c->extras ? r.rwx : (float4 *)nullptr);
            ++launches;
        }
    };
    bool df_velocity = false, df_two_lane = false, df_four_lane = false;
    if (push && c->df_mode == 1) {
        const Rows &r = c->rows;
        // Two lanes per manifold (k_contact_solve_df2) unless disabled; resident waves (measured on MI355X): enough for
        // ~4-5 tasks per wave and sweep while the sweep is latency-bound - more only add polling traffic - and up to every
        // resident slot once the row stream dominates (many islands, millions of points).
        // Lanes per manifold: 4 (k_contact_solve_df4: shortest hop; the default while the sweep is latency-bound), 2 (k_contact_solve_df2)
        // or 1 (k_contact_solve_df: least row traffic, for bandwidth-bound scenes - many islands, millions of points). EDYNHIP_DF_LANES forces one.
        static const uint32_t env_lanes = getenv("EDYNHIP_DF_LANES") ? (uint32_t)atoi(getenv("EDYNHIP_DF_LANES")) : 0u;
        static const bool two_lane_env = !(getenv("EDYNHIP_DF_TWOLANE") && getenv("EDYNHIP_DF_TWOLANE")[0] == '0');
        static const uint32_t env_waves = getenv("EDYNHIP_DF_WAVES") ? (uint32_t)atoi(getenv("EDYNHIP_DF_WAVES")) : 0u;
        // (the multi-lane forms read J_lin on two lanes: ~20 % more row traffic, which only matters once the sweep is
        // bandwidth-bound - then the one-lane kernel is the better one)
        const bool latency_bound = c->df2_waves > 0 && na <= 16u * 32u * c->df2_waves;
        // Four lanes were measured on the settled headline pile (r03): 541 instead of 633 instructions between "inputs arrived" and
        // "published" (1.50 vs 1.66 us per wave-task), but 16 manifolds per wave need two waves per SIMD for the same ~4.5 tasks per
        // wave and sweep, and the two contend for the issue slots exactly while the critical chain runs: 0.60 vs 0.564 ms per solve.
        // Two lanes stay the default; EDYNHIP_DF_LANES=4 selects the four-lane kernel (bit-identical).
        uint32_t lanes = latency_bound ? 2u : 1u;
        if (env_lanes == 1u || env_lanes == 2u || env_lanes == 4u) lanes = env_lanes;
        if (!two_lane_env && lanes > 1u) lanes = 1u;
        if (lanes == 4u && c->df4_waves == 0) lanes = 2u;
        if (lanes == 2u && c->df2_waves == 0) lanes = 1u;
        const bool two_lane = lanes == 2u, four_lane = lanes == 4u;
        df_two_lane = two_lane; df_four_lane = four_lane;
        const uint32_t per_wave = 64u / lanes;
        const uint32_t resident = four_lane ? c->df4_waves : two_lane ? c->df2_waves : c->df_lanes;
        const uint32_t want_waves = env_waves ? env_waves : std::max(four_lane ? 2048u : two_lane ? 1024u : 512u, blocks(na, per_wave * 9));
        const uint32_t grid = std::min(blocks(na, per_wave), std::min(resident, want_waves));
        static const uint32_t env_nap = getenv("EDYNHIP_DF_NAP") ? (uint32_t)atoi(getenv("EDYNHIP_DF_NAP")) : 1u;   // (r04 sweep on one box: 4 -> 1: +0.6 %, 0: the same)
        // (measured, round 5, A/B on one box: the headline pile LOSES - velocity solve 0.553 -> 0.597 ms: the class padding adds 6 % task slots to waves
        //  that were exactly as busy as the chains are long - mixed32k and pile8k win 1.5-3 % of their solves: off by default, EDYNHIP_DF_XCD=1 turns it on)
        static const bool xcd_env = getenv("EDYNHIP_DF_XCD") && getenv("EDYNHIP_DF_XCD")[0] == '1';
        const uint32_t xcd_lists = (xcd_env && two_lane && grid % kXcds == 0 && grid >= 8 * kXcds) ? 1u : 0u;
        DfArgs a{na, grid * per_wave, c->cfg.num_velocity_iterations + 1, c->col_keys_sorted, r.next, r.im, r.rw, rcap, r.dslot, c->cnt, nullptr, df_skip, env_nap, xcd_lists};
        // developer aid: EDYNHIP_DF_TRACE=<file> EDYNHIP_DF_TRACE_STEP=<n> dumps per-task timestamps of the n-th solve
        static const char *trace_path = getenv("EDYNHIP_DF_TRACE");
        static long trace_step = getenv("EDYNHIP_DF_TRACE_STEP") ? atol(getenv("EDYNHIP_DF_TRACE_STEP")) : 100, solve_calls = 0;
        const bool tracing = trace_path && solve_calls++ == trace_step;
        size_t trace_words = 0;
        if (tracing) a.xcd_lists = 0;   // (the trace is laid out by (sweep, round, wave) of the plain assignment)
        if (tracing) {
            const uint32_t rounds = (na + a.stride - 1) / a.stride;
            trace_words = 4 * (size_t)a.sweeps * rounds * (a.stride / per_wave);
            EH_HIP(c, hipMalloc((void **)&a.trace, trace_words * 8));
            EH_HIP(c, hipMemsetAsync(a.trace, 0, trace_words * 8, s));
        }
        void *params[] = {&a};
        // cooperative launch: the runtime guarantees that all `grid` workgroups are resident together, which the
        // hand-off polling relies on
        if (launch_resident(c, df_velocity_fn(lanes, fused_rows), grid, 64, params) == hipSuccess) {
            df_velocity = true;
            ++launches;
        } else {   // e.g. the device is shared and cannot hold the grid: use the per-colour schedule from now on
            (void)hipGetLastError();
            c->df_mode = 0;
            // the mixed schedule has no per-colour form for THIS step (the chains skip the jointed islands): give the step up,
            // the next one takes the per-colour launches
            if (mixed) return set_error(c, EDYNHIP_ERR_INTERNAL, "solve: the runtime refused the resident launch of the mixed schedule");
        }
        if (tracing && df_velocity) {
            std::vector<uint64_t> tr(trace_words); std::vector<uint32_t> keys(na);
            EH_HIP(c, hipStreamSynchronize(s));
            EH_HIP(c, hipMemcpy(tr.data(), a.trace, trace_words * 8, hipMemcpyDeviceToHost));
            EH_HIP(c, hipMemcpy(keys.data(), c->col_keys_sorted, (size_t)na * 4, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(trace_path, "wb")) {
                const uint32_t hdr[4] = {na, a.stride, a.sweeps, per_wave};
                fwrite(hdr, 4, 4, f); fwrite(keys.data(), 4, na, f); fwrite(tr.data(), 8, trace_words, f); fclose(f);
            }
            (void)hipFree(a.trace);
        } else if (tracing) (void)hipFree(a.trace);
    }
    if ((!df_velocity && isl_fused) || (df_velocity && mixed)) {   // mixed: the islands with joints, beside the dataflow launch
        isl_args.iters = c->cfg.num_velocity_iterations;
        if (fused_rows) hipLaunchKernelGGL(k_island_velocity<true>, dim3(kIslGrid), dim3(64), 0, s, isl_args);
        else hipLaunchKernelGGL(k_island_velocity<false>, dim3(kIslGrid), dim3(64), 0, s, isl_args);
        ++launches;
    } else if (!df_velocity) {
        joints_pass(true);
        contacts_pass(true);
        for (uint32_t it = 0; it < c->cfg.num_velocity_iterations; ++it) {
            joints_pass(false);
            contacts_pass(false);
        }
    }
    c->timings.solve_velocity_launches += launches;
    c->stats.solve_schedule = (na + j.n) == 0 ? EDYNHIP_SCHEDULE_NONE
                              : df_velocity ? (mixed ? EDYNHIP_SCHEDULE_MIXED : df_four_lane ? EDYNHIP_SCHEDULE_DATAFLOW4 : df_two_lane ? EDYNHIP_SCHEDULE_DATAFLOW2 : EDYNHIP_SCHEDULE_DATAFLOW1)
                              : isl_fused ? EDYNHIP_SCHEDULE_ISLAND_FUSED : EDYNHIP_SCHEDULE_PER_COLOUR;
    rec(c, 6);
    static const bool pos_df_env = !(getenv("EDYNHIP_DATAFLOW_POS") && getenv("EDYNHIP_DATAFLOW_POS")[0] == '0');
    const uint32_t P = c->cfg.num_position_iterations;
    // one error array per position iteration (DfPosArgs::err_out / err_prev): kMaxDfPosIters of them are allocated
    // (the dataflow position kernel hands positions and orientations over, not origins: worlds with centre-of-mass offsets solve positions per colour)
    const bool pos_df = P > 0 && P <= kMaxDfPosIters && push && c->df_mode == 1 && pos_df_env && !c->b.com;
    hipLaunchKernelGGL(k_integrate, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, dt, c->isl_err, c->isl_done, push ? c->rows.dslot : nullptr, c->rows.first_slot,
                       c->pos_err, pos_df ? P : 0u);
    if (na && !pos_df) hipLaunchKernelGGL(k_store_impulses, dim3(blocks(na, 256)), dim3(256), 0, s, na, c->rows, rcap, mf);
    rec(c, 7);
    const float4 *final_pslot = nullptr;   // set while the bodies' final transforms still live in the hand-off slots
    auto pos_per_colour = [&](uint32_t first_it) {
        for (uint32_t it = first_it; it < c->cfg.num_position_iterations; ++it) {
            for (uint32_t k = 0; k < j.num_colours; ++k) {
                uint32_t a = j.colour_start[k], e = j.colour_start[k + 1];
                if (e > a) hipLaunchKernelGGL(k_pos_joints, dim3(blocks(e - a, 128)), dim3(128), 0, s, a, e, j, c->b, c->isl_err, c->isl_done);
            }
            for (uint32_t k = 0; k < first_tail; ++k) {
                uint32_t a = c->colour_start[k], e = c->colour_end[k];
                if (e > a) hipLaunchKernelGGL(block_pos ? k_pos_contacts<true> : k_pos_contacts<false>, dim3(blocks(2 * (e - a), 128)), dim3(128), 0, s, a, e, c->rows, mf, c->b, c->isl_err, (const uint32_t *)c->isl_done);
            }
            if (tail.n) hipLaunchKernelGGL(block_pos ? k_pos_contacts_tail<true> : k_pos_contacts_tail<false>, dim3(1), dim3(256), 0, s, tail, c->rows, mf, c->b, c->isl_err, (const uint32_t *)c->isl_done);
            if (serial) hipLaunchKernelGGL(block_pos ? k_pos_contacts_serial<true> : k_pos_contacts_serial<false>, dim3(1), dim3(64), 0, s, c->colour_start[kSerialColour], c->colour_end[kSerialColour], c->rows, mf, c->b, c->isl_err, (const uint32_t *)c->isl_done);
            hipLaunchKernelGGL(k_pos_flags, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->isl_err, c->isl_done);
        }
    };
    if (pos_df) {
        static const uint32_t env_pw = getenv("EDYNHIP_DFP_WAVES") ? (uint32_t)atoi(getenv("EDYNHIP_DFP_WAVES")) : 0u;
        const Rows &r = c->rows;
        hipLaunchKernelGGL(k_pos_seed, dim3(blocks(2 * na, 256)), dim3(256), 0, s, na, r, c->b, r.pslot, rcap, mf, df_skip);
        // resident waves: 512 for the block form (round 4: 256 / 768 / 1024 / 2048 within 1 %), 1024 - every SIMD - for the point-by-point form, whose tasks
        // are longer (round 5, one box: 785-791 -> 796-797 steps/s; 384: 743)
        const uint32_t grid = std::min(blocks(na, 32), std::min(c->dfp_waves, env_pw ? env_pw : std::max(block_pos ? 512u : 1024u, blocks(na, 32 * 9))));
        // developer aid: EDYNHIP_DFP_TRACE=<file> EDYNHIP_DF_TRACE_STEP=<n> dumps per-task timestamps of the n-th position solve
        // (same file format as EDYNHIP_DF_TRACE with "sweeps" = position iterations: scripts/df_trace.py reads both)
        static const char *ptrace_path = getenv("EDYNHIP_DFP_TRACE");
        static long ptrace_step = getenv("EDYNHIP_DF_TRACE_STEP") ? atol(getenv("EDYNHIP_DF_TRACE_STEP")) : 100, pos_calls = 0;
        const bool ptracing = ptrace_path && pos_calls++ == ptrace_step;
        const uint32_t prounds = blocks(na, grid * 32u);
        const size_t ptrace_words = 4 * (size_t)prounds * grid;
        uint64_t *ptrace = nullptr;
        if (ptracing) {
            EH_HIP(c, hipMalloc((void **)&ptrace, ptrace_words * P * 8));
            EH_HIP(c, hipMemsetAsync(ptrace, 0, ptrace_words * P * 8, s));
        }
        uint32_t it = 0;
        for (; it < c->cfg.num_position_iterations; ++it) {
            static const bool xcd_env = getenv("EDYNHIP_DF_XCD") && getenv("EDYNHIP_DF_XCD")[0] == '1';
            const uint32_t xcd_lists = (xcd_env && !ptrace && grid % kXcds == 0 && grid >= 8 * kXcds) ? 1u : 0u;
            DfPosArgs a{na, grid * 32u, it, c->col_keys_sorted, r.next, r.pslot, r, mf, c->b, c->pos_err + (size_t)it * c->b.cap,
                        it ? c->pos_err + (size_t)(it - 1) * c->b.cap : nullptr, c->cnt, df_skip, ptrace ? ptrace + ptrace_words * it : nullptr, xcd_lists};
            void *params[] = {&a};
            if (launch_resident(c, df_position_fn(block_pos), grid, 64, params) != hipSuccess) {
                (void)hipGetLastError();
                c->df_mode = 0;
                break;
            }
        }
        if (ptrace) {
            std::vector<uint64_t> tr(ptrace_words * P); std::vector<uint32_t> keys(na);
            EH_HIP(c, hipStreamSynchronize(s));
            EH_HIP(c, hipMemcpy(tr.data(), ptrace, tr.size() * 8, hipMemcpyDeviceToHost));
            EH_HIP(c, hipMemcpy(keys.data(), c->col_keys_sorted, (size_t)na * 4, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(ptrace_path, "wb")) {
                const uint32_t hdr[4] = {na, grid * 32u, P, 32u};
                fwrite(hdr, 4, 4, f); fwrite(keys.data(), 4, na, f); fwrite(tr.data(), 8, tr.size(), f); fclose(f);
            }
            (void)hipFree(ptrace);
        }
        // the bodies' transforms live in the hand-off slots while the dataflow launches run; k_finish picks them up
        if (it == P) {
            final_pslot = r.pslot;
            if (mixed) {   // the islands with joints, on the body records
                isl_args.iters = P;
                hipLaunchKernelGGL(block_pos ? k_island_position<true> : k_island_position<false>, dim3(kIslGrid), dim3(64), 0, s, isl_args);
            }
        } else if (mixed) {
            return set_error(c, EDYNHIP_ERR_INTERNAL, "solve: the runtime refused the resident launch of the mixed schedule (position)");
        } else {   // a cooperative launch was refused: finish per colour (rare; never on an exclusive device)
            hipLaunchKernelGGL(k_pos_writeback, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, r.pslot, r.first_slot);
            if (it > 0) hipLaunchKernelGGL(k_pos_flags, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->pos_err + (size_t)(it - 1) * c->b.cap, c->isl_done);
            pos_per_colour(it);
        }
    } else if (P > 0 && (na || j.n)) {
        if (isl_fused) {
            isl_args.iters = P;
            hipLaunchKernelGGL(block_pos ? k_island_position<true> : k_island_position<false>, dim3(kIslGrid), dim3(64), 0, s, isl_args);
        } else pos_per_colour(0);
    }
    rec(c, 8);
    hipLaunchKernelGGL(k_finish, dim3(blocks(n, 256)), dim3(256), 0, s, n, c->b, c->used, c->m[c->cur ^ 1].seg_start, c->m[c->cur ^ 1].seg_end, c->cnt,
                       final_pslot, c->rows.first_slot, CandLists{c->bvh.cand_list, c->bvh.cand_count, c->bvh.ref_min, c->bvh.ref_max}, c->isl_joint, c->isl_top, c->meshes);
    rec(c, 9);
    ++c->step_index;
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

}  // namespace eh
