// Device-resident mirror of the EnTT component pools used by the step loop, and the per-step
// scratch. One context per GPU. All per-body / per-manifold data are "SoA of float4": one array per
// component (like an EnTT pool), 16 B elements, so body-parallel kernels stream coalesced and the
// solver's by-index gathers fetch a whole component in one transaction.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <array>
#include <map>
#include <vector>
#include "../../include/edynhip.h"
#include "dmesh.hpp"

namespace eh {

constexpr uint32_t kNoColour = 0xFFu;
constexpr uint32_t kMaxColours = 64;
constexpr uint32_t kMaxContactColours = 63;   // contact colours 0..62: (colour, point count) then fits an 8-bit sort key, 0xFF = inactive
// Colours 0..61 are conflict-free sets, solved in parallel. Colour 62 is the SERIAL bucket: a manifold that finds all of 0..61
// taken at one of its bodies (a body with more than 62 coloured contacts - a plate carrying a crowd) goes there, and the bucket is
// solved by one lane, one manifold after the other, after the parallel colours of every sweep. No contact count is an error.
constexpr uint32_t kSerialColour = 62;
constexpr int kMaxPts = 4;

// body flags
constexpr uint32_t BF_KIND_MASK = 0x3u;        // EDYNHIP_KIND_*
constexpr uint32_t BF_SHAPE_SHIFT = 4;         // EDYNHIP_SHAPE_* in bits 4..7
constexpr uint32_t BF_SHAPE_MASK = 0xF0u;
constexpr uint32_t BF_ASLEEP = 0x100u;         // sleeping_tag: the body's island is asleep (island sleeping, solver.hip k_sleep_*)
constexpr uint32_t BF_NOSLEEP = 0x200u;        // sleeping_disabled_tag
constexpr uint32_t BF_REMOVED = 0x400u;        // destroyed entity: the index stays reserved (edynhip_remove_bodies)
// An edge (manifold, joint) sleeps when every procedural endpoint sleeps (an island sleeps as a whole).
__host__ __device__ inline bool edge_asleep(uint32_t fa, uint32_t fb) {
    const bool da = (fa & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC, db = (fb & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC;
    return (da || db) && (!da || (fa & BF_ASLEEP)) && (!db || (fb & BF_ASLEEP));
}

struct Bodies {
    uint32_t n = 0, cap = 0;
    // Gathered state lives in cache-line sized RECORDS (the solver reads bodies by index; random 16-B accesses to
    // separate arrays were measured ~6x slower than one aligned record, scripts/ubench/solve_model.hip):
    //   xf[8*i + 0] pos xyz, w = mass_inv (dynamic) or 0      xf[8*i + 1] orn xyzw
    //   xf[8*i + 2..4] inertia_world_inv rows                   xf[8*i + 5..7] inertia_inv (local) rows      (128 B, one line)
    //   dvw[2*i + 0] delta_linvel xyz, w = effective inv mass (0 for non-procedural)   dvw[2*i + 1] delta_angvel   (32 B)
    float4 *xf = nullptr;
    float4 *dvw = nullptr;
    float4 *linvel = nullptr;   // xyz   (streamed, body-parallel kernels only)
    float4 *angvel = nullptr;   // xyz
    float4 *amin = nullptr;     // AABB min
    float4 *amax = nullptr;     // AABB max
    float4 *shape = nullptr;    // box half extents | sphere radius | plane normal+constant
    // center_of_mass / origin (comp/center_of_mass.hpp, comp/origin.hpp): nullptr while no body has an offset. `pos` is the centre of mass;
    // shapes, contact pivots and joint pivots live in the frame of origin = to_world(-com, pos, orn) (com.w != 0 marks a body that has one)
    float4 *com = nullptr, *origin = nullptr;
    float4 *grav = nullptr;     // per-body gravity
    float2 *mat = nullptr;      // friction, restitution
    float4 *mat2 = nullptr;     // contact_extras materials: spin_friction, roll_friction, stiffness, damping (comp/material.hpp:15-22)
    // material mix table (material_mixing.hpp:36-82), off while mix_K == 0: per body the compact index of its material id (~0u = none),
    // mix_lut[cA * mix_K + cB] = entry for the ORDERED pair (body[0]'s id, body[1]'s id) or -1 - built on the host with the
    // reference's own container semantics (a lookup with the ids the other way round does not always find its entry) -,
    // mix_vals[6 e ..] = restitution, friction, spin_friction, roll_friction, stiffness, damping.
    const uint32_t *mat_cid = nullptr; const int32_t *mix_lut = nullptr; const float *mix_vals = nullptr; uint32_t mix_K = 0;
    uint32_t *flags = nullptr;  // kind | shape << 4
    uint64_t *group = nullptr, *mask = nullptr;
    uint32_t *island = nullptr; // connected-component label (min body index)
};

// Contact manifolds, rebuilt every step in ascending canonical key order (double buffered).
struct Manifolds {
    uint32_t cap = 0;
    uint64_t *skey = nullptr;     // (owner<<32|other)<<1 | swapped   (swapped: body[0] == other), see broadphase.hip
    uint32_t *bodyA = nullptr, *bodyB = nullptr;
    uint32_t *info = nullptr;     // num_points | colour << 8
    uint32_t *seg_start = nullptr, *seg_end = nullptr;   // per body b: the manifolds it owns
    uint32_t *prev_idx = nullptr; // inside a full step: index of the same pair in the previous array (~0u = created now);
                                  // the narrowphase then reads the old points from there instead of a copy
    uint8_t *tree = nullptr;      // 1 = the island union-find hooked on this manifold: the marked manifolds (with the joints) are a
                                  // spanning forest of the contact graph, i.e. a certificate for the island labels - while none of
                                  // them disappears, no island can have split and the labels are updated incrementally (solver.hip)
    // per point slot k (list order, newest first), five float4 at index pt_at(cap, k, m) of five INTERLEAVED base pointers (round 6: one
    // 320-byte record per manifold - pA pB nrm lnrm imp of point 0, then of point 1, ... - so a lane that reaches a manifold through an index
    // (the row preparation through the colour-sorted order, the narrowphase through prev_idx) pulls 3 lines instead of 20: a 16-byte gather
    // costs a whole 128-byte line, scripts/ubench/gather.hip, points.hip)
    float4 *pA = nullptr;         // pivotA xyz, w = distance
    float4 *pB = nullptr;         // pivotB xyz, w = friction
    float4 *nrm = nullptr;        // normal xyz, w = bitcast(attachment)
    float4 *lnrm = nullptr;       // local_normal xyz, w = restitution
    float4 *imp = nullptr;        // normal_impulse, friction_impulse[0], [1], bitcast(lifetime)
    // slot-major, index slot_at(cap, k, m) = k*cap + m:
    uint64_t *pid = nullptr;      // contact events only (else nullptr): the point's id from creation to destruction
    // contact_extras only (else nullptr; allocated when a body gets such a material, edynhip_set_material_extras)
    float4 *xmat = nullptr;       // mixed at creation: roll_friction, spin_friction, stiffness, damping (contact_point_material)
    float4 *ximp = nullptr;       // rolling_friction_impulse[0], [1], spin_friction_impulse, -
};
constexpr bool kPointRecords = true;   // false: the slot-major layout of rounds 1-5 (pA[k*cap + m], ...), kept for A/B runs
constexpr int kPointF = 5;             // float4 per contact point: pA pB nrm lnrm imp
__host__ __device__ __forceinline__ size_t pt_at(uint32_t cap, uint32_t k, uint32_t m) {
    return kPointRecords ? ((size_t)m * 4 + k) * kPointF : (size_t)k * cap + m;
}
__host__ __device__ __forceinline__ size_t slot_at(uint32_t cap, uint32_t k, uint32_t m) { return (size_t)k * cap + m; }
// Extras rows of a point (contact_extras_constraint.cpp:37-78), angular only: J = {0, axis, 0, -axis}. Per point 10 float4 at
// rwx[(k * kXPoint + slot) * cap + p]: rows roll0, roll1, spin as (axis, eff) (I_A^-1 axis, rhs) (I_B^-1 (-axis), impulse) in
// slots 3r..3r+2, and slot 9 = (roll mu, spin mu, -, -); mu = 0: the row does not exist.
constexpr int kXRowF = 3, kXRows = 3, kXPoint = kXRows * kXRowF + 1;

__device__ __forceinline__ const float *mix_lookup(const Bodies &b, uint32_t body0, uint32_t body1) {
    if (b.mix_K == 0) return nullptr;
    const uint32_t c0 = b.mat_cid[body0], c1 = b.mat_cid[body1];
    if (c0 == 0xFFFFFFFFu || c1 == 0xFFFFFFFFu) return nullptr;
    const int32_t e = b.mix_lut[(size_t)c0 * b.mix_K + c1];
    return e < 0 ? nullptr : b.mix_vals + 6 * (size_t)e;
}

// Contact events (EDYNHIP_FLAG_CONTACT_EVENTS; edynhip.h edynhip_contact_event has the same layout).
struct ContactEvent { uint32_t type, step, bodyA, bodyB; uint64_t pid; };
struct EventSink { ContactEvent *buf; uint32_t *count; uint32_t cap; uint32_t step; };   // buf == nullptr: events are off
__device__ __forceinline__ void emit_event(const EventSink &ev, uint32_t type, uint32_t a, uint32_t b, uint64_t pid) {
    const uint32_t i = atomicAdd(ev.count, 1u);   // may run past cap: the host reports the overflow
    if (i < ev.cap) ev.buf[i] = ContactEvent{type, ev.step, a, b, pid};
}

// Joint rows live in SLOTS that mirror the constraints' applied_impulse fields (hinge_constraint.hpp:64-71,
// point_constraint.hpp:28-29): hinge 0..2 linear, 3..4 hinge p/q, 5 limit, 6 bump stop, 7 spring, 8 torque; point 0..2 linear,
// 3 friction torque. Which optional slots carry a row this step is decided by k_prep_joints (rmask).
// 24 slots / 64 parameters are the generic constraint's (6 degrees of freedom x 4 row kinds / 10 floats); the other types use the
// first 9 / 16. edynhip_joints.params and edynhip_set_joint_params carry the first 10 parameters, edynhip_get_joint_impulses the
// first 9 slots.
constexpr int kJointSlots = 24, kJointParams = 64, kJointApiParams = 10, kJointApiSlots = 9, kJointBaseSlots = 9;
struct Joints {
    uint32_t n = 0, cap = 0, num_colours = 0, rows = 0;
    // definitions, in colour-sorted order; orig[] maps back to the caller's index
    uint32_t *orig = nullptr;
    uint32_t *type = nullptr, *bodyA = nullptr, *bodyB = nullptr;
    float4 *pivA = nullptr, *pivB = nullptr;       // object-space pivots
    float4 *axA = nullptr;                          // frame[0] column 0 (hinge axis on A)
    float4 *pA = nullptr, *qA = nullptr;            // frame[0] columns 1, 2
    float4 *axB = nullptr, *pB = nullptr;           // frame[1] columns 0, 1
    float *params = nullptr;                        // [kJointParams][cap]: hinge angle_min, angle_max, limit_restitution, bump_stop_angle,
                                                    // bump_stop_stiffness, torque, speed, rest_angle, stiffness, damping; point: friction_torque
    float *angle = nullptr;                         // [cap] hinge angle tracked across wraps (hinge_constraint.cpp:80-89)
    float *impulse = nullptr;                       // [kJointSlots][cap] applied impulses (warm start)
    // rows (per step)
    float4 *rA = nullptr, *rB = nullptr;            // lever arms
    float4 *wp = nullptr, *wq = nullptr;            // world hinge p, q
    float4 *wax = nullptr;                          // world axis of the optional rows (hinge axis / relative spin direction)
    float4 *wbx = nullptr;                          // cvjoint: axis of the bend-spring row (slot 8)
    float4 *gJ = nullptr;                           // generic: [18][cap] per degree of freedom d three vectors at (3 d + v) * cap + i -
                                                    // linear: axis, rA x axis, rB x axis; angular: axis on A, axis on B, -
    float *eff = nullptr, *rhs = nullptr;           // [kJointSlots][cap]
    float *lo = nullptr, *hi = nullptr;             // [kJointSlots][cap] impulse limits of the optional rows
    uint32_t *rmask = nullptr;                      // [cap] slots that carry a row this step
    uint32_t colour_start[kMaxColours + 1] = {0};   // host copy
};
// Host mirror of the joint definitions by CALLER index (stable; a removed joint stays as a dead entry): the device arrays are
// rebuilt from it - colouring included - whenever joints are added or removed, carrying the applied impulses over.
struct HostJoint {
    int32_t type = 0;
    uint32_t body[2] = {0, 0};
    float pivot[6] = {0}, axis[6] = {0};
    float frame[18] = {0};        // cone / cvjoint: frames A and B, row-major 3x3 (edynhip_set_joint_definition)
    bool has_frames = false;
    float params[kJointParams] = {0};
    float impulse[kJointSlots] = {0};
    float angle = 0;
    bool alive = true;
};

// Solver rows in colour-sorted order p. The per-colour solve kernels are latency / issue bound (a colour of
// a 32k-box pile is only ~10k lanes, one wave per CU), so everything that does not depend on the evolving
// body deltas is precomputed once per step by k_prep_contacts: Jacobian blocks AND the angular impulse
// directions I^-1 J^T. Every value is the result of the same fp32 operations the reference performs inside
// its solve loop, so results stay bit-identical.
// Row r of point slot k of lane p lives at rw[((k*3 + r)*5 + f) * cap + p], r = 0 normal, 1/2 friction tangents:
//   f=0: J_lin.xyz (n or t; J[2] = -J_lin), eff_mass     f=1: J_angA.xyz, rhs      f=2: J_angB.xyz, impulse
//   f=3: inv_IA * J_angA .xyz, mu (normal row only)      f=4: inv_IB * J_angB .xyz
struct Rows {
    uint32_t *order = nullptr;    // p -> manifold index
    uint32_t *bA = nullptr, *bB = nullptr, *np = nullptr;
    uint32_t *label = nullptr;    // island label of the manifold (p-indexed copy: keeps the position kernel's load chain short)
    float4 *rw = nullptr;
    float4 *rwx = nullptr;        // contact_extras rows (kXPoint float4 per point), nullptr unless the world has such materials
    // "Push" hand-off of the body deltas between consecutive manifolds of a body (used when the scene has no joints):
    // lane p reads the deltas of its two bodies from its OWN slots dslot[2*(2p+side) + {0,1}] (coalesced, no dependent
    // gather) and writes the updated deltas into the slots of each body's next manifold in colour order (cyclic).
    float4 *dslot = nullptr;      // [2 sides][2 float4] per lane p
    uint32_t *next = nullptr;     // [2p + side] -> slot (2p' + side') of the same body's next manifold in the sweep;
                                  // bit 31 (kHeadBit) marks THIS slot as the head of its body's chain
    float4 *pslot = nullptr;      // position-solve hand-off slots, 3 float4 per (lane, side): see k_pos_contacts_df
    float *im = nullptr;          // [2p + side] inverse mass of that side's body (0 = read-only body: no hand-off)
    // what the dataflow position solve reads, indexed by the lane like the rows (written by k_prep_contacts): with it the points of a
    // position task need no gather through the manifold index - their addresses follow from p
    float4 *pw = nullptr;         // [(k * kPosF + f) * cap + p]: point k's pivot on A, pivot on B, local normal, (normal, attachment)
    uint32_t *slot_of = nullptr;  // [body * 64 + colour] -> slot, scratch for building `next`
    uint32_t *first_slot = nullptr;   // per body: slot of its lowest-colour manifold (where a sweep leaves its deltas), or ~0
    uint8_t *skip = nullptr;          // [p] mixed schedule: the manifold's island has joints (solved by the island-fused kernels, not on the chains)
};
constexpr int kRowF = 5, kRowsPerPoint = 3, kPosF = 4;
constexpr uint32_t kColUncCap = 16384;   // uncoloured edges one workgroup colours by itself (more: the multi-block rounds)
constexpr uint32_t kHeadBit = 0x80000000u, kSlotMask = 0x7FFFFFFFu;
// Hand-off slot addressing (float4 units). A slot is (lane p, side); its pieces are laid out so that ONE vector load of a
// wave - 64 consecutive p for the velocity slots, 32 consecutive p x 2 sides for the position slots - reads contiguous
// memory: the consumers poll their slots with device-coherent loads that go to the fabric every time, and a poll that
// touches each 128-byte line once instead of four times is a quarter of the traffic.
__host__ __device__ inline size_t dslot_at(uint32_t slot, uint32_t half) {   // velocity: pieces (side, half) = 4 per p
    const uint32_t p = slot >> 1, side = slot & 1u;
    return ((size_t)(p >> 6) << 8) + (((side << 1) + half) << 6) + (p & 63u);
}
__host__ __device__ inline size_t pslot_at(uint32_t slot, uint32_t piece) {   // position: 3 pieces per (p, side)
    const uint32_t p = slot >> 1;
    return (size_t)(p >> 5) * 192u + (piece << 6) + (slot & 63u);
}

struct LBVH {
    uint64_t *keys = nullptr, *keys_sorted = nullptr;   // morton<<32 | body
    uint32_t *parent = nullptr;     // [2n-1]: internal 0..n-2, leaves n-1..2n-2
    uint32_t *left = nullptr, *right = nullptr;   // internal nodes
    uint32_t *rope = nullptr;       // [2n-1] stackless traversal: the node that follows this subtree in a depth-first walk
    uint32_t *split = nullptr;      // [8] the subtrees three levels below the root: the walk's lanes per body (broadphase.hip k_bp_split)
    // candidate lists (broadphase.hip "Verlet lists"): per body its possible partners within a fat margin, and the AABBs
    // they were built from; valid until a body strays from its ref box
    uint32_t *cand_list = nullptr, *cand_count = nullptr;
    float4 *ref_min = nullptr, *ref_max = nullptr;
    bool lists_dirty = true;        // the host changed the set of bodies: rebuild at the next step
    // The lists' look-ahead in steps (the slack a body's list is built with = kListSlack + look-ahead x dt x speed + gravity's share): 6 for
    // scenes whose lists live for many steps; halved (down to 1.5) while the lists are rebuilt in almost every step anyway - a heap in free
    // fall: the slack then only makes every list longer (round 6: polyheap32k, 58 candidates per body and a quarter of the bodies beyond a
    // list's 128 entries at look-ahead 6) - and doubled again once they survive. Any value keeps the pair set exact (k_finish's check).
    float lookahead = 6.0f;
    uint32_t rebuild_hist = 0;      // bit k: the lists were rebuilt k steps ago
    float4 *nmin = nullptr, *nmax = nullptr;      // node boxes [2n-1]
    uint32_t *visit = nullptr;      // refit counters
    uint32_t *np_list = nullptr;    // shaped non-procedural bodies
    uint32_t num_proc = 0, num_np = 0;
    uint32_t age = 0;               // steps since the tree topology was last rebuilt (0 = rebuild now)
};

// Device counters / flags block, mirrored to pinned host memory once or twice per step.
// Broadphase candidate lists (broadphase.hip); k_finish checks them at the end of a step.
struct CandLists { uint32_t *list; uint32_t *count; float4 *ref_min, *ref_max; };   // ref_min.w = the body's slack
constexpr uint32_t kListCap = 128;                // candidates kept per body (broadphase.hip); a body with more walks the tree every step
constexpr int kOwnCap = 64;                       // pair partners an owner keeps in its in-kernel list (broadphase.hip k_bp_pairs); more go through the sorted fallback path
constexpr uint32_t kListOverflow = 0xFFFFFFFFu;   // cand_count value of a body with more candidates than a list holds

constexpr uint32_t kMaxDfPosIters = 8;   // more position iterations than this run on the per-colour schedule

struct Counters {
    uint32_t num_pairs;          // pairs emitted this step
    uint32_t pair_overflow;
    uint32_t num_points;
    uint32_t num_active;         // manifolds with points
    uint32_t uncoloured;         // edges still lacking a colour
    uint32_t colour_overflow;
    uint32_t num_islands;
    uint32_t pairs_changed;      // this step's pair set differs from the previous step's
    uint32_t num_found;          // new manifolds that already existed last step (== previous count <=> none removed)
    uint32_t num_new;            // manifolds created this step (their body pairs are listed in new_edges)
    uint32_t num_extra;          // pair keys beyond an owner's in-LDS list (broadphase fallback path)
    uint32_t num_awake;          // procedural bodies left awake by this step's sleep decisions (island sleeping)
    uint32_t unc_count;          // edges k_col_prepare found uncoloured (listed in col_unc while they fit)
    uint32_t bp_rebuild;         // this step rebuilds the broadphase candidate lists (set by the previous step's k_finish, cleared by k_bp_compact)
    uint32_t df_abort;           // the dataflow solve kernel gave up waiting for a hand-off (never expected; reported as an error)
    uint32_t pairs_differ;       // this step's sorted pair keys differ from the previous step's manifold array (k_bp_compact); 0 = the step runs in place
    uint32_t isl_num;            // island-fused schedule: islands that have constraints this step (solver.hip k_isl_fill)
    uint32_t isl_max_items;      //   and the largest of them, in constraints (read by the NEXT step's schedule decision)
    uint32_t isl_max_jitems;     //   the largest island that has joints
    uint32_t isl_free;           //   active manifolds in islands without joints (mixed schedule: those go to the dataflow launch)
    uint32_t tree_found;         // forest-certificate manifolds (Manifolds::tree) of the previous array that this step's pair set keeps (k_bp_pairs)
    uint32_t tree_marks;         // manifolds marked by this step's island update
    uint32_t tree_total;         // marked manifolds in the current array (NOT reset per step): tree_found == tree_total <=> no island can have split
    uint32_t col_wg_rounds;      // rounds k_col_rounds ran this step (edynhip_stats::colour_rounds counts them with the multi-block rounds)
    uint32_t bp_rebuilt;         // the candidate lists were rebuilt in this step on the device's own flag (k_bp_compact copies bp_rebuild here before clearing it): the host adapts the lists' look-ahead to it
    int32_t bounds_min[3], bounds_max[3];   // ordered-int encoded floats
    // sorted-order ranges per (colour, point count): key = colour*4 + (4 - num_points)
    uint32_t colour_start[4 * kMaxColours], colour_end[4 * kMaxColours];
};

// Per-step stage events; resolved lazily by edynhip_get_timings so that timing adds no host sync.
struct StageTimer {
    static constexpr int kEvents = 11;
    std::vector<hipEvent_t> ev;     // kEvents per recorded step
    uint32_t recorded = 0;          // steps recorded since the last edynhip_step call
    uint32_t capacity = 0;          // steps for which events exist
    hipEvent_t *e = nullptr;        // events of the step being recorded (nullptr = not recording)
    uint32_t mask = 0;              // which of the kEvents are recorded (an event record idles the GPU ~6 us)
};

#define B_POS(b, i) ((b).xf[8 * (size_t)(i)])
#define B_ORN(b, i) ((b).xf[8 * (size_t)(i) + 1])
#define B_IW(b, i, r) ((b).xf[8 * (size_t)(i) + 2 + (r)])
#define B_IL(b, i, r) ((b).xf[8 * (size_t)(i) + 5 + (r)])
// where a body's shape and pivots are anchored: its origin if it has a centre-of-mass offset, else its position
#define B_ORG(b, i) (((b).origin && (b).com[(i)].w != 0.0f) ? from4((b).origin[(i)]) : from4(B_POS(b, i)))
#define B_DV(b, i) ((b).dvw[2 * (size_t)(i)])
#define B_DW(b, i) ((b).dvw[2 * (size_t)(i) + 1])

}  // namespace eh

struct edynhip_ctx {
    edynhip_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    eh::Bodies b;
    eh::Manifolds m[2];
    int cur = 0;                   // index of the current manifold buffer
    uint32_t num_manifolds = 0;    // in m[cur]
    eh::Joints j;
    std::vector<eh::HostJoint> host_joints;   // by caller index (see HostJoint)
    std::vector<uint32_t> pending_redefs;     // joints edited since the device arrays were built (capi.hip flush_joint_redefs)
    eh::Rows rows;
    eh::LBVH bvh;
    float4 *np_ra = nullptr, *np_rb = nullptr, *np_rn = nullptr;   // narrowphase staging: the raw collide() result per manifold
    uint32_t *np_rnum = nullptr;
    uint64_t *pair_keys = nullptr, *pair_keys_sorted = nullptr;
    uint64_t *own_keys = nullptr;                                  // [body][kOwnCap] sorted pair keys of each owner (broadphase.hip)
    uint32_t *own_count = nullptr, *own_offset = nullptr;          // [bodies + 1]
    uint2 *new_edges = nullptr;    // body pairs of manifolds created this step (incremental island update)
    uint32_t *new_edge_m = nullptr; //   and their manifold indices
    uint32_t prev_num_manifolds = 0;
    uint32_t *col_keys = nullptr, *col_keys_sorted = nullptr;   // colour sort (counting sort, solver.hip k_cs_*)
    uint32_t *cs_hist = nullptr, *cs_start = nullptr;            // [256 keys][blocks of 1024 manifolds]
    uint32_t *cs_sup = nullptr;    // [16][256 keys] key counts of every 16 blocks (direct colour sort, solver.hip k_cs_*)
    uint32_t col_lds_edges = 0;   // listed edges k_col_rounds holds in LDS (set at the first colouring)
    uint64_t cc_full_steps = 0, cc_incremental_steps = 0;   // how the island labels were brought up to date, step by step (EDYNHIP_TREE_STATS)
    bool coop_launched = false;   // this context has made a cooperative launch (solver.hip launch_resident)
    uint32_t col_lds_bytes = 0;   // dynamic LDS of that launch: the edges and the hashed mark table
    uint32_t *col_unc = nullptr;                                 // this step's uncoloured edges (k_col_rounds), kColUncCap entries
    uint64_t *used = nullptr;      // per body: colours in use
    uint2 *isl_top = nullptr;      // per island label: (highest colour carried into this step + 1, has an edge to colour) - k_col_tops; cleared like `used`
    uint64_t *best[2] = {nullptr, nullptr};
    float *pos_err = nullptr;      // dataflow position solve: [iteration][island label] max error of that iteration (zeroed by k_integrate)
    float4 *com_store = nullptr, *origin_store = nullptr;   // backing of Bodies::com / origin (attached to `b` once a body has an offset)
    uint32_t *isl_cnt = nullptr, *isl_off = nullptr, *isl_list = nullptr, *isl_items = nullptr, *isl_sorted = nullptr, *isl_joint = nullptr;   // island-fused schedule (solver.hip IslLists)
    uint32_t isl_prep_step = 0xFFFFFFF0u;   // step_index of the last step that bucketed its constraints by island (its isl_max_items reaches cnt_host with the next step's fetch)
    uint32_t last_fetch_step = 0xFFFFFFF0u; // step_index during which fetch_counters last ran
    uint32_t topology_epoch = 0;            // bumped when bodies or joints are added / removed / redefined
    uint32_t isl_lists_epoch = 0xFFFFFFFFu;   // topology_epoch the island lists were built for, while they can be kept (joints only, no contacts, no sleeping)
    uint32_t isl_cache_epoch = 0xFFFFFFFFu, isl_cache_max = 0, isl_cache_free = 0, isl_cache_jmax = 0;   // largest island of a scene whose steps fetch no counters (no shapes, no contacts)
    float *isl_err = nullptr;      // per island label: max position error (as uint bits)
    uint32_t *isl_done = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    eh::Counters *cnt = nullptr;   // device
    eh::Counters *cnt_host = nullptr;   // pinned, device-visible: k_publish_counters writes it, the host spins on cnt_seq
    volatile uint32_t *cnt_seq = nullptr;   // pinned: sequence number of the last published counters
    uint32_t cnt_seq_next = 0;
    eh::StageTimer timer;
    edynhip_timings timings{};
    edynhip_stats stats{};
    uint32_t num_colours = 0;
    uint32_t colour_start[eh::kMaxColours] = {0}, colour_end[eh::kMaxColours] = {0};
    uint32_t colour_split[eh::kMaxColours][3] = {{0}};   // ends of the 4-, 3- and 2-point groups inside each colour's range
    uint32_t num_active = 0;
    std::vector<void *> allocs;
    uint32_t *idx_scratch = nullptr; size_t idx_scratch_cap = 0;   // index lists of the editing entry points (capi.hip index_scratch)
    bool clears_primed = false;    // the previous call ended with k_finish, which pre-clears the next step's scratch
    bool full_step = false;        // inside edynhip_step (all stages back to back): per-step clears are folded into kernels
    float *state_dev = nullptr;    // [max_bodies][13] staging of the packed state (get/set_state run every update in the C++ shim)
    float *state_host = nullptr;   // pinned mirror
    bool sleeping = false;         // EDYNHIP_FLAG_SLEEPING
    bool all_asleep = false;       // the last step left every procedural body asleep and nothing was edited since: steps are no-ops
    bool solve_begin_done = false; // this step's k_cc_flatten already did k_solve_begin's work (solver.hip islands())
    bool has_generic = false;      // some joint is a generic_constraint (k_prep_generic runs)
    bool has_cylinder = false;     // some body is a cylinder_shape (narrowphase.hip k_np_detect_ext runs)
    // convex meshes and polyhedron bodies (mesh.hip)
    bool has_polyhedron = false;   // some body is a polyhedron_shape (k_update_rotated + k_np_detect_poly run)
    struct HostMeshes {
        std::vector<dc::MeshDesc> desc;
        std::vector<float4> vertices, normals, edge_vertices, edge_normals, relevant_normals;
        std::vector<uint32_t> face_first, edge_vidx, edge_faces, relevant_faces, relevant_edges, nb_start, nb_idx;
        void swap(HostMeshes &o) { std::swap(*this, o); }
    } host_meshes;
    dc::Meshes meshes;                       // the same tables on the device (+ the rotated-mesh buffer)
    std::vector<void *> mesh_allocs;
    float4 *rot = nullptr; size_t rot_cap = 0, rot_used = 0;   // rotated meshes of the polyhedron bodies
    uint32_t *rot_off = nullptr;             // per body: its slice of `rot` (~0u: none)
    std::vector<uint32_t> host_rot_off;
    uint32_t *poly_work = nullptr;           // narrowphase.hip PolyBins: bin counters + per-manifold keys + the binned manifold list
    // material mix table, host side: the reference's container (std::map under unordered_pair's comparator) so that lookups behave
    // exactly like its own; ids by body; device buffers are rebuilt on every change (rebuild_mix_table, capi.hip)
    struct MixIdPair { uint32_t first, second; };
    struct MixIdPairLess {
        bool operator()(const MixIdPair &a, const MixIdPair &b) const {   // core/unordered_pair.hpp:32-40
            if (a.first == b.second && a.second == b.first) return false;
            if (a.first == b.first) return a.second < b.second;
            return a.first < b.first;
        }
    };
    std::map<MixIdPair, std::array<float, 6>, MixIdPairLess> host_mix;
    std::vector<uint32_t> host_mat_id;   // per body, 0xFFFF = material::UnassignedID
    uint32_t *d_mat_cid = nullptr; int32_t *d_mix_lut = nullptr; float *d_mix_vals = nullptr; uint32_t mix_lut_cap = 0, mix_vals_cap = 0;
    bool extras = false;           // some body carries a contact_extras material: extras storage exists, per-colour schedule
    uint32_t step_index = 0;       // completed steps
    // contact events: device list of the current edynhip_step call, per-manifold "still there" marks of the previous array
    eh::ContactEvent *events = nullptr; uint32_t *event_count = nullptr; uint32_t event_cap = 0; uint8_t *prev_matched = nullptr;
    // double-buffered read-back (edynhip_snapshot): pinned host copies of the packed state, the event that completes each
    float *snap_host[2] = {nullptr, nullptr}; float *snap_dev[2] = {nullptr, nullptr}; hipEvent_t snap_event[2] = {nullptr, nullptr};
    uint32_t snap_step[2] = {0, 0}, snap_bodies[2] = {0, 0}; int snap_last = -1; hipStream_t snap_stream = nullptr; hipEvent_t snap_ready = nullptr;
    // Bodies that CAN sleep (dynamic, not sleeping_disabled) ever uploaded (removals are not subtracted: an upper bound). While it is zero
    // the island-sleeping stage has nothing to decide - every island carries SL_DISABLED (the reference's exclude_sleeping_disabled views
    // skip such entities, island_manager.cpp:573-623) - and its kernels are not launched: a benchmark scene, sleeping_disabled on every
    // body as SURVEY 8(d) prescribes, then steps like a context created without EDYNHIP_FLAG_SLEEPING.
    uint32_t num_sleepable = 0;
    bool sleep_active() const { return sleeping && num_sleepable > 0; }
    unsigned long long *pp_prof_dev = nullptr; int pp_prof_calls = 0;   // EDYNHIP_PP_PROF (narrowphase.hip): phase ticks of the polyhedron-pair kernels
    bool island_labels_valid = false;   // b.island / sleep_since hold what an island stage (or edynhip_set_sleep_timers) wrote - not before the first step of a new scene
    uint32_t excl_dirty_lo = 0xFFFFFFFFu, excl_dirty_hi = 0;   // body rows of host_excl edited since the last upload (capi.hip flush_exclusions)
    // contact-event prefetch (edynhip_set_event_prefetch): the event list of a step call copied to pinned memory as soon as the last step's
    // narrowphase has run - the caller turns it into registry entities while the solve is still running
    uint32_t evp_max = 0; uint8_t *evp_host = nullptr; hipEvent_t evp_np_done = nullptr, evp_ready = nullptr; bool evp_now = false; int evp_state = 0;   // 0 none, 1 copy enqueued, 2 the call ran no step
    // record snapshots (edynhip_snapshot_records, ABI 15): per slot one device block and its pinned host copy, laid out as
    // [header 64 B | events: rec_event_cap x 24 B | records: bodies x 96 B]; the registry write-back reads the pinned copy in place
    uint8_t *rec_host[2] = {nullptr, nullptr}; uint8_t *rec_dev[2] = {nullptr, nullptr}; hipEvent_t rec_event[2] = {nullptr, nullptr};
    uint32_t rec_step[2] = {0, 0}, rec_bodies[2] = {0, 0}, rec_events_copied[2] = {0, 0}; int rec_last = -1; uint32_t rec_event_cap = 0;
    // Island sleep timers run on the step time stamps the stepper hands to the island manager (stepper_sequential.cpp:60-75,
    // island_manager.cpp:533-539,605-623): sim_clock is the island manager's m_last_time, i.e. the stamp of the PREVIOUS step while
    // a step runs - advanced by fixed_dt per edynhip_step step, set by the caller in edynhip_step_timed (the
    // max_steps_per_update clamp stretches the stamps, not the integration dt).
    double sim_clock = 0;
    // edynhip_set_pair_filter: the user's should_collide predicate (host callback, asked for new candidate pairs: broadphase.hip) and
    // host mirrors of the collision groups / masks for edynhip_default_should_collide
    int (*pair_filter)(void *, uint32_t, uint32_t) = nullptr;
    void *pair_filter_user = nullptr;
    uint32_t *filter_new_idx = nullptr;            // [max_manifolds] positions in the sorted pair list of this step's new candidates / of the rejected ones
    std::vector<uint64_t> host_group, host_mask;
    uint32_t *sleep_state = nullptr, *sleep_action = nullptr;   // per island label: reduction bits / decision
    double *sleep_since = nullptr;                              // per island label: stamp at which its timer started, < 0 = not running
    // the timer that survives an island merge (solver.hip k_sleep_sizes / k_sleep_carry): last step's labels and body count, sizes of last
    // step's islands, the biggest candidate per new label (size << 32 | ~old label), the carried stamps
    uint32_t *sleep_old_label = nullptr, *sleep_size = nullptr, sleep_prev_n = 0;
    unsigned long long *sleep_best = nullptr;
    double *sleep_carried = nullptr;
    int df_mode = -1;              // dataflow velocity solve: -1 = not probed yet, 0 = unavailable/disabled, 1 = in use
    uint32_t df_lanes = 0;         // resident waves of the dataflow velocity kernel
    uint32_t dfp_waves = 0;        // resident waves of the dataflow position kernel
    uint32_t df2_waves = 0;        // resident waves of the two-lane dataflow velocity kernel
    uint32_t df4_waves = 0;        // resident waves of the four-lane dataflow velocity kernel
    bool inplace_step = false;     // broadphase found last step's pair set again and kept the manifold array: islands() has nothing to relabel
    bool points_in_prev = false;   // this step's manifold array holds no copied points yet (see Manifolds::prev_idx)
    bool force_islands = true;     // recompute island labels even if the pair set did not change
    // restitution solver (restitution.hip): on when any body has restitution > 0
    bool has_restitution = false;
    uint32_t restitution_iterations = 8, individual_restitution_iterations = 3;   // settings.hpp:28-29
    uint32_t *rdeg = nullptr, *roff = nullptr, *rcursor = nullptr, *radj = nullptr, *rvisited = nullptr, *rqnext = nullptr;
    uint8_t *rstar = nullptr;
    uint64_t *rbest = nullptr;
    float4 *rimp = nullptr;
    uint32_t *excl = nullptr;      // collision exclusion lists [max_bodies][16], ~0u-terminated; allocated by the first edynhip_exclude_collision
    std::vector<uint32_t> host_excl;   // host mirror of excl (edits are rare: scene construction)
    std::vector<int32_t> host_kind, host_shape;   // per body, for rebuilding the broadphase lists when bodies are appended
};

namespace eh {
// stage entry points (each .hip file implements its own)
int broadphase(edynhip_ctx *c);
int narrowphase(edynhip_ctx *c);
int count_points(edynhip_ctx *c);
int fetch_counters(edynhip_ctx *c, size_t bytes);   // device counters -> cnt_host, waits for them (capi.hip)
int publish_counters(edynhip_ctx *c, size_t bytes, uint32_t *ticket);   // the same in two halves: enqueue the copy ...
int wait_counters(edynhip_ctx *c, uint32_t ticket);                      // ... and wait for it (speculative launches go in between)
int scan_u32(edynhip_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n);   // exclusive prefix sum
int debug_collide(edynhip_ctx *c, uint32_t n, const int32_t *st, const float *sp, const float *pos, const float *orn, float threshold,
                  float *out, uint32_t *count);
int islands(edynhip_ctx *c);
inline EventSink event_sink(const edynhip_ctx *c) { return EventSink{c->events, c->event_count, c->event_cap, c->step_index}; }
int restitution(edynhip_ctx *c);   // restitution.hip: solve_restitution, before gravity and the constraint solver (solver.cpp:397)
int solve(edynhip_ctx *c);
int refresh_derived(edynhip_ctx *c);
int joint_reset_angles(edynhip_ctx *c, const uint8_t *which_dev);   // reset_angle of the marked (sorted-order) joints
int wake_islands_of(edynhip_ctx *c, const std::vector<uint32_t> &bodies);   // capi.hip   // AABBs + world inertias from the current transforms (solver.hip)
// sort helpers (sort.hip)
size_t sort_temp_bytes(uint32_t max_items);
int sort_u64(edynhip_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n, int begin_bit, int end_bit);
int set_error(edynhip_ctx *c, int code, const char *what, hipError_t e = hipSuccess);
int mesh_bind_bodies(edynhip_ctx *c, uint32_t first, uint32_t n, const int32_t *shape_type, const float *shape_param);   // mesh.hip
int update_rotated(edynhip_ctx *c);
}  // namespace eh

#define EH_HIP(c, call)                                                          \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) return eh::set_error((c), EDYNHIP_ERR_HIP, #call, e__); \
    } while (0)
#define EH_TRY(expr)                    \
    do {                                \
        int r__ = (expr);               \
        if (r__ != EDYNHIP_OK) return r__; \
    } while (0)
