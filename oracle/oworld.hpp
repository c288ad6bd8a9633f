// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header). Parity status: see oracle/README.md.
//
// CPU restatement of the reference's per-step loop over plain arrays (no EnTT):
//   /root/reference/src/edyn/simulation/stepper_sequential.cpp:71-102,121-147   step order
//   /root/reference/src/edyn/collision/broadphase.cpp:99-195                    pair maintenance
//   /root/reference/src/edyn/collision/narrowphase.cpp:21-40                    narrowphase driver
//   /root/reference/include/edyn/util/collision_util.hpp:104-276                process_collision
//   /root/reference/src/edyn/util/collision_util.cpp:28-45,205-280,319-475      distances, merge, nearest, create, remove, detect
//   /root/reference/src/edyn/dynamics/solver.cpp:83-215,387-468                 row prep + solver::update
//   /root/reference/src/edyn/constraints/contact_constraint.cpp:15-98           contact rows / position solve
//   /root/reference/src/edyn/constraints/point_constraint.cpp:9-58              point rows
//   /root/reference/src/edyn/constraints/hinge_constraint.cpp:26-213            hinge rows / position solve (no limits/springs)
//   /root/reference/src/edyn/constraints/constraint_row.cpp:6-57                prepare_row / solve / apply
//   /root/reference/src/edyn/constraints/constraint_row_friction.cpp:11-66      friction circle
//   /root/reference/src/edyn/dynamics/island_solver.cpp:76-111,357-376,513-543  warm start, iterations, integrate
//   /root/reference/include/edyn/dynamics/position_solver.hpp:16-51             position correction
//   /root/reference/src/edyn/sys/update_aabbs.cpp:53-78, update_inertias.cpp:12-24
//
// Two solver orders are provided over identical row arithmetic:
//   ORDER_SEQUENTIAL — the reference's order: per island, type-major rows (hinge, point, contact),
//                      all rows then all friction rows per iteration, per-island position iterations.
//                      Manifold order = ascending canonical pair key, point order = the reference's
//                      contact list order (newest first). EnTT pool order itself is not reproducible.
//   ORDER_COLOURED   — the order the GPU uses: deterministic edge colouring (see colour_edges), colours
//                      ascending, joints before contacts; per manifold its normal rows, then its friction rows.
#pragma once
#include <map>
#include <vector>
#include <array>
#include <cstring>
#include "ocollide.hpp"
#include "otree.hpp"

namespace orc {

enum body_kind : int { KIND_DYNAMIC = 0, KIND_KINEMATIC = 1, KIND_STATIC = 2 };
enum joint_type : int { JOINT_POINT = 0, JOINT_HINGE = 1, JOINT_DISTANCE = 2, JOINT_SOFT_DISTANCE = 3, JOINT_CONE = 4, JOINT_CVJOINT = 5, JOINT_GRAVITY = 6, JOINT_GENERIC = 7, JOINT_NULL = 8 /* null_constraint.hpp: no rows, only an island-graph edge */ };
constexpr int kJointSlotsO = 24, kJointParamsO = 64;
// ORDER_EXTERNAL = ORDER_SEQUENTIAL with the visiting order inside each island supplied by the caller (ext_contact_order /
// ext_joint_order): the order the REAL reference used for the same step (island.edges iteration order, which depends on
// EnTT pool history), exported by oracle/ref_world.cpp. With it the restatement and the reference agree bit for bit.
enum solver_order : int { ORDER_SEQUENTIAL = 0, ORDER_COLOURED = 1, ORDER_EXTERNAL = 2 };
constexpr uint32_t kNoColour = 0xFFu;
constexpr uint32_t kMaxColours = 64;
constexpr uint32_t kSerialColour = 62;   // contacts: colours 0..61 are conflict-free, 62 is the serial bucket (edyn_amd/csrc/ctx.hpp)

struct Body {
    int kind = KIND_DYNAMIC;
    vec3 pos{0, 0, 0};
    quat orn{0, 0, 0, 1};
    vec3 linvel{0, 0, 0}, angvel{0, 0, 0};
    float mass_inv = 0;
    mat3 I_inv = kMat3Zero, I_inv_world = kMat3Zero;
    vec3 gravity{0, 0, 0};
    shape sh{};
    bool has_material = true;
    float friction = 0.5f, restitution = 0.0f;
    float spin_friction = 0, roll_friction = 0, stiffness = kLarge, damping = kLarge;   // comp/material.hpp:15-22
    uint32_t material_id = 0xFFFFu;   // material::id (UnassignedID): key into the material mix table (material_mixing.hpp:36-82)
    uint64_t group = ~0ull, mask = ~0ull;
    std::vector<uint32_t> exclusions;   // collision_exclusion (comp/collision_exclusion.hpp:16-31)
    bool removed = false;               // destroyed entity: the index stays reserved
    aabb box{};
    vec3 dv{0, 0, 0}, dw{0, 0, 0};
    uint32_t leaf = DynTree::NIL;
    bool asleep = false;             // sleeping_tag (island_manager.cpp:541-565)
    bool sleeping_disabled = false;  // sleeping_disabled_tag
    // center_of_mass / origin (comp/center_of_mass.hpp, comp/origin.hpp): `pos` is the centre of mass, shapes and every pivot live in the
    // frame of `origin` = to_world(-com, pos, orn); bodies without an offset have neither component and use `pos`
    vec3 com{0, 0, 0}, origin{0, 0, 0};
    bool has_com = false;
    mat3 I_body = kMat3Zero; bool inertia_from_shape = false;   // kept for rigidbody_def::center_of_mass (parallel-axis shift at creation)
    vec3 org() const { return has_com ? origin : pos; }
    void update_origin() { if (has_com) origin = to_world(-com, pos, orn); }   // update_origins.cpp:13-15
    bool procedural() const { return kind == KIND_DYNAMIC; }
    bool rolling() const { return kind == KIND_DYNAMIC && (sh.type == SHAPE_SPHERE || sh.type == SHAPE_CAPSULE || sh.type == SHAPE_CYLINDER); }   // rolling_shapes_tuple_t, shapes.hpp:40-44
    vec3 roll_direction() const {   // roll_direction component: dynamic capsules roll about their axis (rigidbody.cpp:119-130, shapes.hpp:136-139)
        return kind == KIND_DYNAMIC && (sh.type == SHAPE_CAPSULE || sh.type == SHAPE_CYLINDER) ? coordinate_axis_vector(sh.axis) : vec3{0, 0, 0};
    }
};

struct ContactPoint {
    vec3 pivotA, pivotB, normal, local_normal;
    int attachment;
    float distance;
    float friction, restitution;
    float normal_restitution_impulse = 0, friction_restitution_impulse[2] = {0, 0};   // contact_point.hpp:46-58
    uint32_t lifetime;
    float normal_impulse;
    float friction_impulse[2];
    // contact_extras_constraint (contact_point_material / _spin_friction_impulse / _roll_friction_impulse)
    float spin_friction = 0, roll_friction = 0, stiffness = kLarge, damping = kLarge;
    float rolling_impulse[2] = {0, 0}, spin_impulse = 0;
    bool extras() const { return stiffness < kLarge || damping < kLarge || spin_friction > 0 || roll_friction > 0; }   // collision_util.cpp:372-373
    uint64_t id = 0;   // contact events: (step of creation + 1) << 32 | manifold index << 2 | local slot (the GPU's rule)
};

// Contact events as an application observes them through on_construct / on_destroy<contact_manifold | contact_point>.
enum { EV_MANIFOLD_CREATED = 1, EV_MANIFOLD_DESTROYED = 2, EV_POINT_CREATED = 3, EV_POINT_DESTROYED = 4 };
struct ContactEvent { uint32_t type, step, bodyA, bodyB; uint64_t id; };

struct Manifold {
    uint32_t body[2];
    int num_points = 0;
    ContactPoint pt[kMaxContacts];   // list order: newest first
    uint32_t colour = kNoColour;
    bool with_restitution = false;   // contact_manifold_with_restitution: mixed restitution > eps at creation (constraint_util.cpp:83-101)
};

struct Joint {
    int type = JOINT_POINT;
    uint32_t body[2];
    vec3 pivot[2];
    mat3 frame[2] = {kMat3Identity, kMat3Identity};   // hinge: column 0 = axis
    // Optional rows (hinge_constraint.hpp:30-62, point_constraint.hpp:25). hinge params: angle_min, angle_max, limit_restitution,
    // bump_stop_angle, bump_stop_stiffness, torque, speed, rest_angle, stiffness, damping; point: params[0] = friction_torque.
    // cone (cone_constraint.hpp): span_tan[0], span_tan[1], restitution, bump_stop_stiffness, bump_stop_length; slots 0 limit, 1 bump stop.
    // cvjoint (cvjoint_constraint.hpp): twist_min, twist_max, twist_restitution, twist_bump_stop_angle, twist_bump_stop_stiffness,
    // twist_friction_torque, twist_rest_angle, twist_stiffness, twist_damping, rest_direction xyz, bend_stiffness,
    // bend_friction_torque, bend_damping; slots 0..2 linear, 3 twist limit, 4 bump stop, 5 spring, 6 twist friction/damping,
    // 7 bend friction/damping, 8 bend spring.
    // generic (generic_constraint.hpp): 6 degrees of freedom (linear x, y, z along frame[0]'s columns, then angular), 10 floats
    // each: limit_enabled, min, max, limit_restitution, bump_stop_length|angle, bump_stop_stiffness, friction, rest, spring_stiffness,
    // damping; impulse slot 4 * dof + {0 limit, 1 bump stop, 2 spring, 3 friction/damping}.
    float params[kJointParamsO] = {0};
    float angle = 0;   // hinge / cvjoint twist: relative angle tracked across wraps (hinge_constraint.cpp:80-89, cvjoint_constraint.cpp:39-47)
    // applied impulses by SLOT: hinge linear[0..2], hinge[3..4], limit 5, bump_stop 6, spring 7, torque 8; point [0..2], friction 3
    float impulse[kJointSlotsO] = {0};
    bool alive = true; // false: removed (the index stays reserved)
    uint32_t colour = kNoColour;
};

struct Row {
    vec3 J[4];
    float eff_mass, rhs, lower, upper, impulse;
    float inv_mA, inv_mB;
    mat3 inv_IA, inv_IB;
    vec3 *dvA, *dwA, *dvB, *dwB;
};
struct FrictionRow {
    struct { vec3 J[4]; float eff_mass, rhs, impulse; } row[2];
    float mu;
    uint32_t normal_row;
};
struct SpinRow { vec3 J[2]; float eff_mass, rhs, impulse, mu; uint32_t normal_row; };   // constraint_row_spin_friction.hpp
struct RowOptions { float error = 0, erp = 0.2f, restitution = 0; };

// Canonical key of an unordered body pair: (owner << 32) | other, where the OWNER is the procedural body - the one with
// the higher index when both are procedural. Manifolds are kept, coloured and solved in ascending key order. (EnTT's
// pool order is not reproducible, so a canonical order is needed anyway; owner-major order is what a GPU broadphase
// produces without a sort: the owner is the body whose tree query finds the pair, broadphase.cpp:136-171.)
inline uint64_t pair_key_owned(uint32_t owner, uint32_t other) { return ((uint64_t)owner << 32) | other; }
inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// --- row arithmetic (constraint_row.cpp, constraint_row_friction.cpp, constraint_util.cpp:137-158) ---
inline float effective_mass(const vec3 J[4], float imA, const mat3 &iIA, float imB, const mat3 &iIB) {
    float s = dot(J[0], J[0]) * imA + dot(iIA * J[1], J[1]) + dot(J[2], J[2]) * imB + dot(iIB * J[3], J[3]);
    return 1.0f / s;
}
inline float relative_speed(const vec3 J[4], vec3 vA, vec3 wA, vec3 vB, vec3 wB) {
    return dot(J[0], vA) + dot(J[1], wA) + dot(J[2], vB) + dot(J[3], wB);
}
inline void prepare_row(Row &r, const RowOptions &o, vec3 vA, vec3 wA, vec3 vB, vec3 wB) {
    r.eff_mass = effective_mass(r.J, r.inv_mA, r.inv_IA, r.inv_mB, r.inv_IB);
    float relvel = relative_speed(r.J, vA, wA, vB, wB);
    r.rhs = -(o.error * o.erp + relvel * (1 + o.restitution));
}
inline void apply_row_impulse(float imp, Row &r) {
    *r.dvA += r.inv_mA * r.J[0] * imp;
    *r.dvB += r.inv_mB * r.J[2] * imp;
    *r.dwA += r.inv_IA * r.J[1] * imp;
    *r.dwB += r.inv_IB * r.J[3] * imp;
}
inline float solve_row(Row &r) {
    float drel = relative_speed(r.J, *r.dvA, *r.dwA, *r.dvB, *r.dwB);
    float dimp = (r.rhs - drel) * r.eff_mass;
    float imp = r.impulse + dimp;
    if (imp < r.lower) { dimp = r.lower - r.impulse; r.impulse = r.lower; }
    else if (imp > r.upper) { dimp = r.upper - r.impulse; r.impulse = r.upper; }
    else r.impulse = imp;
    return dimp;
}
inline void solve_friction(FrictionRow &f, Row &n) {
    float dimp[2], imp[2];
    for (int i = 0; i < 2; ++i) {
        float drel = relative_speed(f.row[i].J, *n.dvA, *n.dwA, *n.dvB, *n.dwB);
        dimp[i] = (f.row[i].rhs - drel) * f.row[i].eff_mass;
        imp[i] = f.row[i].impulse + dimp[i];
    }
    float len2 = imp[0] * imp[0] + imp[1] * imp[1];
    float max_len = f.mu * n.impulse;
    if (len2 > square(max_len)) {
        float len = std::sqrt(len2);
        if (len > kEps) { imp[0] = imp[0] / len * max_len; imp[1] = imp[1] / len * max_len; }
        else { imp[0] = 0; imp[1] = 0; }
        for (int i = 0; i < 2; ++i) dimp[i] = imp[i] - f.row[i].impulse;
    }
    for (int i = 0; i < 2; ++i) {
        f.row[i].impulse = imp[i];
        *n.dvA += n.inv_mA * f.row[i].J[0] * dimp[i];
        *n.dwA += n.inv_IA * f.row[i].J[1] * dimp[i];
        *n.dvB += n.inv_mB * f.row[i].J[2] * dimp[i];
        *n.dwB += n.inv_IB * f.row[i].J[3] * dimp[i];
    }
}
// ---- the coloured order's own contact-row arithmetic ("fused rows", round 4) ------------------------------------------------------
// The device's velocity solve is a dependency chain: what bounds a step is the number of instructions between
// "a body's deltas arrived" and "its deltas are handed on". The coloured order therefore has its own arithmetic for the normal and
// friction rows of contacts - the same row equations (constraint_row.cpp:24-57, constraint_row_friction.cpp:11-54) written with
// fused multiply-adds and fewer operations; this header is its specification and the device reproduces it bit for bit:
//   * a 3-vector dot product is fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); a relative speed is (lin_A + ang_A) + (lin_B + ang_B);
//   * delta = fma(-relative_speed, eff_mass, rhs * eff_mass);
//   * a normal row clamps the new impulse, min(max(impulse + delta, lower), upper), and applies new - old;
//   * a friction pair scales both impulses by max_len / len (ONE correctly rounded division) when they leave the circle and
//     applies new - old;
//   * an impulse is applied as delta_v = fma(M^-1 J, delta, delta_v) per component.
// The sequential and external orders keep the reference's arithmetic operation for operation (they are what is pinned to the engine
// bit for bit); the coloured order differs from them in the Gauss-Seidel visiting order anyway, and agrees with them - and with the
// engine - within the tolerances of SURVEY 8(d) (tests/test_oracle_physics.py, tests/test_reference_engine.py, lock-step tests).
// Test / mode switch of the coloured order (orc_set_arithmetic): bit 0 = fused velocity rows (above), bit 1 = the per-manifold block
// position correction (contact_solve_position_block below). 0 = the reference's arithmetic in the coloured order too.
// bit 2 (checker-only, no device counterpart): the coloured order keeps the reference's TWO PHASES per iteration over the whole island -
// every normal row (colour after colour), then every friction row, then rolling, then spinning (island_solver.cpp:76-111) - instead of
// finishing each manifold (normals, then its friction rows) before the next. Inside ONE colour the two forms are the same thing (its
// manifolds share no dynamic body), so "two phases per colour" would change nothing; what can be measured is the island-wide form, which
// a device would pay for with two passes over every body's chain per iteration. tests/test_reference_engine.py uses it to put a number on
// how much of the coloured order's distance to the engine comes from this choice (VERDICT r05 missing #5).
enum : int { ARITH_REFERENCE = 0, ARITH_FUSED_VELOCITY = 1, ARITH_BLOCK_POSITION = 2, ARITH_TWO_PHASE = 4 };
inline int g_arith = ARITH_REFERENCE;
inline float dot3_fma(vec3 a, vec3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline vec3 fma3(vec3 w, float s, vec3 acc) { return {std::fmaf(w.x, s, acc.x), std::fmaf(w.y, s, acc.y), std::fmaf(w.z, s, acc.z)}; }
inline float relative_speed_fused(const vec3 J[4], vec3 vA, vec3 wA, vec3 vB, vec3 wB) {
    return (dot3_fma(J[0], vA) + dot3_fma(J[1], wA)) + (dot3_fma(J[2], vB) + dot3_fma(J[3], wB));
}
inline void apply_impulse_fused(float imp, const vec3 J[4], Row &n) {
    *n.dvA = fma3(n.inv_mA * J[0], imp, *n.dvA);
    *n.dwA = fma3(n.inv_IA * J[1], imp, *n.dwA);
    *n.dvB = fma3(n.inv_mB * J[2], imp, *n.dvB);
    *n.dwB = fma3(n.inv_IB * J[3], imp, *n.dwB);
}
inline void solve_normal_fused(Row &r) {
    const float rel = relative_speed_fused(r.J, *r.dvA, *r.dwA, *r.dvB, *r.dwB);
    const float dimp = std::fmaf(-rel, r.eff_mass, r.rhs * r.eff_mass);
    const float nw = std::fmin(std::fmax(r.impulse + dimp, r.lower), r.upper);
    const float applied = nw - r.impulse;
    r.impulse = nw;
    apply_impulse_fused(applied, r.J, r);
}
inline void solve_friction_fused(FrictionRow &f, Row &n) {
    float imp[2];
    for (int i = 0; i < 2; ++i) {
        const float rel = relative_speed_fused(f.row[i].J, *n.dvA, *n.dwA, *n.dvB, *n.dwB);
        imp[i] = f.row[i].impulse + std::fmaf(-rel, f.row[i].eff_mass, f.row[i].rhs * f.row[i].eff_mass);
    }
    const float len2 = std::fmaf(imp[1], imp[1], imp[0] * imp[0]);
    const float max_len = f.mu * n.impulse;
    if (len2 > max_len * max_len) {
        const float len = std::sqrt(len2);
        const float scale = len > kEps ? max_len / len : 0.0f;
        imp[0] *= scale; imp[1] *= scale;
    }
    for (int i = 0; i < 2; ++i) {
        const float applied = imp[i] - f.row[i].impulse;
        f.row[i].impulse = imp[i];
        apply_impulse_fused(applied, f.row[i].J, n);
    }
}
inline void solve_spin_friction(SpinRow &r, Row &n) {   // constraint_row_spin_friction.cpp:5-29
    const float max_len = r.mu * n.impulse;
    const float drel = dot(r.J[0], *n.dwA) + dot(r.J[1], *n.dwB);
    float dimp = (r.rhs - drel) * r.eff_mass;
    const float imp = r.impulse + dimp;
    const float lo = -max_len, hi = max_len;
    if (imp < lo) { dimp = lo - r.impulse; r.impulse = lo; }
    else if (imp > hi) { dimp = hi - r.impulse; r.impulse = hi; }
    else r.impulse = imp;
    *n.dwA += n.inv_IA * r.J[0] * dimp;
    *n.dwB += n.inv_IB * r.J[1] * dimp;
}
inline void warm_start_spin(SpinRow &r, Row &n) {   // :31-36
    *n.dwA += n.inv_IA * r.J[0] * r.impulse;
    *n.dwB += n.inv_IB * r.J[1] * r.impulse;
}
inline void warm_start_friction(FrictionRow &f, Row &n) {
    for (int i = 0; i < 2; ++i) {
        *n.dvA += n.inv_mA * f.row[i].J[0] * f.row[i].impulse;
        *n.dwA += n.inv_IA * f.row[i].J[1] * f.row[i].impulse;
        *n.dvB += n.inv_mB * f.row[i].J[2] * f.row[i].impulse;
        *n.dwB += n.inv_IB * f.row[i].J[3] * f.row[i].impulse;
    }
}

struct StepStats {
    uint32_t num_manifolds = 0, num_points = 0, num_rows = 0, num_islands = 0, num_colours = 0,
             num_joint_colours = 0, colour_rounds = 0;
};

class World {
public:
    float dt = 1.0f / 60.0f;
    int vel_iters = 8, pos_iters = 3;   // context/settings.hpp:22-30 defaults
    vec3 gravity{0, -9.8f, 0};
    int restitution_iters = 8, individual_restitution_iters = 3;   // settings.hpp:28-29
    int order = ORDER_SEQUENTIAL;
    struct ExtContact { uint32_t a, b, slot; };
    std::vector<ExtContact> ext_contact_order;   // ORDER_EXTERNAL: every active contact point, in the reference's visiting order
    std::vector<uint32_t> ext_joint_order;       // ORDER_EXTERNAL: joint indices, in the reference's visiting order
    // ORDER_EXTERNAL, restitution solver: the island edge-list order of the manifolds and every connecting body's adjacency in the real
    // engine's entity graph (ref_world.cpp refw_get_restitution_walk) - one step's worth, consumed by the next solve_restitution()
    struct ExtAdj { uint32_t other, ma, mb; };   // neighbour body; the edge's manifold (body[0], body[1]) or ~0 for a non-contact edge
    std::vector<std::array<uint32_t, 2>> ext_rest_manifolds;
    std::map<uint32_t, std::vector<ExtAdj>> ext_adj;
    bool ext_walk_valid = false;
    std::vector<Body> bodies;
    std::vector<Joint> joints;
    std::map<uint64_t, Manifold> manifolds;
    std::vector<uint32_t> island_label;   // per body; valid for procedural bodies after update_islands()
    StepStats stats;
    // Island sleeping (island_manager.cpp:541-623). Off by default: the benchmark scenes tag every body
    // sleeping_disabled. Islands are identified by their label (lowest body index); the reference keeps
    // island entities and, on a merge, the larger island's timer - a difference only in which timer survives a merge.
    bool sleeping = false;
    uint64_t step_index = 0;
    bool record_events = false;
    mutable std::vector<ContactEvent> events;
    void emit(uint32_t type, const Manifold &m, uint64_t id) const {
        if (record_events) events.push_back(ContactEvent{type, (uint32_t)step_index, m.body[0], m.body[1], id});
    }
    void emit_destroyed(const Manifold &m) const {
        for (int k = 0; k < m.num_points; ++k) emit(EV_POINT_DESTROYED, m, m.pt[k].id);
        emit(EV_MANIFOLD_DESTROYED, m, 0);
    }
    // Island sleep timers run on the step TIME STAMPS the stepper hands to the island manager (stepper_sequential.cpp:60-75,
    // island_manager.cpp:605-623): sim_clock is the stamp of the step being run; step() advances it by fixed dt, step_timed()
    // by the caller's stretched step_dt (the max_steps_per_update clamp scales the stamps, not the integration dt).
    double sim_clock = 0;   // the island manager's m_last_time: the stamp of the last step whose island update has run
    std::vector<uint8_t> split_reset_;     // per label: the island is a part of an island that split in this step
    std::vector<double> sleep_since;       // per label: time stamp at which the island first qualified for sleep, < 0 = not counting
    std::vector<uint64_t> new_keys;        // manifolds created by this step's broadphase (they wake their island)

    uint64_t pair_key(uint32_t a, uint32_t b) const {   // see pair_key_owned
        const bool pa = bodies[a].procedural(), pb = bodies[b].procedural();
        if (pa && pb) return a > b ? pair_key_owned(a, b) : pair_key_owned(b, a);
        return pa ? pair_key_owned(a, b) : pair_key_owned(b, a);
    }

    // rigidbody.cpp:47-191 (make_rigidbody), restricted to the components on the hot path.
    uint32_t add_body(int kind, vec3 pos, quat orn, vec3 linvel, vec3 angvel, float mass, const shape &sh,
                      const mat3 *inertia, float friction, float restitution, bool has_material, uint64_t group,
                      uint64_t mask, const vec3 *grav) {
        Body b;
        b.kind = kind; b.pos = pos; b.orn = orn; b.sh = sh;
        if (kind == KIND_DYNAMIC) {
            b.mass_inv = 1.0f / mass;
            mat3 I = inertia ? *inertia : moment_of_inertia(sh, mass);
            b.I_body = I; b.inertia_from_shape = inertia == nullptr;
            b.I_inv = inverse_symmetric(I);
            mat3 basis = to_mat3(orn);
            b.I_inv_world = basis * b.I_inv * transpose(basis);
        }
        if (kind != KIND_STATIC) { b.linvel = linvel; b.angvel = angvel; }
        vec3 g = grav ? *grav : gravity;
        if (kind == KIND_DYNAMIC) b.gravity = g;
        b.has_material = has_material; b.friction = friction; b.restitution = restitution;
        b.group = group; b.mask = mask;
        if (sh.type != SHAPE_NONE) b.box = shape_aabb(sh, pos, orn);
        uint32_t id = (uint32_t)bodies.size();
        bodies.push_back(b);
        if (sh.type != SHAPE_NONE) {   // broadphase.cpp:75-97 init_new_aabb_entities
            DynTree &t = bodies[id].procedural() ? tree_ : np_tree_;
            bodies[id].leaf = t.create(bodies[id].box, id);
        }
        return id;
    }
    // registry.destroy(body): every edge of the node goes with it - manifolds, contact points, joints - and the islands it
    // touched wake up and are re-examined (island_manager.cpp:47-115). The index stays reserved (a tombstone).
    void remove_body(uint32_t i) {
        Body &b = bodies[i];
        if (b.removed) return;
        std::vector<uint32_t> wake;   // island labels to wake: its own, or (non-procedural) those of its partners
        if (b.procedural() && i < island_label.size()) wake.push_back(island_label[i]);
        for (auto it = manifolds.begin(); it != manifolds.end();) {
            if (it->second.body[0] == i || it->second.body[1] == i) {
                const uint32_t o = it->second.body[0] == i ? it->second.body[1] : it->second.body[0];
                if (bodies[o].procedural() && o < island_label.size()) wake.push_back(island_label[o]);
                emit_destroyed(it->second);   // (the device notices at the next step's broadphase: same step index)
                it = manifolds.erase(it);
            } else ++it;
        }
        for (auto &j : joints)
            if (j.alive && (j.body[0] == i || j.body[1] == i)) {
                const uint32_t o = j.body[0] == i ? j.body[1] : j.body[0];
                if (bodies[o].procedural() && o < island_label.size()) wake.push_back(island_label[o]);
                j.alive = false; joints_coloured_ = false;
            }
        if (b.sh.type != SHAPE_NONE) (b.procedural() ? tree_ : np_tree_).destroy(b.leaf);
        b.removed = true; b.kind = KIND_STATIC; b.sh.type = SHAPE_NONE; b.asleep = false;
        b.linvel = b.angvel = {0, 0, 0}; b.mass_inv = 0; b.gravity = {0, 0, 0};
        for (uint32_t k = 0; k < bodies.size() && k < island_label.size(); ++k)
            if (bodies[k].procedural() && std::find(wake.begin(), wake.end(), island_label[k]) != wake.end()) {
                bodies[k].asleep = false;
                if (island_label[k] < sleep_since.size()) sleep_since[island_label[k]] = -1.0;
            }
    }
    void remove_joint(uint32_t ji) {
        Joint &j = joints[ji];
        if (!j.alive) return;
        j.alive = false; joints_coloured_ = false;
        for (uint32_t e : {j.body[0], j.body[1]}) {   // destroying an edge wakes its island (on_destroy_island_resident)
            if (!bodies[e].procedural() || e >= island_label.size()) continue;
            const uint32_t l = island_label[e];
            for (uint32_t k = 0; k < bodies.size() && k < island_label.size(); ++k)
                if (bodies[k].procedural() && island_label[k] == l) bodies[k].asleep = false;
            if (l < sleep_since.size()) sleep_since[l] = -1.0;
        }
    }
    void set_gravity(vec3 g) {   // gravity_util.cpp:12-20: the setting and every body that carries a gravity component
        gravity = g;
        for (auto &b : bodies) if (b.kind == KIND_DYNAMIC && !b.removed) b.gravity = g;
    }
    // rigidbody_def::center_of_mass right after add_body (rigidbody.cpp:56-87): an inertia derived from the shape is shifted by the
    // parallel-axis theorem (moment_of_inertia.cpp:217-220), then apply_center_of_mass (rigidbody.cpp:517-548): the position given to
    // add_body is the ORIGIN; position and linear velocity move to the centre of mass.
    void set_center_of_mass(uint32_t i, vec3 com, float mass, bool at_creation = true) {   // at_creation = false: edyn::set_center_of_mass on a running world (rigidbody.cpp:364-370): apply_center_of_mass only
        Body &b = bodies[i];
        if (at_creation && b.kind == KIND_DYNAMIC && b.inertia_from_shape) {
            const mat3 d = skew(com);
            const mat3 dd = transpose(d) * d;
            mat3 I;
            for (int r = 0; r < 3; ++r) I.row[r] = b.I_body.row[r] + dd.row[r] * mass;
            b.I_body = I;
            b.I_inv = inverse_symmetric(I);
            mat3 basis = to_mat3(b.orn);
            b.I_inv_world = basis * b.I_inv * transpose(basis);
        }
        const vec3 origin = to_world(-b.com, b.pos, b.orn);
        const vec3 com_world = to_world(com, origin, b.orn);
        if (b.kind != KIND_STATIC) b.linvel += cross(b.angvel, com_world - b.pos);
        b.pos = com_world;
        b.has_com = !(com == vec3{0, 0, 0});
        b.com = b.has_com ? com : vec3{0, 0, 0};
        b.origin = origin;
    }
    uint32_t add_joint(int type, uint32_t a, uint32_t b, vec3 pivotA, vec3 pivotB, vec3 axisA, vec3 axisB) {
        Joint j;
        j.type = type; j.body[0] = a; j.body[1] = b; j.pivot[0] = pivotA; j.pivot[1] = pivotB;
        if (type == JOINT_HINGE) {   // hinge_constraint.cpp:11-17 set_axes
            vec3 p, q;
            plane_space(axisA, p, q); j.frame[0] = mat3_columns(axisA, p, q);
            plane_space(axisB, p, q); j.frame[1] = mat3_columns(axisB, p, q);
        }
        joints.push_back(j);
        joints_coloured_ = false;
        return (uint32_t)joints.size() - 1;
    }

    // One fixed-dt step (stepper_sequential.cpp:121-147 step_simulation order).
    void step() { step_timed(sim_clock + (double)dt); }
    void step_timed(double step_time) {
        pending_stamp_ = step_time;
        broadphase();
        narrowphase();
        update_islands();
        solve();
    }
    bool manifold_asleep(const Manifold &m) const {   // every procedural endpoint sleeps (an island sleeps as a whole)
        const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        bool any_proc = false;
        if (A.procedural()) { if (!A.asleep) return false; any_proc = true; }
        if (B.procedural()) { if (!B.asleep) return false; any_proc = true; }
        return any_proc;
    }
    bool joint_asleep(const Joint &j) const {
        const Body &A = bodies[j.body[0]], &B = bodies[j.body[1]];
        bool any_proc = false;
        if (A.procedural()) { if (!A.asleep) return false; any_proc = true; }
        if (B.procedural()) { if (!B.asleep) return false; any_proc = true; }
        return any_proc;
    }
    void wake_all() { for (auto &b : bodies) b.asleep = false; std::fill(sleep_since.begin(), sleep_since.end(), -1.0); }

    // ---------------- broadphase ----------------
    // settings.should_collide_func (settings.hpp:43, set_should_collide): a user predicate that replaces should_collide_default
    int (*collide_filter)(void *, uint32_t, uint32_t) = nullptr;
    void *collide_filter_user = nullptr;
    bool should_collide(uint32_t a, uint32_t b) const {   // should_collide.cpp:11-57
        if (a == b) return false;
        const Body &A = bodies[a], &B = bodies[b];
        if ((A.group & B.mask) == 0 || (B.group & A.mask) == 0) return false;
        auto excluded = [&](const Body &X, uint32_t other) {   // should_exclude: the body's collision_exclusion list
            for (uint32_t e : X.exclusions) if (e == other) return true;
            return false;
        };
        if (excluded(A, b) || excluded(B, a)) return false;
        return true;
    }
    void exclude_collision(uint32_t a, uint32_t b) {   // util/exclude_collision.cpp:9-35 (both ways, no duplicates, <= 16 each)
        auto one_way = [&](uint32_t x, uint32_t y) {
            auto &l = bodies[x].exclusions;
            if (std::find(l.begin(), l.end(), y) == l.end() && l.size() < 16) l.push_back(y);
        };
        one_way(a, b); one_way(b, a);
    }
    void remove_collision_exclusion(uint32_t a, uint32_t b) {   // :41-60 (the last entry takes the hole)
        auto one_way = [&](uint32_t x, uint32_t y) {
            auto &l = bodies[x].exclusions;
            for (size_t i = l.size(); i; --i) if (l[i - 1] == y) { l[i - 1] = l.back(); l.pop_back(); break; }
        };
        one_way(a, b); one_way(b, a);
    }
    // material mix table (registry.ctx material_mix_table; edyn::insert_material_mixing): an entry for the unordered pair of material
    // ids replaces every mixing rule. Values: restitution, friction, spin_friction, roll_friction, stiffness, damping.
    // The reference keeps the table in a std::map keyed by unordered_pair, whose operator< (core/unordered_pair.hpp:32-40) treats
    // {a, b} and {b, a} as equivalent but orders everything else by (first, second) - not a strict weak ordering, so whether a
    // lookup with the ids in the other order FINDS its entry depends on the tree's shape. That is what the reference simulates
    // with; it is reproduced by using the same container with the same comparator and the same insertion sequence (same libstdc++).
    struct IdPair { uint32_t first, second; };
    struct IdPairLess {
        bool operator()(const IdPair &a, const IdPair &b) const {
            if (a.first == b.second && a.second == b.first) return false;
            if (a.first == b.first) return a.second < b.second;
            return a.first < b.first;
        }
    };
    std::map<IdPair, std::array<float, 6>, IdPairLess> mix_table;
    const std::array<float, 6> *mix_entry(const Body &A, const Body &B) const {   // key = {material of body[0], material of body[1]}
        auto it = mix_table.find(IdPair{A.material_id, B.material_id});
        return it == mix_table.end() ? nullptr : &it->second;
    }
    bool tags_restitution(uint32_t a, uint32_t b) const {   // constraint_util.cpp:83-101
        const Body &A = bodies[a], &B = bodies[b];
        if (!(A.has_material && B.has_material)) return false;
        if (const auto *e = mix_entry(A, B)) return (*e)[0] > kEps;
        return std::min(A.restitution, B.restitution) > kEps;
    }
    void broadphase() {
        const float sep = kContactBreakingThreshold * 1.3f;   // broadphase.hpp:18
        const vec3 sep_off = vec3{1, 1, 1} * -sep;
        const vec3 q_off = vec3{1, 1, 1} * -kContactBreakingThreshold;   // broadphase.hpp:15
        new_keys.clear();
        for (auto it = manifolds.begin(); it != manifolds.end();) {   // destroy_separated_manifolds (sleeping manifolds excluded)
            const aabb &b0 = bodies[it->second.body[0]].box, &b1 = bodies[it->second.body[1]].box;
            if (!manifold_asleep(it->second) && !intersect(b0.inset(sep_off), b1)) { emit_destroyed(it->second); it = manifolds.erase(it); }
            else ++it;
        }
        for (auto &b : bodies) {   // move_aabbs
            if (b.sh.type == SHAPE_NONE || b.asleep) continue;
            if (b.procedural()) tree_.move(b.leaf, b.box);
            else if (b.kind == KIND_KINEMATIC) np_tree_.move(b.leaf, b.box);
        }
        // EnTT views iterate a pool back to front, i.e. most recently created body first.
        for (uint32_t k = (uint32_t)bodies.size(); k-- > 0;) {
            const Body &b = bodies[k];
            if (!b.procedural() || b.sh.type == SHAPE_NONE || b.asleep) continue;   // sleeping bodies do not query
            const aabb q = b.box.inset(q_off);
            auto visit_tree = [&](const DynTree &t) {
                t.query(q, [&](uint32_t leaf) {
                    uint32_t other = t.payload(leaf);
                    if (collide_filter ? !collide_filter(collide_filter_user, k, other) : !should_collide(k, other)) return;   // broadphase.cpp:145
                    uint64_t key = pair_key(k, other);
                    if (manifolds.count(key)) return;
                    if (!intersect(q, bodies[other].box)) return;
                    Manifold m; m.body[0] = k; m.body[1] = other;   // constraint_util.cpp:60-102
                    m.with_restitution = tags_restitution(k, other);
                    manifolds.emplace(key, m);
                    new_keys.push_back(key);
                    emit(EV_MANIFOLD_CREATED, m, 0);
                });
            };
            visit_tree(tree_);
            visit_tree(np_tree_);
        }
    }

    // ---------------- narrowphase ----------------
    void detect(const Manifold &m, coll_result &res) const {   // collision_util.cpp:440-475
        const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        const vec3 off = vec3{1, 1, 1} * -kContactBreakingThreshold;
        res.num_points = 0;
        if (!intersect(A.box.inset(off), B.box)) return;
        coll_ctx ctx{A.org(), A.orn, B.org(), B.orn, kCollisionThreshold};
        collide(A.sh, B.sh, ctx, res);
    }
    static size_t find_nearest(const ContactPoint &cp, const coll_result &res) {   // collision_util.cpp:233-255
        float best = square(kContactCachingThreshold);
        size_t idx = res.num_points;
        for (size_t i = 0; i < res.num_points; ++i) {
            float dA = length_sqr(res.point[i].pivotA - cp.pivotA);
            float dB = length_sqr(res.point[i].pivotB - cp.pivotB);
            if (dA < best) { best = dA; idx = i; }
            if (dB < best) { best = dB; idx = i; }
        }
        return idx;
    }
    // collision_util.cpp:257-280. Note: compares against the result's pivotA for BOTH bodies, as the reference does.
    size_t find_nearest_rolling(const coll_result &res, vec3 cp_pivot, vec3 origin, quat orn, vec3 angvel) const {
        size_t idx = res.num_points;
        quat prev_orn = integrate(orn, angvel, -dt);
        vec3 prev_pivot = to_world(cp_pivot, origin, prev_orn);
        float best = square(kContactCachingThreshold);
        for (size_t i = 0; i < res.num_points; ++i) {
            vec3 pA = to_world(res.point[i].pivotA, origin, orn);
            float d2 = distance_sqr(pA, prev_pivot);
            if (d2 < best) { best = d2; idx = i; }
        }
        return idx;
    }
    void set_local_normal(const Manifold &m, ContactPoint &cp) const {
        if (cp.attachment != NA_NONE) {
            const quat orn = bodies[m.body[cp.attachment == NA_ON_A ? 0 : 1]].orn;
            cp.local_normal = rotate(conjugate(orn), cp.normal);
        } else cp.local_normal = {0, 0, 0};
    }
    void merge_point(const Manifold &m, const coll_point &rp, ContactPoint &cp) const {   // collision_util.cpp:205-231
        cp.pivotA = rp.pivotA; cp.pivotB = rp.pivotB; cp.normal = rp.normal;
        cp.distance = rp.distance; cp.attachment = rp.attachment;
        set_local_normal(m, cp);
    }
    ContactPoint make_point(const Manifold &m, const coll_point &rp) const {   // collision_util.cpp:319-395
        ContactPoint cp{};
        cp.pivotA = rp.pivotA; cp.pivotB = rp.pivotB; cp.normal = rp.normal;
        cp.attachment = rp.attachment; cp.distance = rp.distance;
        set_local_normal(m, cp);
        const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        if (const auto *e = mix_entry(A, B)) {   // assign_material_properties, collision_util.cpp:293-299
            cp.restitution = (*e)[0]; cp.friction = (*e)[1]; cp.spin_friction = (*e)[2]; cp.roll_friction = (*e)[3];
            cp.stiffness = (*e)[4]; cp.damping = (*e)[5];
            return cp;
        }
        cp.friction = std::sqrt(A.friction * B.friction);            // material_mixing.hpp:16-18
        cp.restitution = std::min(A.restitution, B.restitution);     // :12-14
        cp.roll_friction = std::max(A.roll_friction, B.roll_friction);   // :24-26 (assign_material_properties, collision_util.cpp:309-315)
        cp.spin_friction = std::max(A.spin_friction, B.spin_friction);   // :20-22
        if (A.stiffness < kLarge || B.stiffness < kLarge) {
            cp.stiffness = 1 / (1 / A.stiffness + 1 / B.stiffness);      // :28-30
            cp.damping = 1 / (1 / A.damping + 1 / B.damping);            // :32-34
        }
        return cp;
    }
    bool should_remove(const ContactPoint &cp, const Body &A, const Body &B) const {   // collision_util.cpp:397-413
        const float thr = kContactBreakingThreshold, thr2 = thr * thr;
        vec3 pA = to_world(cp.pivotA, A.org(), A.orn), pB = to_world(cp.pivotB, B.org(), B.orn);
        vec3 d = pA - pB;
        float nd = dot(d, cp.normal);
        vec3 td = d - nd * cp.normal;
        return nd > thr || length_sqr(td) > thr2;
    }
    void process_collision(Manifold &m, const coll_result &res, uint32_t midx = 0) const {   // collision_util.hpp:104-276
        const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        const size_t R = res.num_points;
        bool merged[kMaxContacts] = {false, false, false, false};
        bool removed[kMaxContacts] = {false, false, false, false};
        const int n_old = m.num_points;
        size_t num_points = (size_t)n_old;
        for (int i = 0; i < n_old; ++i) {
            ContactPoint &cp = m.pt[i];
            ++cp.lifetime;
            size_t nearest = find_nearest(cp, res);
            if (nearest == R && A.rolling()) nearest = find_nearest_rolling(res, cp.pivotA, A.org(), A.orn, A.angvel);
            if (nearest == R && B.rolling()) nearest = find_nearest_rolling(res, cp.pivotB, B.org(), B.orn, B.angvel);
            if (nearest < R && !merged[nearest]) {
                merge_point(m, res.point[nearest], cp);
                merged[nearest] = true;
            } else if (should_remove(cp, A, B)) {
                removed[i] = true;
                --num_points;
            }
        }
        bool all_merged = true;
        for (size_t r = 0; r < R; ++r) all_merged &= merged[r];

        struct Local { coll_point point; int old_index = -1; insert_type type = insert_type::none; };
        Local local[kMaxContacts];
        if (!all_merged) {
            if (num_points > 0) {
                int k = 0;
                for (int i = 0; i < n_old; ++i) {
                    if (removed[i]) continue;
                    local[k].point = {m.pt[i].pivotA, m.pt[i].pivotB, m.pt[i].normal, m.pt[i].distance, NA_NONE};
                    local[k].old_index = i;
                    ++k;
                }
            } else {
                ++num_points;
                local[0].point = res.point[0];
                local[0].type = insert_type::append;
                merged[0] = true;
            }
            for (size_t r = 0; r < R; ++r) {
                if (merged[r]) continue;
                const coll_point &rp = res.point[r];
                vec3 piv[kMaxContacts];
                for (size_t i = 0; i < num_points; ++i) piv[i] = local[i].point.pivotA;
                insert_result ir = insertion_point_index(piv, kMaxContacts, num_points, rp.pivotA);
                if (ir.type == insert_type::none) {
                    for (size_t i = 0; i < num_points; ++i) piv[i] = local[i].point.pivotB;
                    ir = insertion_point_index(piv, kMaxContacts, num_points, rp.pivotB);
                }
                if (ir.type != insert_type::none) { local[ir.index].point = rp; local[ir.index].type = ir.type; }
            }
        }
        // Resolve: destroyed old points leave the list, new points are pushed at the head in creation order.
        bool dead[kMaxContacts];
        for (int i = 0; i < n_old; ++i) dead[i] = removed[i];
        ContactPoint created[kMaxContacts];
        int n_created = 0;
        auto create = [&](size_t slot, const coll_point &p) {
            ContactPoint &cp = created[n_created++];
            cp = make_point(m, p);
            cp.id = ((uint64_t)(step_index + 1) << 32) | ((uint64_t)midx << 2) | (uint64_t)slot;
            emit(EV_POINT_CREATED, m, cp.id);
        };
        if (!all_merged) {
            for (size_t i = 0; i < num_points; ++i) {
                Local &lp = local[i];
                switch (lp.type) {
                case insert_type::none: break;
                case insert_type::append: create(i, lp.point); break;
                case insert_type::similar:
                    if (lp.old_index < 0) create(i, lp.point);
                    else merge_point(m, lp.point, m.pt[lp.old_index]);
                    break;
                case insert_type::replace:
                    if (lp.old_index >= 0) dead[lp.old_index] = true;
                    create(i, lp.point);
                    break;
                }
            }
        }
        for (int i = 0; i < n_old; ++i) if (dead[i]) emit(EV_POINT_DESTROYED, m, m.pt[i].id);
        ContactPoint out[kMaxContacts];
        int n_out = 0;
        for (int i = n_created - 1; i >= 0; --i) out[n_out++] = created[i];
        for (int i = 0; i < n_old; ++i) if (!dead[i]) out[n_out++] = m.pt[i];
        m.num_points = n_out;
        for (int i = 0; i < n_out; ++i) m.pt[i] = out[i];
        if (n_out == 0) m.colour = kNoColour;   // inactive pairs hold no solver colour
    }
    void narrowphase() {
        for (auto &kv : manifolds) {   // update_contact_distances, collision_util.cpp:28-45 (narrowphase.cpp:31 excludes sleeping)
            Manifold &m = kv.second;
            if (manifold_asleep(m)) continue;
            const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
            for (int i = 0; i < m.num_points; ++i) {
                vec3 pA = to_world(m.pt[i].pivotA, A.org(), A.orn), pB = to_world(m.pt[i].pivotB, B.org(), B.orn);
                m.pt[i].distance = dot(m.pt[i].normal, pA - pB);
            }
        }
        uint32_t midx = 0;   // index in canonical order = the device's manifold index
        for (auto &kv : manifolds) {
            const uint32_t mi = midx++;
            if (manifold_asleep(kv.second)) continue;
            coll_result res;
            detect(kv.second, res);
            process_collision(kv.second, res, mi);
        }
    }

    // ---------------- islands (connected components over procedural bodies) ----------------
    void update_islands() {
        const uint32_t n = (uint32_t)bodies.size();
        const std::vector<uint32_t> prev_label = island_label;   // last step's partition (split detection below)
        island_label.resize(n);
        for (uint32_t i = 0; i < n; ++i) island_label[i] = i;
        auto find = [&](uint32_t x) {
            while (island_label[x] != x) { island_label[x] = island_label[island_label[x]]; x = island_label[x]; }
            return x;
        };
        auto unite = [&](uint32_t a, uint32_t b) {
            if (!bodies[a].procedural() || !bodies[b].procedural()) return;   // static/kinematic nodes do not connect
            uint32_t ra = find(a), rb = find(b);
            if (ra == rb) return;
            if (ra < rb) island_label[rb] = ra; else island_label[ra] = rb;
        };
        for (auto &kv : manifolds) unite(kv.second.body[0], kv.second.body[1]);   // every manifold is a graph edge
        for (auto &j : joints) if (j.alive) unite(j.body[0], j.body[1]);
        uint32_t count = 0;
        for (uint32_t i = 0; i < n; ++i) {
            island_label[i] = find(i);
            if (bodies[i].procedural() && island_label[i] == i) ++count;
        }
        stats.num_islands = count;
        // merge_islands (island_manager.cpp:297-350): when islands merge, the BIGGEST of them - nodes + edges - survives with its
        // sleep_timestamp and takes the others in. Labels here are lowest body indices, so the surviving timer has to be carried to the
        // merged island's label: per new island, the timer of the biggest of last step's islands it is made of, size = procedural bodies
        // + edges (manifolds that existed before this step, joints); ties: the lowest old label. (The engine's count also has each
        // island's non-procedural nodes - a set with history: a static body stays in it after its last edge is gone, until a split
        // rebuilds it - and among equals it keeps the first in an ECS iteration order; both only matter between islands of nearly the same
        // size. Demonstrated against the engine with a four-box scene: tests/test_reference_engine.py::test_island_merge_keeps_the_bigger_*.)
        if (sleeping) {
            sleep_since.resize(n, -1.0);
            std::vector<uint32_t> size_old(n, 0);
            auto old_label = [&](uint32_t i) { return i < prev_label.size() ? prev_label[i] : i; };
            for (uint32_t i = 0; i < n && i < prev_label.size(); ++i) if (bodies[i].procedural() && !bodies[i].removed && prev_label[i] < n) ++size_old[prev_label[i]];
            std::vector<uint64_t> fresh(new_keys);
            std::sort(fresh.begin(), fresh.end());
            for (auto &kv : manifolds) {
                if (std::binary_search(fresh.begin(), fresh.end(), kv.first)) continue;   // created this step: in no island yet
                const uint32_t a = kv.second.body[0], b = kv.second.body[1];
                const uint32_t l = old_label(bodies[a].procedural() ? a : b);
                if (l < n) ++size_old[l];
            }
            for (auto &j : joints) if (j.alive) { const uint32_t l = old_label(bodies[j.body[0]].procedural() ? j.body[0] : j.body[1]); if (l < n) ++size_old[l]; }
            std::vector<double> carried(n, -1.0);
            std::vector<uint32_t> best(n, 0xFFFFFFFFu);
            for (uint32_t r = 0; r < n && r < prev_label.size(); ++r) {
                if (prev_label[r] != r || !bodies[r].procedural() || bodies[r].removed) continue;   // last step's roots
                const uint32_t L = island_label[r];
                if (best[L] == 0xFFFFFFFFu || size_old[r] > size_old[best[L]]) { best[L] = r; carried[L] = sleep_since[r]; }
            }
            for (uint32_t i = 0; i < n; ++i) sleep_since[i] = (bodies[i].procedural() && island_label[i] == i) ? carried[i] : -1.0;
        }
        // split_islands (island_manager.cpp:411-447): when an island falls apart, the largest component is MOVED into the island entity
        // - a move assignment of a freshly built `island`, whose sleep_timestamp is empty - and the others become new islands: every part
        // starts its sleep timer again.
        split_reset_.assign(n, 0);
        for (uint32_t i = 0; i < n && i < prev_label.size(); ++i) {
            if (!bodies[i].procedural() || bodies[i].removed) continue;
            const uint32_t r = prev_label[i];   // last step's root of this body: a body of the same old island
            if (r < n && bodies[r].procedural() && !bodies[r].removed && island_label[r] != island_label[i]) { split_reset_[island_label[i]] = 1; split_reset_[island_label[r]] = 1; }
        }
        if (sleeping) update_sleep();   // island_manager::update: put_islands_to_sleep() still sees the PREVIOUS step's stamp ...
        sim_clock = pending_stamp_;     // ... and only then m_last_time = timestamp (island_manager.cpp:533-539)
    }
    // wake_up / put_islands_to_sleep (island_manager.cpp:524-539, 573-623). An island that received a new edge, or
    // that holds both sleeping and awake bodies (a merge), wakes as a whole. An awake island whose bodies are all below
    // the thresholds starts / continues its timer and goes to sleep once it has run for more than island_time_to_sleep;
    // the timer compares the PREVIOUS update's time stamps, i.e. (steps elapsed) * dt.
    void update_sleep() {
        const uint32_t n = (uint32_t)bodies.size();
        enum { FAST = 1, DISABLED = 2, HAS_ASLEEP = 4, HAS_AWAKE = 8, WAKE = 16 };
        std::vector<uint8_t> st(n, 0);
        sleep_since.resize(n, -1.0);
        const float lin2 = 0.005f * 0.005f, ang = kPi / 48.0f, ang2 = ang * ang;   // config/constants.hpp:41-42
        for (uint32_t i = 0; i < n; ++i) {
            const Body &b = bodies[i];
            if (!b.procedural()) continue;
            uint8_t &s = st[island_label[i]];
            if (length_sqr(b.linvel) > lin2 || length_sqr(b.angvel) > ang2) s |= FAST;
            if (b.sleeping_disabled) s |= DISABLED;
            s |= b.asleep ? HAS_ASLEEP : HAS_AWAKE;
        }
        for (uint64_t key : new_keys) {
            auto it = manifolds.find(key);
            if (it == manifolds.end()) continue;
            const uint32_t a = it->second.body[0], b = it->second.body[1];
            st[bodies[a].procedural() ? island_label[a] : island_label[b]] |= WAKE;
        }
        std::vector<uint8_t> action(n, 0);   // 0 keep, 1 awake, 2 sleep
        for (uint32_t i = 0; i < n; ++i) {
            if (!bodies[i].procedural() || island_label[i] != i) { sleep_since[i] = -1; continue; }
            if (i < split_reset_.size() && split_reset_[i]) sleep_since[i] = -1;   // a part of an island that split: its timer starts again
            const uint8_t s = st[i];
            const bool wake = (s & WAKE) || ((s & HAS_ASLEEP) && (s & HAS_AWAKE));
            if ((s & HAS_ASLEEP) && !(s & HAS_AWAKE) && !wake) { action[i] = 0; continue; }   // stays asleep
            action[i] = 1;
            if (!(s & DISABLED) && !(s & FAST)) {
                if (sleep_since[i] < 0) sleep_since[i] = sim_clock;
                else if (sim_clock - sleep_since[i] > 2.0) { action[i] = 2; sleep_since[i] = -1; }   // island_time_to_sleep
            } else sleep_since[i] = -1;
        }
        for (uint32_t i = 0; i < n; ++i) {
            Body &b = bodies[i];
            if (!b.procedural()) continue;
            const uint8_t a = action[island_label[i]];
            if (a == 1) b.asleep = false;
            else if (a == 2) { b.asleep = true; b.linvel = {0, 0, 0}; b.angvel = {0, 0, 0}; }
        }
    }

    // ---------------- colouring (shared spec with the GPU; see DESIGN.md "Colouring") ----------------
    // Edges = constraints between two bodies; only procedural endpoints constrain the colour. Each round every
    // uncoloured edge whose priority is the maximum among the uncoloured edges at all of its procedural endpoints
    // takes the lowest colour free at those endpoints. Decisions in a round depend only on the state before it.
    template <typename EdgeAt>
    uint32_t colour_edges(uint32_t num_edges, EdgeAt &&edge, std::vector<uint64_t> &used, bool serial_bucket) {
        std::vector<uint64_t> best(bodies.size(), 0);
        uint32_t rounds = 0;
        for (;;) {
            bool any = false;
            for (uint32_t e = 0; e < num_edges; ++e) {
                uint32_t a, b; uint32_t *col;
                edge(e, a, b, col);
                if (!col || *col != kNoColour) continue;
                any = true;
                uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);   // lower index wins == sequential first-fit in canonical order
                if (bodies[a].procedural()) best[a] = std::max(best[a], pr);
                if (bodies[b].procedural()) best[b] = std::max(best[b], pr);
            }
            if (!any) break;
            ++rounds;
            std::vector<std::pair<uint32_t, uint32_t>> chosen;
            for (uint32_t e = 0; e < num_edges; ++e) {
                uint32_t a, b; uint32_t *col;
                edge(e, a, b, col);
                if (!col || *col != kNoColour) continue;
                uint64_t pr = (uint64_t)(0xFFFFFFFFu - e);   // lower index wins == sequential first-fit in canonical order
                bool pa = bodies[a].procedural(), pb = bodies[b].procedural();
                if ((pa && best[a] != pr) || (pb && best[b] != pr)) continue;
                uint64_t busy = (pa ? used[a] : 0) | (pb ? used[b] : 0);
                // colours 0..61 are conflict-free sets; what finds none of them free (a body with more than 62 coloured
                // contacts) goes to colour 62, the bucket the device solves serially - and this loop visits serially anyway
                uint32_t c = 0;
                if (serial_bucket) {
                    while (c < kSerialColour && (busy >> c & 1)) ++c;
                } else {   // joints: 64 colours, more is an error on the device
                    while (c < kMaxColours && (busy >> c & 1)) ++c;
                    if (c >= kMaxColours) { colour_overflow_ = true; c = kMaxColours - 1; }
                }
                *col = c;
                if (pa) used[a] |= 1ull << c;
                if (pb) used[b] |= 1ull << c;
            }
            std::fill(best.begin(), best.end(), 0);
        }
        return rounds;
    }
    void colour_contacts() {
        std::vector<Manifold *> ms;
        ms.reserve(manifolds.size());
        for (auto &kv : manifolds) ms.push_back(&kv.second);
        std::vector<uint64_t> used(bodies.size(), 0);
        // In every island that has an edge to colour in this step (a new or re-activated contact) the top colour carried over from
        // the last step is released and first-fit again: colour classes freed by vanished contacts are reclaimed when the island next
        // changes, so the colour count does not drift up - and an island in which nothing happened keeps its colouring untouched.
        // Per island, not per world: an island is coloured - and therefore solved - the same way whatever else the world holds, so a
        // shard of the world steps exactly like the whole (edyn_amd/parallel.py).
        std::vector<uint32_t> top(bodies.size(), 0);   // per island label: its highest carried colour + 1
        std::vector<uint8_t> changed(bodies.size(), 0);
        auto label_of = [&](const Manifold &m) { return bodies[m.body[0]].procedural() ? island_label[m.body[0]] : island_label[m.body[1]]; };
        for (Manifold *m : ms) {
            if (manifold_asleep(*m) || m->num_points == 0) continue;
            if (m->colour != kNoColour) top[label_of(*m)] = std::max(top[label_of(*m)], m->colour + 1);
            else changed[label_of(*m)] = 1;
        }
        for (Manifold *m : ms) {
            if (manifold_asleep(*m)) {   // out of the solve, but it keeps its colour for when the island wakes
                if (m->colour != kNoColour) for (int s = 0; s < 2; ++s) if (bodies[m->body[s]].procedural()) used[m->body[s]] |= 1ull << m->colour;
                continue;
            }
            if (m->num_points == 0) { m->colour = kNoColour; continue; }   // inactive edges hold no colour
            if (m->colour != kNoColour && changed[label_of(*m)] && top[label_of(*m)] >= 2 && m->colour + 1 == top[label_of(*m)]) m->colour = kNoColour;
            if (m->colour != kNoColour) {
                for (int s = 0; s < 2; ++s) if (bodies[m->body[s]].procedural()) used[m->body[s]] |= 1ull << m->colour;
            }
        }
        stats.colour_rounds = colour_edges((uint32_t)ms.size(), [&](uint32_t e, uint32_t &a, uint32_t &b, uint32_t *&col) {
            a = ms[e]->body[0]; b = ms[e]->body[1];
            col = (ms[e]->num_points > 0 && !manifold_asleep(*ms[e])) ? &ms[e]->colour : nullptr;
        }, used, true);
        uint32_t nc = 0;
        for (Manifold *m : ms) if (m->colour != kNoColour) nc = std::max(nc, m->colour + 1);
        stats.num_colours = nc;
    }
    void colour_joints() {
        if (joints_coloured_) return;
        for (auto &j : joints) j.colour = kNoColour;
        std::vector<uint64_t> used(bodies.size(), 0);
        std::vector<uint32_t> live;   // removed joints keep their index but take no colour
        for (uint32_t e = 0; e < joints.size(); ++e) if (joints[e].alive) live.push_back(e);
        colour_edges((uint32_t)live.size(), [&](uint32_t e, uint32_t &a, uint32_t &b, uint32_t *&col) {
            a = joints[live[e]].body[0]; b = joints[live[e]].body[1]; col = &joints[live[e]].colour;
        }, used, false);
        uint32_t nc = 0;
        for (auto &j : joints) if (j.alive) nc = std::max(nc, j.colour + 1);
        stats.num_joint_colours = nc;
        joints_coloured_ = true;
    }

    // ---------------- solver ----------------
    struct BodyRef {   // solver.cpp:83-147: static => zero velocity; non-procedural => zero inverse mass, dummy deltas
        vec3 pos; quat orn; vec3 linvel, angvel; float inv_m; mat3 inv_I; vec3 *dv, *dw;
        vec3 roll_dir{0, 0, 0};
        vec3 org_{0, 0, 0};
        vec3 org() const { return org_; }   // constraint_body::origin (constraints/constraint_body.hpp:10-18)
    };
    BodyRef body_ref(uint32_t i) {
        Body &b = bodies[i];
        BodyRef r;
        r.pos = b.pos; r.orn = b.orn; r.roll_dir = b.roll_direction(); r.org_ = b.org();
        if (b.procedural()) { r.inv_m = b.mass_inv; r.inv_I = b.I_inv_world; r.dv = &b.dv; r.dw = &b.dw; }
        else { r.inv_m = 0; r.inv_I = kMat3Zero; r.dv = &dummy_dv_; r.dw = &dummy_dw_; }
        if (b.kind == KIND_STATIC) { r.linvel = {0, 0, 0}; r.angvel = {0, 0, 0}; }
        else { r.linvel = b.linvel; r.angvel = b.angvel; }
        return r;
    }
    static void finish_row(Row &r, const RowOptions &o, const BodyRef &A, const BodyRef &B) {
        r.inv_mA = A.inv_m; r.inv_IA = A.inv_I; r.inv_mB = B.inv_m; r.inv_IB = B.inv_I;
        r.dvA = A.dv; r.dwA = A.dw; r.dvB = B.dv; r.dwB = B.dw;
        prepare_row(r, o, A.linvel, A.angvel, B.linvel, B.angvel);
    }
    // contact_constraint.cpp:15-56
    struct ExtraRows { bool roll = false, spin = false; FrictionRow rr; SpinRow sr; };
    void prepare_contact(const ContactPoint &cp, const BodyRef &A, const BodyRef &B, Row &nr, FrictionRow &fr, ExtraRows *ex = nullptr, int num_points = 1) {
        vec3 pAw = to_world(cp.pivotA, A.org(), A.orn), pBw = to_world(cp.pivotB, B.org(), B.orn);
        vec3 rA = pAw - A.pos, rB = pBw - B.pos;
        const vec3 n = cp.normal;
        nr.J[0] = n; nr.J[1] = cross(rA, n); nr.J[2] = -n; nr.J[3] = -cross(rB, n);
        nr.impulse = cp.normal_impulse;
        nr.lower = 0; nr.upper = kLarge;
        RowOptions o;
        o.restitution = 0;   // solver.cpp:282-283: restitution solver enabled => rows carry zero restitution
        if (cp.distance > 0) o.error = cp.distance / dt;
        if (cp.extras() && cp.distance < 0 && cp.stiffness < kLarge) {   // soft contact, contact_extras_constraint.cpp:16-35
            const vec3 vA = A.linvel + cross(A.angvel, rA), vB = B.linvel + cross(B.angvel, rB);
            const float normal_relvel = dot(vA - vB, n);
            const float spring_force = -cp.distance * cp.stiffness / (float)num_points;
            const float damper_force = -normal_relvel * cp.damping / (float)num_points;
            nr.upper = std::max(spring_force + damper_force, 0.0f) * dt;
            o.error = -kLarge;
        }
        finish_row(nr, o, A, B);
        if (ex) {
            ex->roll = cp.extras() && cp.roll_friction > 0;
            ex->spin = cp.extras() && cp.spin_friction > 0;
            if (ex->roll) {   // :37-64
                ex->rr.mu = cp.roll_friction;
                vec3 t[2];
                plane_space(n, t[0], t[1]);
                for (int i = 0; i < 2; ++i) {
                    auto &ri = ex->rr.row[i];
                    // a body with a rolling direction scales the axis down by the projection of that direction on it
                    // (both directions are rotated by body A's orientation, as the reference does, :50-55)
                    for (const vec3 &rd : {A.roll_dir, B.roll_dir})
                        if (rd.x != 0 || rd.y != 0 || rd.z != 0) t[i] *= dot(rotate(A.orn, rd), t[i]);
                    ri.J[0] = {0, 0, 0}; ri.J[1] = t[i]; ri.J[2] = {0, 0, 0}; ri.J[3] = -t[i];
                    ri.impulse = cp.rolling_impulse[i];
                    const float s = dot(A.inv_I * ri.J[1], ri.J[1]) + dot(B.inv_I * ri.J[3], ri.J[3]);
                    ri.eff_mass = s > kEps ? 1.0f / s : 0.0f;
                    ri.rhs = -relative_speed(ri.J, A.linvel, A.angvel, B.linvel, B.angvel);
                }
            }
            if (ex->spin) {   // :66-78
                SpinRow &sr = ex->sr;
                sr.mu = cp.spin_friction;
                sr.J[0] = n; sr.J[1] = -n;
                sr.impulse = cp.spin_impulse;
                const float s = dot(A.inv_I * sr.J[0], sr.J[0]) + dot(B.inv_I * sr.J[1], sr.J[1]);
                sr.eff_mass = 1.0f / s;
                sr.rhs = -(dot(sr.J[0], A.angvel) + dot(sr.J[1], B.angvel));
            }
        }
        fr.mu = cp.friction;
        vec3 t[2];
        plane_space(n, t[0], t[1]);
        for (int i = 0; i < 2; ++i) {
            auto &ri = fr.row[i];
            ri.J[0] = t[i]; ri.J[1] = cross(rA, t[i]); ri.J[2] = -t[i]; ri.J[3] = -cross(rB, t[i]);
            ri.impulse = cp.friction_impulse[i];
            ri.eff_mass = effective_mass(ri.J, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
            ri.rhs = -relative_speed(ri.J, A.linvel, A.angvel, B.linvel, B.angvel);
        }
    }
    // point_constraint.cpp:9-46 / hinge_constraint.cpp:26-178. Returns the number of rows; slot[r] = the applied-impulse slot
    // row r reads and (after the solve) writes (store_applied_impulses, point_constraint.cpp:48-58, hinge_constraint.cpp:215-257).
    static constexpr int kMaxJointRows = kJointSlotsO;
    static float acos_cr(float x) { return g_libm_trig ? std::acos(x) : (float)std::acos((double)x); }
    static float atan2_cr(float y, float x) { return g_libm_trig ? std::atan2(y, x) : (float)std::atan2((double)y, (double)x); }
    static float asin_cr(float x) { return g_libm_trig ? std::asin(x) : (float)std::asin((double)x); }
    static quat shortest_arc(vec3 v0, vec3 v1) {   // quaternion.cpp:25-38
        const vec3 c = cross(v0, v1);
        const float d = dot(v0, v1);
        if (d <= -1 + kEps) {
            vec3 n, m;
            plane_space(v0, n, m);
            return {n.x, n.y, n.z, 0};
        }
        const float s = std::sqrt((1 + d) * 2);
        const float rs = 1 / s;
        return normalize(quat{c.x * rs, c.y * rs, c.z * rs, s * 0.5f});
    }
    static float cvjoint_relative_angle(const Joint &j, quat ornA, quat ornB, vec3 twist_axisA, vec3 twist_axisB) {   // cvjoint_constraint.cpp:25-37
        const quat arc = shortest_arc(twist_axisB, twist_axisA);
        const vec3 angle_axisB = rotate(conjugate(ornA) * arc * ornB, j.frame[1].column(1));
        return atan2_cr(dot(angle_axisB, j.frame[0].column(2)), dot(angle_axisB, j.frame[0].column(1)));
    }
    static void track_angle(Joint &j, float new_angle) {   // update_angle, :39-47 (the hinge's rule)
        const float previous = normalize_angle(j.angle);
        const float d0 = new_angle - previous;
        const float d1 = d0 + kPi2 * (d0 < 0 ? 1.0f : -1.0f);
        j.angle += std::fabs(d0) < std::fabs(d1) ? d0 : d1;
    }
    static float normalize_angle(float a) {   // math.hpp:53-63
        a = std::fmod(a, kPi2);
        if (a < -kPi) return a + kPi2;
        if (a > kPi) return a - kPi2;
        return a;
    }
    int prepare_joint(Joint &j, const BodyRef &A, const BodyRef &B, Row *rows, int *slot) {
        if (j.type == JOINT_NULL) return 0;
        vec3 pA = to_world(j.pivot[0], A.org(), A.orn), pB = to_world(j.pivot[1], B.org(), B.orn);
        vec3 rA = pA - A.pos, rB = pB - B.pos;
        if (j.type == JOINT_GENERIC) {   // generic_constraint.cpp:10-258
            const vec3 pivot_offset = pB - pA;
            int n = 0;
            auto add = [&](const vec3 (&J)[4], int sl, float lo, float hi, const RowOptions &o) {
                Row &r = rows[n];
                for (int k = 0; k < 4; ++k) r.J[k] = J[k];
                r.lower = lo; r.upper = hi; r.impulse = j.impulse[sl];
                finish_row(r, o, A, B);
                slot[n] = sl; ++n;
            };
            const vec3 axisA_x = rotate(A.orn, j.frame[0].column(0)), axisB_x = rotate(B.orn, j.frame[1].column(0));
            for (int d = 0; d < 6; ++d) {
                const float *P = j.params + 10 * d;
                const bool limit_enabled = P[0] != 0, angular = d >= 3;
                const float vmin = P[1], vmax = P[2], limit_restitution = P[3], bump_len = P[4], bump_stiffness = P[5], friction = P[6],
                            rest = P[7], spring_stiffness = P[8], damping = P[9];
                const bool non_zero_limit = vmin < vmax;
                vec3 J[4];
                float current;
                vec3 axA, axB;
                if (!angular) {
                    const vec3 axisA = rotate(A.orn, j.frame[0].column(d));
                    J[0] = axisA; J[1] = cross(rA, axisA); J[2] = -axisA; J[3] = -cross(rB, axisA);
                    current = dot(pivot_offset, axisA);
                } else {
                    const int i = d - 3;
                    if (i == 0) {
                        const quat arc = shortest_arc(axisB_x, axisA_x);
                        const vec3 angle_axisB = rotate(conjugate(A.orn) * arc * B.orn, j.frame[1].column(1));
                        current = atan2_cr(dot(angle_axisB, j.frame[0].column(2)), dot(angle_axisB, j.frame[0].column(1)));
                        axA = axisA_x; axB = axisB_x;
                    } else {
                        const vec3 other = rotate(A.orn, j.frame[0].column(i == 1 ? 2 : 1));
                        const float cos_angle = std::min(std::max(dot(axisB_x, other), -1.0f), 1.0f);
                        current = kPi * 0.5f - acos_cr(cos_angle);
                        vec3 axis = cross(other, axisB_x);
                        if (!try_normalize(axis)) axis = i == 1 ? vec3{0, 0, 1} : vec3{0, 1, 0};
                        axA = axB = -axis;
                    }
                    J[0] = {0, 0, 0}; J[1] = axA; J[2] = {0, 0, 0}; J[3] = -axB;
                }
                if (limit_enabled) {
                    RowOptions o;
                    float lo = -kLarge, hi = kLarge;
                    if (non_zero_limit) {
                        float limit_error;
                        const float mid = (vmin + vmax) / 2.0f;
                        if (current < mid) { limit_error = vmin - current; lo = -kLarge; hi = 0; }
                        else { limit_error = vmax - current; lo = 0; hi = kLarge; }
                        if (angular) o.error = limit_error / dt;
                        else { if (current > vmin && current < vmax) o.error = limit_error / dt; o.erp = 0.9f; }
                        o.restitution = limit_restitution;
                    } else if (angular) {
                        o.error = -current / dt;
                    }
                    add(J, 4 * d, lo, hi, o);
                }
                if (limit_enabled && non_zero_limit && bump_stiffness > 0 && bump_len > 0) {
                    float defl = 0;
                    const float bmin = vmin + bump_len, bmax = vmax - bump_len;
                    if (current < bmin) defl = current - bmin;
                    else if (current > bmax) defl = current - bmax;
                    const float imp = bump_stiffness * defl * dt;
                    RowOptions o; o.error = -defl / dt;
                    add(J, 4 * d + 1, std::min(imp, 0.0f), std::max(0.0f, imp), o);
                }
                if (spring_stiffness > 0) {
                    const float defl = current - rest;
                    const float imp = spring_stiffness * defl * dt;
                    RowOptions o; o.error = -defl / dt;
                    add(J, 4 * d + 2, std::min(imp, 0.0f), std::max(0.0f, imp), o);
                }
                if (friction > 0 || damping > 0) {
                    float fi = friction * dt;
                    if (damping > 0) {
                        const float rel = angular ? dot(A.angvel, axA) - dot(B.angvel, axB) : relative_speed(J, A.linvel, A.angvel, B.linvel, B.angvel);
                        fi += std::fabs(rel) * damping * dt;
                    }
                    add(J, 4 * d + 3, -fi, fi, RowOptions{});
                }
            }
            return n;
        }
        if (j.type == JOINT_GRAVITY) {   // gravity_constraint.cpp:6-28: Newtonian attraction as an impulse-limited row
            const vec3 d = A.pos - B.pos;
            const float l2 = std::max(length_sqr(d), kEps);
            const float l = std::sqrt(l2);
            const vec3 dn = d / l;
            const float F = kGravitationalConstant / (l2 * A.inv_m * B.inv_m);
            const float P = F * dt;
            Row &r = rows[0];
            r.J[0] = dn; r.J[1] = {0, 0, 0}; r.J[2] = -dn; r.J[3] = -vec3{0, 0, 0};
            r.lower = -P; r.upper = P; r.impulse = j.impulse[0];
            RowOptions o; o.error = kLarge;
            finish_row(r, o, A, B);
            slot[0] = 0;
            return 1;
        }
        if (j.type == JOINT_CONE) {   // cone_constraint.cpp:12-95
            const vec3 pivotB_world = pB;
            const vec3 pivotB_in_A = to_object(pivotB_world, A.org(), A.orn);
            const vec3 pf = to_object(pivotB_in_A, j.pivot[0], j.frame[0]);
            const float scaling_y = 1.0f / j.params[0], scaling_z = 1.0f / j.params[1];
            const vec3 ps = pf * vec3{1, scaling_y, scaling_z};
            const float proj_yz_len_sqr = ps.y * ps.y + ps.z * ps.z;
            vec3 normal_scaled, tangent_scaled;
            if (proj_yz_len_sqr > kEps) {
                normal_scaled = normalize(vec3{-std::sqrt(proj_yz_len_sqr), ps.y, ps.z});
                tangent_scaled = normalize(vec3{0, -ps.z, ps.y});
            } else {
                normal_scaled = normalize(vec3{-1, 1, 0});
                tangent_scaled = normalize(vec3{0, 0, 1});
            }
            const float error = dot(ps, normal_scaled);
            const vec3 dir_on_cone{-normal_scaled.x, normal_scaled.y, normal_scaled.z};
            const float cone_proj = dot(ps, dir_on_cone);
            const vec3 point_on_cone_scaled = dir_on_cone * cone_proj;
            const vec3 descale{1, 1 / scaling_y, 1 / scaling_z};
            const vec3 point_on_cone = point_on_cone_scaled * descale;
            const vec3 pivotA = to_world(point_on_cone, j.pivot[0], j.frame[0]);
            const vec3 pivotA_world = to_world(pivotA, A.org(), A.orn);
            const vec3 tangent = normalize(tangent_scaled * descale);
            const vec3 normal = normalize(cross(tangent, point_on_cone));
            const vec3 nw = rotate(A.orn, j.frame[0] * normal);
            const vec3 cA = pivotA_world - A.pos, cB = pivotB_world - B.pos;
            auto cone_row = [&](int n, float lo, float hi, const RowOptions &o) {
                Row &r = rows[n];
                r.J[0] = nw; r.J[1] = cross(cA, nw); r.J[2] = -nw; r.J[3] = -cross(cB, nw);
                r.lower = lo; r.upper = hi; r.impulse = j.impulse[n];
                finish_row(r, o, A, B);
                slot[n] = n;
            };
            RowOptions o; o.error = -error / dt; o.restitution = j.params[2];
            cone_row(0, 0, kLarge, o);
            if (j.params[3] > 0 && j.params[4] > 0) {
                const float deflection = j.params[4] + error;
                const float spring_impulse = j.params[3] * deflection * dt;
                RowOptions ob; ob.error = -deflection / dt;
                cone_row(1, 0, std::max(0.0f, spring_impulse), ob);
                return 2;
            }
            return 1;
        }
        if (j.type == JOINT_CVJOINT) {   // cvjoint_constraint.cpp:49-222
            const float *P = j.params;
            const float twist_min = P[0], twist_max = P[1], twist_restitution = P[2], bump_angle = P[3], bump_stiffness = P[4],
                        twist_friction_torque = P[5], twist_rest_angle = P[6], twist_stiffness = P[7], twist_damping = P[8],
                        bend_stiffness = P[12], bend_friction_torque = P[13], bend_damping = P[14];
            const vec3 rest_direction{P[9], P[10], P[11]};
            mat3 sA = skew(rA), sB = skew(rB);
            int n = 0;
            for (int i = 0; i < 3; ++i) {
                Row &r = rows[n];
                r.J[0] = kMat3Identity.row[i]; r.J[1] = -sA.row[i]; r.J[2] = -kMat3Identity.row[i]; r.J[3] = sB.row[i];
                r.lower = -kLarge; r.upper = kLarge; r.impulse = j.impulse[n];
                finish_row(r, RowOptions{}, A, B);
                slot[n] = n; ++n;
            }
            const vec3 tA = rotate(A.orn, j.frame[0].column(0)), tB = rotate(B.orn, j.frame[1].column(0));
            auto ang_row = [&](vec3 axA, vec3 axB, int sl, float lo, float hi, const RowOptions &o) {
                Row &r = rows[n];
                r.J[0] = {0, 0, 0}; r.J[1] = axA; r.J[2] = {0, 0, 0}; r.J[3] = -axB;
                r.lower = lo; r.upper = hi; r.impulse = j.impulse[sl];
                finish_row(r, o, A, B);
                slot[n] = sl; ++n;
            };
            const bool has_limit = twist_min < twist_max;
            {
                const float angle = cvjoint_relative_angle(j, A.orn, B.orn, tA, tB);
                RowOptions o;
                float lo = -kLarge, hi = kLarge;
                if (has_limit) {
                    track_angle(j, angle);
                    float limit_error;
                    const float mid = (twist_min + twist_max) / 2.0f;
                    if (j.angle < mid) { limit_error = twist_min - j.angle; lo = -kLarge; hi = 0; }
                    else { limit_error = twist_max - j.angle; lo = 0; hi = kLarge; }
                    if (j.angle > twist_min && j.angle < twist_max) o.error = limit_error / dt;
                    o.restitution = twist_restitution;
                }
                ang_row(tA, tB, 3, lo, hi, o);
            }
            if (has_limit && bump_stiffness > 0 && bump_angle > 0) {
                float defl = 0;
                const float bmin = twist_min + bump_angle, bmax = twist_max - bump_angle;
                if (j.angle < bmin) defl = j.angle - bmin;
                else if (j.angle > bmax) defl = j.angle - bmax;
                const float imp = bump_stiffness * defl * dt;
                RowOptions o; o.error = -defl / dt;
                ang_row(tA, tB, 4, std::min(imp, 0.0f), std::max(0.0f, imp), o);
            }
            if (has_limit && twist_stiffness > 0) {
                const float defl = j.angle - twist_rest_angle;
                const float imp = twist_stiffness * defl * dt;
                RowOptions o; o.error = -defl / dt;
                ang_row(tA, tB, 5, std::min(imp, 0.0f), std::max(0.0f, imp), o);
            }
            if (has_limit && (twist_friction_torque > 0 || twist_damping > 0)) {
                float fi = twist_friction_torque * dt;
                if (twist_damping > 0) {
                    const float relvel = dot(A.angvel, tA) - dot(B.angvel, tB);
                    fi += std::fabs(relvel) * twist_damping * dt;
                }
                ang_row(tA, tB, 6, -fi, fi, RowOptions{});
            }
            if (bend_friction_torque > 0 || bend_damping > 0) {
                const vec3 twA = dot(A.angvel, tA) * tA, twB = dot(B.angvel, tB) * tB;
                const vec3 angvel_rel = (A.angvel - twA) - (B.angvel - twB);
                const float angspd_rel = length(angvel_rel);
                vec3 axis;
                if (angspd_rel > kEps) axis = angvel_rel / angspd_rel;
                else axis = rotate(A.orn, j.frame[0].column(1));
                float fi = bend_friction_torque * dt;
                if (twist_damping > 0) fi += std::fabs(angspd_rel) * bend_damping * dt;   // (the reference tests twist_damping here, :199)
                ang_row(axis, axis, 7, -fi, fi, RowOptions{});
            }
            if (bend_stiffness > 0) {
                vec3 bend_axis = cross(rotate(A.orn, rest_direction), tB);
                const float len = length(bend_axis);
                const float angle = asin_cr(len);
                if (len > kEps) bend_axis /= len;
                else bend_axis = rotate(A.orn, j.frame[0].column(1));
                const float imp = bend_stiffness * angle * dt;
                RowOptions o; o.error = -angle / dt;
                ang_row(bend_axis, bend_axis, 8, std::min(imp, 0.0f), std::max(0.0f, imp), o);
            }
            return n;
        }
        if (j.type == JOINT_DISTANCE || j.type == JOINT_SOFT_DISTANCE) {
            // distance_constraint.cpp:7-31 (params[0] = distance) / soft_distance_constraint.cpp:8-62 (distance, stiffness, damping)
            vec3 d = pA - pB;
            const float dist_sqr = length_sqr(d);
            auto dir_row = [&](int n, vec3 dir, vec3 p, vec3 q, float lo, float hi, const RowOptions &o) {
                Row &r = rows[n];
                r.J[0] = dir; r.J[1] = p; r.J[2] = -dir; r.J[3] = -q;
                r.lower = lo; r.upper = hi; r.impulse = j.impulse[n];
                finish_row(r, o, A, B);
                slot[n] = n;
            };
            if (j.type == JOINT_DISTANCE) {
                if (!(dist_sqr > kEps)) d = vec3{1, 0, 0};
                RowOptions o; o.error = 0.5f * (dist_sqr - j.params[0] * j.params[0]) / dt;
                dir_row(0, d, cross(rA, d), cross(rB, d), -kLarge, kLarge, o);
                return 1;
            }
            const float dist = std::sqrt(dist_sqr);
            vec3 dn;
            if (dist_sqr > kEps) dn = d / dist; else dn = vec3{1, 0, 0};
            const vec3 p = cross(rA, dn), q = cross(rB, dn);
            const float spring_impulse = j.params[1] * (j.params[0] - dist) * dt;
            RowOptions os; os.error = spring_impulse > 0 ? -kLarge : kLarge;
            dir_row(0, dn, p, q, std::min(spring_impulse, 0.0f), std::max(0.0f, spring_impulse), os);
            const vec3 Jd[4] = {dn, p, -dn, -q};
            const float relspd = relative_speed(Jd, A.linvel, A.angvel, B.linvel, B.angvel);
            const float damping_impulse = j.params[2] * relspd * dt;
            dir_row(1, dn, p, q, -std::fabs(damping_impulse), std::fabs(damping_impulse), RowOptions{});
            return 2;
        }
        mat3 sA = skew(rA), sB = skew(rB);
        int n = 0;
        for (int i = 0; i < 3; ++i) {
            Row &r = rows[n];
            r.J[0] = kMat3Identity.row[i]; r.J[1] = -sA.row[i]; r.J[2] = -kMat3Identity.row[i]; r.J[3] = sB.row[i];
            r.lower = -kScalarMax; r.upper = kScalarMax;
            r.impulse = j.impulse[n];
            RowOptions o;
            if (j.type == JOINT_POINT) o.error = (pA[i] - pB[i]) / dt;
            finish_row(r, o, A, B);
            slot[n] = n;
            ++n;
        }
        auto axial_row = [&](vec3 ax, int sl, float lo, float hi, const RowOptions &o) {
            Row &r = rows[n];
            r.J[0] = {0, 0, 0}; r.J[1] = ax; r.J[2] = {0, 0, 0}; r.J[3] = -ax;
            r.lower = lo; r.upper = hi;
            r.impulse = j.impulse[sl];
            finish_row(r, o, A, B);
            slot[n] = sl;
            ++n;
        };
        if (j.type == JOINT_POINT) {
            const float friction_torque = j.params[0];
            if (friction_torque > 0) {   // point_constraint.cpp:33-46
                vec3 spin = A.angvel - B.angvel;
                const float lsqr = length_sqr(spin);
                if ((double)lsqr > 1e-18) {   // try_normalize, vector3.hpp:239-248
                    spin /= std::sqrt(lsqr);
                    const float fi = friction_torque * dt;
                    axial_row(spin, 3, -fi, fi, RowOptions{});
                }
            }
            return n;
        }
        vec3 p = rotate(A.orn, j.frame[0].column(1)), q = rotate(A.orn, j.frame[0].column(2));
        axial_row(p, 3, -kScalarMax, kScalarMax, RowOptions{});
        axial_row(q, 4, -kScalarMax, kScalarMax, RowOptions{});
        const float angle_min = j.params[0], angle_max = j.params[1], limit_restitution = j.params[2], bump_stop_angle = j.params[3],
                    bump_stop_stiffness = j.params[4], torque = j.params[5], speed = j.params[6], rest_angle = j.params[7],
                    stiffness = j.params[8], damping = j.params[9];
        const bool has_limit = angle_min < angle_max, has_spring = stiffness > 0, has_torque = torque > 0 || damping > 0;
        vec3 hinge_axis{0, 0, 0};
        if (has_limit || has_spring || has_torque) hinge_axis = rotate(A.orn, j.frame[0].column(0));
        if (has_limit || has_spring) {   // hinge_constraint.cpp:80-89
            const vec3 angle_axisB = rotate(B.orn, j.frame[1].column(1));
            const float current = atan2_cr(dot(angle_axisB, q), dot(angle_axisB, p));
            const float previous = normalize_angle(j.angle);
            const float d0 = current - previous;
            const float d1 = d0 + kPi2 * (d0 < 0 ? 1.0f : -1.0f);
            j.angle += std::fabs(d0) < std::fabs(d1) ? d0 : d1;
        }
        if (has_limit) {   // :91-142
            RowOptions o;
            float lo, hi, limit_error;
            const float halfway = (angle_min + angle_max) / 2.0f;
            if (j.angle < halfway) { limit_error = angle_min - j.angle; lo = -kLarge; hi = 0; }
            else { limit_error = angle_max - j.angle; lo = 0; hi = kLarge; }
            o.error = limit_error / dt;
            o.restitution = limit_restitution;
            axial_row(hinge_axis, 5, lo, hi, o);
            if (bump_stop_stiffness > 0 && bump_stop_angle > 0) {
                float defl = 0;
                const float bmin = angle_min + bump_stop_angle, bmax = angle_max - bump_stop_angle;
                if (j.angle < bmin) defl = j.angle - bmin;
                else if (j.angle > bmax) defl = j.angle - bmax;
                if (defl != 0) {
                    const float imp = bump_stop_stiffness * defl * dt;
                    RowOptions ob; ob.error = -defl / dt;
                    axial_row(hinge_axis, 6, std::min(imp, 0.0f), std::max(0.0f, imp), ob);
                }
            }
        }
        if (has_spring) {   // :144-158
            const float defl = j.angle - rest_angle;
            const float imp = stiffness * defl * dt;
            RowOptions o; o.error = -defl / dt;
            axial_row(hinge_axis, 7, std::min(imp, 0.0f), std::max(0.0f, imp), o);
        }
        if (has_torque) {   // :160-178
            float ti = torque * dt;
            if (damping > 0) {
                const float relvel = dot(A.angvel, hinge_axis) - dot(B.angvel, hinge_axis);
                ti += std::fabs(relvel) * damping * dt;
            }
            RowOptions o; o.error = -speed;
            axial_row(hinge_axis, 8, -ti, ti, o);
        }
        return n;
    }
    void reset_joint_angle(Joint &j) {   // hinge_constraint::reset_angle, hinge_constraint.cpp:19-24
        const Body &A = bodies[j.body[0]], &B = bodies[j.body[1]];
        if (j.type == JOINT_CVJOINT) {   // cvjoint_constraint::reset_angle, cvjoint_constraint.cpp:12-23
            j.angle = cvjoint_relative_angle(j, A.orn, B.orn, rotate(A.orn, j.frame[0].column(0)), rotate(B.orn, j.frame[1].column(0)));
            return;
        }
        const vec3 p = rotate(A.orn, j.frame[0].column(1)), q = rotate(A.orn, j.frame[0].column(2));
        const vec3 angle_axisB = rotate(B.orn, j.frame[1].column(1));
        j.angle = atan2_cr(dot(angle_axisB, q), dot(angle_axisB, p));
    }

    // position_solver.hpp:16-51. Transforms of non-procedural bodies are left untouched (the reference
    // re-normalises a static body's quaternion here, a no-op for unit quaternions up to 1 ulp).
    struct PosSolver {
        Body *A, *B;
        float inv_mA, inv_mB;
        mat3 inv_IA, inv_IB;
        float max_error = 0;
        void bind(Body &a, Body &b) {
            A = &a; B = &b;
            inv_mA = a.procedural() ? a.mass_inv : 0; inv_IA = a.procedural() ? a.I_inv_world : kMat3Zero;
            inv_mB = b.procedural() ? b.mass_inv : 0; inv_IB = b.procedural() ? b.I_inv_world : kMat3Zero;
        }
        void solve(const vec3 J[4], float error) {
            float em = effective_mass(J, inv_mA, inv_IA, inv_mB, inv_IB);
            float corr = error * kContactPositionCorrectionRate * em;
            if (A->procedural()) apply(*A, inv_mA, inv_IA, J[0], J[1], corr);
            if (B->procedural()) apply(*B, inv_mB, inv_IB, J[2], J[3], corr);
            A->update_origin(); B->update_origin();   // position_solver.hpp:34-41: origins follow the corrected transforms
            max_error = std::max(std::fabs(error), max_error);
        }
        static void apply(Body &b, float inv_m, mat3 &inv_I, vec3 Jl, vec3 Ja, float corr) {
            b.pos += inv_m * Jl * corr;
            vec3 ang = inv_I * Ja * corr;
            b.orn = b.orn + quaternion_derivative(b.orn, ang);
            b.orn = normalize(b.orn);
            mat3 basis = to_mat3(b.orn);
            inv_I = basis * b.I_inv * transpose(basis);
            b.I_inv_world = inv_I;
        }
    };
    void contact_solve_position(Manifold &m, ContactPoint &cp, PosSolver &ps) {   // contact_constraint.cpp:58-90
        if (cp.extras() && cp.stiffness < kLarge) return;   // soft contacts take no position correction (contact_extras_constraint.cpp:81-86)
        Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        ps.bind(A, B);
        vec3 pAw = to_world(cp.pivotA, A.org(), A.orn), pBw = to_world(cp.pivotB, B.org(), B.orn);
        if (cp.attachment == NA_ON_A) cp.normal = rotate(A.orn, cp.local_normal);
        else if (cp.attachment == NA_ON_B) cp.normal = rotate(B.orn, cp.local_normal);
        cp.distance = dot(pAw - pBw, cp.normal);
        vec3 rA = pAw - A.pos, rB = pBw - B.pos;
        if (cp.distance > -kEps) return;
        float error = -cp.distance;
        vec3 J[4] = {cp.normal, cross(rA, cp.normal), -cp.normal, -cross(rB, cp.normal)};
        ps.solve(J, error);
    }
    // The coloured order's own arithmetic for the position correction of a contact point (see "fused rows" above): the correction of
    // contact_constraint.cpp:58-90 / position_solver.hpp:16-51 with
    //   R = the rotation matrix of the unit orientation, built without to_mat3's renormalising division, used for the pivot, the normal
    //       and the inertia product I_w Ja = R (I_l (R^T Ja)); every dot product an fma chain;
    //   effective mass 1 / ((lin_A + ang_A) + (lin_B + ang_B)); the orientation re-normalised with ONE division, q * (1 / |q|).
    // The bodies' world inertia is not read; it is rebuilt (reference arithmetic) after each correction for the joints' position solve.
    static mat3 basis_unit(quat q) {
        const float xs = q.x + q.x, ys = q.y + q.y, zs = q.z + q.z;
        const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
        const float xx = q.x * xs, yy = q.y * ys, zz = q.z * zs;
        return {{{1.0f - (yy + zz), std::fmaf(q.x, ys, -wz), std::fmaf(q.x, zs, wy)},
                 {std::fmaf(q.x, ys, wz), 1.0f - (xx + zz), std::fmaf(q.y, zs, -wx)},
                 {std::fmaf(q.x, zs, -wy), std::fmaf(q.y, zs, wx), 1.0f - (xx + yy)}}};
    }
    static vec3 mv_fma(const mat3 &m, vec3 v) { return {dot3_fma(m.row[0], v), dot3_fma(m.row[1], v), dot3_fma(m.row[2], v)}; }
    static vec3 mtv_fma(const mat3 &m, vec3 v) {
        return {std::fmaf(m.row[2].x, v.z, std::fmaf(m.row[1].x, v.y, m.row[0].x * v.x)), std::fmaf(m.row[2].y, v.z, std::fmaf(m.row[1].y, v.y, m.row[0].y * v.x)),
                std::fmaf(m.row[2].z, v.z, std::fmaf(m.row[1].z, v.y, m.row[0].z * v.x))};
    }
    static vec3 cross_fma(vec3 a, vec3 b) { return {std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x))}; }
    // The unit of the coloured order's position solve is the MANIFOLD ("block correction", round 4): the corrections of its <= 4 points
    // are all evaluated from the transforms the manifold was entered with and applied together - one translation and one orientation
    // update (one re-normalisation) per body and manifold instead of one per point. Between two points of one manifold the reference's
    // Gauss-Seidel lets the second see the first's correction (contact_constraint.cpp:58-90 called point after point); with the 0.2
    // correction rate that coupling is a second-order term (< 0.2^2 of the penetration), well inside SURVEY 8(d)'s lock-step bound, and
    // it is what makes a position task a short dependency chain on the device: a point costs one rotation of pivot and normal and the
    // inertia product; the quaternion update, its square root and division are paid once per manifold. Specification (the device's
    // pos_manifold_block computes exactly this, bit for bit):
    //   RA, RB = basis_unit of the entry orientations; per point, in list order: world pivots, normal, distance, Jacobian, I_w Ja,
    //   effective mass and correction exactly as in the per-point form above; each body's translation is the sum, in list order, of the
    //   ROUNDED products (inv_m Jl) corr_i, its rotation vector the sum of the rounded products (I_w Ja_i) corr_i; then
    //   pos += sum, q = orn + quaternion_derivative(orn, rotation sum), q * (1 / |q|). A manifold without a penetrating point changes nothing.
    void contact_solve_position_block(Manifold &m, PosSolver &ps) {
        Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        const mat3 RA = basis_unit(A.orn), RB = basis_unit(B.orn);
        const float imA = A.procedural() ? A.mass_inv : 0.0f, imB = B.procedural() ? B.mass_inv : 0.0f;
        const mat3 IlA = A.procedural() ? A.I_inv : kMat3Zero, IlB = B.procedural() ? B.I_inv : kMat3Zero;
        vec3 tA{0, 0, 0}, rotA{0, 0, 0}, tB{0, 0, 0}, rotB{0, 0, 0};
        bool any = false;
        for (int i = 0; i < m.num_points; ++i) {
            ContactPoint &cp = m.pt[i];
            if (cp.extras() && cp.stiffness < kLarge) continue;   // soft contacts take no position correction
            const vec3 pAw = mv_fma(RA, cp.pivotA) + A.org(), pBw = mv_fma(RB, cp.pivotB) + B.org();
            if (cp.attachment == NA_ON_A) cp.normal = mv_fma(RA, cp.local_normal);
            else if (cp.attachment == NA_ON_B) cp.normal = mv_fma(RB, cp.local_normal);
            const vec3 n = cp.normal;
            cp.distance = dot3_fma(pAw - pBw, n);
            if (cp.distance > -kEps) continue;
            const vec3 rA = pAw - A.pos, rB = pBw - B.pos;
            const vec3 JlA = n, JaA = cross_fma(rA, n), JlB = -n, JaB = -cross_fma(rB, n);
            const vec3 wA = mv_fma(RA, mv_fma(IlA, mtv_fma(RA, JaA))), wB = mv_fma(RB, mv_fma(IlB, mtv_fma(RB, JaB)));
            const float sA = dot3_fma(JlA, JlA) * imA + dot3_fma(wA, JaA), sB = dot3_fma(JlB, JlB) * imB + dot3_fma(wB, JaB);
            const float em = 1.0f / (sA + sB);
            const float corr = (-cp.distance * 0.2f) * em;
            ps.max_error = std::max(std::fabs(cp.distance), ps.max_error);
            tA = tA + (imA * JlA) * corr; rotA = rotA + wA * corr;
            tB = tB + (imB * JlB) * corr; rotB = rotB + wB * corr;
            any = true;
        }
        if (!any) return;
        auto apply = [&](Body &b, vec3 t, vec3 rot) {
            if (!b.procedural()) return;
            b.pos = b.pos + t;
            const quat q = b.orn + quaternion_derivative(b.orn, rot);
            const float l2 = std::fmaf(q.w, q.w, std::fmaf(q.z, q.z, std::fmaf(q.y, q.y, q.x * q.x)));
            const float rl = 1.0f / std::sqrt(l2);
            b.orn = quat{q.x * rl, q.y * rl, q.z * rl, q.w * rl};
            const mat3 basis = to_mat3(b.orn);
            b.I_inv_world = basis * b.I_inv * transpose(basis);   // update_inertia (reference arithmetic): what the joints' position solve reads
            b.update_origin();
        };
        apply(A, tA, rotA);
        apply(B, tB, rotB);
    }
    void generic_solve_position(Joint &j, PosSolver &ps) {   // generic_constraint.cpp:260-290: the limited linear degrees of freedom
        Body &A = bodies[j.body[0]], &B = bodies[j.body[1]];
        ps.bind(A, B);
        for (int i = 0; i < 3; ++i) {
            const float *P = j.params + 10 * i;
            if (P[0] == 0) continue;
            const vec3 pA = to_world(j.pivot[0], A.org(), A.orn), pB = to_world(j.pivot[1], B.org(), B.orn);
            const vec3 pivot_offset = pB - pA, rA = pA - A.pos, rB = pB - B.pos;
            const vec3 axisA = rotate(A.orn, j.frame[0].column(i));
            const float proj = dot(pivot_offset, axisA);
            float error = 0;
            if (proj < P[1]) error = proj - P[1];
            else if (proj > P[2]) error = proj - P[2];
            vec3 J[4] = {axisA, cross(rA, axisA), -axisA, -cross(rB, axisA)};
            ps.solve(J, error);
        }
    }
    void cvjoint_solve_position(Joint &j, PosSolver &ps) {   // cvjoint_constraint.cpp:224-265
        Body &A = bodies[j.body[0]], &B = bodies[j.body[1]];
        ps.bind(A, B);
        const vec3 tA = rotate(A.orn, j.frame[0].column(0)), tB = rotate(B.orn, j.frame[1].column(0));
        const float angle = cvjoint_relative_angle(j, A.orn, B.orn, tA, tB);
        const float twist_min = j.params[0], twist_max = j.params[1];
        float twist_error = 0;
        if (twist_min < twist_max) {
            track_angle(j, angle);
            if (j.angle < twist_min) twist_error = j.angle - twist_min;
            else if (j.angle > twist_max) twist_error = j.angle - twist_max;
        } else {
            twist_error = angle;
        }
        { vec3 J[4] = {{0, 0, 0}, tA, {0, 0, 0}, -tB}; ps.solve(J, twist_error); }
        vec3 pA = to_world(j.pivot[0], A.org(), A.orn), pB = to_world(j.pivot[1], B.org(), B.orn);
        vec3 dir = pA - pB;
        const float err = length(dir);
        if (err > kEps) {
            dir /= err;
            vec3 rA = pA - A.pos, rB = pB - B.pos;
            vec3 J[4] = {dir, cross(rA, dir), -dir, -cross(rB, dir)};
            ps.solve(J, -err);
        }
    }
    void hinge_solve_position(Joint &j, PosSolver &ps) {   // hinge_constraint.cpp:180-213
        Body &A = bodies[j.body[0]], &B = bodies[j.body[1]];
        ps.bind(A, B);
        vec3 axisA = rotate(A.orn, j.frame[0].column(0)), axisB = rotate(B.orn, j.frame[1].column(0));
        vec3 p, q;
        plane_space(axisA, p, q);
        vec3 u = cross(axisA, axisB);
        float e = dot(u, p);
        if (std::fabs(e) > kEps) { vec3 J[4] = {{0, 0, 0}, p, {0, 0, 0}, -p}; ps.solve(J, e); }
        e = dot(u, q);
        if (std::fabs(e) > kEps) { vec3 J[4] = {{0, 0, 0}, q, {0, 0, 0}, -q}; ps.solve(J, e); }
        vec3 pA = to_world(j.pivot[0], A.org(), A.orn), pB = to_world(j.pivot[1], B.org(), B.orn);
        vec3 dir = pA - pB;
        float err = length(dir);
        if (err > kEps) {
            dir /= err;
            vec3 rA = pA - A.pos, rB = pB - B.pos;
            vec3 J[4] = {dir, cross(rA, dir), -dir, -cross(rB, dir)};
            ps.solve(J, -err);
        }
    }

    void integrate_body(Body &b) {   // island_solver.cpp:357-376
        b.linvel += b.dv; b.angvel += b.dw;
        b.pos += b.linvel * dt;
        b.orn = integrate(b.orn, b.angvel, dt);
        b.dv = {0, 0, 0}; b.dw = {0, 0, 0};
    }

    void refresh_derived() {   // update_aabbs (dynamic + kinematic, update_aabbs.cpp:53-78), update_inertias (dynamic, update_inertias.cpp:12-24)
        for (auto &b : bodies) {
            if (b.asleep) continue;   // update_origins / update_aabbs / update_inertias views exclude sleeping entities
            b.update_origin();        // solver.cpp:453: before the AABBs (the position solve left the origins of untouched bodies where the integration found them)
            if (b.sh.type != SHAPE_NONE && b.kind != KIND_STATIC) b.box = shape_aabb(b.sh, b.org(), b.orn);
            if (b.kind == KIND_DYNAMIC) {
                mat3 basis = to_mat3(b.orn);
                b.I_inv_world = basis * b.I_inv * transpose(basis);
            }
        }
    }

    // ---------------- restitution solver (restitution_solver.cpp:31-408) ----------------
    // Shock propagation before the constraint solver: per island and iteration, find the manifold that closes fastest; if it
    // closes faster than 0.005 m/s, walk the island breadth-first from the faster of its two bodies and, at every procedural
    // body, solve the manifolds around it that are still closing (rows with the contact's restitution, impulses from zero,
    // `individual_restitution_iters` Gauss-Seidel sweeps) and apply the velocity changes at once. The reference walks its
    // entity graph in adjacency-list order (an artefact of insertion history); here every choice is canonical: manifolds in
    // ascending pair-key order, ties to the lower key, neighbours in that same order - the order the device pass uses too.
    float manifold_min_relvel(const Manifold &m) const {   // :31-81
        const Body &A = bodies[m.body[0]], &B = bodies[m.body[1]];
        const vec3 vA = A.kind == KIND_STATIC ? vec3{0, 0, 0} : A.linvel, wA = A.kind == KIND_STATIC ? vec3{0, 0, 0} : A.angvel;
        const vec3 vB = B.kind == KIND_STATIC ? vec3{0, 0, 0} : B.linvel, wB = B.kind == KIND_STATIC ? vec3{0, 0, 0} : B.angvel;
        float mn = kScalarMax;
        for (int i = 0; i < m.num_points; ++i) {
            const ContactPoint &cp = m.pt[i];
            const vec3 pA = to_world(cp.pivotA, A.org(), A.orn), pB = to_world(cp.pivotB, B.org(), B.orn);
            const vec3 rA = pA - A.pos, rB = pB - B.pos;
            const vec3 velA = vA + cross(wA, rA), velB = vB + cross(wB, rB);
            mn = std::min(dot(velA - velB, cp.normal), mn);
        }
        return mn;
    }
    void restitution_solve_star(const std::vector<Manifold *> &ms) {   // solve_manifolds, :149-314
        std::vector<Row> rows;
        std::vector<FrictionRow> fric;
        std::vector<ContactPoint *> cps;
        for (Manifold *m : ms) {
            BodyRef A = body_ref(m->body[0]), B = body_ref(m->body[1]);
            for (int i = 0; i < m->num_points; ++i) {
                ContactPoint &cp = m->pt[i];
                const vec3 pA = to_world(cp.pivotA, A.org(), A.orn), pB = to_world(cp.pivotB, B.org(), B.orn);
                const vec3 rA = pA - A.pos, rB = pB - B.pos, n = cp.normal;
                Row r;
                r.J[0] = n; r.J[1] = cross(rA, n); r.J[2] = -n; r.J[3] = -cross(rB, n);
                r.lower = 0; r.upper = kLarge; r.impulse = 0;
                RowOptions o; o.restitution = cp.restitution;
                finish_row(r, o, A, B);
                FrictionRow f;
                f.mu = cp.friction; f.normal_row = (uint32_t)rows.size();
                vec3 t[2];
                plane_space(n, t[0], t[1]);
                for (int k = 0; k < 2; ++k) {
                    f.row[k].J[0] = t[k]; f.row[k].J[1] = cross(rA, t[k]); f.row[k].J[2] = -t[k]; f.row[k].J[3] = -cross(rB, t[k]);
                    f.row[k].eff_mass = effective_mass(f.row[k].J, A.inv_m, A.inv_I, B.inv_m, B.inv_I);
                    f.row[k].rhs = -relative_speed(f.row[k].J, A.linvel, A.angvel, B.linvel, B.angvel);
                    f.row[k].impulse = 0;
                }
                rows.push_back(r); fric.push_back(f); cps.push_back(&cp);
            }
        }
        for (int it = 0; it < individual_restitution_iters; ++it)
            for (size_t k = 0; k < rows.size(); ++k) {
                const float d = solve_row(rows[k]);
                apply_row_impulse(d, rows[k]);
                solve_friction(fric[k], rows[fric[k].normal_row]);
            }
        for (size_t k = 0; k < rows.size(); ++k) {
            cps[k]->normal_restitution_impulse = rows[k].impulse;
            cps[k]->friction_restitution_impulse[0] = fric[k].row[0].impulse;
            cps[k]->friction_restitution_impulse[1] = fric[k].row[1].impulse;
        }
        for (Manifold *m : ms)
            for (uint32_t bi : m->body) {
                Body &b = bodies[bi];
                if (b.kind == KIND_STATIC) continue;
                b.linvel += b.dv; b.angvel += b.dw;
                b.dv = {0, 0, 0}; b.dw = {0, 0, 0};
            }
        dummy_dv_ = {0, 0, 0}; dummy_dw_ = {0, 0, 0};
    }
    void solve_restitution() {   // :388-408
        if (restitution_iters <= 0) return;
        bool any = false;
        for (auto &kv : manifolds) if (kv.second.with_restitution) { any = true; break; }
        if (!any) return;
        const uint32_t n = (uint32_t)bodies.size();
        // canonical adjacency: every body's manifolds in ascending key order (std::map order)
        std::vector<std::vector<Manifold *>> adj(n);
        std::map<uint32_t, std::vector<Manifold *>> by_island;
        auto label_of = [&](uint32_t a, uint32_t b) { return bodies[a].procedural() ? island_label[a] : island_label[b]; };
        const bool ext = order == ORDER_EXTERNAL && ext_walk_valid;
        ext_walk_valid = false;
        if (ext) {   // the island edge-list order of the real engine (the fastest-manifold search keeps the first minimum)
            for (auto &ab : ext_rest_manifolds) {
                auto it = manifolds.find(pair_key(ab[0], ab[1]));
                if (it == manifolds.end()) { ext_order_mismatch = true; continue; }
                Manifold &m = it->second;
                if (manifold_asleep(m)) continue;
                by_island[label_of(m.body[0], m.body[1])].push_back(&m);
            }
        }
        for (auto &kv : manifolds) {
            Manifold &m = kv.second;
            if (manifold_asleep(m)) continue;   // island_view excludes sleeping islands
            adj[m.body[0]].push_back(&m); adj[m.body[1]].push_back(&m);
            if (!ext) by_island[label_of(m.body[0], m.body[1])].push_back(&m);
        }
        const float threshold = -0.005f;
        std::vector<uint8_t> island_done(n, 0);
        for (int it = 0; it < restitution_iters; ++it) {
            bool all_solved = true;
            for (auto &isl : by_island) {
                // (the reference re-examines every island in every iteration, solved or not: an island that was quiet can be
                // hit again later - islands do not interact, so skipping is equivalent only within one iteration)
                float min_relvel = kScalarMax;
                Manifold *fastest = nullptr;
                for (Manifold *m : isl.second) {
                    if (!m->with_restitution) continue;
                    const float r = manifold_min_relvel(*m);
                    if (r < min_relvel) { min_relvel = r; fastest = m; }
                }
                if (!fastest || min_relvel > threshold) continue;   // solved
                all_solved = false;
                const Body &FA = bodies[fastest->body[0]], &FB = bodies[fastest->body[1]];
                const float sA = FA.kind == KIND_STATIC ? 0.0f : length_sqr(FA.linvel), sB = FB.kind == KIND_STATIC ? 0.0f : length_sqr(FB.linvel);
                uint32_t start;
                if (sA > sB) start = FA.procedural() ? fastest->body[0] : fastest->body[1];
                else start = FB.procedural() ? fastest->body[1] : fastest->body[0];
                if (ext) {   // entity_graph::traverse (entity_graph.hpp:356-422) over the real engine's adjacency order
                    std::vector<uint8_t> seen(n, 0);
                    std::vector<uint32_t> to_visit{start};
                    while (!to_visit.empty()) {
                        const uint32_t node = to_visit.back();
                        to_visit.pop_back();
                        seen[node] = 1;
                        if (!bodies[node].procedural()) continue;   // non-connecting: neither solved from nor walked through
                        auto ait = ext_adj.find(node);
                        if (ait == ext_adj.end()) continue;
                        std::vector<Manifold *> star;
                        for (const ExtAdj &e : ait->second) {   // graph.visit_edges(node): adjacency by adjacency, edge by edge
                            if (e.ma == 0xFFFFFFFFu) continue;
                            auto it = manifolds.find(pair_key(e.ma, e.mb));
                            if (it == manifolds.end()) { ext_order_mismatch = true; continue; }
                            if (manifold_min_relvel(it->second) < threshold) star.push_back(&it->second);
                        }
                        if (!star.empty()) restitution_solve_star(star);
                        for (const ExtAdj &e : ait->second)   // neighbours in adjacency order, each put at the FRONT (breadth first)
                            if (!seen[e.other]) { to_visit.insert(to_visit.begin(), e.other); seen[e.other] = 1; }
                    }
                    continue;
                }
                std::vector<uint8_t> visited(n, 0);
                std::vector<uint32_t> queue{start};
                visited[start] = 1;
                for (size_t qi = 0; qi < queue.size(); ++qi) {   // breadth-first (entity_graph.hpp:356-422)
                    const uint32_t node = queue[qi];
                    std::vector<Manifold *> star;
                    for (Manifold *m : adj[node]) if (manifold_min_relvel(*m) < threshold) star.push_back(m);
                    if (!star.empty()) restitution_solve_star(star);
                    for (Manifold *m : adj[node]) {
                        const uint32_t o = m->body[0] == node ? m->body[1] : m->body[0];
                        if (!visited[o] && bodies[o].procedural()) { visited[o] = 1; queue.push_back(o); }
                    }
                }
            }
            if (all_solved) break;
        }
    }

    void solve() {
        dummy_dv_ = {0, 0, 0}; dummy_dw_ = {0, 0, 0};
        solve_restitution();
        for (auto &b : bodies)   // apply_gravity.hpp:12-17
            if (b.kind == KIND_DYNAMIC && !b.asleep && b.gravity != vec3{0, 0, 0}) b.linvel += b.gravity * dt;
        if (order == ORDER_COLOURED) solve_coloured(); else solve_sequential();
        refresh_derived();
        uint32_t np = 0;
        for (auto &kv : manifolds) np += kv.second.num_points;
        stats.num_manifolds = (uint32_t)manifolds.size();
        stats.num_points = np;
        ++step_index;
    }

    // Reference order (island_solver.cpp:513-543 per island).
    void solve_sequential() {
        std::map<uint32_t, std::vector<Manifold *>> isl_m;
        std::map<uint32_t, std::vector<Joint *>> isl_j;
        std::map<uint32_t, std::vector<uint32_t>> isl_b;
        auto label_of = [&](uint32_t a, uint32_t b) { return bodies[a].procedural() ? island_label[a] : island_label[b]; };
        for (uint32_t i = 0; i < bodies.size(); ++i) if (bodies[i].procedural() && !bodies[i].asleep) isl_b[island_label[i]].push_back(i);   // solver.cpp:408 excludes sleeping islands
        for (auto &kv : manifolds) isl_m[label_of(kv.second.body[0], kv.second.body[1])].push_back(&kv.second);
        for (auto &j : joints) if (j.alive) isl_j[label_of(j.body[0], j.body[1])].push_back(&j);
        stats.num_rows = 0;
        for (auto &ib : isl_b) {
            const uint32_t label = ib.first;
            std::vector<Row> rows;
            std::vector<FrictionRow> fric;
            struct JSpan { Joint *j; int first, n; int slot[kMaxJointRows]; };
            std::vector<JSpan> jrows;
            std::vector<std::pair<ContactPoint *, uint32_t>> crows;   // point, normal row index
            auto &js = isl_j[label];
            // (manifold, slot) pairs in visiting order: list order of the canonical manifold sequence, or the caller's order
            std::vector<std::pair<Manifold *, int>> cps;
            if (order == ORDER_EXTERNAL) {
                std::vector<Joint *> ordered;
                for (uint32_t ji : ext_joint_order) {
                    Joint *j = &joints[ji];
                    if (label_of(j->body[0], j->body[1]) == label) ordered.push_back(j);
                }
                if (ordered.size() != js.size()) ext_order_mismatch = true;
                js = ordered;
                for (const ExtContact &e : ext_contact_order) {
                    if (label_of(e.a, e.b) != label) continue;
                    auto it = manifolds.find(pair_key(e.a, e.b));
                    if (it == manifolds.end() || (int)e.slot >= it->second.num_points) { ext_order_mismatch = true; continue; }
                    cps.push_back({&it->second, (int)e.slot});
                }
                size_t expect = 0;
                for (Manifold *m : isl_m[label]) expect += (size_t)m->num_points;
                if (expect != cps.size()) ext_order_mismatch = true;
            } else {
                // constraints_tuple order: every contact_constraint, then every contact_extras_constraint
                for (int pass = 0; pass < 2; ++pass)
                    for (Manifold *m : isl_m[label]) for (int i = 0; i < m->num_points; ++i) if ((int)m->pt[i].extras() == pass) cps.push_back({m, i});
            }
            std::vector<FrictionRow> roll;
            std::vector<SpinRow> spin;
            std::vector<ContactPoint *> roll_cp, spin_cp;
            for (int type : {JOINT_GRAVITY, JOINT_DISTANCE, JOINT_SOFT_DISTANCE, JOINT_HINGE, JOINT_GENERIC, JOINT_CVJOINT, JOINT_CONE, JOINT_POINT})   // constraints_tuple order (constraint.hpp:23-34)
                for (Joint *j : js) {
                    if (j->type != type) continue;
                    Row tmp[kMaxJointRows];
                    JSpan sp; sp.j = j; sp.first = (int)rows.size();
                    sp.n = prepare_joint(*j, body_ref(j->body[0]), body_ref(j->body[1]), tmp, sp.slot);
                    jrows.push_back(sp);
                    for (int i = 0; i < sp.n; ++i) rows.push_back(tmp[i]);
                }
            for (auto &cp : cps) {
                Manifold *m = cp.first;
                BodyRef A = body_ref(m->body[0]), B = body_ref(m->body[1]);
                Row nr; FrictionRow fr; ExtraRows ex;
                prepare_contact(m->pt[cp.second], A, B, nr, fr, &ex, m->num_points);
                fr.normal_row = (uint32_t)rows.size();
                crows.push_back({&m->pt[cp.second], fr.normal_row});
                if (ex.roll) { ex.rr.normal_row = fr.normal_row; roll.push_back(ex.rr); roll_cp.push_back(&m->pt[cp.second]); }
                if (ex.spin) { ex.sr.normal_row = fr.normal_row; spin.push_back(ex.sr); spin_cp.push_back(&m->pt[cp.second]); }
                rows.push_back(nr);
                fric.push_back(fr);
            }
            stats.num_rows += (uint32_t)rows.size();
            for (auto &r : rows) apply_row_impulse(r.impulse, r);                 // warm start
            for (auto &f : fric) warm_start_friction(f, rows[f.normal_row]);
            for (auto &f : roll) warm_start_friction(f, rows[f.normal_row]);
            for (auto &r : spin) warm_start_spin(r, rows[r.normal_row]);
            for (int it = 0; it < vel_iters; ++it) {   // island_solver.cpp:94-111
                for (auto &r : rows) { float d = solve_row(r); apply_row_impulse(d, r); }
                for (auto &f : fric) solve_friction(f, rows[f.normal_row]);
                for (auto &f : roll) solve_friction(f, rows[f.normal_row]);
                for (auto &r : spin) solve_spin_friction(r, rows[r.normal_row]);
            }
            for (uint32_t b : ib.second) integrate_body(bodies[b]);
            for (auto &jr : jrows)                                                 // assign_applied_impulses
                for (int i = 0; i < jr.n; ++i) jr.j->impulse[jr.slot[i]] = rows[jr.first + i].impulse;
            for (size_t k = 0; k < crows.size(); ++k) {
                crows[k].first->normal_impulse = rows[crows[k].second].impulse;
                crows[k].first->friction_impulse[0] = fric[k].row[0].impulse;
                crows[k].first->friction_impulse[1] = fric[k].row[1].impulse;
            }
            for (size_t k = 0; k < roll.size(); ++k) { roll_cp[k]->rolling_impulse[0] = roll[k].row[0].impulse; roll_cp[k]->rolling_impulse[1] = roll[k].row[1].impulse; }
            for (size_t k = 0; k < spin.size(); ++k) spin_cp[k]->spin_impulse = spin[k].impulse;
            for (int it = 0; it < pos_iters; ++it) {
                // The per-type position passes are the ARGUMENTS of one function call (max_variadic(solve_each<C>(...)...),
                // island_solver.cpp:324-333): their order is unspecified in C++, and the reference as built here (g++, x86-64)
                // evaluates them right to left - the constraint types run in REVERSE tuple order: contact_extras, contact,
                // cvjoint, generic, hinge. Pinned to that build (tests/test_reference_engine.py mixes the types).
                PosSolver hs, cs;
                for (int pass = 1; pass >= 0; --pass)
                    for (auto &cp : cps) if ((int)cp.first->pt[cp.second].extras() == pass) contact_solve_position(*cp.first, cp.first->pt[cp.second], cs);
                for (Joint *j : js) if (j->type == JOINT_CVJOINT) cvjoint_solve_position(*j, hs);
                for (Joint *j : js) if (j->type == JOINT_GENERIC) generic_solve_position(*j, hs);
                for (Joint *j : js) if (j->type == JOINT_HINGE) hinge_solve_position(*j, hs);
                if (std::max(hs.max_error, cs.max_error) < 0.005f) break;
            }
        }
    }

    // GPU order. Rows live per constraint; deltas live on the bodies, exactly as above.
    void solve_coloured() {
        colour_joints();
        colour_contacts();
        struct CRows { Manifold *m; Row nr[kMaxContacts]; FrictionRow fr[kMaxContacts]; ExtraRows ex[kMaxContacts]; };
        struct JRows { Joint *j; int n; Row r[kMaxJointRows]; int slot[kMaxJointRows]; };
        std::vector<std::vector<CRows>> cc(stats.num_colours);
        std::vector<std::vector<JRows>> jc(stats.num_joint_colours);
        stats.num_rows = 0;
        for (auto &j : joints) {
            if (!j.alive || joint_asleep(j)) continue;
            JRows jr; jr.j = &j;
            jr.n = prepare_joint(j, body_ref(j.body[0]), body_ref(j.body[1]), jr.r, jr.slot);
            stats.num_rows += jr.n;
            jc[j.colour].push_back(jr);
        }
        for (auto &kv : manifolds) {
            Manifold &m = kv.second;
            if (m.num_points == 0 || manifold_asleep(m)) continue;
            CRows cr; cr.m = &m;
            BodyRef A = body_ref(m.body[0]), B = body_ref(m.body[1]);
            for (int i = 0; i < m.num_points; ++i) prepare_contact(m.pt[i], A, B, cr.nr[i], cr.fr[i], &cr.ex[i], m.num_points);
            stats.num_rows += m.num_points;
            cc[m.colour].push_back(cr);
        }
        for (auto &col : jc) for (auto &jr : col) for (int i = 0; i < jr.n; ++i) apply_row_impulse(jr.r[i].impulse, jr.r[i]);
        const bool fused = (g_arith & ARITH_FUSED_VELOCITY) != 0, block = (g_arith & ARITH_BLOCK_POSITION) != 0, two_phase = (g_arith & ARITH_TWO_PHASE) != 0;
        if (two_phase) {   // warm_start(row_cache&) and solve(row_cache&) of the reference, the rows of each kind in colour order
            for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) apply_row_impulse(cr.nr[i].impulse, cr.nr[i]);
            for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) warm_start_friction(cr.fr[i], cr.nr[i]);
            for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].roll) warm_start_friction(cr.ex[i].rr, cr.nr[i]);
            for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].spin) warm_start_spin(cr.ex[i].sr, cr.nr[i]);
        } else
        for (auto &col : cc) for (auto &cr : col) {
            if (fused) {
                for (int i = 0; i < cr.m->num_points; ++i) apply_impulse_fused(cr.nr[i].impulse, cr.nr[i].J, cr.nr[i]);
                for (int i = 0; i < cr.m->num_points; ++i) for (int t = 0; t < 2; ++t) apply_impulse_fused(cr.fr[i].row[t].impulse, cr.fr[i].row[t].J, cr.nr[i]);
            } else {
                for (int i = 0; i < cr.m->num_points; ++i) apply_row_impulse(cr.nr[i].impulse, cr.nr[i]);
                for (int i = 0; i < cr.m->num_points; ++i) warm_start_friction(cr.fr[i], cr.nr[i]);
            }
            for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].roll) warm_start_friction(cr.ex[i].rr, cr.nr[i]);
            for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].spin) warm_start_spin(cr.ex[i].sr, cr.nr[i]);
        }
        for (int it = 0; it < vel_iters; ++it) {
            for (auto &col : jc) for (auto &jr : col) for (int i = 0; i < jr.n; ++i) { float d = solve_row(jr.r[i]); apply_row_impulse(d, jr.r[i]); }
            if (two_phase) {
                for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) { float d = solve_row(cr.nr[i]); apply_row_impulse(d, cr.nr[i]); }
                for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) solve_friction(cr.fr[i], cr.nr[i]);
                for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].roll) solve_friction(cr.ex[i].rr, cr.nr[i]);
                for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].spin) solve_spin_friction(cr.ex[i].sr, cr.nr[i]);
                continue;
            }
            for (auto &col : cc) for (auto &cr : col) {
                if (fused) {
                    for (int i = 0; i < cr.m->num_points; ++i) solve_normal_fused(cr.nr[i]);
                    for (int i = 0; i < cr.m->num_points; ++i) solve_friction_fused(cr.fr[i], cr.nr[i]);
                } else {
                    for (int i = 0; i < cr.m->num_points; ++i) { float d = solve_row(cr.nr[i]); apply_row_impulse(d, cr.nr[i]); }
                    for (int i = 0; i < cr.m->num_points; ++i) solve_friction(cr.fr[i], cr.nr[i]);
                }
                // inside a manifold the reference's order of row kinds: normals, friction, rolling, spinning
                for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].roll) solve_friction(cr.ex[i].rr, cr.nr[i]);
                for (int i = 0; i < cr.m->num_points; ++i) if (cr.ex[i].spin) solve_spin_friction(cr.ex[i].sr, cr.nr[i]);
            }
        }
        for (auto &b : bodies) if (b.kind == KIND_DYNAMIC && !b.asleep) integrate_body(b);
        for (auto &col : jc) for (auto &jr : col) for (int i = 0; i < jr.n; ++i) jr.j->impulse[jr.slot[i]] = jr.r[i].impulse;
        for (auto &col : cc) for (auto &cr : col) for (int i = 0; i < cr.m->num_points; ++i) {
            cr.m->pt[i].normal_impulse = cr.nr[i].impulse;
            cr.m->pt[i].friction_impulse[0] = cr.fr[i].row[0].impulse;
            cr.m->pt[i].friction_impulse[1] = cr.fr[i].row[1].impulse;
            if (cr.ex[i].roll) { cr.m->pt[i].rolling_impulse[0] = cr.ex[i].rr.row[0].impulse; cr.m->pt[i].rolling_impulse[1] = cr.ex[i].rr.row[1].impulse; }
            if (cr.ex[i].spin) cr.m->pt[i].spin_impulse = cr.ex[i].sr.impulse;
        }
        // Position iterations: per-island early-out, islands keyed by label.
        std::vector<uint8_t> done(bodies.size(), 0);
        std::vector<float> err(bodies.size(), 0.0f);
        auto label_of = [&](uint32_t a, uint32_t b) { return bodies[a].procedural() ? island_label[a] : island_label[b]; };
        for (int it = 0; it < pos_iters; ++it) {
            std::fill(err.begin(), err.end(), 0.0f);
            for (auto &col : jc) for (auto &jr : col) {
                if (jr.j->type != JOINT_HINGE && jr.j->type != JOINT_CVJOINT && jr.j->type != JOINT_GENERIC) continue;
                uint32_t l = label_of(jr.j->body[0], jr.j->body[1]);
                if (done[l]) continue;
                PosSolver ps;
                if (jr.j->type == JOINT_HINGE) hinge_solve_position(*jr.j, ps);
                else if (jr.j->type == JOINT_GENERIC) generic_solve_position(*jr.j, ps);
                else cvjoint_solve_position(*jr.j, ps);
                err[l] = std::max(err[l], ps.max_error);
            }
            for (auto &col : cc) for (auto &cr : col) {
                uint32_t l = label_of(cr.m->body[0], cr.m->body[1]);
                if (done[l]) continue;
                PosSolver ps;
                if (block) contact_solve_position_block(*cr.m, ps);
                else for (int i = 0; i < cr.m->num_points; ++i) contact_solve_position(*cr.m, cr.m->pt[i], ps);
                err[l] = std::max(err[l], ps.max_error);
            }
            for (size_t l = 0; l < done.size(); ++l) if (err[l] < 0.005f) done[l] = 1;
        }
    }

    bool colour_overflow() const { return colour_overflow_; }
    bool ext_order_mismatch = false;   // ORDER_EXTERNAL: the supplied order did not cover exactly this step's constraints

private:
    DynTree tree_, np_tree_;
    vec3 dummy_dv_{0, 0, 0}, dummy_dw_{0, 0, 0};
    bool joints_coloured_ = false;
    double pending_stamp_ = 0;
    bool colour_overflow_ = false;
};

}  // namespace orc
