/* edynhip.h — C-ABI of the MI355X-native stepper that replaces Edyn's per-step simulation loop.
 *
 * The reference (xissburg/edyn v1.3.1) has no FFI layer: its seam is the C++ API over a caller-owned
 * entt::registry. This C-ABI sits UNDER a header-only C++ shim (include/edyn/edyn.hpp in this repo)
 * that keeps edyn::attach / edyn::update / edyn::make_rigidbody, so each entry point below cites the
 * reference interface it stands in for:
 *
 *   edynhip_create / edynhip_destroy   <- edyn::attach / edyn::detach            include/edyn/edyn.hpp:66-77, src/edyn/edyn.cpp:73-197
 *                                         + settings{fixed_dt, iterations, gravity} include/edyn/context/settings.hpp:21-57
 *   edynhip_set_bodies                 <- edyn::make_rigidbody(registry, def)     include/edyn/util/rigidbody.hpp:29-93, src/edyn/util/rigidbody.cpp:47-191
 *   edynhip_set_joints                 <- edyn::make_constraint<point|hinge>      include/edyn/util/constraint_util.hpp:38-54
 *   edynhip_step                       <- edyn::step_simulation / one fixed step of edyn::update
 *                                                                                 src/edyn/simulation/stepper_sequential.cpp:71-102,121-147
 *   edynhip_get_state / set_state      <- reading/patching position, orientation, linvel, angvel pools
 *                                                                                 include/edyn/comp/{position,orientation,linvel,angvel}.hpp
 *   edynhip_get_manifolds / set_       <- contact_manifold + contact_point* components
 *                                                                                 include/edyn/collision/contact_manifold.hpp:14-22, contact_point.hpp:17-58
 *   edynhip_get_timings / get_stats    <- profile_timers / profile_counters       include/edyn/context/profile.hpp:8-27
 *
 * Conventions: plain pointers and sizes, host arrays are borrowed for the duration of the call only,
 * every function returns 0 on success or a negative edynhip_status; nothing throws across the boundary.
 * A context is bound to one GPU and is single-threaded, like the reference's sequential stepper.
 * There is NO CPU fallback: if no gfx950 device is usable, edynhip_create fails with EDYNHIP_ERR_NO_DEVICE.
 */
#ifndef EDYNHIP_H
#define EDYNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct edynhip_ctx edynhip_ctx;

typedef enum {
    EDYNHIP_OK = 0,
    EDYNHIP_ERR_INVALID = -1,       /* bad argument */
    EDYNHIP_ERR_NO_DEVICE = -2,     /* no usable HIP device */
    EDYNHIP_ERR_HIP = -3,           /* a HIP runtime call failed (see edynhip_last_error) */
    EDYNHIP_ERR_CAPACITY = -4,      /* pair / manifold capacity exceeded */
    EDYNHIP_ERR_COLOURS = -5,       /* a body has more joints than joint colours (64); contacts have no such limit */
    EDYNHIP_ERR_UNSUPPORTED = -6,   /* feature outside the hot-path scope (e.g. a cylinder / polyhedron / mesh shape) */
    EDYNHIP_ERR_INTERNAL = -7       /* a device-side invariant failed (e.g. the dataflow solve timed out waiting for a hand-off) */
} edynhip_status;

/* rigidbody_kind (include/edyn/util/rigidbody.hpp:22-27) */
enum { EDYNHIP_KIND_DYNAMIC = 0, EDYNHIP_KIND_KINEMATIC = 1, EDYNHIP_KIND_STATIC = 2 };
/* shapes on the hot path; NONE = amorphous body */
enum { EDYNHIP_SHAPE_NONE = 0, EDYNHIP_SHAPE_BOX = 1, EDYNHIP_SHAPE_SPHERE = 2, EDYNHIP_SHAPE_PLANE = 3,
       EDYNHIP_SHAPE_CAPSULE = 4, /* shape_param = radius, half_length, axis (0 x, 1 y, 2 z): shapes/capsule_shape.hpp:17-30 */
       EDYNHIP_SHAPE_CYLINDER = 5, /* shape_param = radius, half_length, axis: shapes/cylinder_shape.hpp:22-25 */
       EDYNHIP_SHAPE_POLYHEDRON = 6 /* shape_param[0] = id of a mesh made with edynhip_create_convex_mesh: shapes/polyhedron_shape.hpp:11-43 */ };
enum { EDYNHIP_JOINT_POINT = 0, EDYNHIP_JOINT_HINGE = 1,
       EDYNHIP_JOINT_DISTANCE = 2,       /* distance_constraint.cpp:7-31; params[0] = distance; impulse slot 0 */
       EDYNHIP_JOINT_SOFT_DISTANCE = 3,  /* soft_distance_constraint.cpp:8-62; params = distance, stiffness, damping; slots 0 spring, 1 damping */
       EDYNHIP_JOINT_CONE = 4,           /* cone_constraint.cpp:12-104; frames + params through edynhip_set_joint_definition */
       EDYNHIP_JOINT_CVJOINT = 5,        /* cvjoint_constraint.cpp:12-302; frames + params through edynhip_set_joint_definition */
       EDYNHIP_JOINT_GRAVITY = 6,        /* gravity_constraint.cpp:6-34: Newtonian attraction between the two bodies; impulse slot 0 */
       EDYNHIP_JOINT_GENERIC = 7,        /* generic_constraint.cpp:10-330; frames + 6 x 10 parameters through edynhip_set_generic_definition */
       EDYNHIP_JOINT_NULL = 8            /* null_constraint.hpp:12-17: no rows - an edge of the island graph that keeps two bodies in one island */ };
/* contact_normal_attachment (include/edyn/collision/contact_normal_attachment.hpp:17-21) */
enum { EDYNHIP_ATTACH_NONE = 0, EDYNHIP_ATTACH_ON_A = 1, EDYNHIP_ATTACH_ON_B = 2 };

typedef struct {
    int32_t device;                  /* HIP device ordinal */
    uint32_t max_bodies;
    uint32_t max_manifolds;          /* 0 = 16 * max_bodies + 1024 */
    uint32_t max_joints;
    float fixed_dt;                  /* settings.fixed_dt, default 1/60 */
    uint32_t num_velocity_iterations;/* settings.num_solver_velocity_iterations, default 8 */
    uint32_t num_position_iterations;/* settings.num_solver_position_iterations, default 3 */
    float gravity[3];                /* settings.gravity, default (0,-9.8,0) */
    uint32_t flags;                  /* EDYNHIP_FLAG_* */
} edynhip_config;

enum {
    EDYNHIP_FLAG_TIMING = 1u,        /* record per-stage HIP events (edynhip_get_timings); each event costs ~6 us of idle GPU */
    EDYNHIP_FLAG_TIMING_SOLVE = 16u, /* record only the two events around the velocity solve (solve_velocity_ms) */
    EDYNHIP_FLAG_SLEEPING = 4u,      /* island sleeping / waking (island_manager.cpp:524-623); off = every body sleeping_disabled */
    EDYNHIP_FLAG_CONTACT_EVENTS = 32u, /* record manifold / contact point creation and destruction (edynhip_get_contact_events) */
    /* Contact arithmetic of the solve. Default (neither flag): every contact row and every contact position correction is evaluated with the
       reference's operations in the reference's order (constraint_row.cpp:24-57, constraint_row_friction.cpp:11-54, contact_constraint.cpp:58-90,
       position_solver.hpp:16-51) - the stepper then differs from the reference in the Gauss-Seidel VISITING order only. The two flags opt in
       to faster forms of the same equations; each is a stated deviation (tests/test_arithmetic_fork.py):
         FUSED_VELOCITY_ROWS  fused multiply-adds, one division per friction-circle clamp (an fp-level change: ~1e-7 m/s per step);
         BLOCK_POSITION       the <= 4 points of a manifold corrected as one block from the transforms the manifold was entered with
                              (an algorithmic change: up to ~3e-4 m per step; SURVEY 8(d)(3) is NOT met in this mode). */
    EDYNHIP_FLAG_FUSED_VELOCITY_ROWS = 64u,
    EDYNHIP_FLAG_BLOCK_POSITION = 128u,
    EDYNHIP_FLAG_EXCLUSIVE_DEVICE = 8u /* promise: nothing else launches work on this device while a step runs (one stepper per
                                        GPU). The resident-grid solver kernels are then launched plainly instead of
                                        cooperatively (~0.1 ms per step less idle GPU). Without the promise the default,
                                        cooperative launches, is always safe. */
};

/* Scene description, one entry per body, index = body id. Arrays are packed row-major
 * (pos[n][3], orn[n][4] xyzw, ...), i.e. the memory layout of the corresponding EnTT pools. */
typedef struct {
    const int32_t *kind;             /* EDYNHIP_KIND_* */
    const float *pos;                /* [n][3] */
    const float *orn;                /* [n][4] */
    const float *linvel;             /* [n][3] */
    const float *angvel;             /* [n][3] */
    const float *mass;               /* [n]   (dynamic only) */
    const float *inertia;            /* [n][9] local inertia tensor, used where has_inertia[i] != 0; may be NULL */
    const uint8_t *has_inertia;      /* [n] or NULL: 0 = derive from mass and shape (moment_of_inertia.cpp) */
    const int32_t *shape_type;       /* EDYNHIP_SHAPE_* */
    const float *shape_param;        /* [n][4]: box half_extents xyz | sphere radius | plane normal xyz + constant */
    const float *friction;           /* [n] material.friction */
    const float *restitution;        /* [n] material.restitution; any value > 0 turns the restitution solver on
                                        (src/edyn/dynamics/restitution_solver.cpp:86-408) */
    const uint64_t *group;           /* [n] collision_filter.group (all ones = default) */
    const uint64_t *mask;            /* [n] collision_filter.mask */
    const float *gravity;            /* [n][3] per-body gravity or NULL = config gravity */
    const uint8_t *sleeping_disabled;/* [n] or NULL: sleeping_disabled_tag (only meaningful with EDYNHIP_FLAG_SLEEPING) */
    const float *center_of_mass;     /* [n][3] or NULL (ABI 8): rigidbody_def::center_of_mass in the shape's frame; `pos` is then the ORIGIN - the
                                        stepper moves position / linear velocity to the centre of mass and shifts a shape-derived inertia
                                        (util/rigidbody.cpp:56-87,517-548); edynhip_get_state returns the centre of mass, as the reference's position */
} edynhip_bodies;

typedef struct {
    const int32_t *type;             /* EDYNHIP_JOINT_* */
    const uint32_t *body;            /* [n][2] */
    const float *pivot;              /* [n][2][3] object-space pivots */
    const float *axis;               /* [n][2][3] hinge axes in object space (ignored for point joints) */
    const float *params;             /* [n][10] or NULL (all zero). hinge_constraint (hinge_constraint.hpp:30-62): angle_min, angle_max,
                                        limit_restitution, bump_stop_angle, bump_stop_stiffness, torque, speed, rest_angle, stiffness,
                                        damping - limits are active when angle_min < angle_max, the spring when stiffness > 0, the
                                        torque row when torque > 0 or damping > 0. point_constraint (point_constraint.hpp:25):
                                        [0] = friction_torque. */
} edynhip_joints;

/* settings that may change on a running world (include/edyn/context/settings.hpp:22-30) */
typedef struct {
    float fixed_dt;
    uint32_t num_velocity_iterations;
    uint32_t num_position_iterations;
    float gravity[3];
    uint32_t num_restitution_iterations;             /* default 8; 0 turns the restitution solver off */
    uint32_t num_individual_restitution_iterations;  /* default 3 */
} edynhip_params;

/* One contact point; mirrors contact_point + contact_point_geometry + _material + _impulse. */
typedef struct {
    float pivotA[3], pivotB[3], normal[3], local_normal[3];
    float distance, friction, restitution;
    int32_t attachment;
    uint32_t lifetime;
    float normal_impulse, friction_impulse[2];
} edynhip_point;

/* One contact manifold (body pair) with its <= 4 points in the reference's list order (newest first). */
typedef struct {
    uint32_t body[2];
    uint32_t num_points;
    uint32_t colour;                 /* solver colour of this pair (0xFF = none) */
    edynhip_point pt[4];
} edynhip_manifold;

/* Stage selector for edynhip_run_stages: the reference's stage names (profile.hpp:8-18). */
enum {
    EDYNHIP_STAGE_BROADPHASE = 1u,
    EDYNHIP_STAGE_NARROWPHASE = 2u,
    EDYNHIP_STAGE_ISLANDS = 4u,
    EDYNHIP_STAGE_SOLVE = 8u,
    EDYNHIP_STAGE_ALL = 15u
};

typedef struct {
    /* milliseconds, accumulated over the steps of the last edynhip_step call (EDYNHIP_FLAG_TIMING) */
    float broadphase_ms, narrowphase_ms, islands_ms, colouring_ms, prepare_ms, solve_velocity_ms,
          integrate_ms, solve_position_ms, finish_ms, step_ms;
    uint32_t solve_velocity_launches;  /* kernel launches inside solve_velocity_ms */
    uint32_t steps;
} edynhip_timings;

typedef struct {
    uint32_t num_bodies, num_manifolds, num_points, num_active_manifolds, num_islands, num_colours,
             num_joint_colours, colour_rounds, num_joints, num_joint_rows;   /* colour_rounds: rounds the last step's contact colouring took (the one-workgroup rounds + the multi-block ones) */
    uint32_t solve_schedule;         /* EDYNHIP_SCHEDULE_*: which velocity-solve kernels the last step launched */
    uint32_t colour_size[64];        /* manifolds per solver colour in the last step */
} edynhip_stats;

/* Velocity-solve schedules (edynhip_stats::solve_schedule). The results are bit-identical across them. */
enum {
    EDYNHIP_SCHEDULE_NONE = 0,          /* nothing to solve */
    EDYNHIP_SCHEDULE_DATAFLOW2 = 1,     /* k_contact_solve_df2: one launch per step, two lanes per manifold */
    EDYNHIP_SCHEDULE_DATAFLOW1 = 2,     /* k_contact_solve_df: one launch per step, one lane per manifold */
    EDYNHIP_SCHEDULE_ISLAND_FUSED = 3,  /* k_island_velocity: one wave per island (scenes with joints / contact_extras rows) */
    EDYNHIP_SCHEDULE_MIXED = 4,         /* dataflow launch for the islands without joints + k_island_velocity for the others */
    EDYNHIP_SCHEDULE_PER_COLOUR = 5,    /* k_contact_solve / k_joint_solve: one launch per colour and sweep */
    EDYNHIP_SCHEDULE_DATAFLOW4 = 6      /* k_contact_solve_df4: one launch per step, four lanes per manifold */
};

edynhip_ctx *edynhip_create(const edynhip_config *cfg, int *status_out);
void edynhip_destroy(edynhip_ctx *ctx);
const char *edynhip_last_error(const edynhip_ctx *ctx);   /* ctx may be NULL: error of the last failed create */

/* Optional: run on a caller-owned hipStream_t (e.g. torch's current stream). NULL = the ctx's own stream. */
int edynhip_set_stream(edynhip_ctx *ctx, void *hip_stream);

int edynhip_set_bodies(edynhip_ctx *ctx, uint32_t n, const edynhip_bodies *bodies);
int edynhip_set_joints(edynhip_ctx *ctx, uint32_t n, const edynhip_joints *joints);
/* Append `n` bodies after the existing ones (indices num_bodies .. num_bodies+n-1). Existing bodies, their contact
 * manifolds (cached impulses, colours) and joints are untouched: this is registry.create + make_rigidbody on a running
 * world (src/edyn/util/rigidbody.cpp:18-161; the reference's island worker receives the new entities through
 * registry_operation insertions, src/edyn/simulation/simulation_worker.cpp). To remove bodies: edynhip_remove_bodies, which keeps
 * every other index by leaving a tombstone. */
int edynhip_add_bodies(edynhip_ctx *ctx, uint32_t n, const edynhip_bodies *bodies);

/* Joints on a running world: make_constraint / registry.destroy(constraint entity) (include/edyn/util/constraint_util.hpp:38-54,
 * island_manager.cpp:68-97). Joint indices are the order of creation and stay valid for the life of the world (a removed joint
 * keeps its index); applied impulses and hinge angles of the other joints are carried over. *first_index = index of the first
 * joint added. edynhip_set_joint_params = registry.patch<hinge|point_constraint> followed by reset_angle. */
int edynhip_add_joints(edynhip_ctx *ctx, uint32_t n, const edynhip_joints *joints, uint32_t *first_index);
int edynhip_remove_joints(edynhip_ctx *ctx, uint32_t n, const uint32_t *joint_indices);
int edynhip_set_joint_params(edynhip_ctx *ctx, uint32_t joint_index, const float *params10);
/* registry.destroy(rigid body) on a running world (src/edyn/edyn.cpp:148-197 hooks, island_manager.cpp:47-115): the body's
 * manifolds and contact points disappear with the next step, joints attached to it are removed, the islands it touched wake
 * up. The body INDEX stays reserved (the slot reads back as a shapeless static body); every other index, manifold, warm-start
 * impulse and sleep timer is untouched. */
int edynhip_remove_bodies(edynhip_ctx *ctx, uint32_t n, const uint32_t *body_indices);
/* settings on a running world WITHOUT losing state: set_fixed_dt, set_solver_velocity/position_iterations, set_gravity
 * (src/edyn/edyn.cpp:203-207, src/edyn/config/solver_iteration_config.cpp:9-75, src/edyn/util/gravity_util.cpp:12-20 - like the
 * reference, set_gravity also replaces the gravity of every dynamic body). */
int edynhip_get_params(edynhip_ctx *ctx, edynhip_params *out);
int edynhip_set_params(edynhip_ctx *ctx, const edynhip_params *params);

/* Advance `nsteps` fixed-dt steps. Returns after the work is enqueued and error flags were checked. */
int edynhip_step(edynhip_ctx *ctx, uint32_t nsteps);
/* The same with explicit step time stamps: step i runs at first_step_time + i * step_dt. The stamps only feed the island
 * sleep timers; integration always uses fixed_dt. This is stepper_sequential::update's behaviour when more steps are due
 * than max_steps_per_update allows: the steps that do run get stretched stamps (stepper_sequential.cpp:60-75). */
int edynhip_step_timed(edynhip_ctx *ctx, uint32_t nsteps, double first_step_time, double step_dt);
/* Run a subset of one step's stages (parity tests). */
int edynhip_run_stages(edynhip_ctx *ctx, uint32_t stage_mask);
/* Block until all enqueued work of this ctx has finished. */
int edynhip_synchronize(edynhip_ctx *ctx);

int edynhip_get_state(edynhip_ctx *ctx, float *pos, float *orn, float *linvel, float *angvel);
int edynhip_set_state(edynhip_ctx *ctx, const float *pos, const float *orn, const float *linvel, const float *angvel);
/* collision_exclusion lists (include/edyn/util/exclude_collision.hpp:20-47, src/edyn/util/exclude_collision.cpp:9-71;
 * evaluated by should_collide_default, src/edyn/collision/should_collide.cpp:11-57): no NEW manifold is created for an
 * excluded pair (symmetric; at most 16 partners per body, EDYNHIP_ERR_CAPACITY beyond). A manifold that already exists
 * lives on until its AABBs separate, as in the reference. */
int edynhip_exclude_collision(edynhip_ctx *ctx, uint32_t body_a, uint32_t body_b);
int edynhip_remove_collision_exclusion(edynhip_ctx *ctx, uint32_t body_a, uint32_t body_b);
/* settings.should_collide_func / edyn::set_should_collide (include/edyn/collision/should_collide.hpp:8-18, include/edyn/context/settings.hpp:43):
 * the user's predicate that REPLACES should_collide_default - the reference calls it for every candidate of a body's tree query
 * (src/edyn/collision/broadphase.cpp:143-153) and creates a manifold only when it says yes. A host callback cannot run inside the
 * device broadphase, so a context with a filter takes a slow path in every step that has candidates for NEW manifolds: the device
 * broadphase runs WITHOUT its group / mask / exclusion test, the new candidate pairs (AABBs already found overlapping, no manifold yet)
 * travel to the host, `filter(user, body, other)` is asked once per pair - body = the querying (procedural) body, body[0] of the
 * manifold that would be made - the rejected pairs are taken out of the step's pair list, and the step goes on. A rejected pair is
 * asked about again in every step in which it is still a candidate, an existing manifold lives on until its AABBs separate - both as
 * in the reference. The predicate must be a function of the pair (the reference asks more often - also for pairs that have a
 * manifold or whose boxes then do not overlap - and in another order). edynhip_default_should_collide evaluates what the device would
 * have (collision groups / masks, exclusion lists) for callbacks that extend the default. filter = NULL restores the device test.
 * Costs one device-to-host copy and a stream synchronisation per step that has new candidates. Worlds over several devices:
 * edynhip_world_set_pair_filter (global body indices). */
typedef int (*edynhip_pair_filter)(void *user, uint32_t body, uint32_t other);
int edynhip_set_pair_filter(edynhip_ctx *ctx, edynhip_pair_filter filter, void *user);
int edynhip_default_should_collide(edynhip_ctx *ctx, uint32_t body_a, uint32_t body_b);   /* 1 / 0, negative = error */
/* Recompute every awake body's AABB and world-space inverse inertia from its current transform: the reference's public
 * update_aabbs(registry) / update_inertias(registry) (include/edyn/sys/update_aabbs.hpp:18-25, update_inertias.hpp:18-28).
 * A step does this at its end (solver.cpp:456-465); after edynhip_set_state the derived state is stale until then -
 * exactly as in the reference, where the broadphase of the next step still sees the old AABB - unless this is called. */
int edynhip_refresh_derived(edynhip_ctx *ctx);
/* Pack (pos3, orn4, linvel3, angvel3) = 13 floats per body into DEVICE memory `dst` (for RCCL gathers). */
int edynhip_pack_state_device(edynhip_ctx *ctx, void *dst_device, uint32_t first_body, uint32_t count);
/* Derived per-body state: aabb[n][6] (min,max), inertia_world_inv[n][9], island label[n]; any may be NULL. */
int edynhip_get_derived(edynhip_ctx *ctx, float *aabb, float *inertia_world_inv, uint32_t *island);

/* Manifolds are kept (and returned) in ascending canonical order: key = (owner << 32) | other, where the owner is the
 * pair's procedural (dynamic) body - the one with the higher index when both are dynamic. edynhip_set_manifolds expects
 * records in that order. (EnTT's pool order is not reproducible; the solver visits manifolds in this order.) */
int edynhip_num_manifolds(edynhip_ctx *ctx, uint32_t *n);
int edynhip_get_manifolds(edynhip_ctx *ctx, edynhip_manifold *out, uint32_t capacity, uint32_t *n);
int edynhip_set_manifolds(edynhip_ctx *ctx, const edynhip_manifold *in, uint32_t n);
/* Canonical broadphase pairs: keys[i] = (max(body)<<32 | min(body)), ascending. */
int edynhip_get_pairs(edynhip_ctx *ctx, uint64_t *keys, uint32_t capacity, uint32_t *n);
/* Per joint (by caller index, removed joints read 0) 10 floats: the applied impulses by slot - hinge: linear[3], hinge[2], limit,
 * bump_stop, spring, torque; point: applied[3], friction - and the tracked hinge angle (hinge_constraint.hpp:62-71). */
int edynhip_get_joint_impulses(edynhip_ctx *ctx, float *impulses10);

/* Frames (row-major 3x3, first COLUMN = the cone direction / the twist axis) and the parameter block of a cone or cvjoint
 * constraint (created with identity frames by edynhip_set_joints / add_joints); resets the cvjoint's twist angle.
 *   cone   : span_tan[0], span_tan[1], restitution, bump_stop_stiffness, bump_stop_length (cone_constraint.hpp:19-49);
 *            impulse slots: 0 limit, 1 bump stop. frame_b is ignored.
 *   cvjoint: twist_min, twist_max, twist_restitution, twist_bump_stop_angle, twist_bump_stop_stiffness, twist_friction_torque,
 *            twist_rest_angle, twist_stiffness, twist_damping, rest_direction[3], bend_stiffness, bend_friction_torque,
 *            bend_damping (cvjoint_constraint.hpp:20-102); impulse slots: 0..2 linear, 3 twist limit, 4 twist bump stop,
 *            5 twist spring, 6 twist friction / damping, 7 bend friction / damping, 8 bend spring; [9] of
 *            edynhip_get_joint_impulses = the tracked twist angle. */
int edynhip_set_joint_definition(edynhip_ctx *ctx, uint32_t joint, const float *frame_a9, const float *frame_b9, const float *params16);

/* generic_constraint: frames (row-major 3x3) and, per degree of freedom d (0..2 translation along frame_a's columns, 3..5 rotation:
 * twist about x, then the two bending angles), 10 floats dof[10 d + ..]: limit_enabled, min, max, limit_restitution,
 * bump_stop_length (angle), bump_stop_stiffness, friction force (torque), rest offset (angle), spring_stiffness, damping
 * (generic_constraint.hpp:23-70). Created by edynhip_set_joints / add_joints with identity frames and every degree of freedom
 * free. edynhip_get_joint_slot_impulses: out[n][24], slot 4 d + {0 limit, 1 bump stop, 2 spring, 3 friction / damping} for
 * generic constraints, the first 9 slots as documented above for the other types. */
int edynhip_set_generic_definition(edynhip_ctx *ctx, uint32_t joint, const float *frame_a9, const float *frame_b9, const float *dof60);
int edynhip_get_joint_slot_impulses(edynhip_ctx *ctx, float *impulses24);

/* contact_extras materials: material::{spin_friction, roll_friction, stiffness, damping} of bodies [first, first + n)
 * (comp/material.hpp:15-22; any array may be NULL = the default 0, 0, large_scalar, large_scalar). Contact points created
 * from then on mix them (material_mixing.hpp:20-34) and, where a mixed value is not the default, get the rolling-friction
 * pair, the spinning-friction row and / or the force-limited ("soft") normal row of contact_extras_constraint
 * (contact_extras_constraint.cpp:12-110, constraint_row_spin_friction.cpp:5-36); soft contacts take no position
 * correction. A world with such materials solves on the per-colour schedule. roll_direction components are not modelled.
 * edynhip_get_point_extras: out[4 * i + k][7] = rolling impulse[2], spin impulse, mixed roll / spin coefficient, stiffness,
 * damping of point k of manifold i in edynhip_get_manifolds order. */
int edynhip_set_material_extras(edynhip_ctx *ctx, uint32_t first, uint32_t n, const float *spin_friction, const float *roll_friction,
                                const float *stiffness, const float *damping);
int edynhip_get_point_extras(edynhip_ctx *ctx, float *out7, uint32_t capacity_manifolds, uint32_t *n);

/* Material ids and the material mix table (material::id, comp/material.hpp:27-31; material_mix_table, dynamics/material_mixing.hpp:36-82;
 * edyn::insert_material_mixing, util/insert_material_mixing.hpp:17): an entry for a pair of ids replaces every mixing rule for
 * contact points created from then on - material6 = restitution, friction, spin_friction, roll_friction, stiffness, damping.
 * ids: 16 bits, 0xFFFF = unassigned. Faithful to the reference down to its lookup: the table is keyed by the ids of
 * (manifold body[0], body[1]) in a std::map under unordered_pair's comparator, so an entry inserted as (a, b) is not always found
 * for a pair that meets as (b, a) - insert both orders if in doubt (core/unordered_pair.hpp:32-40). */
int edynhip_set_material_ids(edynhip_ctx *ctx, uint32_t first, uint32_t n, const uint32_t *ids);
int edynhip_insert_material_mixing(edynhip_ctx *ctx, uint32_t id0, uint32_t id1, const float *material6);

/* Contact events (EDYNHIP_FLAG_CONTACT_EVENTS): what an application observes in the reference through
 * registry.on_construct / on_destroy<contact_manifold> (make_contact_manifold, constraint_util.cpp:60-102;
 * broadphase::destroy_separated_manifolds, broadphase.cpp:99-134) and <contact_point> (create_contact_point,
 * collision_util.cpp:311-388; destroy_contact_point, :390-430; contact_started_tag, narrowphase.cpp:111-130).
 * The events of all steps of the last edynhip_step / edynhip_step_timed call, in no particular order within a step.
 * A point keeps its id from creation to destruction: id = (step of creation + 1) << 32 | manifold index << 2 | slot
 * (points injected with edynhip_set_manifolds: high word 0). If more events occurred than the context can hold
 * (5 x max_manifolds), `n` is capped and EDYNHIP_ERR_CAPACITY is returned: resynchronise from edynhip_get_manifolds. */
enum { EDYNHIP_EVENT_MANIFOLD_CREATED = 1, EDYNHIP_EVENT_MANIFOLD_DESTROYED = 2, EDYNHIP_EVENT_POINT_CREATED = 3, EDYNHIP_EVENT_POINT_DESTROYED = 4 };
typedef struct {
    uint32_t type;       /* EDYNHIP_EVENT_* */
    uint32_t step;       /* step index (steps since edynhip_create / edynhip_set_bodies) in which it happened */
    uint32_t body[2];    /* the manifold's bodies, as in edynhip_manifold.body */
    uint64_t point_id;   /* point events; 0 for manifold events */
} edynhip_contact_event;
int edynhip_get_contact_events(edynhip_ctx *ctx, edynhip_contact_event *out, uint32_t capacity, uint32_t *n);
/* ids[4 * i + k] = id of point k of manifold i in edynhip_get_manifolds order (0 where there is no point). */
int edynhip_get_point_ids(edynhip_ctx *ctx, uint64_t *ids, uint32_t capacity_manifolds, uint32_t *n);

/* Double-buffered state read-back - the analogue of the asynchronous stepper handing finished steps to the main thread
 * (simulation_worker.cpp:406-444): edynhip_snapshot enqueues, behind the steps issued so far, a copy of the packed state
 * (13 floats / body) into one of two pinned host buffers and returns at once; edynhip_snapshot_read waits for THAT copy
 * only (not for steps enqueued after it) and unpacks it. step -> snapshot -> step -> snapshot_read hands the host the
 * first step's result while the second one runs. `step_index` (may be NULL) = steps completed when the snapshot was taken. */
int edynhip_snapshot(edynhip_ctx *ctx);
int edynhip_snapshot_read(edynhip_ctx *ctx, float *pos, float *orn, float *linvel, float *angvel, uint32_t *step_index);

/* The registry write-back, read in place (ABI 15). The reference has no write-back - the registry IS its storage
 * (stepper_sequential.cpp:28-119) - so what a registry-facing caller pays per update beyond the step is this copy. One packed
 * 96-byte record per body, written by the device into pinned host memory the context owns (two slots, alternating): no
 * per-call staging arrays, no second scatter, and everything a per-body host loop would otherwise compute or fetch separately:
 *   pos orn linvel angvel   the simulated state (what solver::update leaves in the components, solver.cpp:331-339)
 *   present_pos/orn         update_presentation.cpp:56-84 evaluated at `present_dt` from that state: pos + linvel * dt and
 *                           integrate(orn, angvel, dt) (math/quaternion.cpp:7-22) with the stepper's own integrate()
 *   origin                  update_origins.cpp:13-15 (meaningful when EDYNHIP_RECORD_HAS_ORIGIN is set)
 *   flags                   EDYNHIP_RECORD_*: dynamic body / sleeping_tag (island_manager.cpp:541-565) / has a centre-of-mass offset / removed
 * and the contact events of the last step call (edynhip_get_contact_events' list), the first `max_events` of them.
 * edynhip_snapshot_records enqueues, behind the steps issued so far, the pack and the copy (on a side stream) and returns at once;
 * edynhip_snapshot_map waits for THAT copy only and hands out pointers into the pinned slot, valid until the next-but-one
 * edynhip_snapshot_records call. step -> snapshot_records -> step -> snapshot_map hands over the first step's result while the
 * second one runs (simulation_worker.cpp:406-444). `total_events` > `num_events`: the list was cut - in a synchronous caller
 * edynhip_get_contact_events still returns all of them (until the next step call), otherwise resynchronise from the manifolds. */
enum { EDYNHIP_RECORD_DYNAMIC = 1u, EDYNHIP_RECORD_ASLEEP = 2u, EDYNHIP_RECORD_HAS_ORIGIN = 4u, EDYNHIP_RECORD_REMOVED = 8u };
typedef struct {
    float pos[3], orn[4], linvel[3], angvel[3];
    float present_pos[3], present_orn[4];
    float origin[3];
    uint32_t flags;
} edynhip_body_record;
typedef struct {
    const edynhip_body_record *records;    /* [num_bodies], index = body id */
    uint32_t num_bodies;
    uint32_t step_index;                   /* steps completed when the snapshot was taken */
    const edynhip_contact_event *events;   /* [num_events]; NULL when the context records no events */
    uint32_t num_events, total_events;
} edynhip_record_view;
/* flags: EDYNHIP_SNAPSHOT_DIRECT - the caller is going to wait for THIS snapshot right away (a synchronous write-back): the pack kernels store
 * straight into the pinned slot on the stepper's stream instead of handing a device buffer to the copy engine on the side stream (no
 * second stream, no event between the two: ~0.04 ms sooner on the headline pile); without it the copy overlaps whatever is enqueued next. */
enum { EDYNHIP_SNAPSHOT_DIRECT = 1u };
int edynhip_snapshot_records(edynhip_ctx *ctx, float present_dt, uint32_t max_events, uint32_t flags);
int edynhip_snapshot_map(edynhip_ctx *ctx, edynhip_record_view *view);
/* Contact-event prefetch (ABI 15). In the reference contact points become registry entities INSIDE the step, during the narrowphase
 * (create_contact_point collision_util.cpp:311-388, destroy_contact_point :390-430, narrowphase.cpp:21-40) - before the solver moves
 * anything. Every event of a step is emitted by its broadphase and narrowphase, so with the prefetch enabled (max_events > 0) a step
 * call copies its event list (the count and the first max_events events) to pinned memory as soon as the LAST step's narrowphase has
 * run, on a side stream; edynhip_prefetched_events waits for that copy only, so the caller creates / destroys the contact entities
 * while the islands, solve and finish stages of that step are still running. `total_events` > `num_events`: the list was cut -
 * edynhip_get_contact_events returns all of it (and waits for the step). The pointers stay valid until the next step call. */
int edynhip_set_event_prefetch(edynhip_ctx *ctx, uint32_t max_events);
int edynhip_prefetched_events(edynhip_ctx *ctx, const edynhip_contact_event **events, uint32_t *num_events, uint32_t *total_events);

/* Test hook: run the device closest-feature routine on `n` independent shape pairs (no world state involved).
 * shape_type[n][2], shape_param[n][2][4], pos[n][2][3], orn[n][2][4]; out_points[n][4][11] =
 * (pivotA3, pivotB3, normal3, distance, attachment) per point, out_count[n]. Replaces nothing in the reference: it
 * exposes edyn::collide(shA, shB, ctx, result) (include/edyn/collision/collide.hpp) for parity tests. */
int edynhip_debug_collide(edynhip_ctx *ctx, uint32_t n, const int32_t *shape_type, const float *shape_param, const float *pos,
                          const float *orn, float threshold, float *out_points, uint32_t *out_count);

/* Island sleeping (EDYNHIP_FLAG_SLEEPING): asleep[n] = 1 where the body carries sleeping_tag; wake_all = wake_up_entity on
 * everything (also implied by edynhip_set_state). island_manager.cpp:541-565, util/island_util.cpp:61-66. */
int edynhip_get_asleep(edynhip_ctx *ctx, uint8_t *asleep);
int edynhip_wake_all(edynhip_ctx *ctx);
/* wake_up_entity (util/rigidbody.cpp:409-415 -> island_manager.cpp wake_up_island): wakes the islands of the listed bodies. */
int edynhip_wake_bodies(edynhip_ctx *ctx, uint32_t n, const uint32_t *indices);
/* edyn::set_center_of_mass on a running world (util/rigidbody.cpp:364-370, apply_center_of_mass :517-548): the body's position and linear
 * velocity move to the new centre of mass (given in the shape's frame; zero removes the offset), its origin - shape, contact and joint
 * pivots - stays where it is, its inertia is not changed. */
int edynhip_set_center_of_mass(edynhip_ctx *ctx, uint32_t body, const float *com3);

int edynhip_get_timings(edynhip_ctx *ctx, edynhip_timings *out);
int edynhip_get_stats(edynhip_ctx *ctx, edynhip_stats *out);
/* State a caller carries from one context into another (the C++ shim re-creates a context to grow it): the joints' applied
 * impulses - all 24 slots per joint, the layout of edynhip_get_joint_slot_impulses - and tracked angles (the 10th value of
 * edynhip_get_joint_impulses; NULL keeps the angles), by caller joint index; and the sleeping tags of the bodies (as
 * edynhip_get_asleep returns them; sleeping bodies get zero velocities, island timers of awake islands restart). */
int edynhip_set_joint_warm_start(edynhip_ctx *ctx, const float *impulses24, const float *angles);
int edynhip_set_asleep(edynhip_ctx *ctx, const uint8_t *asleep);
/* ... and the island sleep TIMERS (island::sleep_timestamp, island_manager.cpp:605-623; ABI 14): island_label[n] = every body's island (the
 * lowest body index of the island, as edynhip_get_derived returns it), since[n] = by island label, the time stamp at which the island first
 * met the sleep thresholds (negative or NaN: no timer running; entries that are not labels are ignored), clock = the stamp of the last step.
 * A context that receives them (after its bodies, manifolds and sleeping tags) continues the timers where the other context left them - its
 * first step relabels from the given labels, so an island that merges or splits in that very step follows the rules of a running world -
 * instead of starting every timer again. A no-op without EDYNHIP_FLAG_SLEEPING. */
int edynhip_get_sleep_timers(edynhip_ctx *ctx, uint32_t *island_label, double *since, double *clock);
int edynhip_set_sleep_timers(edynhip_ctx *ctx, const uint32_t *island_label, const double *since, double clock);

uint32_t edynhip_abi_version(void);

/* polyhedron_shape's convex_mesh (include/edyn/shapes/convex_mesh.hpp:17-70): vertices[num_vertices][3], the faces' vertex indices
 * (counter-clockwise seen from outside) and faces[num_faces][2] = (first index, vertex count). Does what convex_mesh::initialize does
 * (src/edyn/shapes/convex_mesh.cpp:10-30): moves the vertices so that the centroid is the origin, derives face normals, unique edges,
 * vertex adjacency and the relevant (direction-unique) faces and edges. The mesh belongs to the context and is shared by every body
 * whose shape is EDYNHIP_SHAPE_POLYHEDRON with shape_param[0] = *mesh_id (the shared_ptr<convex_mesh> of the reference); meshes are
 * created before the bodies that use them. flags: EDYNHIP_MESH_INITIALIZED = the vertices come from a convex_mesh on which
 * initialize() already ran (they are relative to the centroid): they are taken as they are and only the derived arrays are built.
 * Limits: closed mesh of positive volume, at most 32 vertices per face (support polygons are held in fixed storage on the device). */
enum { EDYNHIP_MESH_INITIALIZED = 1u };
int edynhip_create_convex_mesh(edynhip_ctx *ctx, uint32_t num_vertices, const float *vertices, uint32_t num_indices, const uint32_t *indices,
                               uint32_t num_faces, const uint32_t *faces, uint32_t flags, uint32_t *mesh_id);
/* The derived arrays of a mesh (parity tests; out == NULL returns the element count): float fields [count][3], index fields uint32. */
enum { EDYNHIP_MESH_VERTICES = 0, EDYNHIP_MESH_NORMALS = 1, EDYNHIP_MESH_RELEVANT_NORMALS = 2, EDYNHIP_MESH_EDGE_VERTICES = 3,
       EDYNHIP_MESH_EDGE_NORMALS = 4, EDYNHIP_MESH_EDGES = 5, EDYNHIP_MESH_EDGE_FACES = 6, EDYNHIP_MESH_RELEVANT_FACES = 7,
       EDYNHIP_MESH_RELEVANT_EDGES = 8, EDYNHIP_MESH_NEIGHBORS_START = 9, EDYNHIP_MESH_NEIGHBOR_INDICES = 10,
       EDYNHIP_MESH_INERTIA_SUMS = 11 /* 7 floats: the face sums of moment_of_inertia_polyhedron (volume, xx, yy, zz, yz, zx, xy) */ };
int edynhip_get_convex_mesh(edynhip_ctx *ctx, uint32_t mesh_id, int field, void *out, uint32_t capacity, uint32_t *count);

/* Measurement aid for the roofline report (bench.py): streams `bytes` of device memory with 16-byte loads from every CU (read_gbs)
 * and copies them device-to-device (copy_gbs, read + write bytes counted), best of five runs each, on the context's stream.
 * Not part of the step path. */
int edynhip_measure_bandwidth(edynhip_ctx *ctx, uint64_t bytes, float *read_gbs, float *copy_gbs);

/* ------------------------------------------------------------------------------------------------ Multi-GPU world (ABI 12)
 * ONE simulation over several GPUs of a node at island granularity - the reference's own unit of parallelism
 * (src/edyn/dynamics/solver.cpp:408-428 runs every island as its own task; islands share only non-procedural bodies, which a
 * step never writes: include/edyn/comp/island.hpp:34-41). An application reaches it through edyn::attach with
 * init_config::devices (include/edyn/edyn.hpp of this repository; the reference's attach: include/edyn/edyn.hpp:66-70).
 * Every device gets one edynhip_ctx (a "shard") that owns the dynamic bodies of its islands plus a replica of every non-dynamic
 * body; shards step concurrently (one private host thread each) with NO exchange inside a step; after each step the integrated
 * state of every shard is copied into pinned host memory and from there into the world's arrays (edynhip_world_get_state) - the
 * registry write-back. Islands of different shards that approach each other are noticed from island bounding boxes reduced ON
 * THE DEVICE (edyn_amd/csrc/multi.hip), and the world re-partitions, carrying contact manifolds (warm-start impulses, colours),
 * joint impulses / angles, sleeping tags, exclusions, joint definitions and meshes to the islands' new owners. A shard computes
 * for its islands bit for bit what one context computes for the whole world (tests/cpp/multi.cpp).
 * Body / joint indices of this interface are GLOBAL (the caller's order). `devices` may name a device more than once (several
 * shards on one GPU: functional tests). Scene description calls (meshes, bodies, joints, definitions, exclusions - in this order)
 * come before the first step; a later edynhip_world_set_bodies starts a new world. */
typedef struct edynhip_world edynhip_world;
typedef struct {
    uint32_t num_shards, num_bodies;
    uint32_t steps;                  /* steps taken since creation */
    uint32_t approach_checks;        /* island-box sweeps (device reduction + host sweep) */
    uint32_t repartitions;           /* times the shards were rebuilt because islands of different shards met (or on request) */
    uint32_t bodies_per_shard[16];   /* bodies whose state each shard reports (shard 0 also reports the replicated ones) */
    uint32_t rebalances;             /* meetings that took the full, load-balanced re-partition because the heaviest shard exceeded 1.5 x the mean (ABI 15) */
} edynhip_world_stats;
edynhip_world *edynhip_world_create(const edynhip_config *cfg /* .device is ignored; capacities 0 = sized per shard */,
                                    const int32_t *devices, uint32_t num_devices, int *status_out);
void edynhip_world_destroy(edynhip_world *w);
const char *edynhip_world_last_error(const edynhip_world *w);   /* w may be NULL: error of the last failed create */
int edynhip_world_create_convex_mesh(edynhip_world *w, uint32_t num_vertices, const float *vertices, uint32_t num_indices, const uint32_t *indices,
                                     uint32_t num_faces, const uint32_t *faces, uint32_t flags, uint32_t *mesh_id);
int edynhip_world_set_bodies(edynhip_world *w, uint32_t n, const edynhip_bodies *bodies);
/* The scene description (joints, joint definitions, exclusions) is what the shards are (re)built from, with the state of
 * edynhip_world_set_bodies: these three calls come before the first step. On a world that has been stepped they return EDYNHIP_ERR_UNSUPPORTED
 * (a rebuild would reset the simulation to its initial state); describe the scene again - edynhip_world_set_bodies with the current state -
 * to edit a running world. The initial island probe loads the whole scene onto devices[0]: a scene must fit one GPU's memory once. */
int edynhip_world_set_joints(edynhip_world *w, uint32_t n, const edynhip_joints *joints);
/* edynhip_set_joint_definition (generic = 0: params[16]) / edynhip_set_generic_definition (generic = 1: params[60]) by global joint index */
int edynhip_world_set_joint_definition(edynhip_world *w, uint32_t joint, const float *frame_a9, const float *frame_b9, const float *params, int generic);
int edynhip_world_exclude_collision(edynhip_world *w, uint32_t body_a, uint32_t body_b);
/* settings.should_collide_func on a world over several devices (ABI 14): `filter(user, body, other)` is asked with GLOBAL body indices by
 * whichever shard holds the candidate pair (see edynhip_set_pair_filter for when and how often); the shards step on their own host
 * threads, the calls are serialised. May be set before or after the shards exist; filter = NULL restores the device test.
 * edynhip_world_default_should_collide: collision groups / masks and exclusion lists of the scene, in global indices. */
int edynhip_world_set_pair_filter(edynhip_world *w, edynhip_pair_filter filter, void *user);
int edynhip_world_default_should_collide(edynhip_world *w, uint32_t body_a, uint32_t body_b);
/* edyn::step_simulation on every shard, then the gather, the approach test and - when islands of different shards have met - the
 * re-partition. Returns when the state of the last step is in the world's arrays. */
int edynhip_world_step(edynhip_world *w, uint32_t nsteps);
int edynhip_world_get_state(edynhip_world *w, float *pos, float *orn, float *linvel, float *angvel);   /* any may be NULL */
int edynhip_world_get_partition(edynhip_world *w, int32_t *shard_of_body);   /* -1 = replicated (not dynamic) */
int edynhip_world_repartition(edynhip_world *w);                             /* re-partition now (load balance / tests) */
int edynhip_world_get_manifolds(edynhip_world *w, edynhip_manifold *out, uint32_t capacity, uint32_t *n);   /* global indices, canonical order */
int edynhip_world_get_stats(edynhip_world *w, edynhip_world_stats *out);
edynhip_ctx *edynhip_world_context(edynhip_world *w, uint32_t shard);        /* a shard's context (read-only use: statistics, timings) */

/* The pieces of the above that processes owning ONE GPU each use (edyn_amd/parallel.py ShardedWorld over torch.distributed / RCCL):
 * the island partitioner - longest-processing-time-first over the summed weights, deterministic (islands by descending weight, ties
 * by ascending label, each to the lightest rank, ties to the lowest rank); rank_of[i] = -1 for non-dynamic bodies. weights may be
 * NULL (1 per body). Host only: no GPU needed. */
int edynhip_partition_islands(uint32_t n, const uint32_t *island_label, const int32_t *kind, const double *weights, uint32_t world_size, int32_t *rank_of);
/* island boxes [num][6] (min, max) grown by `margin` on every side that overlap: pairs (label_a < label_b), sorted, without
 * duplicates; any_owner = 0: only pairs of different owners. pairs may be NULL (count only). Host only. */
int edynhip_island_boxes_overlap(uint32_t num_islands, const float *boxes6, const uint32_t *labels, const int32_t *owner, float margin, int any_owner,
                                 uint32_t *pairs, uint32_t capacity, uint32_t *num_pairs);
/* per island of the context (label = its lowest body index, as edynhip_get_derived returns them) the union of the AABBs of its
 * shaped dynamic bodies, reduced on the device after the last step. labels / boxes6 may be NULL (count only). */
int edynhip_get_island_boxes(edynhip_ctx *ctx, uint32_t *labels, float *boxes6, uint32_t capacity, uint32_t *num_islands);

#ifdef __cplusplus
}
#endif
#endif /* EDYNHIP_H */
