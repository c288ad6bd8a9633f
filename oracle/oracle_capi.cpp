// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// C ABI over the CPU restatement so that tests/ (ctypes) and bench.py's cpu_baseline leg can drive it.
// The record layouts mirror include/edynhip.h so the same numpy dtypes describe both sides.
#include "oworld.hpp"
#include <chrono>
#include <thread>

using namespace orc;

extern "C" {

struct orc_point_rec {
    float pivotA[3], pivotB[3], normal[3], local_normal[3];
    float distance, friction, restitution;
    int32_t attachment;
    uint32_t lifetime;
    float normal_impulse, friction_impulse[2];
};
struct orc_manifold_rec {
    uint32_t body[2];
    uint32_t num_points;
    uint32_t colour;
    orc_point_rec pt[4];
};

static vec3 v3(const float *p) { return {p[0], p[1], p[2]}; }
static void put3(float *d, vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

void *orc_world_create(float dt, int vel_iters, int pos_iters, const float *g, int order) {
    World *w = new World();
    w->dt = dt; w->vel_iters = vel_iters; w->pos_iters = pos_iters; w->gravity = v3(g); w->order = order;
    return w;
}
void orc_world_destroy(void *h) { delete (World *)h; }

static shape make_shape(int type, const float *p) {
    shape s;
    s.type = type;
    if (type == SHAPE_BOX) s.half_extents = v3(p);
    else if (type == SHAPE_SPHERE) s.radius = p[0];
    else if (type == SHAPE_PLANE) { s.normal = v3(p); s.constant = p[3]; }
    else if (type == SHAPE_CAPSULE || type == SHAPE_CYLINDER) { s.radius = p[0]; s.half_length = p[1]; s.axis = (int)p[2]; }
    else if (type == SHAPE_POLYHEDRON) s.mesh = (int)p[0];
    return s;
}
// convex meshes live in a process-wide registry (a polyhedron_shape holds a shared_ptr<convex_mesh>); shape_param[0] = the id
int orc_create_mesh(uint32_t nv, const float *verts, uint32_t nidx, const uint32_t *indices, uint32_t nfaces, const uint32_t *faces) {
    auto m = std::make_shared<ConvexMesh>();
    for (uint32_t i = 0; i < nv; ++i) m->vertices.push_back(v3(verts + 3 * i));
    m->indices.assign(indices, indices + nidx);
    m->faces.assign(faces, faces + 2 * nfaces);
    m->initialize();
    mesh_registry().push_back(m);
    return (int)mesh_registry().size() - 1;
}
// what: 0 vertices, 1 normals, 2 relevant_normals, 3 edge_vertices, 4 edge_normals (floats x3); 5 edges, 6 edge_faces, 7 relevant_faces,
// 8 relevant_edges, 9 neighbors_start, 10 neighbor_indices (uint32). out == nullptr: returns the element count.
static uint32_t mesh_field(const ConvexMesh &m, int what, void *out) {
    const std::vector<vec3> *fv[5] = {&m.vertices, &m.normals, &m.relevant_normals, &m.edge_vertices, &m.edge_normals};
    const std::vector<uint32_t> *uv[6] = {&m.edges, &m.edge_faces, &m.relevant_faces, &m.relevant_edges, &m.neighbors_start, &m.neighbor_indices};
    if (what < 5) {
        if (out) for (size_t i = 0; i < fv[what]->size(); ++i) put3((float *)out + 3 * i, (*fv[what])[i]);
        return (uint32_t)fv[what]->size();
    }
    if (out) std::copy(uv[what - 5]->begin(), uv[what - 5]->end(), (uint32_t *)out);
    return (uint32_t)uv[what - 5]->size();
}
uint32_t orc_mesh_get(int id, int what, void *out) { return mesh_field(*mesh_registry()[id], what, out); }
void orc_mesh_inertia(int id, float mass, float *out9) {
    const mat3 I = mesh_registry()[id]->inertia(mass);
    for (int r = 0; r < 3; ++r) put3(out9 + 3 * r, I.row[r]);
}

uint32_t orc_add_body(void *h, int kind, const float *pos, const float *orn, const float *linvel, const float *angvel,
                      float mass, int shape_type, const float *shape_param, const float *inertia9, float friction,
                      float restitution, int has_material, uint64_t group, uint64_t mask, const float *grav) {
    World *w = (World *)h;
    mat3 I;
    if (inertia9) I = {{{inertia9[0], inertia9[1], inertia9[2]}, {inertia9[3], inertia9[4], inertia9[5]}, {inertia9[6], inertia9[7], inertia9[8]}}};
    vec3 g = grav ? v3(grav) : w->gravity;
    return w->add_body(kind, v3(pos), quat{orn[0], orn[1], orn[2], orn[3]}, v3(linvel), v3(angvel), mass,
                       make_shape(shape_type, shape_param), inertia9 ? &I : nullptr, friction, restitution,
                       has_material != 0, group, mask, &g);
}
void orc_set_ext_restitution_walk(void *h, const uint32_t *manifolds2, uint32_t nm, const uint32_t *adj, uint32_t nwords) {
    World *w = (World *)h;
    w->ext_rest_manifolds.clear(); w->ext_adj.clear();
    for (uint32_t i = 0; i < nm; ++i) w->ext_rest_manifolds.push_back({manifolds2[2 * i], manifolds2[2 * i + 1]});
    for (uint32_t p = 0; p + 1 < nwords;) {
        const uint32_t body = adj[p], count = adj[p + 1];
        p += 2;
        auto &lst = w->ext_adj[body];
        for (uint32_t k = 0; k < count && p + 2 < nwords + 0u + 1u; ++k, p += 3) lst.push_back({adj[p], adj[p + 1], adj[p + 2]});
    }
    w->ext_walk_valid = true;
}
void orc_set_center_of_mass(void *h, uint32_t body, const float *com, float mass) { ((World *)h)->set_center_of_mass(body, v3(com), mass); }
void orc_move_center_of_mass(void *h, uint32_t body, const float *com) { ((World *)h)->set_center_of_mass(body, v3(com), 0.0f, false); }
uint32_t orc_add_joint(void *h, int type, uint32_t a, uint32_t b, const float *pivotA, const float *pivotB,
                       const float *axisA, const float *axisB) {
    return ((World *)h)->add_joint(type, a, b, v3(pivotA), v3(pivotB), v3(axisA), v3(axisB));
}
// ORDER_EXTERNAL: the visiting order of the next step(s), as exported by ref_world.cpp (3 uint32 per contact entry).
void orc_set_ext_order(void *h, const uint32_t *contacts3, uint32_t nc, const uint32_t *joints, uint32_t nj) {
    World *w = (World *)h;
    w->ext_contact_order.clear(); w->ext_joint_order.clear();
    for (uint32_t i = 0; i < nc; ++i) w->ext_contact_order.push_back({contacts3[3 * i], contacts3[3 * i + 1], contacts3[3 * i + 2]});
    for (uint32_t i = 0; i < nj; ++i) w->ext_joint_order.push_back(joints[i]);
    w->ext_order_mismatch = false;
}
int orc_ext_order_mismatch(void *h) { return ((World *)h)->ext_order_mismatch ? 1 : 0; }
// Process-wide: integrate() calls the C library's sinf/cosf like the reference instead of the correctly rounded value.
void orc_set_restitution_iterations(void *h, int iters, int individual) { World *w = (World *)h; w->restitution_iters = iters; w->individual_restitution_iters = individual; }
void orc_set_libm_trig(int on) { g_libm_trig = on != 0; }
void orc_set_fused_rows(int on) { g_arith = on ? (ARITH_FUSED_VELOCITY | ARITH_BLOCK_POSITION) : ARITH_REFERENCE; }
void orc_set_arithmetic(int mode) { g_arith = mode & 7; }
int orc_get_arithmetic() { return g_arith; }
void orc_set_should_collide(void *h, int (*fn)(void *, uint32_t, uint32_t), void *user) { auto *w = (World *)h; w->collide_filter = fn; w->collide_filter_user = user; }
int orc_default_should_collide(void *h, uint32_t a, uint32_t b) { return ((World *)h)->should_collide(a, b) ? 1 : 0; }
void orc_exclude_collision(void *h, uint32_t a, uint32_t b) { ((World *)h)->exclude_collision(a, b); }
void orc_remove_collision_exclusion(void *h, uint32_t a, uint32_t b) { ((World *)h)->remove_collision_exclusion(a, b); }
// island sleeping (off by default)
void orc_set_sleeping(void *h, int enable) { ((World *)h)->sleeping = enable != 0; }
void orc_set_sleeping_disabled(void *h, uint32_t body, int disabled) { ((World *)h)->bodies[body].sleeping_disabled = disabled != 0; }
void orc_wake_all(void *h) { ((World *)h)->wake_all(); }
void orc_get_asleep(void *h, uint8_t *out) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) out[i] = w->bodies[i].asleep ? 1 : 0;
}
void orc_step(void *h, int n) { World *w = (World *)h; for (int i = 0; i < n; ++i) w->step(); }
void orc_run_stage(void *h, int stage) {
    World *w = (World *)h;
    switch (stage) {
    case 0: w->broadphase(); break;
    case 1: w->narrowphase(); break;
    case 2: w->update_islands(); break;
    case 3: w->solve(); break;
    }
}
uint32_t orc_num_bodies(void *h) { return (uint32_t)((World *)h)->bodies.size(); }
void orc_get_state(void *h, float *pos, float *orn, float *linvel, float *angvel) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        const Body &b = w->bodies[i];
        put3(pos + 3 * i, b.pos);
        orn[4 * i] = b.orn.x; orn[4 * i + 1] = b.orn.y; orn[4 * i + 2] = b.orn.z; orn[4 * i + 3] = b.orn.w;
        put3(linvel + 3 * i, b.linvel); put3(angvel + 3 * i, b.angvel);
    }
}
void orc_set_state(void *h, const float *pos, const float *orn, const float *linvel, const float *angvel) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        Body &b = w->bodies[i];
        b.pos = v3(pos + 3 * i);
        b.orn = {orn[4 * i], orn[4 * i + 1], orn[4 * i + 2], orn[4 * i + 3]};
        b.linvel = v3(linvel + 3 * i); b.angvel = v3(angvel + 3 * i);
        b.update_origin();
    }
}
void orc_refresh_derived(void *h) { ((World *)h)->refresh_derived(); }
void orc_get_derived(void *h, float *aabb6, float *inertia_world9, uint32_t *island) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        const Body &b = w->bodies[i];
        if (aabb6) { put3(aabb6 + 6 * i, b.box.min); put3(aabb6 + 6 * i + 3, b.box.max); }
        if (inertia_world9) for (int r = 0; r < 3; ++r) put3(inertia_world9 + 9 * i + 3 * r, b.I_inv_world.row[r]);
        if (island) island[i] = i < w->island_label.size() ? w->island_label[i] : (uint32_t)i;
    }
}
uint32_t orc_num_manifolds(void *h) { return (uint32_t)((World *)h)->manifolds.size(); }
void orc_get_manifolds(void *h, orc_manifold_rec *out) {
    World *w = (World *)h;
    size_t k = 0;
    for (auto &kv : w->manifolds) {
        const Manifold &m = kv.second;
        orc_manifold_rec &r = out[k++];
        std::memset(&r, 0, sizeof(r));
        r.body[0] = m.body[0]; r.body[1] = m.body[1]; r.num_points = (uint32_t)m.num_points; r.colour = m.colour;
        for (int i = 0; i < m.num_points; ++i) {
            const ContactPoint &c = m.pt[i];
            orc_point_rec &p = r.pt[i];
            put3(p.pivotA, c.pivotA); put3(p.pivotB, c.pivotB); put3(p.normal, c.normal); put3(p.local_normal, c.local_normal);
            p.distance = c.distance; p.friction = c.friction; p.restitution = c.restitution;
            p.attachment = c.attachment; p.lifetime = c.lifetime;
            p.normal_impulse = c.normal_impulse; p.friction_impulse[0] = c.friction_impulse[0]; p.friction_impulse[1] = c.friction_impulse[1];
        }
    }
}
void orc_set_manifolds(void *h, const orc_manifold_rec *in, uint32_t n) {
    World *w = (World *)h;
    w->manifolds.clear();
    // the colours come with the records; the number of colours in use (whose top one the next step releases, colour_contacts)
    // is part of the colouring state and follows from them - edynhip_set_manifolds does the same
    uint32_t nc = 0;
    for (uint32_t k = 0; k < n; ++k) if (in[k].colour != kNoColour && in[k].num_points > 0) nc = std::max(nc, in[k].colour + 1);
    w->stats.num_colours = nc;
    for (uint32_t k = 0; k < n; ++k) {
        const orc_manifold_rec &r = in[k];
        Manifold m;
        m.body[0] = r.body[0]; m.body[1] = r.body[1]; m.num_points = (int)r.num_points; m.colour = r.colour;
        for (int i = 0; i < m.num_points; ++i) {
            ContactPoint &c = m.pt[i];
            const orc_point_rec &p = r.pt[i];
            c.pivotA = v3(p.pivotA); c.pivotB = v3(p.pivotB); c.normal = v3(p.normal); c.local_normal = v3(p.local_normal);
            c.distance = p.distance; c.friction = p.friction; c.restitution = p.restitution;
            c.attachment = p.attachment; c.lifetime = p.lifetime;
            c.id = ((uint64_t)k << 2) | (uint64_t)i;   // injected points: high word 0 (edynhip_set_manifolds does the same)
            {   // the record carries no contact_extras data: mixed as for a new point (as edynhip_set_manifolds does)
                const Body &A = w->bodies[m.body[0]], &B = w->bodies[m.body[1]];
                c.roll_friction = std::max(A.roll_friction, B.roll_friction); c.spin_friction = std::max(A.spin_friction, B.spin_friction);
                if (A.stiffness < kLarge || B.stiffness < kLarge) { c.stiffness = 1 / (1 / A.stiffness + 1 / B.stiffness); c.damping = 1 / (1 / A.damping + 1 / B.damping); }
            }
            c.normal_impulse = p.normal_impulse; c.friction_impulse[0] = p.friction_impulse[0]; c.friction_impulse[1] = p.friction_impulse[1];
        }
        m.with_restitution = w->tags_restitution(m.body[0], m.body[1]);
        w->manifolds.emplace(w->pair_key(m.body[0], m.body[1]), m);
    }
}
// per joint 10 floats: the 9 applied-impulse slots (hinge: linear[3], hinge[2], limit, bump_stop, spring, torque; point:
// applied[3], friction) + the tracked hinge angle - the layout of ref_world.cpp's refw_get_joint_impulses
void orc_get_joint_impulses(void *h, float *imp10) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->joints.size(); ++i) {
        const bool alive = w->joints[i].alive;   // a removed joint reads 0, like the device side
        for (int k = 0; k < 9; ++k) imp10[10 * i + k] = alive ? w->joints[i].impulse[k] : 0.0f;
        imp10[10 * i + 9] = alive ? w->joints[i].angle : 0.0f;
    }
}
void orc_set_joint_params(void *h, uint32_t joint, const float *p) {
    World *w = (World *)h;
    Joint &j = w->joints[joint];
    for (int k = 0; k < 10; ++k) j.params[k] = p[k];
    if (j.type == JOINT_HINGE) w->reset_joint_angle(j);
}
void orc_remove_body(void *h, uint32_t body) { ((World *)h)->remove_body(body); }
void orc_remove_joint(void *h, uint32_t joint) { ((World *)h)->remove_joint(joint); }
void orc_set_params(void *h, float dt, int vel_iters, int pos_iters, const float *g) {
    World *w = (World *)h;
    w->dt = dt; w->vel_iters = vel_iters; w->pos_iters = pos_iters;
    if (g) w->set_gravity(v3(g));
}
void orc_step_timed(void *h, int n, double first_time, double step_dt) {
    World *w = (World *)h;
    for (int i = 0; i < n; ++i) w->step_timed(first_time + step_dt * i);
}
void orc_get_stats(void *h, uint32_t *out7) {
    const StepStats &s = ((World *)h)->stats;
    out7[0] = s.num_manifolds; out7[1] = s.num_points; out7[2] = s.num_rows; out7[3] = s.num_islands;
    out7[4] = s.num_colours; out7[5] = s.num_joint_colours; out7[6] = s.colour_rounds;
}
// Wall-clock seconds for `n` steps (single thread) — used by bench.py's cpu_baseline leg.
double orc_time_steps(void *h, int n) {
    World *w = (World *)h;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) w->step();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------- leaf functions for golden-vector and cross-check tests ----------------
// out points: per point 11 floats (pivotA3, pivotB3, normal3, distance, attachment)
int orc_collide(int typeA, const float *paramA, const float *posA, const float *ornA, int typeB, const float *paramB,
                const float *posB, const float *ornB, float threshold, float *out) {
    coll_ctx ctx{v3(posA), {ornA[0], ornA[1], ornA[2], ornA[3]}, v3(posB), {ornB[0], ornB[1], ornB[2], ornB[3]}, threshold};
    coll_result r;
    collide(make_shape(typeA, paramA), make_shape(typeB, paramB), ctx, r);
    for (size_t i = 0; i < r.num_points; ++i) {
        float *o = out + 11 * i;
        put3(o, r.point[i].pivotA); put3(o + 3, r.point[i].pivotB); put3(o + 6, r.point[i].normal);
        o[9] = r.point[i].distance; o[10] = (float)r.point[i].attachment;
    }
    return (int)r.num_points;
}
int orc_intersect_line_aabb(const float *p0, const float *p1, const float *bmin, const float *bmax, float *s) {
    return (int)intersect_line_aabb({p0[0], p0[1]}, {p1[0], p1[1]}, {bmin[0], bmin[1]}, {bmax[0], bmax[1]}, s[0], s[1]);
}
void orc_plane_space(const float *n, float *p, float *q) { vec3 a, b; plane_space(v3(n), a, b); put3(p, a); put3(q, b); }
void orc_integrate(const float *q, const float *w, float dt, float *out) {
    quat r = integrate({q[0], q[1], q[2], q[3]}, v3(w), dt);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_rotate(const float *q, const float *v, float *out) { put3(out, rotate({q[0], q[1], q[2], q[3]}, v3(v))); }
// returns type | index<<8 ; num_points updated in place
int orc_insertion_point_index(const float *pts, int *num_points, const float *np) {
    vec3 p[4];
    for (int i = 0; i < 4; ++i) p[i] = v3(pts + 3 * i);
    size_t n = (size_t)*num_points;
    insert_result r = insertion_point_index(p, 4, n, v3(np));
    *num_points = (int)n;
    return (int)r.type | ((int)(r.index & 0xFF) << 8);
}
float orc_closest_segment_segment(const float *p1, const float *q1, const float *p2, const float *q2, float *st,
                                  float *c, int *num) {
    float s, t, sp = 0, tp = 0; vec3 c1, c2, c1p{0, 0, 0}, c2p{0, 0, 0}; size_t n = 0;
    float d = closest_point_segment_segment(v3(p1), v3(q1), v3(p2), v3(q2), s, t, c1, c2, &n, &sp, &tp, &c1p, &c2p);
    st[0] = s; st[1] = t; st[2] = sp; st[3] = tp;
    put3(c, c1); put3(c + 3, c2); put3(c + 6, c1p); put3(c + 9, c2p);
    *num = (int)n;
    return d;
}
void orc_box_support_feature(const float *h, const float *dir, float threshold, int *feature, int *index, float *proj) {
    box_support_feature_local(v3(h), v3(dir), *feature, *index, *proj, threshold);
}
float orc_box_support_projection(const float *h, const float *pos, const float *orn, const float *dir) {
    return box_support_projection(v3(h), v3(pos), {orn[0], orn[1], orn[2], orn[3]}, v3(dir));
}
// rowdata: J[12], inv_mA, inv_mB, inv_IA[9], inv_IB[9], error, erp, restitution, lower, upper, impulse  (38 floats)
// vel: vA,wA,vB,wB (12) ; delta in/out: dvA,dwA,dvB,dwB (12). out: eff_mass, rhs, new impulse, delta_impulse
void orc_row_prepare_solve(const float *rd, const float *vel, float *delta, float *out) {
    Row r;
    for (int i = 0; i < 4; ++i) r.J[i] = v3(rd + 3 * i);
    r.inv_mA = rd[12]; r.inv_mB = rd[13];
    for (int k = 0; k < 3; ++k) { r.inv_IA.row[k] = v3(rd + 14 + 3 * k); r.inv_IB.row[k] = v3(rd + 23 + 3 * k); }
    RowOptions o; o.error = rd[32]; o.erp = rd[33]; o.restitution = rd[34];
    r.lower = rd[35]; r.upper = rd[36]; r.impulse = rd[37];
    vec3 d[4] = {v3(delta), v3(delta + 3), v3(delta + 6), v3(delta + 9)};
    r.dvA = &d[0]; r.dwA = &d[1]; r.dvB = &d[2]; r.dwB = &d[3];
    prepare_row(r, o, v3(vel), v3(vel + 3), v3(vel + 6), v3(vel + 9));
    float di = solve_row(r);
    apply_row_impulse(di, r);
    out[0] = r.eff_mass; out[1] = r.rhs; out[2] = r.impulse; out[3] = di;
    for (int i = 0; i < 4; ++i) put3(delta + 3 * i, d[i]);
}
int orc_should_collide(uint64_t groupA, uint64_t maskA, uint64_t groupB, uint64_t maskB) {
    return ((groupA & maskB) != 0 && (groupB & maskA) != 0) ? 1 : 0;
}
// frames + the full parameter block of a joint (cone / cvjoint): frames row-major 3x3, params[16] (see oworld.hpp Joint)
void orc_set_joint_definition(void *h, uint32_t joint, const float *fA, const float *fB, const float *params16) {
    World *w = (World *)h;
    Joint &j = w->joints[joint];
    for (int r = 0; r < 3; ++r) { j.frame[0].row[r] = v3(fA + 3 * r); j.frame[1].row[r] = v3(fB + 3 * r); }
    for (int k = 0; k < 16; ++k) j.params[k] = params16[k];
    if (j.type == JOINT_HINGE || j.type == JOINT_CVJOINT) w->reset_joint_angle(j);
}
// generic_constraint: frames + 6 x 10 parameters (oworld.hpp Joint::params); impulses of all 24 slots
void orc_set_generic_definition(void *h, uint32_t joint, const float *fA, const float *fB, const float *dof60) {
    World *w = (World *)h;
    Joint &j = w->joints[joint];
    for (int r = 0; r < 3; ++r) { j.frame[0].row[r] = v3(fA + 3 * r); j.frame[1].row[r] = v3(fB + 3 * r); }
    for (int k = 0; k < 60; ++k) j.params[k] = dof60[k];
}
void orc_get_joint_impulses24(void *h, float *out24) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->joints.size(); ++i)
        for (int k = 0; k < 24; ++k) out24[24 * i + k] = w->joints[i].alive ? w->joints[i].impulse[k] : 0.0f;
}
// what a world carries into another (edynhip_set_joint_warm_start / edynhip_set_asleep do the same on the device)
void orc_set_joint_warm_start(void *h, const float *imp24, const float *angles) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->joints.size(); ++i) {
        if (!w->joints[i].alive) continue;
        for (int k = 0; k < 24; ++k) w->joints[i].impulse[k] = imp24[24 * i + k];
        if (angles) w->joints[i].angle = angles[i];
    }
}
void orc_set_asleep(void *h, const uint8_t *flags) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        Body &b = w->bodies[i];
        if (b.kind != KIND_DYNAMIC || b.removed || b.sleeping_disabled) continue;
        b.asleep = flags[i] != 0;
        if (b.asleep) { b.linvel = {0, 0, 0}; b.angvel = {0, 0, 0}; }
    }
}
// material ids and the mix table (edyn::insert_material_mixing)
void orc_set_material_id(void *h, uint32_t body, uint32_t id) { ((World *)h)->bodies[body].material_id = id; }
void orc_insert_material_mixing(void *h, uint32_t id0, uint32_t id1, const float *m6) {
    World *w = (World *)h;
    w->mix_table[World::IdPair{id0, id1}] = {m6[0], m6[1], m6[2], m6[3], m6[4], m6[5]};
}
// contact_extras materials and impulses
void orc_set_material_extras(void *h, uint32_t body, float spin, float roll, float stiffness, float damping) {
    Body &b = ((World *)h)->bodies[body];
    b.spin_friction = spin; b.roll_friction = roll; b.stiffness = stiffness; b.damping = damping;
}
void orc_get_point_extras(void *h, float *out7) {   // [manifold][4][7]: rolling impulse 0/1, spin impulse, roll mu, spin mu, stiffness, damping
    World *w = (World *)h;
    size_t m = 0;
    for (auto &kv : w->manifolds) {
        for (int k = 0; k < 4; ++k) {
            float *o = out7 + (4 * m + k) * 7;
            for (int i = 0; i < 7; ++i) o[i] = 0;
            if (k >= kv.second.num_points) continue;
            const ContactPoint &c = kv.second.pt[k];
            o[0] = c.rolling_impulse[0]; o[1] = c.rolling_impulse[1]; o[2] = c.spin_impulse;
            o[3] = c.roll_friction; o[4] = c.spin_friction; o[5] = c.stiffness; o[6] = c.damping;
        }
        ++m;
    }
}
// contact events (test counterpart of edynhip_get_contact_events / edynhip_get_point_ids)
void orc_record_events(void *h, int on) { World *w = (World *)h; w->record_events = on != 0; w->events.clear(); }
void orc_clear_events(void *h) { ((World *)h)->events.clear(); }
uint32_t orc_num_events(void *h) { return (uint32_t)((World *)h)->events.size(); }
void orc_get_events(void *h, ContactEvent *out) {
    World *w = (World *)h;
    for (size_t i = 0; i < w->events.size(); ++i) out[i] = w->events[i];
}
void orc_get_point_ids(void *h, uint64_t *ids) {   // [4 * manifold + k], canonical manifold order
    World *w = (World *)h;
    size_t m = 0;
    for (auto &kv : w->manifolds) {
        for (int k = 0; k < 4; ++k) ids[4 * m + k] = k < kv.second.num_points ? kv.second.pt[k].id : 0;
        ++m;
    }
}
uint32_t orc_sizeof_manifold_rec() { return (uint32_t)sizeof(orc_manifold_rec); }

}  // extern "C"

// Batch form of orc_collide for the randomized device-vs-oracle narrowphase test.
static std::vector<uint8_t> g_batch_flags;
extern "C" void orc_collide_batch(uint32_t n, const int32_t *st, const float *sp, const float *pos, const float *orn, float threshold,
                                  float *out, uint32_t *count) {
    g_batch_flags.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        g_poly_flags = 0;
        count[i] = (uint32_t)orc_collide(st[2 * i], sp + 8 * i, pos + 6 * i, orn + 8 * i, st[2 * i + 1], sp + 8 * i + 4, pos + 6 * i + 3,
                                         orn + 8 * i + 4, threshold, out + (size_t)i * 44);
        g_batch_flags[i] = (uint8_t)g_poly_flags;
    }
}
// per pair of the last orc_collide_batch: bit 0 = the pair took a path on which the reference reads uninitialised variables
// (collide_polyhedron_polyhedron.cpp:98-100 when no edge pair spans a Minkowski face) - those pairs cannot be compared with it
extern "C" void orc_last_batch_flags(uint8_t *out) { std::copy(g_batch_flags.begin(), g_batch_flags.end(), out); }

// Twins of oracle/ref_xcheck.cpp's ref_tree_run / ref_friction_solve over the restatement (same argument layouts).
extern "C" uint32_t orc_tree_run(uint32_t nops, const int32_t *ops, const float *boxes, uint32_t *hits, uint32_t max_hits, uint8_t *moved) {
    DynTree tree;
    std::vector<uint32_t> id_of;
    uint32_t nh = 0;
    for (uint32_t i = 0; i < nops; ++i) {
        aabb box{v3(boxes + 6 * i), v3(boxes + 6 * i + 3)};
        int op = ops[2 * i], hnd = ops[2 * i + 1];
        moved[i] = 0;
        if (op == 0) {
            if ((size_t)hnd >= id_of.size()) id_of.resize(hnd + 1, DynTree::NIL);
            id_of[hnd] = tree.create(box, (uint32_t)hnd);
        } else if (op == 1) {
            moved[i] = tree.move(id_of[hnd], box) ? 1 : 0;
        } else if (op == 2) {
            tree.destroy(id_of[hnd]);
            id_of[hnd] = DynTree::NIL;
        } else {
            tree.query(box, [&](uint32_t id) { if (nh < max_hits) hits[nh++] = tree.payload(id); });
            if (nh < max_hits) hits[nh++] = 0xFFFFFFFFu;
        }
    }
    return nh;
}
extern "C" void orc_friction_solve(const float *nd, const float *fd, float *delta, int warm, int sweeps, float *out) {
    Row r;
    for (int i = 0; i < 4; ++i) r.J[i] = v3(nd + 3 * i);
    r.inv_mA = nd[12]; r.inv_mB = nd[13];
    for (int k = 0; k < 3; ++k) { r.inv_IA.row[k] = v3(nd + 14 + 3 * k); r.inv_IB.row[k] = v3(nd + 23 + 3 * k); }
    r.impulse = nd[32];
    vec3 d[4] = {v3(delta), v3(delta + 3), v3(delta + 6), v3(delta + 9)};
    r.dvA = &d[0]; r.dwA = &d[1]; r.dvB = &d[2]; r.dwB = &d[3];
    FrictionRow f;
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 4; ++i) f.row[k].J[i] = v3(fd + 15 * k + 3 * i);
        f.row[k].eff_mass = fd[15 * k + 12]; f.row[k].rhs = fd[15 * k + 13]; f.row[k].impulse = fd[15 * k + 14];
    }
    f.mu = fd[30];
    f.normal_row = 0;
    if (warm) warm_start_friction(f, r);
    for (int s = 0; s < sweeps; ++s) solve_friction(f, r);
    out[0] = f.row[0].impulse; out[1] = f.row[1].impulse;
    for (int i = 0; i < 4; ++i) put3(delta + 3 * i, d[i]);
}
