// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shims over the REAL reference functions, for the few reference translation units that
// compile without EnTT (geom.cpp, quaternion.cpp, constraint_row.cpp, box_shape.cpp, triangle.cpp,
// shape_util.cpp). Built by `make ref` from the sources where they lie under /root/reference into
// oracle/_ref/libedynref.so; tests compare the restatement (liboracle.so) against these bit for bit.
// Nothing from /root/reference is copied into this repository.
#include <edyn/math/geom.hpp>
#include <edyn/math/quaternion.hpp>
#include <edyn/math/vector2.hpp>
#include <edyn/shapes/box_shape.hpp>
#include <edyn/constraints/constraint_row.hpp>
#include <edyn/constraints/constraint_row_options.hpp>

using namespace edyn;
static vector3 v3(const float *p) { return {p[0], p[1], p[2]}; }
static void put3(float *d, vector3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

extern "C" {
int ref_intersect_line_aabb(const float *p0, const float *p1, const float *bmin, const float *bmax, float *s) {
    return (int)intersect_line_aabb(vector2{p0[0], p0[1]}, vector2{p1[0], p1[1]}, vector2{bmin[0], bmin[1]},
                                    vector2{bmax[0], bmax[1]}, s[0], s[1]);
}
void ref_plane_space(const float *n, float *p, float *q) { vector3 a, b; plane_space(v3(n), a, b); put3(p, a); put3(q, b); }
void ref_integrate(const float *q, const float *w, float dt, float *out) {
    quaternion r = integrate(quaternion{q[0], q[1], q[2], q[3]}, v3(w), dt);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void ref_rotate(const float *q, const float *v, float *out) { put3(out, rotate(quaternion{q[0], q[1], q[2], q[3]}, v3(v))); }
int ref_insertion_point_index(const float *pts, int *num_points, const float *np) {
    std::array<vector3, 4> p;
    for (int i = 0; i < 4; ++i) p[i] = v3(pts + 3 * i);
    size_t n = (size_t)*num_points;
    auto r = insertion_point_index(p, n, v3(np));
    *num_points = (int)n;
    int type = 0;
    switch (r.type) {
    case point_insertion_type::none: type = 0; break;
    case point_insertion_type::append: type = 1; break;
    case point_insertion_type::similar: type = 2; break;
    case point_insertion_type::replace: type = 3; break;
    }
    return type | ((int)(r.index & 0xFF) << 8);
}
float ref_closest_segment_segment(const float *p1, const float *q1, const float *p2, const float *q2, float *st,
                                  float *c, int *num) {
    scalar s, t, sp = 0, tp = 0; vector3 c1, c2, c1p{0, 0, 0}, c2p{0, 0, 0}; size_t n = 0;
    float d = closest_point_segment_segment(v3(p1), v3(q1), v3(p2), v3(q2), s, t, c1, c2, &n, &sp, &tp, &c1p, &c2p);
    st[0] = s; st[1] = t; st[2] = sp; st[3] = tp;
    put3(c, c1); put3(c + 3, c2); put3(c + 6, c1p); put3(c + 9, c2p);
    *num = (int)n;
    return d;
}
void ref_box_support_feature(const float *h, const float *dir, float threshold, int *feature, int *index, float *proj) {
    box_shape b{v3(h)};
    box_feature f; size_t idx; scalar p;
    b.support_feature(v3(dir), f, idx, p, threshold);
    *feature = (int)f; *index = (int)idx; *proj = p;
}
float ref_box_support_projection(const float *h, const float *pos, const float *orn, const float *dir) {
    box_shape b{v3(h)};
    return b.support_projection(v3(pos), quaternion{orn[0], orn[1], orn[2], orn[3]}, v3(dir));
}
void ref_row_prepare_solve(const float *rd, const float *vel, float *delta, float *out) {
    constraint_row r;
    for (int i = 0; i < 4; ++i) r.J[i] = v3(rd + 3 * i);
    r.inv_mA = rd[12]; r.inv_mB = rd[13];
    for (int k = 0; k < 3; ++k) { r.inv_IA.row[k] = v3(rd + 14 + 3 * k); r.inv_IB.row[k] = v3(rd + 23 + 3 * k); }
    constraint_row_options o; o.error = rd[32]; o.erp = rd[33]; o.restitution = rd[34];
    r.lower_limit = rd[35]; r.upper_limit = rd[36]; r.impulse = rd[37];
    delta_linvel dv[2] = {delta_linvel{v3(delta)}, delta_linvel{v3(delta + 6)}};
    delta_angvel dw[2] = {delta_angvel{v3(delta + 3)}, delta_angvel{v3(delta + 9)}};
    r.dvA = &dv[0]; r.dwA = &dw[0]; r.dvB = &dv[1]; r.dwB = &dw[1];
    prepare_row(r, o, v3(vel), v3(vel + 3), v3(vel + 6), v3(vel + 9));
    float di = solve(r);
    apply_row_impulse(di, r);
    out[0] = r.eff_mass; out[1] = r.rhs; out[2] = r.impulse; out[3] = di;
    put3(delta, dv[0]); put3(delta + 3, dw[0]); put3(delta + 6, dv[1]); put3(delta + 9, dw[1]);
}
}

// ---------------------------------------------------------------------------------------------------
// Leaves that need the EnTT-dependent translation units (built against oracle/entt_min): closest-feature
// routines, collision_result, dynamic_tree, friction rows.
#include <edyn/collision/collide.hpp>
#include <edyn/collision/dynamic_tree.hpp>
#include <edyn/constraints/constraint_row_friction.hpp>
#include <edyn/util/aabb_util.hpp>
#include <edyn/shapes/polyhedron_shape.hpp>
#include <edyn/shapes/convex_mesh.hpp>
#include <edyn/dynamics/moment_of_inertia.hpp>
#include <memory>
#include <variant>
#include <vector>

namespace {
using ref_shape = std::variant<std::monostate, box_shape, sphere_shape, plane_shape, capsule_shape, cylinder_shape, polyhedron_shape>;
std::vector<std::shared_ptr<convex_mesh>> g_ref_meshes;
ref_shape make_ref_shape(int type, const float *p) {
    if (type == 6) return polyhedron_shape{g_ref_meshes[(int)p[0]]};
    if (type == 1) return box_shape{v3(p)};
    if (type == 2) return sphere_shape{p[0]};
    if (type == 3) return plane_shape{v3(p), p[3]};
    if (type == 4) return capsule_shape{p[0], p[1], (coordinate_axis)(int)p[2]};
    if (type == 5) return cylinder_shape{p[0], p[1], (coordinate_axis)(int)p[2]};
    return std::monostate{};
}
}  // namespace

std::shared_ptr<edyn::convex_mesh> ref_mesh(int id) { return g_ref_meshes[id]; }
extern "C" {
int ref_create_mesh(uint32_t nv, const float *verts, uint32_t nidx, const uint32_t *indices, uint32_t nfaces, const uint32_t *faces) {
    auto m = std::make_shared<convex_mesh>();
    for (uint32_t i = 0; i < nv; ++i) m->vertices.push_back(v3(verts + 3 * i));
    m->indices.assign(indices, indices + nidx);
    m->faces.assign(faces, faces + 2 * nfaces);
    m->initialize();
    g_ref_meshes.push_back(m);
    return (int)g_ref_meshes.size() - 1;
}
uint32_t ref_mesh_get(int id, int what, void *out) {   // fields as orc_mesh_get
    const convex_mesh &m = *g_ref_meshes[id];
    const std::vector<vector3> *fv[5] = {&m.vertices, &m.normals, nullptr, &m.edge_vertices, &m.edge_normals};
    const std::vector<uint32_t> *uv[6] = {&m.edges, &m.edge_faces, &m.relevant_faces, &m.relevant_edges, &m.neighbors_start, &m.neighbor_indices};
    if (what == 2) {
        if (out) for (size_t i = 0; i < m.relevant_faces.size(); ++i) put3((float *)out + 3 * i, m.relevant_normals[i]);
        return (uint32_t)m.relevant_faces.size();
    }
    if (what < 5) {
        if (out) for (size_t i = 0; i < fv[what]->size(); ++i) put3((float *)out + 3 * i, (*fv[what])[i]);
        return (uint32_t)fv[what]->size();
    }
    if (out) std::copy(uv[what - 5]->begin(), uv[what - 5]->end(), (uint32_t *)out);
    return (uint32_t)uv[what - 5]->size();
}
void ref_mesh_inertia(int id, float mass, float *out9) {
    const auto I = moment_of_inertia(polyhedron_shape{g_ref_meshes[id]}, mass);
    for (int r = 0; r < 3; ++r) put3(out9 + 3 * r, I.row[r]);
}
// Same signature as orc_collide_batch: per pair shape types st[2], shape params sp[2][4], pos[2][3], orn[2][4];
// out per point 11 floats (pivotA, pivotB, normal, distance, normal_attachment). collide.hpp:43-330 overloads.
void ref_collide_batch(uint32_t n, const int32_t *st, const float *sp, const float *pos, const float *orn, float threshold,
                       float *out, uint32_t *count) {
    for (uint32_t i = 0; i < n; ++i) {
        auto shA = make_ref_shape(st[2 * i], sp + 8 * i), shB = make_ref_shape(st[2 * i + 1], sp + 8 * i + 4);
        const float *pa = pos + 6 * i, *pb = pa + 3, *qa = orn + 8 * i, *qb = qa + 4;
        collision_result result;
        // the rotated mesh as update_rotated_meshes leaves it: orientation * rotated_mesh_list::orientation (identity)
        rotated_mesh rotA, rotB;
        if (auto *pa_ = std::get_if<polyhedron_shape>(&shA)) { rotA = make_rotated_mesh(*pa_->mesh, quaternion{qa[0], qa[1], qa[2], qa[3]} * quaternion_identity); pa_->rotated = &rotA; }
        if (auto *pb_ = std::get_if<polyhedron_shape>(&shB)) { rotB = make_rotated_mesh(*pb_->mesh, quaternion{qb[0], qb[1], qb[2], qb[3]} * quaternion_identity); pb_->rotated = &rotB; }
        std::visit([&](auto &&a) {
            std::visit([&](auto &&b) {
                using A = std::decay_t<decltype(a)>;
                using B = std::decay_t<decltype(b)>;
                if constexpr (!std::is_same_v<A, std::monostate> && !std::is_same_v<B, std::monostate> &&
                              !(std::is_same_v<A, plane_shape> && std::is_same_v<B, plane_shape>)) {
                    quaternion ornA{qa[0], qa[1], qa[2], qa[3]}, ornB{qb[0], qb[1], qb[2], qb[3]};
                    collision_context ctx{v3(pa), ornA, shape_aabb(a, v3(pa), ornA), v3(pb), ornB, shape_aabb(b, v3(pb), ornB), threshold};
                    collide(a, b, ctx, result);
                }
            }, shB);
        }, shA);
        count[i] = (uint32_t)result.num_points;
        for (size_t k = 0; k < result.num_points; ++k) {
            float *o = out + (size_t)i * 44 + 11 * k;
            put3(o, result.point[k].pivotA); put3(o + 3, result.point[k].pivotB); put3(o + 6, result.point[k].normal);
            o[9] = result.point[k].distance; o[10] = (float)(int)result.point[k].normal_attachment;
        }
    }
}

// Scripted dynamic_tree session (dynamic_tree.cpp:41-339, query_tree.hpp:9-42).
// ops: nops x {op, handle}; op 0 create(handle = payload), 1 move(handle), 2 destroy(handle), 3 query.
// boxes: nops x 6 (min, max). hits: visited payloads in visit order, each query terminated by 0xFFFFFFFF;
// moved[i] = move()'s return value. Returns the number of uint32 written to hits.
uint32_t ref_tree_run(uint32_t nops, const int32_t *ops, const float *boxes, uint32_t *hits, uint32_t max_hits, uint8_t *moved) {
    dynamic_tree tree;
    std::vector<tree_node_id_t> id_of;
    uint32_t nh = 0;
    for (uint32_t i = 0; i < nops; ++i) {
        AABB box{v3(boxes + 6 * i), v3(boxes + 6 * i + 3)};
        int op = ops[2 * i], hnd = ops[2 * i + 1];
        moved[i] = 0;
        if (op == 0) {
            if ((size_t)hnd >= id_of.size()) id_of.resize(hnd + 1, null_tree_node_id);
            id_of[hnd] = tree.create(box, entt::entity{(uint32_t)hnd});
        } else if (op == 1) {
            moved[i] = tree.move(id_of[hnd], box) ? 1 : 0;
        } else if (op == 2) {
            tree.destroy(id_of[hnd]);
            id_of[hnd] = null_tree_node_id;
        } else {
            tree.query(box, [&](tree_node_id_t id) { if (nh < max_hits) hits[nh++] = entt::to_integral(tree.get_node(id).entity); });
            if (nh < max_hits) hits[nh++] = 0xFFFFFFFFu;
        }
    }
    return nh;
}

// Friction pair against its normal row (constraint_row_friction.cpp:11-66): warm_start (optional) then `sweeps` x solve_friction.
// normal: J[12], inv_mA, inv_mB, inv_IA[9], inv_IB[9], impulse (33) ; fric: 2 x {J[12], eff_mass, rhs, impulse} (30), mu (1)
// delta in/out: dvA, dwA, dvB, dwB (12) ; out: the two friction impulses.
void ref_friction_solve(const float *nd, const float *fd, float *delta, int warm, int sweeps, float *out) {
    std::vector<constraint_row> rows(1);
    constraint_row &r = rows[0];
    for (int i = 0; i < 4; ++i) r.J[i] = v3(nd + 3 * i);
    r.inv_mA = nd[12]; r.inv_mB = nd[13];
    for (int k = 0; k < 3; ++k) { r.inv_IA.row[k] = v3(nd + 14 + 3 * k); r.inv_IB.row[k] = v3(nd + 23 + 3 * k); }
    r.impulse = nd[32];
    delta_linvel dv[2] = {delta_linvel{v3(delta)}, delta_linvel{v3(delta + 6)}};
    delta_angvel dw[2] = {delta_angvel{v3(delta + 3)}, delta_angvel{v3(delta + 9)}};
    r.dvA = &dv[0]; r.dwA = &dw[0]; r.dvB = &dv[1]; r.dwB = &dw[1];
    constraint_row_friction f;
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 4; ++i) f.row[k].J[i] = v3(fd + 15 * k + 3 * i);
        f.row[k].eff_mass = fd[15 * k + 12]; f.row[k].rhs = fd[15 * k + 13]; f.row[k].impulse = fd[15 * k + 14];
    }
    f.friction_coefficient = fd[30];
    f.normal_row_index = 0;
    if (warm) warm_start(f, rows);
    for (int s = 0; s < sweeps; ++s) solve_friction(f, rows);
    out[0] = f.row[0].impulse; out[1] = f.row[1].impulse;
    put3(delta, dv[0]); put3(delta + 3, dw[0]); put3(delta + 6, dv[1]); put3(delta + 9, dw[1]);
}
}
