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
