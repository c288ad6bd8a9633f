// cylinder_shape on the device: support functions, AABB, inertia and the closest-feature routines of its pairs.
//   /root/reference/include/edyn/shapes/cylinder_shape.hpp:1-64, src/edyn/shapes/cylinder_shape.cpp:1-59
//   /root/reference/src/edyn/util/shape_util.cpp:307-349, src/edyn/util/aabb_util.cpp:72-79, src/edyn/dynamics/moment_of_inertia.cpp:27-44
//   /root/reference/src/edyn/math/geom.cpp:24-33,172-215,217-439,441-474,476-728,772-798 (disc / circle-line / circle-circle geometry)
//   /root/reference/src/edyn/collision/collide/collide_cylinder_{plane,sphere,cylinder,box}.cpp, collide_capsule_cylinder.cpp
// These routines run in their own kernel (narrowphase.hip k_np_detect_ext), launched only for worlds that contain a cylinder,
// on the manifolds that involve one: they are written with plain arrays and loops (cylinder-cylinder alone runs up to twelve Newton
// iterations of circle-line / circle-circle searches) and would cost the register-resident box / sphere / capsule kernel its
// occupancy. Every value is produced by the reference's operations in the reference's order; sin / cos / atan2 of the Newton
// iterations are evaluated in double and rounded once (sin_cr / cos_cr / atan2_cr), the convention of integrate() and the hinge.
#pragma once
#include "dcollide.hpp"

namespace dc {

struct CylSh { float radius, half_length; int axis; };
DI CylSh cyl_of(float4 s) { return CylSh{s.x, s.y, (int)s.z}; }
DI f2 mk2(float x, float y) { return {x, y}; }
DI f3 axis_vec(int axis) { return axis == 0 ? mk3(1, 0, 0) : (axis == 1 ? mk3(0, 1, 0) : mk3(0, 0, 1)); }
DI float atan2_cr(float y, float x) { return (float)atan2((double)y, (double)x); }
DI float to_sign(bool b) { return b ? 1.0f : -1.0f; }
DI f3 &operator/=(f3 &a, float s) { const float z = 1.0f / s; a.x *= z; a.y *= z; a.z *= z; return a; }   // vector3 operator/=
// vector2.hpp
DI f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
DI f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
DI f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
DI f2 operator/(f2 a, float s) { return {a.x / s, a.y / s}; }
DI float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
DI float length_sqr(f2 a) { return dot(a, a); }
DI float length(f2 a) { return sqrtf(length_sqr(a)); }
DI float distance_sqr(f2 a, f2 b) { return length_sqr(a - b); }
DI f2 orthogonal(f2 v) { return {-v.y, v.x}; }
DI f2 normalize(f2 v) { return v / length(v); }
DI f3 project_direction(f3 v, f3 n) { return v - n * dot(v, n); }
DI f3 quaternion_x(q4 q) { return rotate(q, mk3(1, 0, 0)); }
DI f3 quaternion_y(q4 q) { return rotate(q, mk3(0, 1, 0)); }
DI f3 quaternion_z(q4 q) { return rotate(q, mk3(0, 0, 1)); }
DI float closest_point_segment(f3 q0, f3 q1, f3 p, float &t, f3 &q) {   // geom.cpp:12-22
    const f3 v = q1 - q0, w = p - q0;
    const float a = dot(w, v), b = dot(v, v);
    t = clamp_unit(a / b);
    q = q0 + v * t;
    return length_sqr(p - q);
}
DI float closest_point_line(f3 q0, f3 dir, f3 p, float &t, f3 &r) {   // geom.cpp:35-44
    const f3 w = p - q0;
    const float a = dot(w, dir), b = dot(dir, dir);
    t = a / b;
    r = q0 + dir * t;
    return length_sqr(p - r);
}
DI void closest_point_segment_segment(f3 p1, f3 q1, f3 p2, f3 q2, float &, float &, f3 &c1, f3 &c2, int *num, float *, float *, f3 *c1p, f3 *c2p) {
    closest_segment_segment<true>(p1, q1, p2, q2, c1, c2, *num, *c1p, *c2p);   // geom.cpp:73-170 (dcollide.hpp)
}
DI void capsule_vertices(const CylSh &s, f3 pos, q4 orn, f3 (&out)[2]) {   // capsule_shape::get_vertices
    const f3 dir = rotate(orn, axis_vec(s.axis));
    out[0] = pos + dir * s.half_length;
    out[1] = pos - dir * s.half_length;
}
DI float capsule_support_projection(const f3 (&v)[2], float radius, f3 dir) { return fmaxf(dot(v[0], dir), dot(v[1], dir)) + radius; }   // shape_util.cpp:297-300

enum cyl_feature : int { CF_FACE = 0, CF_SIDE_EDGE = 1, CF_CAP_EDGE = 2 };

// ---- shape functions
DI f3 cylinder_support_point_local(float radius, float half_length, int axis, f3 dir) {   // shape_util.cpp:307-330
    const int ai = axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    const float planar_len_sq = dir[o0] * dir[o0] + dir[o1] * dir[o1];
    f3 sup = mk3(0, 0, 0);
    sup[ai] = dir[ai] < 0 ? -half_length : half_length;
    if (planar_len_sq > kEps) {
        const float d = radius / sqrtf(planar_len_sq);
        sup[o0] = dir[o0] * d;
        sup[o1] = dir[o1] * d;
    } else {
        sup[o0] = radius;
        sup[o1] = 0;
    }
    return sup;
}
DI f3 cylinder_support_point(float radius, float half_length, int axis, q4 orn, f3 dir) {   // :332-337
    const f3 local_dir = rotate(conjugate(orn), dir);
    return rotate(orn, cylinder_support_point_local(radius, half_length, axis, local_dir));
}
DI f3 cylinder_support_point(const CylSh &s, f3 pos, q4 orn, f3 dir) {   // :339-342
    return pos + cylinder_support_point(s.radius, s.half_length, s.axis, orn, dir);
}
DI float cylinder_support_projection(const CylSh &s, f3 pos, q4 orn, f3 dir) {   // :344-349
    const f3 local_dir = rotate(conjugate(orn), dir);
    const f3 pt = cylinder_support_point_local(s.radius, s.half_length, s.axis, local_dir);
    return dot(pos, dir) + dot(pt, local_dir);
}
DI void cylinder_support_feature_local(const CylSh &s, f3 dir, int &feature, int &index, float threshold) {   // cylinder_shape.cpp:15-50
    const int ai = s.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    const float ortho_dir_len_sqr = dir[o0] * dir[o0] + dir[o1] * dir[o1];
    const float proj_cap_face_sqr = 4.0f * s.radius * s.radius * ortho_dir_len_sqr;
    if (proj_cap_face_sqr < threshold * threshold) {
        feature = CF_FACE;
        index = dir[ai] > 0 ? 0 : 1;
        return;
    }
    const float proj_side_edge = fabsf(2.0f * s.half_length * dir[ai]);
    if (proj_side_edge < threshold) {
        feature = CF_SIDE_EDGE;
        return;
    }
    feature = CF_CAP_EDGE;
    index = dir[ai] > 0 ? 0 : 1;
}
DI void cylinder_support_feature(const CylSh &s, f3 /*pos*/, q4 orn, f3 axis_dir, int &feature, int &index, float threshold) {   // :52-57
    cylinder_support_feature_local(s, rotate(conjugate(orn), axis_dir), feature, index, threshold);
}
DI void cylinder_vertices(const CylSh &s, f3 pos, q4 orn, f3 out[2]) {   // cylinder_shape.hpp:33-39
    const f3 dir = rotate(orn, axis_vec(s.axis));
    out[0] = pos + dir * s.half_length;
    out[1] = pos - dir * s.half_length;
}
DI box3 cylinder_aabb(const CylSh &s, f3 pos, q4 orn) {   // aabb_util.cpp:72-79
    const f3 ptx = cylinder_support_point(s.radius, s.half_length, s.axis, orn, mk3(1, 0, 0));
    const f3 pty = cylinder_support_point(s.radius, s.half_length, s.axis, orn, mk3(0, 1, 0));
    const f3 ptz = cylinder_support_point(s.radius, s.half_length, s.axis, orn, mk3(0, 0, 1));
    const f3 v{ptx.x, pty.y, ptz.z};
    return {pos - v, pos + v};
}
DI f3 cylinder_inertia_diag(const CylSh &s, float mass) {   // moment_of_inertia.cpp:27-44,167-169 (the diagonal)
    const float len = s.half_length * 2, radius = s.radius;
    const float xx = 0.5f * mass * radius * radius;
    const float yy_zz = 1.0f / 12.0f * mass * (3.0f * radius * radius + len * len);
    return s.axis == 0 ? mk3(xx, yy_zz, yy_zz) : (s.axis == 1 ? mk3(yy_zz, xx, yy_zz) : mk3(yy_zz, yy_zz, xx));
}

// ---- geometry
DI float distance_sqr_line(f3 q0, f3 dir, f3 p) {   // geom.cpp:24-33
    const f3 w = p - q0;
    const float a = dot(w, dir), b = dot(dir, dir);
    const float t = a / b;
    const f3 q = q0 + dir * t;
    return length_sqr(p - q);
}
DI float closest_point_disc(f3 dpos, q4 dorn, float radius, int axis, f3 p, f3 &q) {   // :172-192
    const f3 normal = rotate(dorn, axis_vec(axis));
    const float ln = dot(p - dpos, normal);
    const f3 p_proj = p - normal * ln;
    const f3 d = p_proj - dpos;
    const float l2 = length_sqr(d);
    if (l2 < radius * radius) {
        q = p_proj;
        return ln * ln;
    }
    const float l = sqrtf(l2);
    const f3 dn = d / l;
    q = dpos + dn * radius;
    return length_sqr(p - q);
}
DI int intersect_line_circle(f2 p0, f2 p1, float radius, float &s0, float &s1) {   // :194-215
    const f2 d = p1 - p0;
    const float dl2 = length_sqr(d);
    const float dp = dot(d, p0);
    const float delta = dp * dp - dl2 * (dot(p0, p0) - radius * radius);
    if (delta < 0) return 0;
    if (delta > kEps) {
        const float delta_sqrt = sqrtf(delta);
        const float dl2_inv = 1 / dl2;
        s0 = -(dp + delta_sqrt) * dl2_inv;
        s1 = -(dp - delta_sqrt) * dl2_inv;
        return 2;
    }
    s0 = -dp * dl2;
    return 1;
}
DI f3 support_point_circle(f3 pos, q4 orn, float radius, int axis, f3 dir) {   // :772-798
    const int ni = axis, t0 = (ni + 1) % 3, t1 = (ni + 2) % 3;
    const f3 local_dir = rotate(conjugate(orn), dir);
    const float len_plane_sqr = local_dir[t0] * local_dir[t0] + local_dir[t1] * local_dir[t1];
    f3 sup = mk3(0, 0, 0);
    if (len_plane_sqr > kEps) {
        const float d = radius / sqrtf(len_plane_sqr);
        sup[ni] = 0; sup[t0] = local_dir[t0] * d; sup[t1] = local_dir[t1] * d;
    } else {
        sup[ni] = 0; sup[t0] = radius; sup[t1] = 0;
    }
    return pos + rotate(orn, sup);
}
DI float closest_point_circle_line(f3 cpos, q4 corn, float radius, int axis, f3 p0, f3 p1, int &num_points,
                                       float &s0, f3 &rc0, f3 &rl0, float &s1, f3 &rc1, f3 &rl1, f3 &normal,
                                       float threshold = kSupportTolerance) {   // :217-439
    const f3 q0 = to_object(p0, cpos, corn), q1 = to_object(p1, cpos, corn);
    const f3 qv = q1 - q0;
    const float qv_len_sqr = length_sqr(qv);
    const int ni = axis, t0 = (ni + 1) % 3, t1 = (ni + 2) % 3;
    const float qv_proj_len = length(mk2(qv[t0], qv[t1]));
    const float diameter = square(radius);
    const f2 q0_proj{q0[t0], q0[t1]}, q1_proj{q1[t0], q1[t1]};
    if (qv_proj_len > kEps && fabsf(qv[ni] / qv_proj_len) * diameter < threshold) {
        const f3 tangent = cross(qv, axis_vec(axis));
        normal = cross(qv, tangent);
        normal = rotate(corn, normal);
        normal = normalize(normal);
        num_points = intersect_line_circle(q0_proj, q1_proj, radius, s0, s1);
        if (num_points > 0) {
            const f3 rl0_local = q0 + qv * s0;
            f3 rc0_local = rl0_local;
            rc0_local[ni] = 0;
            rl0 = cpos + rotate(corn, rl0_local);
            rc0 = cpos + rotate(corn, rc0_local);
            float dist2 = square(rl0_local[ni]);
            if (num_points > 1) {
                const f3 rl1_local = q0 + qv * s1;
                f3 rc1_local = rl1_local;
                rc1_local[ni] = 0;
                rl1 = cpos + rotate(corn, rl1_local);
                rc1 = cpos + rotate(corn, rc1_local);
                dist2 = fminf(dist2, square(rl1_local[ni]));
            }
            return dist2;
        } else {
            closest_point_line(p0, p1 - p0, cpos, s0, rl0);
            const f3 proj = project_plane(rl0, cpos, normal);
            const f3 dir = normalize(proj - cpos);
            rc0 = cpos + dir * radius;
            const f3 d = rl0 - rc0;
            const float dl2 = length_sqr(d);
            if (dl2 > kEps) normal = d / sqrtf(dl2);
            else normal = dir;
            num_points = 1;
            return dl2;
        }
    }
    if (length_sqr(q0_proj) <= kEps && length_sqr(q1_proj) <= kEps) {
        num_points = 1;
        normal = axis == 0 ? quaternion_y(corn) : (axis == 1 ? quaternion_z(corn) : quaternion_x(corn));
        s0 = -q0[ni] / qv[ni];
        rc0 = cpos + normal * radius;
        rl0 = lerp(p0, p1, s0);
        return radius * radius;
    }
    const f3 q_plane = q0 - (q0[ni] / qv[ni]) * qv;
    const float initial_theta = atan2_cr(q_plane[t0], q_plane[t1]);
    const float qv_len_sqr_inv = 1.0f / qv_len_sqr;
    float theta = initial_theta;
    for (int i = 0; i < 20; ++i) {
        const float sin_theta = sin_cr(theta), cos_theta = cos_cr(theta);
        f3 q_theta = mk3(0, 0, 0), d_q_theta = mk3(0, 0, 0), dd_q_theta = mk3(0, 0, 0);
        q_theta[ni] = 0; q_theta[t0] = sin_theta * radius; q_theta[t1] = cos_theta * radius;
        d_q_theta[ni] = 0; d_q_theta[t0] = cos_theta * radius; d_q_theta[t1] = -sin_theta * radius;
        dd_q_theta[ni] = 0; dd_q_theta[t0] = -sin_theta * radius; dd_q_theta[t1] = -cos_theta * radius;
        const f3 c_theta = q0 + dot(q_theta - q0, qv) * qv_len_sqr_inv * qv;
        const f3 d_c_theta = dot(d_q_theta, qv) * qv_len_sqr_inv * qv;
        const f3 dd_c_theta = dot(dd_q_theta, qv) * qv_len_sqr_inv * qv;
        const f3 d_theta = q_theta - c_theta;
        const f3 d_d_theta = d_q_theta - d_c_theta;
        const f3 dd_d_theta = dd_q_theta - dd_c_theta;
        const float d_f_theta = dot(d_theta, d_d_theta);
        const float dd_f_theta = dot(d_d_theta, d_d_theta) + dot(dd_d_theta, d_theta);
        const float delta = d_f_theta / dd_f_theta;
        theta -= delta;
        if (fabsf(delta) < kPi * 1.0f / 180.0f) break;
    }
    const float closest_sin_theta = sin_cr(theta), closest_cos_theta = cos_cr(theta);
    f3 rc0_local = mk3(0, 0, 0);
    rc0_local[ni] = 0; rc0_local[t0] = closest_sin_theta * radius; rc0_local[t1] = closest_cos_theta * radius;
    f3 rl0_local;
    const float dist_sqr = closest_point_line(q0, qv, rc0_local, s0, rl0_local);
    rc0 = cpos + rotate(corn, rc0_local);
    rl0 = cpos + rotate(corn, rl0_local);
    f3 tangent = mk3(0, 0, 0);
    tangent[ni] = 0; tangent[t0] = closest_cos_theta; tangent[t1] = -closest_sin_theta;
    normal = cross(tangent, qv);
    const float normal_len_sqr = length_sqr(normal);
    if (normal_len_sqr > kEps) {
        normal /= sqrtf(normal_len_sqr);
        normal = rotate(corn, normal);
    } else if (dist_sqr > kEps) {
        normal = (rl0 - rc0) / sqrtf(dist_sqr);
    } else {
        normal[ni] = 0; normal[t0] = closest_sin_theta; normal[t1] = closest_cos_theta;
        normal = rotate(corn, normal);
    }
    num_points = 1;
    return dist_sqr;
}
DI int intersect_circle_circle(f2 posA, float radiusA, f2 posB, float radiusB, f2 &res0, f2 &res1) {   // :441-474
    const f2 u = posB - posA;
    const float lu2 = length_sqr(u);
    const float rsum = radiusA + radiusB, rsub = radiusA - radiusB;
    if (lu2 < kEps && rsub < kEps) {
        res0 = posA + mk2(1, 0) * radiusA;
        res1 = posB - mk2(1, 0) * radiusB;
        return 2;
    }
    if (lu2 < rsub * rsub || lu2 > rsum * rsum) return 0;
    const float lu2_inv = 1.0f / lu2;
    const float s = ((radiusA * radiusA - radiusB * radiusB) * lu2_inv + 1.0f) * 0.5f;
    const float t = sqrtf(fmaxf(0.0f, radiusA * radiusA * lu2_inv - s * s));
    const f2 v = orthogonal(u);
    const f2 su = s * u, tv = t * v;
    res0 = posA + su + tv;
    res1 = posA + su - tv;
    return t > kEps ? 2 : 1;
}
DI float closest_point_circle_circle(f3 posA, q4 ornA, float radiusA, int axisA, f3 posB, q4 ornB, float radiusB, int axisB,
                                         int &num_points, f3 &rA0, f3 &rB0, f3 &rA1, f3 &rB1, f3 &normal) {   // :476-728
    const f3 normalA = rotate(ornA, axis_vec(axisA)), normalB = rotate(ornB, axis_vec(axisB));
    const int nA = axisA, tA0 = (nA + 1) % 3, tA1 = (nA + 2) % 3;
    const int nB = axisB, tB0 = (nB + 1) % 3, tB1 = (nB + 2) % 3;
    const f3 posB_in_A = to_object(posB, posA, ornA);
    if (!(length_sqr(cross(normalA, normalB)) > kEps)) {   // parallel
        normal = normalB;
        const f2 posB_in_A_proj{posB_in_A[tA0], posB_in_A[tA1]};
        f2 c0, c1;
        const int np = intersect_circle_circle(mk2(0, 0), radiusA, posB_in_A_proj, radiusB, c0, c1);
        if (np > 0) {
            num_points = np;
            f3 rA0_local = mk3(0, 0, 0);
            rA0_local[nA] = 0; rA0_local[tA0] = c0.x; rA0_local[tA1] = c0.y;
            f3 rB0_local = rA0_local;
            rB0_local[nA] = posB_in_A[nA];
            rA0 = to_world(rA0_local, posA, ornA);
            rB0 = to_world(rB0_local, posA, ornA);
            if (np > 1) {
                f3 rA1_local = mk3(0, 0, 0);
                rA1_local[nA] = 0; rA1_local[tA0] = c1.x; rA1_local[tA1] = c1.y;
                f3 rB1_local = rA1_local;
                rB1_local[nA] = posB_in_A[nA];
                rA1 = to_world(rA1_local, posA, ornA);
                rB1 = to_world(rB1_local, posA, ornA);
            }
            return square(posB_in_A[nA]);
        } else {
            num_points = 1;
            f2 dir = posB_in_A_proj;
            const float dir_len_sqr = length_sqr(dir);
            f3 tanA = mk3(0, 0, 0);
            tanA[tA0] = 1;
            if (dir_len_sqr > kEps) {
                { const float z = 1.0f / sqrtf(dir_len_sqr); dir.x *= z; dir.y *= z; }   // vector2 operator/=
                const f3 pointA = tanA * radiusA;
                const f3 pointB_in_A = posB_in_A + tanA * radiusB;
                const bool A_contains_B = length_sqr(mk2(pointB_in_A[tA0], pointB_in_A[tA1])) < radiusA * radiusA;
                const bool B_contains_A = distance_sqr(mk2(pointA[tA0], pointA[tA1]), posB_in_A_proj) < radiusB * radiusB;
                f3 dirA = mk3(0, 0, 0), dirB = mk3(0, 0, 0);
                dirA[nA] = 0; dirA[tA0] = dir.x; dirA[tA1] = dir.y;
                dirB[nB] = 0; dirB[tB0] = dir.x; dirB[tB1] = dir.y;
                dirA *= B_contains_A ? -1.0f : 1.0f;
                dirB *= (B_contains_A || (!A_contains_B && !B_contains_A)) ? -1.0f : 1.0f;
                rA0 = to_world(dirA * radiusA, posA, ornA);
                rB0 = to_world(posB_in_A + dirB * radiusB, posA, ornA);
                return distance_sqr(rA0, rB0);
            } else {
                rA0 = to_world(tanA * radiusA, posA, ornA);
                rB0 = to_world(tanA * radiusB, posA, ornA);
                return distance_sqr(rA0, rB0);
            }
        }
    }
    const q4 ornB_in_A = conjugate(ornA) * ornB;
    f3 u, v;
    if (axisA == 0) { u = quaternion_z(ornB_in_A); v = quaternion_y(ornB_in_A); }
    else if (axisA == 1) { u = quaternion_x(ornB_in_A); v = quaternion_z(ornB_in_A); }
    else { u = quaternion_y(ornB_in_A); v = quaternion_x(ornB_in_A); }
    const f3 sup_pos = support_point_circle(posB_in_A, ornB_in_A, radiusB, axisB, axis_vec(axisA));
    const f3 sup_neg = support_point_circle(posB_in_A, ornB_in_A, radiusB, axisB, -axis_vec(axisA));
    const f3 sup = fabsf(sup_pos[nA]) < fabsf(sup_neg[nA]) ? sup_pos : sup_neg;
    const f3 sup_in_B = to_object(sup, posB_in_A, ornB_in_A);
    const float initial_phi = atan2_cr(sup_in_B[tA0], sup_in_B[tA1]);
    float phi = initial_phi;
    for (int i = 0; i < 20; ++i) {
        const float cos_phi = cos_cr(phi), sin_phi = sin_cr(phi);
        const f3 p_phi = posB_in_A + (u * cos_phi + v * sin_phi) * radiusB;
        const f3 d_p_phi = (u * -sin_phi + v * cos_phi) * radiusB;
        const f3 dd_p_phi = (u * -cos_phi + v * -sin_phi) * radiusB;
        const float theta = atan2_cr(p_phi[tA0], p_phi[tA1]);
        const float cos_theta = cos_cr(theta), sin_theta = sin_cr(theta);
        f3 q_theta = mk3(0, 0, 0), d_q_theta = mk3(0, 0, 0), dd_q_theta = mk3(0, 0, 0);
        q_theta[nA] = 0; q_theta[tA0] = sin_theta * radiusA; q_theta[tA1] = cos_theta * radiusA;
        d_q_theta[nA] = 0; d_q_theta[tA0] = cos_theta * radiusA; d_q_theta[tA1] = -sin_theta * radiusA;
        dd_q_theta[nA] = 0; dd_q_theta[tA0] = -sin_theta * radiusA; dd_q_theta[tA1] = -cos_theta * radiusA;
        const f3 d_phi = p_phi - q_theta;
        const f3 d_d_phi = d_p_phi - d_q_theta;
        const f3 dd_d_phi = dd_p_phi - dd_q_theta;
        const float d_f_phi = dot(d_phi, d_d_phi);
        const float dd_f_phi = dot(d_d_phi, d_d_phi) + dot(dd_d_phi, d_phi);
        const float delta = d_f_phi / dd_f_phi;
        phi -= delta;
        if (fabsf(delta) < kPi * 1.0f / 180.0f) break;
    }
    const float cos_phi = cos_cr(phi), sin_phi = sin_cr(phi);
    rB0 = posB_in_A + (u * cos_phi + v * sin_phi) * radiusB;
    const float theta = atan2_cr(rB0[tA0], rB0[tA1]);
    const float cos_theta = cos_cr(theta), sin_theta = sin_cr(theta);
    rA0 = mk3(0, 0, 0);
    rA0[nA] = 0; rA0[tA0] = sin_theta * radiusA; rA0[tA1] = cos_theta * radiusA;
    rA0 = to_world(rA0, posA, ornA);
    rB0 = to_world(rB0, posA, ornA);
    const f3 dir = rA0 - rB0;
    const float dist_sqr = length_sqr(dir);
    f3 tangentA = mk3(0, 0, 0);
    tangentA[nA] = 0; tangentA[tA0] = cos_theta; tangentA[tA1] = -sin_theta;
    const f3 tangentB = u * -sin_phi + v * cos_phi;
    normal = cross(tangentA, tangentB);
    const float normal_len_sqr = length_sqr(normal);
    if (normal_len_sqr > kEps) {
        normal /= sqrtf(normal_len_sqr);
        normal = rotate(ornA, normal);
    } else if (dist_sqr > kEps) {
        normal = dir / sqrtf(dist_sqr);
    } else {
        normal[nA] = 0; normal[tA0] = sin_theta; normal[tA1] = cos_theta;
        normal = rotate(ornA, normal);
    }
    num_points = 1;
    return dist_sqr;
}

// ---- collide(cylinder, plane)   collide_cylinder_plane.cpp:7-86
DI void collide_cylinder_plane(const CylSh &shA, f3 pn, float pc, const Ctx &ctx, CResult &result) {
    const f3 posA = ctx.posA; const q4 ornA = ctx.ornA;
    const f3 normal = pn, center = normal * pc;
    const float projA = -cylinder_support_projection(shA, posA, ornA, -normal);
    const float distance = projA - pc;
    if (distance > ctx.threshold) return;
    int featureA; int feature_indexA = 0;
    cylinder_support_feature(shA, posA, ornA, -normal, featureA, feature_indexA, kSupportTolerance);
    CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    point.normal = normal; point.distance = distance; point.attachment = NA_ON_B;
    const int ai = shA.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureA == CF_FACE) {
        const float multipliers[4] = {0, 1, 0, -1};
        const float pivotA_axis = shA.half_length * to_sign(feature_indexA == 0);
        for (int i = 0; i < 4; ++i) {
            point.pivotA[ai] = pivotA_axis;
            point.pivotA[o0] = shA.radius * multipliers[i];
            point.pivotA[o1] = shA.radius * multipliers[(i + 1) % 4];
            const f3 pivotA_world = to_world(point.pivotA, posA, ornA);
            point.pivotB = project_plane(pivotA_world, center, normal);
            point.distance = dot(pivotA_world - point.pivotB, normal);
            res_maybe_add(result, point);
        }
    } else {
        const f3 cyl_axis = rotate(ornA, axis_vec(shA.axis));
        f3 cyl_vertices[2]; int num_vertices = 0;
        if (featureA == CF_CAP_EDGE) {
            cyl_vertices[0] = posA + cyl_axis * shA.half_length * to_sign(feature_indexA == 0);
            num_vertices = 1;
        } else {
            cyl_vertices[0] = posA - cyl_axis * shA.half_length;
            cyl_vertices[1] = posA + cyl_axis * shA.half_length;
            num_vertices = 2;
        }
        const f3 dirA = normalize(project_direction(-normal, cyl_axis));
        for (int i = 0; i < num_vertices; ++i) {
            const f3 pivotA_world = cyl_vertices[i] + dirA * shA.radius;
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = project_plane(pivotA_world, center, normal);
            point.distance = dot(pivotA_world - point.pivotB, normal);
            res_maybe_add(result, point);
        }
    }
}

// ---- collide(cylinder, sphere)   collide_cylinder_sphere.cpp:8-86
DI void collide_cylinder_sphere(const CylSh &shA, float radiusB, const Ctx &ctx, CResult &result) {
    const f3 posA = ctx.posA, posB = ctx.posB; const q4 ornA = ctx.ornA, ornB = ctx.ornB;
    const float threshold = ctx.threshold;
    const f3 cyl_axis = rotate(ornA, axis_vec(shA.axis));
    const f3 cyl_vertices[2] = {posA + cyl_axis * shA.half_length, posA - cyl_axis * shA.half_length};
    const f3 v = cyl_vertices[1] - cyl_vertices[0];
    const f3 w = posB - cyl_vertices[0];
    const float denom = dot(v, v);
    const float t = dot(w, v) / denom;
    if (t > 0 && t < 1) {
        const f3 p_cyl = cyl_vertices[0] + v * t;
        const f3 dir = p_cyl - posB;
        const float dist_sqr = length_sqr(dir);
        const float min_dist = shA.radius + radiusB + threshold;
        if (dist_sqr > min_dist * min_dist) return;
        const float dist = sqrtf(dist_sqr);
        const f3 normal = dist_sqr > kEps ? dir / dist : mk3(0, 1, 0);
        CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
        point.pivotA = rotate(conjugate(ornA), p_cyl - normal * shA.radius - posA);
        point.pivotB = rotate(conjugate(ornB), normal * radiusB);
        point.distance = dist - shA.radius - radiusB;
        point.normal = normal;
        point.attachment = NA_NONE;
        res_add(result, point);
        return;
    }
    const int cyl_face_idx = t < 0.5f ? 0 : 1;
    const f3 disc_pos = cyl_vertices[cyl_face_idx];
    f3 closest;
    const float dist_sqr = closest_point_disc(disc_pos, ornA, shA.radius, shA.axis, posB, closest);
    const float min_dist = radiusB + threshold;
    if (dist_sqr > min_dist * min_dist) return;
    f3 normal = closest - posB;
    const float n_len_sqr = length_sqr(normal);
    const float n_len = sqrtf(n_len_sqr);
    normal = n_len_sqr > kEps ? normal / n_len : cyl_axis * to_sign(t > 0.5f);
    CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    point.pivotA = rotate(conjugate(ornA), closest - posA);
    point.pivotB = rotate(conjugate(ornB), normal * radiusB);
    point.distance = n_len - radiusB;
    point.normal = normal;
    const f3 sphere_proj = project_plane(posB, posA, cyl_axis);
    point.attachment = distance_sqr(sphere_proj, posA) < shA.radius * shA.radius ? NA_ON_A : NA_NONE;
    res_add(result, point);
}

// ---- collide(cylinder, cylinder)   collide_cylinder_cylinder.cpp:15-513
DI void collide_cylinder_cylinder(const CylSh &shA, const CylSh &shB, const Ctx &ctx, CResult &result) {
    const f3 posA = ctx.posA, posB = ctx.posB; const q4 ornA = ctx.ornA, ornB = ctx.ornB;
    const f3 axisA = rotate(ornA, axis_vec(shA.axis)), axisB = rotate(ornB, axis_vec(shB.axis));
    const f3 verticesA[2] = {posA + axisA * shA.half_length, posA - axisA * shA.half_length};
    const f3 verticesB[2] = {posB + axisB * shB.half_length, posB - axisB * shB.half_length};
    f3 sep_axis = mk3(0, 0, 0);
    float distance = -kScalarMax;
    {   // A's faces
        f3 dir = axisA;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.half_length);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // B's faces
        f3 dir = axisB;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // axis vs axis
        f3 dir = cross(axisA, axisB);
        if (try_normalize(dir)) {
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -(dot(posA, -dir) + shA.radius);
            const float projB = dot(posB, dir) + shB.radius;
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    for (int i = 0; i < 2; ++i)   // face edges vs the other's side edge
        for (int j = 0; j < 2; ++j) {
            const bool is_circleA = j == 0;
            const f3 circle_pos = is_circleA ? verticesA[i] : verticesB[i];
            int num_points; float s0, s1; f3 closest_circle[2], closest_line[2], dir;
            const q4 orn = is_circleA ? ornA : ornB;
            const float radius = is_circleA ? shA.radius : shB.radius;
            const int axis = is_circleA ? shA.axis : shB.axis;
            const f3 *vertices = is_circleA ? verticesB : verticesA;
            closest_point_circle_line(circle_pos, orn, radius, axis, vertices[0], vertices[1], num_points, s0, closest_circle[0],
                                      closest_line[0], s1, closest_circle[1], closest_line[1], dir, kSupportTolerance);
            if (num_points == 2) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    for (int i = 0; i < 2; ++i)   // face edges vs face edges
        for (int j = 0; j < 2; ++j) {
            int num_points; f3 closestA[2], closestB[2], dir;
            closest_point_circle_circle(verticesA[i], ornA, shA.radius, shA.axis, verticesB[j], ornB, shB.radius, shB.axis, num_points,
                                        closestA[0], closestB[0], closestA[1], closestB[1], dir);
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    if (distance > ctx.threshold) return;
    int featureA, featureB; int feature_indexA = 0, feature_indexB = 0;
    cylinder_support_feature(shA, posA, ornA, -sep_axis, featureA, feature_indexA, kSupportTolerance);
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportTolerance);
    CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    auto get_local_distance = [&](f3 pivotA, f3 pivotB) {
        return dot(to_world(pivotA, posA, ornA) - to_world(pivotB, posB, ornB), sep_axis);
    };
    const int aA = shA.axis, oA0 = (aA + 1) % 3, oA1 = (aA + 2) % 3;
    const int aB = shB.axis, oB0 = (aB + 1) % 3, oB1 = (aB + 2) % 3;
    if (featureA == CF_FACE && featureB == CF_FACE) {
        const f3 posA_in_B = to_object(posA, posB, ornB);
        const q4 ornA_in_B = conjugate(ornB) * ornA;
        point.attachment = NA_ON_B;
        f2 intersection[2];
        const f2 centerA{posA_in_B[oB0], posA_in_B[oB1]};
        int num_points = intersect_circle_circle(centerA, shA.radius, mk2(0, 0), shB.radius, intersection[0], intersection[1]);
        auto from_B_pivot = [&](float bx, float by, float pivotA_axis, float pivotB_axis, bool maybe) {
            point.pivotB[aB] = pivotB_axis; point.pivotB[oB0] = bx; point.pivotB[oB1] = by;
            point.pivotA = to_object(point.pivotB, posA_in_B, ornA_in_B);
            point.pivotA[aA] = pivotA_axis;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            if (maybe) res_maybe_add(result, point); else res_add(result, point);
        };
        if (num_points > 0) {
            const float merge_distance = kBreakingThreshold;
            if (num_points > 1 && distance_sqr(intersection[0], intersection[1]) < merge_distance * merge_distance) {
                num_points = 1;
                intersection[0] = (intersection[0] + intersection[1]) * 0.5f;
            }
            const float pivotA_axis = shA.half_length * to_sign(feature_indexA == 0);
            const float pivotB_axis = shB.half_length * to_sign(feature_indexB == 0);
            for (int i = 0; i < num_points; ++i) from_B_pivot(intersection[i].x, intersection[i].y, pivotA_axis, pivotB_axis, false);
            const float dist_sqr = length_sqr(centerA);
            if (num_points > 1) {
                f2 dir = normalize(orthogonal(intersection[1] - intersection[0]));
                if (dot(dir, centerA) < 0) dir = dir * -1.0f;
                { const f2 extraA = centerA - dir * shA.radius; from_B_pivot(extraA.x, extraA.y, pivotA_axis, pivotB_axis, false); }
                { const f2 extraB = dir * shB.radius; from_B_pivot(extraB.x, extraB.y, pivotA_axis, pivotB_axis, false); }
            } else if (dist_sqr < shB.radius * shB.radius || dist_sqr < shA.radius * shA.radius) {
                f2 dir = normalize(centerA);
                if (shA.radius < shB.radius) { const f2 e = centerA - dir * shA.radius; from_B_pivot(e.x, e.y, pivotA_axis, pivotB_axis, false); }
                else { const f2 e = dir * shB.radius; from_B_pivot(e.x, e.y, pivotA_axis, pivotB_axis, false); }
                dir = orthogonal(dir);
                if (shA.radius < shB.radius) {
                    const f2 e0 = centerA + dir * shA.radius; from_B_pivot(e0.x, e0.y, pivotA_axis, pivotB_axis, false);
                    const f2 e1 = centerA - dir * shA.radius; from_B_pivot(e1.x, e1.y, pivotA_axis, pivotB_axis, false);
                } else {
                    const f2 e0 = dir * shB.radius; from_B_pivot(e0.x, e0.y, pivotA_axis, pivotB_axis, false);
                    const f2 e1 = -dir * shB.radius; from_B_pivot(e1.x, e1.y, pivotA_axis, pivotB_axis, false);
                }
            }
        } else {
            const f3 circle_pointA = posA + quaternion_z(ornA) * shA.radius;
            const f3 circle_pointB = posB + quaternion_z(ornB) * shB.radius;
            const float multipliers[4] = {0, 1, 0, -1};
            if (distance_sqr_line(posA, axisA, circle_pointB) < shA.radius * shA.radius) {
                const f3 posB_in_A = to_object(posB, posA, ornA);
                const q4 ornB_in_A = conjugate(ornA) * ornB;
                for (int i = 0; i < 4; ++i) {
                    point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
                    point.pivotB[oB0] = shB.radius * multipliers[i];
                    point.pivotB[oB1] = shB.radius * multipliers[(i + 1) % 4];
                    point.pivotA = to_world(point.pivotB, posB_in_A, ornB_in_A);
                    point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
                    point.distance = get_local_distance(point.pivotA, point.pivotB);
                    res_maybe_add(result, point);
                }
            } else if (distance_sqr_line(posB, axisB, circle_pointA) < shB.radius * shB.radius) {
                for (int i = 0; i < 4; ++i) {
                    point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
                    point.pivotA[oA0] = shA.radius * multipliers[i];
                    point.pivotA[oA1] = shA.radius * multipliers[(i + 1) % 4];
                    point.pivotB = to_world(point.pivotA, posA_in_B, ornA_in_B);
                    point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
                    point.distance = get_local_distance(point.pivotA, point.pivotB);
                    res_maybe_add(result, point);
                }
            }
        }
    } else if (featureA == CF_FACE && featureB == CF_CAP_EDGE) {
        const f3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        if (!(distance_sqr_line(posA, axisA, supportB) > square(shA.radius))) {
            const f3 pivotA_world = project_plane(supportB, verticesA[feature_indexA], sep_axis);
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = to_object(supportB, posB, ornB);
            point.attachment = NA_ON_A;
            res_maybe_add(result, point);
        }
    } else if (featureA == CF_CAP_EDGE && featureB == CF_FACE) {
        const f3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        if (!(distance_sqr_line(posB, axisB, supportA) > square(shB.radius))) {
            point.pivotA = to_object(supportA, posA, ornA);
            const f3 pivotB_world = project_plane(supportA, verticesB[feature_indexB], sep_axis);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            point.attachment = NA_ON_B;
            res_maybe_add(result, point);
        }
    } else if (featureA == CF_FACE && featureB == CF_SIDE_EDGE) {
        point.attachment = NA_ON_A;
        const f3 v0 = to_object(verticesB[0], posA, ornA), v1 = to_object(verticesB[1], posA, ornA);
        const f2 v0_proj{v0[oA0], v0[oA1]}, v1_proj{v1[oA0], v1[oA1]};
        float s[2];
        const int num_points = intersect_line_circle(v0_proj, v1_proj, shA.radius, s[0], s[1]);
        for (int i = 0; i < num_points; ++i) {
            s[i] = clamp_unit(s[i]);
            point.pivotA = lerp(v0, v1, s[i]);
            point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
            const f3 normalB = rotate(conjugate(ornB), sep_axis);
            point.pivotB = axis_vec(shB.axis) * shB.half_length * (1 - 2 * s[i]) + normalB * shB.radius;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            res_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_FACE) {
        point.attachment = NA_ON_B;
        const f3 v0 = to_object(verticesA[0], posB, ornB), v1 = to_object(verticesA[1], posB, ornB);
        const f2 v0_proj{v0[oB0], v0[oB1]}, v1_proj{v1[oB0], v1[oB1]};
        float s[2];
        const int num_points = intersect_line_circle(v0_proj, v1_proj, shB.radius, s[0], s[1]);
        for (int i = 0; i < num_points; ++i) {
            s[i] = clamp_unit(s[i]);
            point.pivotB = lerp(v0, v1, s[i]);
            point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
            const f3 normalA = rotate(conjugate(ornA), sep_axis);
            point.pivotA = axis_vec(shA.axis) * shA.half_length * (1 - 2 * s[i]) - normalA * shA.radius;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            res_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_SIDE_EDGE) {
        point.attachment = NA_NONE;
        float s[2], t[2]; f3 closestA[2], closestB[2]; int num_points = 0;
        closest_point_segment_segment(verticesA[0], verticesA[1], verticesB[0], verticesB[1], s[0], t[0], closestA[0], closestB[0], &num_points,
                                      &s[1], &t[1], &closestA[1], &closestB[1]);
        for (int i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closestA[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closestB[i] + sep_axis * shB.radius, posB, ornB);
            res_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_CAP_EDGE) {
        const f3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        f3 pivotA; float t;
        closest_point_segment(verticesA[0], verticesA[1], supportB, t, pivotA);
        point.pivotA = to_object(pivotA - sep_axis * shA.radius, posA, ornA);
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = NA_NONE;
        res_add(result, point);
    } else if (featureB == CF_SIDE_EDGE && featureA == CF_CAP_EDGE) {
        const f3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        f3 pivotB; float t;
        closest_point_segment(verticesB[0], verticesB[1], supportA, t, pivotB);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(pivotB + sep_axis * shB.radius, posB, ornB);
        point.attachment = NA_NONE;
        res_add(result, point);
    } else if (featureA == CF_CAP_EDGE && featureB == CF_CAP_EDGE) {
        const f3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        const f3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = NA_NONE;
        res_add(result, point);
    }
}

// ---- collide(cylinder, box)   collide_cylinder_box.cpp:17-427
DI void collide_cylinder_box(const CylSh &shA, f3 hB, const Ctx &ctx, CResult &result) {
    const f3 posA = ctx.posA, posB = ctx.posB; const q4 ornA = ctx.ornA, ornB = ctx.ornB;
    const f3 box_axes[3] = {quaternion_x(ornB), quaternion_y(ornB), quaternion_z(ornB)};
    const f3 cyl_axis = rotate(ornA, axis_vec(shA.axis));
    const f3 cyl_vertices[2] = {posA + cyl_axis * shA.half_length, posA - cyl_axis * shA.half_length};
    f3 sep_axis = mk3(0, 0, 0);
    float distance = -kScalarMax;
    for (int i = 0; i < 3; ++i) {   // box faces
        f3 dir = box_axes[i];
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = dot(posB, dir) + hB[i];
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // cylinder cap faces
        f3 dir = cyl_axis;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.half_length);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 3; ++i) {   // box edges vs cylinder side edges
        f3 dir = cross(box_axes[i], cyl_axis);
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 8; ++i) {   // box vertices vs cylinder side edges
        const f3 vertex = to_world(box_vertex(hB, i), posB, ornB);
        f3 closest; float t;
        closest_point_line(posA, cyl_axis, vertex, t, closest);
        f3 dir = closest - vertex;
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.radius);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 2; ++i) {   // cylinder cap edges vs box edges
        const f3 circle_position = cyl_vertices[i];
        for (int j = 0; j < 12; ++j) {
            f3 edge_vertices[2];
            edge_world(hB, j, posB, ornB, edge_vertices);
            int num_points; float s[2]; f3 closest_circle[2], closest_line[2], dir;
            closest_point_circle_line(circle_position, ornA, shA.radius, shA.axis, edge_vertices[0], edge_vertices[1], num_points, s[0],
                                      closest_circle[0], closest_line[0], s[1], closest_circle[1], closest_line[1], dir, kSupportTolerance);
            if (num_points == 2) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = box_support_projection(hB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    if (distance > ctx.threshold) return;
    int featureA; int feature_indexA = 0;
    cylinder_support_feature(shA, posA, ornA, -sep_axis, featureA, feature_indexA, kSupportTolerance);
    int featureB, fiB; float projB_unused;
    support_feature(hB, posB, ornB, mk3(0, 0, 0), sep_axis, featureB, fiB, projB_unused, kSupportTolerance);
    const int feature_indexB = (int)fiB;
    CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    const int ai = shA.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureA == CF_FACE && featureB == BF_FACE) {
        const float sign_faceA = to_sign(feature_indexA == 0);
        f3 verticesB_local[4], verticesB_world[4];
        for (int i = 0; i < 4; ++i) {
            verticesB_local[i] = box_vertex(hB, kFaceIdx[feature_indexB * 4 + i]);
            verticesB_world[i] = to_world(verticesB_local[i], posB, ornB);
        }
        point.attachment = NA_ON_B;
        int num_edge_intersections = 0;
        f3 last_edge[2] = {{0, 0, 0}, {0, 0, 0}};
        for (int vertex_idx = 0; vertex_idx < 4; ++vertex_idx) {
            const int next_vertex_idx = (vertex_idx + 1) % 4;
            const f3 v0w = verticesB_world[vertex_idx], v1w = verticesB_world[next_vertex_idx];
            const f3 v0A = to_object(v0w, posA, ornA), v1A = to_object(v1w, posA, ornA);
            const f2 v0A_proj{v0A[o0], v0A[o1]}, v1A_proj{v1A[o0], v1A[o1]};
            float s[2];
            const int num_points = intersect_line_circle(v0A_proj, v1A_proj, shA.radius, s[0], s[1]);
            if (num_points == 0) continue;
            if (num_points == 1 && (s[0] < 0 || s[0] > 1)) continue;
            if (num_points == 2 && ((s[0] < 0 && s[1] < 0) || (s[0] > 1 && s[1] > 1))) continue;
            ++num_edge_intersections;
            last_edge[0] = v0w; last_edge[1] = v1w;
            const f3 v0B = verticesB_local[vertex_idx], v1B = verticesB_local[next_vertex_idx];
            const float pivotA_axis = shA.half_length * sign_faceA;
            for (int pt_idx = 0; pt_idx < num_points; ++pt_idx) {
                const float t = s[pt_idx];
                if (!(t < 1)) continue;
                const float u = clamp_unit(t);
                point.pivotA = lerp(v0A, v1A, u);
                point.pivotB = lerp(v0B, v1B, u);
                point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
                point.pivotA[ai] = pivotA_axis;
                res_maybe_add(result, point);
            }
        }
        const f3 posA_in_B = to_object(posA, posB, ornB);
        const q4 ornA_in_B = conjugate(ornB) * ornA;
        const f3 face_normal_local = face_normal((int)feature_indexB);
        if (num_edge_intersections == 0) {
            if (point_in_quad_prism(verticesB_local, face_normal_local, posA_in_B)) {
                const float multipliers[4] = {0, 1, 0, -1};
                for (int i = 0; i < 4; ++i) {
                    const int j = (i + 1) % 4;
                    point.pivotA[ai] = shA.half_length * sign_faceA;
                    point.pivotA[o0] = shA.radius * multipliers[i];
                    point.pivotA[o1] = shA.radius * multipliers[j];
                    const f3 pivotA_in_B = to_world(point.pivotA, posA_in_B, ornA_in_B);
                    point.distance = dot(pivotA_in_B - verticesB_local[0], face_normal_local);
                    point.pivotB = project_plane(pivotA_in_B, verticesB_local[0], face_normal_local);
                    res_maybe_add(result, point);
                }
            }
        } else if (num_edge_intersections == 1) {
            f2 edge_in_A[2];
            for (int i = 0; i < 2; ++i) {
                const f3 l = to_object(last_edge[i], posA, ornA);
                edge_in_A[i] = mk2(l[o0], l[o1]);
            }
            const f2 edge_dir = edge_in_A[1] - edge_in_A[0];
            f2 tangent = normalize(orthogonal(edge_dir));
            const f3 posB_in_A = to_object(posB, posA, ornA);
            const f2 box_face_center{posB_in_A[o0], posB_in_A[o1]};
            if (dot(tangent, box_face_center) < 0) tangent = tangent * -1.0f;
            point.pivotA[ai] = shA.half_length * to_sign(feature_indexA == 0);
            point.pivotA[o0] = tangent.x * shA.radius;
            point.pivotA[o1] = tangent.y * shA.radius;
            const f3 pivotA_in_B = to_world(point.pivotA, posA_in_B, ornA_in_B);
            point.pivotB = project_plane(pivotA_in_B, verticesB_local[0], face_normal_local);
            point.distance = dot(pivotA_in_B - verticesB_local[0], face_normal_local);
            res_maybe_add(result, point);
        }
    } else if (featureA == CF_FACE && featureB == BF_EDGE) {
        const f3 verticesB_local[2] = {box_vertex(hB, kEdgeIdx[feature_indexB * 2]), box_vertex(hB, kEdgeIdx[feature_indexB * 2 + 1])};
        const f3 verticesB_world[2] = {to_world(verticesB_local[0], posB, ornB), to_world(verticesB_local[1], posB, ornB)};
        point.attachment = NA_ON_A;
        const f3 v0A = to_object(verticesB_world[0], posA, ornA), v1A = to_object(verticesB_world[1], posA, ornA);
        const f2 v0A_proj{v0A[o0], v0A[o1]}, v1A_proj{v1A[o0], v1A[o1]};
        float s[2];
        const int num_points = intersect_line_circle(v0A_proj, v1A_proj, shA.radius, s[0], s[1]);
        const float sign_faceA = to_sign(feature_indexA == 0);
        const float pivotA_axis = shA.half_length * sign_faceA;
        for (int pt_idx = 0; pt_idx < num_points; ++pt_idx) {
            const float t = clamp_unit(s[pt_idx]);
            point.pivotA = lerp(v0A, v1A, t);
            point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
            point.pivotA[ai] = pivotA_axis;
            point.pivotB = lerp(verticesB_local[0], verticesB_local[1], t);
            res_maybe_add(result, point);
        }
    } else if (featureA == CF_FACE && featureB == BF_VERTEX) {
        const float sign_faceA = to_sign(feature_indexA == 0);
        point.pivotB = box_vertex(hB, (int)feature_indexB);
        const f3 pivotB_world = to_world(point.pivotB, posB, ornB);
        if (!(distance_sqr_line(posA, cyl_axis, pivotB_world) > square(shA.radius))) {
            const float pivotA_axis = shA.half_length * sign_faceA;
            point.pivotA = to_object(pivotB_world, posA, ornA);
            point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
            point.pivotA[ai] = pivotA_axis;
            point.attachment = NA_ON_A;
            res_maybe_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_FACE) {
        const f3 fnormal = face_normal_world((int)feature_indexB, ornB);
        f3 face_vertices[4];
        face_world(hB, (int)feature_indexB, posB, ornB, face_vertices);
        point.attachment = NA_ON_B;
        const f3 edge_vertices[2] = {cyl_vertices[0] - sep_axis * shA.radius, cyl_vertices[1] - sep_axis * shA.radius};
        const f3 fcenter = face_center(hB, (int)feature_indexB, posB, ornB);
        const m3 fbasis = face_basis((int)feature_indexB, ornB);
        const f2 half_extents = face_half_extents(hB, (int)feature_indexB);
        const f3 e0 = to_object(edge_vertices[0], fcenter, fbasis), e1 = to_object(edge_vertices[1], fcenter, fbasis);
        const f2 p0{e0.x, e0.z}, p1{e1.x, e1.z};
        float s[2];
        const int num_points = intersect_line_aabb(p0, p1, -half_extents, half_extents, s[0], s[1]);
        for (int i = 0; i < num_points; ++i) {
            const float t = clamp_unit(s[i]);
            const f3 edge_pivot = lerp(edge_vertices[0], edge_vertices[1], t);
            point.distance = dot(edge_pivot - face_vertices[0], fnormal);
            const f3 pivot_on_face = edge_pivot - fnormal * point.distance;
            point.pivotA = to_object(edge_pivot, posA, ornA);
            point.pivotB = to_object(pivot_on_face, posB, ornB);
            res_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_EDGE) {
        point.attachment = NA_NONE;
        f3 box_edge[2];
        edge_world(hB, (int)feature_indexB, posB, ornB, box_edge);
        float s[2], t[2]; f3 closestA[2], closestB[2]; int num_points = 0;
        closest_point_segment_segment(cyl_vertices[0], cyl_vertices[1], box_edge[0], box_edge[1], s[0], t[0], closestA[0], closestB[0], &num_points,
                                      &s[1], &t[1], &closestA[1], &closestB[1]);
        for (int i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closestA[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closestB[i], posB, ornB);
            res_add(result, point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_VERTEX) {
        point.pivotB = box_vertex(hB, (int)feature_indexB);
        const f3 pivotB_world = to_world(point.pivotB, posB, ornB);
        f3 closest; float t;
        closest_point_segment(cyl_vertices[0], cyl_vertices[1], pivotB_world, t, closest);
        point.pivotA = to_object(closest - sep_axis * shA.radius, posA, ornA);
        point.attachment = NA_NONE;
        res_add(result, point);
    } else if (featureA == CF_CAP_EDGE) {
        const f3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(supportA - sep_axis * distance, posB, ornB);
        point.attachment = featureB == BF_FACE ? NA_ON_B : NA_NONE;
        res_maybe_add(result, point);
    }
}

// ---- collide(capsule, cylinder)   collide_capsule_cylinder.cpp:10-247
DI void collide_capsule_cylinder(const CylSh &shA, const CylSh &shB, const Ctx &ctx, CResult &result) {
    const f3 posA = mk3(0, 0, 0); const q4 ornA = ctx.ornA;
    const f3 posB = ctx.posB - ctx.posA; const q4 ornB = ctx.ornB;
    f3 capsule_vertices_[2], cylinder_vertices_[2];
    capsule_vertices(shA, posA, ornA, capsule_vertices_);
    cylinder_vertices(shB, posB, ornB, cylinder_vertices_);
    const f3 cap_axis = normalize(capsule_vertices_[1] - capsule_vertices_[0]);
    const f3 cyl_axis = normalize(cylinder_vertices_[1] - cylinder_vertices_[0]);
    float distance = -kScalarMax;
    f3 sep_axis = mk3(0, 0, 0);
    {   // cylinder cap faces
        f3 dir = cyl_axis;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // cylinder edge vs capsule edge
        f3 dir = cross(cyl_axis, cap_axis);
        if (try_normalize(dir)) {
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = dot(posA, dir) - shA.radius;
            const float projB = dot(posB, dir) + shB.radius;
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    for (int k = 0; k < 2; ++k) {   // cylinder edge vs capsule vertices
        const f3 vertex = capsule_vertices_[k];
        f3 closest; float t;
        closest_point_line(posB, cyl_axis, vertex, t, closest);
        f3 dir = vertex - closest;
        if (!try_normalize(dir)) continue;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 2; ++i) {   // cylinder caps vs capsule edge
        float s[2]; int num_points; f3 closest_circle[2], closest_line[2], dir;
        closest_point_circle_line(cylinder_vertices_[i], ornB, shB.radius, shB.axis, capsule_vertices_[0], capsule_vertices_[1], num_points, s[0],
                                  closest_circle[0], closest_line[0], s[1], closest_circle[1], closest_line[1], dir);
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 2; ++i)   // cylinder caps vs capsule vertices
        for (int j = 0; j < 2; ++j) {
            const f3 vertex = capsule_vertices_[j];
            f3 closest;
            closest_point_disc(cylinder_vertices_[i], ornB, shB.radius, shB.axis, vertex, closest);
            f3 dir = closest - vertex;
            if (!try_normalize(dir)) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    if (distance > ctx.threshold) return;
    const float proj_capsule_vertices[2] = {dot(capsule_vertices_[0], sep_axis), dot(capsule_vertices_[1], sep_axis)};
    const bool is_capsule_edge = fabsf(proj_capsule_vertices[0] - proj_capsule_vertices[1]) < kSupportTolerance;
    int featureB; int feature_indexB = 0;
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportTolerance);
    CPoint point{}; point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    if (featureB == CF_FACE) {
        point.attachment = NA_ON_B;
        if (is_capsule_edge) {
            const f3 v0 = to_object(capsule_vertices_[0], posB, ornB), v1 = to_object(capsule_vertices_[1], posB, ornB);
            f2 v0_proj, v1_proj;
            if (shB.axis == 0) { v0_proj = {v0.z, v0.y}; v1_proj = {v1.z, v1.y}; }
            else if (shB.axis == 1) { v0_proj = {v0.z, v0.x}; v1_proj = {v1.z, v1.x}; }
            else { v0_proj = {v0.y, v0.x}; v1_proj = {v1.y, v1.x}; }
            float s[2];
            const int num_points = intersect_line_circle(v0_proj, v1_proj, shB.radius, s[0], s[1]);
            for (int i = 0; i < num_points; ++i) {
                const float t = clamp_unit(s[i]);
                const f3 pivotA_world = lerp(capsule_vertices_[0], capsule_vertices_[1], t) - sep_axis * shA.radius;
                const f3 pivotB_world = project_plane(pivotA_world, cylinder_vertices_[feature_indexB], sep_axis);
                point.pivotA = to_object(pivotA_world, posA, ornA);
                point.pivotB = to_object(pivotB_world, posB, ornB);
                point.distance = dot(pivotA_world - pivotB_world, sep_axis);
                res_add(result, point);
            }
        } else {
            const f3 closest_capsule_vertex = proj_capsule_vertices[0] < proj_capsule_vertices[1] ? capsule_vertices_[0] : capsule_vertices_[1];
            const f3 pivotA_world = closest_capsule_vertex - sep_axis * shA.radius;
            const f3 pivotB_world = project_plane(closest_capsule_vertex, cylinder_vertices_[feature_indexB], sep_axis);
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            res_add(result, point);
        }
    } else if (featureB == CF_SIDE_EDGE) {
        point.attachment = NA_NONE;
        float s[2], t[2]; f3 closest_capsule[2], closest_cylinder[2]; int num_points = 0;
        closest_point_segment_segment(capsule_vertices_[0], capsule_vertices_[1], cylinder_vertices_[0], cylinder_vertices_[1], s[0], t[0],
                                      closest_capsule[0], closest_cylinder[0], &num_points, &s[1], &t[1], &closest_capsule[1], &closest_cylinder[1]);
        for (int i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closest_capsule[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closest_cylinder[i] + sep_axis * shB.radius, posB, ornB);
            res_add(result, point);
        }
    } else {
        point.attachment = NA_NONE;
        const f3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotB = to_object(supportB, posB, ornB);
        point.pivotA = to_object(supportB + sep_axis * distance, posA, ornA);
        res_add(result, point);
    }
}


// Pairs that involve a cylinder, incl. swap_collide (collide.hpp:369-374); other pairs leave the result untouched (returns false).
DI bool collide_ext(int tA, float4 sA, int tB, float4 sB, const Ctx &c, CResult &r) {
    if (tA != SHAPE_CYLINDER && tB != SHAPE_CYLINDER) return false;
    r.num = 0;
    const Ctx sw{c.posB, c.ornB, c.posA, c.ornA, c.threshold};
    bool swapped = false;
    if (tA == SHAPE_CYLINDER && tB == SHAPE_PLANE) collide_cylinder_plane(cyl_of(sA), from4(sB), sB.w, c, r);
    else if (tA == SHAPE_PLANE && tB == SHAPE_CYLINDER) { collide_cylinder_plane(cyl_of(sB), from4(sA), sA.w, sw, r); swapped = true; }
    else if (tA == SHAPE_CYLINDER && tB == SHAPE_SPHERE) collide_cylinder_sphere(cyl_of(sA), sB.x, c, r);
    else if (tA == SHAPE_SPHERE && tB == SHAPE_CYLINDER) { collide_cylinder_sphere(cyl_of(sB), sA.x, sw, r); swapped = true; }
    else if (tA == SHAPE_CYLINDER && tB == SHAPE_CYLINDER) collide_cylinder_cylinder(cyl_of(sA), cyl_of(sB), c, r);
    else if (tA == SHAPE_CYLINDER && tB == SHAPE_BOX) collide_cylinder_box(cyl_of(sA), from4(sB), c, r);
    else if (tA == SHAPE_BOX && tB == SHAPE_CYLINDER) { collide_cylinder_box(cyl_of(sB), from4(sA), sw, r); swapped = true; }
    else if (tA == SHAPE_CAPSULE && tB == SHAPE_CYLINDER) collide_capsule_cylinder(cyl_of(sA), cyl_of(sB), c, r);
    else if (tA == SHAPE_CYLINDER && tB == SHAPE_CAPSULE) { collide_capsule_cylinder(cyl_of(sB), cyl_of(sA), sw, r); swapped = true; }
    if (swapped)
        for (int i = 0; i < r.num; ++i) cp_swap(r.pt[i]);
    return true;
}

}  // namespace dc
