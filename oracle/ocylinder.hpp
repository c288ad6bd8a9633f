// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// cylinder_shape: support functions, AABB, inertia and the closest-feature routines of its pairs, restated from
//   /root/reference/include/edyn/shapes/cylinder_shape.hpp:1-64, src/edyn/shapes/cylinder_shape.cpp:1-59
//   /root/reference/src/edyn/util/shape_util.cpp:307-349 (cylinder_support_point / _projection)
//   /root/reference/src/edyn/util/aabb_util.cpp:72-79 (cylinder_aabb), src/edyn/dynamics/moment_of_inertia.cpp:27-44,167-169
//   /root/reference/src/edyn/math/geom.cpp:24-33 (distance_sqr_line), :172-192 (closest_point_disc), :194-215 (intersect_line_circle),
//       :217-439 (closest_point_circle_line), :441-474 (intersect_circle_circle), :476-728 (closest_point_circle_circle),
//       :772-798 (support_point_circle)
//   /root/reference/src/edyn/collision/collide/collide_cylinder_plane.cpp:7-86, collide_cylinder_sphere.cpp:8-86,
//       collide_cylinder_cylinder.cpp:15-513, collide_cylinder_box.cpp:17-427, collide_capsule_cylinder.cpp:10-247
// The Newton iterations of the circle routines call sin / cos / atan2: evaluated through sin_cr / cos_cr / atan2_cr (correctly
// rounded by default - what the device computes -, the C library's float versions under set_libm_trig, which is what the
// reference engine calls: tests/test_reference_engine.py pins these routines to the engine's own with that switch).
#pragma once
// (included by ocollide.hpp after the capsule routines: needs coll_point / coll_result / coll_ctx, closest_point_line, closest_point_segment)

namespace orc {

enum cyl_feature : int { CF_FACE = 0, CF_SIDE_EDGE = 1, CF_CAP_EDGE = 2 };

inline float atan2_cr(float y, float x) { return g_libm_trig ? std::atan2(y, x) : (float)std::atan2((double)y, (double)x); }
inline float to_sign(bool b) { return b ? 1.0f : -1.0f; }
// vector2.hpp
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
inline vec2 operator/(vec2 a, float s) { return {a.x / s, a.y / s}; }
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float length_sqr(vec2 a) { return dot(a, a); }
inline float length(vec2 a) { return std::sqrt(length_sqr(a)); }
inline float distance_sqr(vec2 a, vec2 b) { return length_sqr(a - b); }
inline vec2 orthogonal(vec2 v) { return {-v.y, v.x}; }
inline vec2 normalize(vec2 v) { return v / length(v); }
inline vec3 project_direction(vec3 v, vec3 n) { return v - n * dot(v, n); }   // math/vector3.hpp project_direction
inline quat quat_mul(quat a, quat b) { return a * b; }
inline vec3 quaternion_x(quat q) { return rotate(q, vec3{1, 0, 0}); }
inline vec3 quaternion_y(quat q) { return rotate(q, vec3{0, 1, 0}); }
inline vec3 quaternion_z(quat q) { return rotate(q, vec3{0, 0, 1}); }

// ---- shape functions
inline vec3 cylinder_support_point_local(float radius, float half_length, int axis, vec3 dir) {   // shape_util.cpp:307-330
    const int ai = axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    const float planar_len_sq = dir[o0] * dir[o0] + dir[o1] * dir[o1];
    vec3 sup{0, 0, 0};
    sup[ai] = dir[ai] < 0 ? -half_length : half_length;
    if (planar_len_sq > kEps) {
        const float d = radius / std::sqrt(planar_len_sq);
        sup[o0] = dir[o0] * d;
        sup[o1] = dir[o1] * d;
    } else {
        sup[o0] = radius;
        sup[o1] = 0;
    }
    return sup;
}
inline vec3 cylinder_support_point(float radius, float half_length, int axis, quat orn, vec3 dir) {   // :332-337
    const vec3 local_dir = rotate(conjugate(orn), dir);
    return rotate(orn, cylinder_support_point_local(radius, half_length, axis, local_dir));
}
inline vec3 cylinder_support_point(const shape &s, vec3 pos, quat orn, vec3 dir) {   // :339-342
    return pos + cylinder_support_point(s.radius, s.half_length, s.axis, orn, dir);
}
inline float cylinder_support_projection(const shape &s, vec3 pos, quat orn, vec3 dir) {   // :344-349
    const vec3 local_dir = rotate(conjugate(orn), dir);
    const vec3 pt = cylinder_support_point_local(s.radius, s.half_length, s.axis, local_dir);
    return dot(pos, dir) + dot(pt, local_dir);
}
inline void cylinder_support_feature_local(const shape &s, vec3 dir, int &feature, size_t &index, float threshold) {   // cylinder_shape.cpp:15-50
    const int ai = s.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    const float ortho_dir_len_sqr = dir[o0] * dir[o0] + dir[o1] * dir[o1];
    const float proj_cap_face_sqr = 4.0f * s.radius * s.radius * ortho_dir_len_sqr;
    if (proj_cap_face_sqr < threshold * threshold) {
        feature = CF_FACE;
        index = dir[ai] > 0 ? 0 : 1;
        return;
    }
    const float proj_side_edge = std::fabs(2.0f * s.half_length * dir[ai]);
    if (proj_side_edge < threshold) {
        feature = CF_SIDE_EDGE;
        return;
    }
    feature = CF_CAP_EDGE;
    index = dir[ai] > 0 ? 0 : 1;
}
inline void cylinder_support_feature(const shape &s, vec3 /*pos*/, quat orn, vec3 axis_dir, int &feature, size_t &index, float threshold) {   // :52-57
    cylinder_support_feature_local(s, rotate(conjugate(orn), axis_dir), feature, index, threshold);
}
inline void cylinder_vertices(const shape &s, vec3 pos, quat orn, vec3 out[2]) {   // cylinder_shape.hpp:33-39
    const vec3 dir = rotate(orn, coordinate_axis_vector(s.axis));
    out[0] = pos + dir * s.half_length;
    out[1] = pos - dir * s.half_length;
}
inline aabb cylinder_aabb(const shape &s, vec3 pos, quat orn) {   // aabb_util.cpp:72-79
    const vec3 ptx = cylinder_support_point(s.radius, s.half_length, s.axis, orn, vec3{1, 0, 0});
    const vec3 pty = cylinder_support_point(s.radius, s.half_length, s.axis, orn, vec3{0, 1, 0});
    const vec3 ptz = cylinder_support_point(s.radius, s.half_length, s.axis, orn, vec3{0, 0, 1});
    const vec3 v{ptx.x, pty.y, ptz.z};
    return {pos - v, pos + v};
}
inline mat3 cylinder_inertia(const shape &s, float mass) {   // moment_of_inertia.cpp:27-44,167-169
    const float len = s.half_length * 2, radius = s.radius;
    const float xx = 0.5f * mass * radius * radius;
    const float yy_zz = 1.0f / 12.0f * mass * (3.0f * radius * radius + len * len);
    return diagonal(s.axis == 0 ? vec3{xx, yy_zz, yy_zz} : (s.axis == 1 ? vec3{yy_zz, xx, yy_zz} : vec3{yy_zz, yy_zz, xx}));
}

// ---- geometry
inline float distance_sqr_line(vec3 q0, vec3 dir, vec3 p) {   // geom.cpp:24-33
    const vec3 w = p - q0;
    const float a = dot(w, dir), b = dot(dir, dir);
    const float t = a / b;
    const vec3 q = q0 + dir * t;
    return length_sqr(p - q);
}
inline float closest_point_disc(vec3 dpos, quat dorn, float radius, int axis, vec3 p, vec3 &q) {   // :172-192
    const vec3 normal = rotate(dorn, coordinate_axis_vector(axis));
    const float ln = dot(p - dpos, normal);
    const vec3 p_proj = p - normal * ln;
    const vec3 d = p_proj - dpos;
    const float l2 = length_sqr(d);
    if (l2 < radius * radius) {
        q = p_proj;
        return ln * ln;
    }
    const float l = std::sqrt(l2);
    const vec3 dn = d / l;
    q = dpos + dn * radius;
    return length_sqr(p - q);
}
inline size_t intersect_line_circle(vec2 p0, vec2 p1, float radius, float &s0, float &s1) {   // :194-215
    const vec2 d = p1 - p0;
    const float dl2 = length_sqr(d);
    const float dp = dot(d, p0);
    const float delta = dp * dp - dl2 * (dot(p0, p0) - radius * radius);
    if (delta < 0) return 0;
    if (delta > kEps) {
        const float delta_sqrt = std::sqrt(delta);
        const float dl2_inv = 1 / dl2;
        s0 = -(dp + delta_sqrt) * dl2_inv;
        s1 = -(dp - delta_sqrt) * dl2_inv;
        return 2;
    }
    s0 = -dp * dl2;
    return 1;
}
inline vec3 support_point_circle(vec3 pos, quat orn, float radius, int axis, vec3 dir) {   // :772-798
    const int ni = axis, t0 = (ni + 1) % 3, t1 = (ni + 2) % 3;
    const vec3 local_dir = rotate(conjugate(orn), dir);
    const float len_plane_sqr = local_dir[t0] * local_dir[t0] + local_dir[t1] * local_dir[t1];
    vec3 sup{0, 0, 0};
    if (len_plane_sqr > kEps) {
        const float d = radius / std::sqrt(len_plane_sqr);
        sup[ni] = 0; sup[t0] = local_dir[t0] * d; sup[t1] = local_dir[t1] * d;
    } else {
        sup[ni] = 0; sup[t0] = radius; sup[t1] = 0;
    }
    return pos + rotate(orn, sup);
}
inline float closest_point_circle_line(vec3 cpos, quat corn, float radius, int axis, vec3 p0, vec3 p1, size_t &num_points,
                                       float &s0, vec3 &rc0, vec3 &rl0, float &s1, vec3 &rc1, vec3 &rl1, vec3 &normal,
                                       float threshold = kSupportFeatureTolerance) {   // :217-439
    const vec3 q0 = to_object(p0, cpos, corn), q1 = to_object(p1, cpos, corn);
    const vec3 qv = q1 - q0;
    const float qv_len_sqr = length_sqr(qv);
    const int ni = axis, t0 = (ni + 1) % 3, t1 = (ni + 2) % 3;
    const float qv_proj_len = length(vec2{qv[t0], qv[t1]});
    const float diameter = square(radius);
    const vec2 q0_proj{q0[t0], q0[t1]}, q1_proj{q1[t0], q1[t1]};
    if (qv_proj_len > kEps && std::fabs(qv[ni] / qv_proj_len) * diameter < threshold) {
        const vec3 tangent = cross(qv, coordinate_axis_vector(axis));
        normal = cross(qv, tangent);
        normal = rotate(corn, normal);
        normal = normalize(normal);
        num_points = intersect_line_circle(q0_proj, q1_proj, radius, s0, s1);
        if (num_points > 0) {
            const vec3 rl0_local = q0 + qv * s0;
            vec3 rc0_local = rl0_local;
            rc0_local[ni] = 0;
            rl0 = cpos + rotate(corn, rl0_local);
            rc0 = cpos + rotate(corn, rc0_local);
            float dist2 = square(rl0_local[ni]);
            if (num_points > 1) {
                const vec3 rl1_local = q0 + qv * s1;
                vec3 rc1_local = rl1_local;
                rc1_local[ni] = 0;
                rl1 = cpos + rotate(corn, rl1_local);
                rc1 = cpos + rotate(corn, rc1_local);
                dist2 = std::min(dist2, square(rl1_local[ni]));
            }
            return dist2;
        } else {
            closest_point_line(p0, p1 - p0, cpos, s0, rl0);
            const vec3 proj = project_plane(rl0, cpos, normal);
            const vec3 dir = normalize(proj - cpos);
            rc0 = cpos + dir * radius;
            const vec3 d = rl0 - rc0;
            const float dl2 = length_sqr(d);
            if (dl2 > kEps) normal = d / std::sqrt(dl2);
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
    const vec3 q_plane = q0 - (q0[ni] / qv[ni]) * qv;
    const float initial_theta = atan2_cr(q_plane[t0], q_plane[t1]);
    const float qv_len_sqr_inv = 1.0f / qv_len_sqr;
    float theta = initial_theta;
    for (size_t i = 0; i < 20; ++i) {
        const float sin_theta = sin_cr(theta), cos_theta = cos_cr(theta);
        vec3 q_theta{0, 0, 0}, d_q_theta{0, 0, 0}, dd_q_theta{0, 0, 0};
        q_theta[ni] = 0; q_theta[t0] = sin_theta * radius; q_theta[t1] = cos_theta * radius;
        d_q_theta[ni] = 0; d_q_theta[t0] = cos_theta * radius; d_q_theta[t1] = -sin_theta * radius;
        dd_q_theta[ni] = 0; dd_q_theta[t0] = -sin_theta * radius; dd_q_theta[t1] = -cos_theta * radius;
        const vec3 c_theta = q0 + dot(q_theta - q0, qv) * qv_len_sqr_inv * qv;
        const vec3 d_c_theta = dot(d_q_theta, qv) * qv_len_sqr_inv * qv;
        const vec3 dd_c_theta = dot(dd_q_theta, qv) * qv_len_sqr_inv * qv;
        const vec3 d_theta = q_theta - c_theta;
        const vec3 d_d_theta = d_q_theta - d_c_theta;
        const vec3 dd_d_theta = dd_q_theta - dd_c_theta;
        const float d_f_theta = dot(d_theta, d_d_theta);
        const float dd_f_theta = dot(d_d_theta, d_d_theta) + dot(dd_d_theta, d_theta);
        const float delta = d_f_theta / dd_f_theta;
        theta -= delta;
        if (std::fabs(delta) < kPi * 1.0f / 180.0f) break;
    }
    const float closest_sin_theta = sin_cr(theta), closest_cos_theta = cos_cr(theta);
    vec3 rc0_local{0, 0, 0};
    rc0_local[ni] = 0; rc0_local[t0] = closest_sin_theta * radius; rc0_local[t1] = closest_cos_theta * radius;
    vec3 rl0_local;
    const float dist_sqr = closest_point_line(q0, qv, rc0_local, s0, rl0_local);
    rc0 = cpos + rotate(corn, rc0_local);
    rl0 = cpos + rotate(corn, rl0_local);
    vec3 tangent{0, 0, 0};
    tangent[ni] = 0; tangent[t0] = closest_cos_theta; tangent[t1] = -closest_sin_theta;
    normal = cross(tangent, qv);
    const float normal_len_sqr = length_sqr(normal);
    if (normal_len_sqr > kEps) {
        normal /= std::sqrt(normal_len_sqr);
        normal = rotate(corn, normal);
    } else if (dist_sqr > kEps) {
        normal = (rl0 - rc0) / std::sqrt(dist_sqr);
    } else {
        normal[ni] = 0; normal[t0] = closest_sin_theta; normal[t1] = closest_cos_theta;
        normal = rotate(corn, normal);
    }
    num_points = 1;
    return dist_sqr;
}
inline size_t intersect_circle_circle(vec2 posA, float radiusA, vec2 posB, float radiusB, vec2 &res0, vec2 &res1) {   // :441-474
    const vec2 u = posB - posA;
    const float lu2 = length_sqr(u);
    const float rsum = radiusA + radiusB, rsub = radiusA - radiusB;
    if (lu2 < kEps && rsub < kEps) {
        res0 = posA + vec2{1, 0} * radiusA;
        res1 = posB - vec2{1, 0} * radiusB;
        return 2;
    }
    if (lu2 < rsub * rsub || lu2 > rsum * rsum) return 0;
    const float lu2_inv = 1.0f / lu2;
    const float s = ((radiusA * radiusA - radiusB * radiusB) * lu2_inv + 1.0f) * 0.5f;
    const float t = std::sqrt(std::max(0.0f, radiusA * radiusA * lu2_inv - s * s));
    const vec2 v = orthogonal(u);
    const vec2 su = s * u, tv = t * v;
    res0 = posA + su + tv;
    res1 = posA + su - tv;
    return t > kEps ? 2 : 1;
}
inline float closest_point_circle_circle(vec3 posA, quat ornA, float radiusA, int axisA, vec3 posB, quat ornB, float radiusB, int axisB,
                                         size_t &num_points, vec3 &rA0, vec3 &rB0, vec3 &rA1, vec3 &rB1, vec3 &normal) {   // :476-728
    const vec3 normalA = rotate(ornA, coordinate_axis_vector(axisA)), normalB = rotate(ornB, coordinate_axis_vector(axisB));
    const int nA = axisA, tA0 = (nA + 1) % 3, tA1 = (nA + 2) % 3;
    const int nB = axisB, tB0 = (nB + 1) % 3, tB1 = (nB + 2) % 3;
    const vec3 posB_in_A = to_object(posB, posA, ornA);
    if (!(length_sqr(cross(normalA, normalB)) > kEps)) {   // parallel
        normal = normalB;
        const vec2 posB_in_A_proj{posB_in_A[tA0], posB_in_A[tA1]};
        vec2 c0, c1;
        const size_t np = intersect_circle_circle(vec2{0, 0}, radiusA, posB_in_A_proj, radiusB, c0, c1);
        if (np > 0) {
            num_points = np;
            vec3 rA0_local{0, 0, 0};
            rA0_local[nA] = 0; rA0_local[tA0] = c0.x; rA0_local[tA1] = c0.y;
            vec3 rB0_local = rA0_local;
            rB0_local[nA] = posB_in_A[nA];
            rA0 = to_world(rA0_local, posA, ornA);
            rB0 = to_world(rB0_local, posA, ornA);
            if (np > 1) {
                vec3 rA1_local{0, 0, 0};
                rA1_local[nA] = 0; rA1_local[tA0] = c1.x; rA1_local[tA1] = c1.y;
                vec3 rB1_local = rA1_local;
                rB1_local[nA] = posB_in_A[nA];
                rA1 = to_world(rA1_local, posA, ornA);
                rB1 = to_world(rB1_local, posA, ornA);
            }
            return square(posB_in_A[nA]);
        } else {
            num_points = 1;
            vec2 dir = posB_in_A_proj;
            const float dir_len_sqr = length_sqr(dir);
            vec3 tanA{0, 0, 0};
            tanA[tA0] = 1;
            if (dir_len_sqr > kEps) {
                { const float z = 1.0f / std::sqrt(dir_len_sqr); dir.x *= z; dir.y *= z; }   // vector2 operator/=
                const vec3 pointA = tanA * radiusA;
                const vec3 pointB_in_A = posB_in_A + tanA * radiusB;
                const bool A_contains_B = length_sqr(vec2{pointB_in_A[tA0], pointB_in_A[tA1]}) < radiusA * radiusA;
                const bool B_contains_A = distance_sqr(vec2{pointA[tA0], pointA[tA1]}, posB_in_A_proj) < radiusB * radiusB;
                vec3 dirA{0, 0, 0}, dirB{0, 0, 0};
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
    const quat ornB_in_A = conjugate(ornA) * ornB;
    vec3 u, v;
    if (axisA == 0) { u = quaternion_z(ornB_in_A); v = quaternion_y(ornB_in_A); }
    else if (axisA == 1) { u = quaternion_x(ornB_in_A); v = quaternion_z(ornB_in_A); }
    else { u = quaternion_y(ornB_in_A); v = quaternion_x(ornB_in_A); }
    const vec3 sup_pos = support_point_circle(posB_in_A, ornB_in_A, radiusB, axisB, coordinate_axis_vector(axisA));
    const vec3 sup_neg = support_point_circle(posB_in_A, ornB_in_A, radiusB, axisB, -coordinate_axis_vector(axisA));
    const vec3 sup = std::fabs(sup_pos[nA]) < std::fabs(sup_neg[nA]) ? sup_pos : sup_neg;
    const vec3 sup_in_B = to_object(sup, posB_in_A, ornB_in_A);
    const float initial_phi = atan2_cr(sup_in_B[tA0], sup_in_B[tA1]);
    float phi = initial_phi;
    for (size_t i = 0; i < 20; ++i) {
        const float cos_phi = cos_cr(phi), sin_phi = sin_cr(phi);
        const vec3 p_phi = posB_in_A + (u * cos_phi + v * sin_phi) * radiusB;
        const vec3 d_p_phi = (u * -sin_phi + v * cos_phi) * radiusB;
        const vec3 dd_p_phi = (u * -cos_phi + v * -sin_phi) * radiusB;
        const float theta = atan2_cr(p_phi[tA0], p_phi[tA1]);
        const float cos_theta = cos_cr(theta), sin_theta = sin_cr(theta);
        vec3 q_theta{0, 0, 0}, d_q_theta{0, 0, 0}, dd_q_theta{0, 0, 0};
        q_theta[nA] = 0; q_theta[tA0] = sin_theta * radiusA; q_theta[tA1] = cos_theta * radiusA;
        d_q_theta[nA] = 0; d_q_theta[tA0] = cos_theta * radiusA; d_q_theta[tA1] = -sin_theta * radiusA;
        dd_q_theta[nA] = 0; dd_q_theta[tA0] = -sin_theta * radiusA; dd_q_theta[tA1] = -cos_theta * radiusA;
        const vec3 d_phi = p_phi - q_theta;
        const vec3 d_d_phi = d_p_phi - d_q_theta;
        const vec3 dd_d_phi = dd_p_phi - dd_q_theta;
        const float d_f_phi = dot(d_phi, d_d_phi);
        const float dd_f_phi = dot(d_d_phi, d_d_phi) + dot(dd_d_phi, d_phi);
        const float delta = d_f_phi / dd_f_phi;
        phi -= delta;
        if (std::fabs(delta) < kPi * 1.0f / 180.0f) break;
    }
    const float cos_phi = cos_cr(phi), sin_phi = sin_cr(phi);
    rB0 = posB_in_A + (u * cos_phi + v * sin_phi) * radiusB;
    const float theta = atan2_cr(rB0[tA0], rB0[tA1]);
    const float cos_theta = cos_cr(theta), sin_theta = sin_cr(theta);
    rA0 = vec3{0, 0, 0};
    rA0[nA] = 0; rA0[tA0] = sin_theta * radiusA; rA0[tA1] = cos_theta * radiusA;
    rA0 = to_world(rA0, posA, ornA);
    rB0 = to_world(rB0, posA, ornA);
    const vec3 dir = rA0 - rB0;
    const float dist_sqr = length_sqr(dir);
    vec3 tangentA{0, 0, 0};
    tangentA[nA] = 0; tangentA[tA0] = cos_theta; tangentA[tA1] = -sin_theta;
    const vec3 tangentB = u * -sin_phi + v * cos_phi;
    normal = cross(tangentA, tangentB);
    const float normal_len_sqr = length_sqr(normal);
    if (normal_len_sqr > kEps) {
        normal /= std::sqrt(normal_len_sqr);
        normal = rotate(ornA, normal);
    } else if (dist_sqr > kEps) {
        normal = dir / std::sqrt(dist_sqr);
    } else {
        normal[nA] = 0; normal[tA0] = sin_theta; normal[tA1] = cos_theta;
        normal = rotate(ornA, normal);
    }
    num_points = 1;
    return dist_sqr;
}

// ---- collide(cylinder, plane)   collide_cylinder_plane.cpp:7-86
inline void collide_cylinder_plane(const shape &shA, vec3 pn, float pc, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA; const quat ornA = ctx.ornA;
    const vec3 normal = pn, center = normal * pc;
    const float projA = -cylinder_support_projection(shA, posA, ornA, -normal);
    const float distance = projA - pc;
    if (distance > ctx.threshold) return;
    int featureA; size_t feature_indexA = 0;
    cylinder_support_feature(shA, posA, ornA, -normal, featureA, feature_indexA, kSupportFeatureTolerance);
    coll_point point{};
    point.normal = normal; point.distance = distance; point.attachment = NA_ON_B;
    const int ai = shA.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureA == CF_FACE) {
        const float multipliers[4] = {0, 1, 0, -1};
        const float pivotA_axis = shA.half_length * to_sign(feature_indexA == 0);
        for (int i = 0; i < 4; ++i) {
            point.pivotA[ai] = pivotA_axis;
            point.pivotA[o0] = shA.radius * multipliers[i];
            point.pivotA[o1] = shA.radius * multipliers[(i + 1) % 4];
            const vec3 pivotA_world = to_world(point.pivotA, posA, ornA);
            point.pivotB = project_plane(pivotA_world, center, normal);
            point.distance = dot(pivotA_world - point.pivotB, normal);
            result.maybe_add_point(point);
        }
    } else {
        const vec3 cyl_axis = rotate(ornA, coordinate_axis_vector(shA.axis));
        vec3 cyl_vertices[2]; int num_vertices = 0;
        if (featureA == CF_CAP_EDGE) {
            cyl_vertices[0] = posA + cyl_axis * shA.half_length * to_sign(feature_indexA == 0);
            num_vertices = 1;
        } else {
            cyl_vertices[0] = posA - cyl_axis * shA.half_length;
            cyl_vertices[1] = posA + cyl_axis * shA.half_length;
            num_vertices = 2;
        }
        const vec3 dirA = normalize(project_direction(-normal, cyl_axis));
        for (int i = 0; i < num_vertices; ++i) {
            const vec3 pivotA_world = cyl_vertices[i] + dirA * shA.radius;
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = project_plane(pivotA_world, center, normal);
            point.distance = dot(pivotA_world - point.pivotB, normal);
            result.maybe_add_point(point);
        }
    }
}

// ---- collide(cylinder, sphere)   collide_cylinder_sphere.cpp:8-86
inline void collide_cylinder_sphere(const shape &shA, float radiusB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA, posB = ctx.posB; const quat ornA = ctx.ornA, ornB = ctx.ornB;
    const float threshold = ctx.threshold;
    const vec3 cyl_axis = rotate(ornA, coordinate_axis_vector(shA.axis));
    const vec3 cyl_vertices[2] = {posA + cyl_axis * shA.half_length, posA - cyl_axis * shA.half_length};
    const vec3 v = cyl_vertices[1] - cyl_vertices[0];
    const vec3 w = posB - cyl_vertices[0];
    const float denom = dot(v, v);
    const float t = dot(w, v) / denom;
    if (t > 0 && t < 1) {
        const vec3 p_cyl = cyl_vertices[0] + v * t;
        const vec3 dir = p_cyl - posB;
        const float dist_sqr = length_sqr(dir);
        const float min_dist = shA.radius + radiusB + threshold;
        if (dist_sqr > min_dist * min_dist) return;
        const float dist = std::sqrt(dist_sqr);
        const vec3 normal = dist_sqr > kEps ? dir / dist : vec3{0, 1, 0};
        coll_point point{};
        point.pivotA = rotate(conjugate(ornA), p_cyl - normal * shA.radius - posA);
        point.pivotB = rotate(conjugate(ornB), normal * radiusB);
        point.distance = dist - shA.radius - radiusB;
        point.normal = normal;
        point.attachment = NA_NONE;
        result.add_point(point);
        return;
    }
    const size_t cyl_face_idx = t < 0.5f ? 0 : 1;
    const vec3 disc_pos = cyl_vertices[cyl_face_idx];
    vec3 closest;
    const float dist_sqr = closest_point_disc(disc_pos, ornA, shA.radius, shA.axis, posB, closest);
    const float min_dist = radiusB + threshold;
    if (dist_sqr > min_dist * min_dist) return;
    vec3 normal = closest - posB;
    const float n_len_sqr = length_sqr(normal);
    const float n_len = std::sqrt(n_len_sqr);
    normal = n_len_sqr > kEps ? normal / n_len : cyl_axis * to_sign(t > 0.5f);
    coll_point point{};
    point.pivotA = rotate(conjugate(ornA), closest - posA);
    point.pivotB = rotate(conjugate(ornB), normal * radiusB);
    point.distance = n_len - radiusB;
    point.normal = normal;
    const vec3 sphere_proj = project_plane(posB, posA, cyl_axis);
    point.attachment = distance_sqr(sphere_proj, posA) < shA.radius * shA.radius ? NA_ON_A : NA_NONE;
    result.add_point(point);
}

// ---- collide(cylinder, cylinder)   collide_cylinder_cylinder.cpp:15-513
inline void collide_cylinder_cylinder(const shape &shA, const shape &shB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA, posB = ctx.posB; const quat ornA = ctx.ornA, ornB = ctx.ornB;
    const vec3 axisA = rotate(ornA, coordinate_axis_vector(shA.axis)), axisB = rotate(ornB, coordinate_axis_vector(shB.axis));
    const vec3 verticesA[2] = {posA + axisA * shA.half_length, posA - axisA * shA.half_length};
    const vec3 verticesB[2] = {posB + axisB * shB.half_length, posB - axisB * shB.half_length};
    vec3 sep_axis{0, 0, 0};
    float distance = -kScalarMax;
    {   // A's faces
        vec3 dir = axisA;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.half_length);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // B's faces
        vec3 dir = axisB;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // axis vs axis
        vec3 dir = cross(axisA, axisB);
        if (try_normalize(dir)) {
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -(dot(posA, -dir) + shA.radius);
            const float projB = dot(posB, dir) + shB.radius;
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    for (size_t i = 0; i < 2; ++i)   // face edges vs the other's side edge
        for (size_t j = 0; j < 2; ++j) {
            const bool is_circleA = j == 0;
            const vec3 circle_pos = is_circleA ? verticesA[i] : verticesB[i];
            size_t num_points; float s0, s1; vec3 closest_circle[2], closest_line[2], dir;
            const quat orn = is_circleA ? ornA : ornB;
            const float radius = is_circleA ? shA.radius : shB.radius;
            const int axis = is_circleA ? shA.axis : shB.axis;
            const vec3 *vertices = is_circleA ? verticesB : verticesA;
            closest_point_circle_line(circle_pos, orn, radius, axis, vertices[0], vertices[1], num_points, s0, closest_circle[0],
                                      closest_line[0], s1, closest_circle[1], closest_line[1], dir, kSupportFeatureTolerance);
            if (num_points == 2) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    for (size_t i = 0; i < 2; ++i)   // face edges vs face edges
        for (size_t j = 0; j < 2; ++j) {
            size_t num_points; vec3 closestA[2], closestB[2], dir;
            closest_point_circle_circle(verticesA[i], ornA, shA.radius, shA.axis, verticesB[j], ornB, shB.radius, shB.axis, num_points,
                                        closestA[0], closestB[0], closestA[1], closestB[1], dir);
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    if (distance > ctx.threshold) return;
    int featureA, featureB; size_t feature_indexA = 0, feature_indexB = 0;
    cylinder_support_feature(shA, posA, ornA, -sep_axis, featureA, feature_indexA, kSupportFeatureTolerance);
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportFeatureTolerance);
    coll_point point{};
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    auto get_local_distance = [&](vec3 pivotA, vec3 pivotB) {
        return dot(to_world(pivotA, posA, ornA) - to_world(pivotB, posB, ornB), sep_axis);
    };
    const int aA = shA.axis, oA0 = (aA + 1) % 3, oA1 = (aA + 2) % 3;
    const int aB = shB.axis, oB0 = (aB + 1) % 3, oB1 = (aB + 2) % 3;
    if (featureA == CF_FACE && featureB == CF_FACE) {
        const vec3 posA_in_B = to_object(posA, posB, ornB);
        const quat ornA_in_B = conjugate(ornB) * ornA;
        point.attachment = NA_ON_B;
        vec2 intersection[2];
        const vec2 centerA{posA_in_B[oB0], posA_in_B[oB1]};
        size_t num_points = intersect_circle_circle(centerA, shA.radius, vec2{0, 0}, shB.radius, intersection[0], intersection[1]);
        auto from_B_pivot = [&](float bx, float by, float pivotA_axis, float pivotB_axis, bool maybe) {
            point.pivotB[aB] = pivotB_axis; point.pivotB[oB0] = bx; point.pivotB[oB1] = by;
            point.pivotA = to_object(point.pivotB, posA_in_B, ornA_in_B);
            point.pivotA[aA] = pivotA_axis;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            if (maybe) result.maybe_add_point(point); else result.add_point(point);
        };
        if (num_points > 0) {
            const float merge_distance = kContactBreakingThreshold;
            if (num_points > 1 && distance_sqr(intersection[0], intersection[1]) < merge_distance * merge_distance) {
                num_points = 1;
                intersection[0] = (intersection[0] + intersection[1]) * 0.5f;
            }
            const float pivotA_axis = shA.half_length * to_sign(feature_indexA == 0);
            const float pivotB_axis = shB.half_length * to_sign(feature_indexB == 0);
            for (size_t i = 0; i < num_points; ++i) from_B_pivot(intersection[i].x, intersection[i].y, pivotA_axis, pivotB_axis, false);
            const float dist_sqr = length_sqr(centerA);
            if (num_points > 1) {
                vec2 dir = normalize(orthogonal(intersection[1] - intersection[0]));
                if (dot(dir, centerA) < 0) dir = dir * -1.0f;
                { const vec2 extraA = centerA - dir * shA.radius; from_B_pivot(extraA.x, extraA.y, pivotA_axis, pivotB_axis, false); }
                { const vec2 extraB = dir * shB.radius; from_B_pivot(extraB.x, extraB.y, pivotA_axis, pivotB_axis, false); }
            } else if (dist_sqr < shB.radius * shB.radius || dist_sqr < shA.radius * shA.radius) {
                vec2 dir = normalize(centerA);
                if (shA.radius < shB.radius) { const vec2 e = centerA - dir * shA.radius; from_B_pivot(e.x, e.y, pivotA_axis, pivotB_axis, false); }
                else { const vec2 e = dir * shB.radius; from_B_pivot(e.x, e.y, pivotA_axis, pivotB_axis, false); }
                dir = orthogonal(dir);
                if (shA.radius < shB.radius) {
                    const vec2 e0 = centerA + dir * shA.radius; from_B_pivot(e0.x, e0.y, pivotA_axis, pivotB_axis, false);
                    const vec2 e1 = centerA - dir * shA.radius; from_B_pivot(e1.x, e1.y, pivotA_axis, pivotB_axis, false);
                } else {
                    const vec2 e0 = dir * shB.radius; from_B_pivot(e0.x, e0.y, pivotA_axis, pivotB_axis, false);
                    const vec2 e1 = -dir * shB.radius; from_B_pivot(e1.x, e1.y, pivotA_axis, pivotB_axis, false);
                }
            }
        } else {
            const vec3 circle_pointA = posA + quaternion_z(ornA) * shA.radius;
            const vec3 circle_pointB = posB + quaternion_z(ornB) * shB.radius;
            const float multipliers[4] = {0, 1, 0, -1};
            if (distance_sqr_line(posA, axisA, circle_pointB) < shA.radius * shA.radius) {
                const vec3 posB_in_A = to_object(posB, posA, ornA);
                const quat ornB_in_A = conjugate(ornA) * ornB;
                for (size_t i = 0; i < 4; ++i) {
                    point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
                    point.pivotB[oB0] = shB.radius * multipliers[i];
                    point.pivotB[oB1] = shB.radius * multipliers[(i + 1) % 4];
                    point.pivotA = to_world(point.pivotB, posB_in_A, ornB_in_A);
                    point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
                    point.distance = get_local_distance(point.pivotA, point.pivotB);
                    result.maybe_add_point(point);
                }
            } else if (distance_sqr_line(posB, axisB, circle_pointA) < shB.radius * shB.radius) {
                for (size_t i = 0; i < 4; ++i) {
                    point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
                    point.pivotA[oA0] = shA.radius * multipliers[i];
                    point.pivotA[oA1] = shA.radius * multipliers[(i + 1) % 4];
                    point.pivotB = to_world(point.pivotA, posA_in_B, ornA_in_B);
                    point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
                    point.distance = get_local_distance(point.pivotA, point.pivotB);
                    result.maybe_add_point(point);
                }
            }
        }
    } else if (featureA == CF_FACE && featureB == CF_CAP_EDGE) {
        const vec3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        if (!(distance_sqr_line(posA, axisA, supportB) > square(shA.radius))) {
            const vec3 pivotA_world = project_plane(supportB, verticesA[feature_indexA], sep_axis);
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = to_object(supportB, posB, ornB);
            point.attachment = NA_ON_A;
            result.maybe_add_point(point);
        }
    } else if (featureA == CF_CAP_EDGE && featureB == CF_FACE) {
        const vec3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        if (!(distance_sqr_line(posB, axisB, supportA) > square(shB.radius))) {
            point.pivotA = to_object(supportA, posA, ornA);
            const vec3 pivotB_world = project_plane(supportA, verticesB[feature_indexB], sep_axis);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            point.attachment = NA_ON_B;
            result.maybe_add_point(point);
        }
    } else if (featureA == CF_FACE && featureB == CF_SIDE_EDGE) {
        point.attachment = NA_ON_A;
        const vec3 v0 = to_object(verticesB[0], posA, ornA), v1 = to_object(verticesB[1], posA, ornA);
        const vec2 v0_proj{v0[oA0], v0[oA1]}, v1_proj{v1[oA0], v1[oA1]};
        float s[2];
        const size_t num_points = intersect_line_circle(v0_proj, v1_proj, shA.radius, s[0], s[1]);
        for (size_t i = 0; i < num_points; ++i) {
            s[i] = clamp_unit(s[i]);
            point.pivotA = lerp(v0, v1, s[i]);
            point.pivotA[aA] = shA.half_length * to_sign(feature_indexA == 0);
            const vec3 normalB = rotate(conjugate(ornB), sep_axis);
            point.pivotB = coordinate_axis_vector(shB.axis) * shB.half_length * (1 - 2 * s[i]) + normalB * shB.radius;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            result.add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_FACE) {
        point.attachment = NA_ON_B;
        const vec3 v0 = to_object(verticesA[0], posB, ornB), v1 = to_object(verticesA[1], posB, ornB);
        const vec2 v0_proj{v0[oB0], v0[oB1]}, v1_proj{v1[oB0], v1[oB1]};
        float s[2];
        const size_t num_points = intersect_line_circle(v0_proj, v1_proj, shB.radius, s[0], s[1]);
        for (size_t i = 0; i < num_points; ++i) {
            s[i] = clamp_unit(s[i]);
            point.pivotB = lerp(v0, v1, s[i]);
            point.pivotB[aB] = shB.half_length * to_sign(feature_indexB == 0);
            const vec3 normalA = rotate(conjugate(ornA), sep_axis);
            point.pivotA = coordinate_axis_vector(shA.axis) * shA.half_length * (1 - 2 * s[i]) - normalA * shA.radius;
            point.distance = get_local_distance(point.pivotA, point.pivotB);
            result.add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_SIDE_EDGE) {
        point.attachment = NA_NONE;
        float s[2], t[2]; vec3 closestA[2], closestB[2]; size_t num_points = 0;
        closest_point_segment_segment(verticesA[0], verticesA[1], verticesB[0], verticesB[1], s[0], t[0], closestA[0], closestB[0], &num_points,
                                      &s[1], &t[1], &closestA[1], &closestB[1]);
        for (size_t i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closestA[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closestB[i] + sep_axis * shB.radius, posB, ornB);
            result.add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == CF_CAP_EDGE) {
        const vec3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        vec3 pivotA; float t;
        closest_point_segment(verticesA[0], verticesA[1], supportB, t, pivotA);
        point.pivotA = to_object(pivotA - sep_axis * shA.radius, posA, ornA);
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = NA_NONE;
        result.add_point(point);
    } else if (featureB == CF_SIDE_EDGE && featureA == CF_CAP_EDGE) {
        const vec3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        vec3 pivotB; float t;
        closest_point_segment(verticesB[0], verticesB[1], supportA, t, pivotB);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(pivotB + sep_axis * shB.radius, posB, ornB);
        point.attachment = NA_NONE;
        result.add_point(point);
    } else if (featureA == CF_CAP_EDGE && featureB == CF_CAP_EDGE) {
        const vec3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        const vec3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = NA_NONE;
        result.add_point(point);
    }
}

// ---- collide(cylinder, box)   collide_cylinder_box.cpp:17-427
inline void collide_cylinder_box(const shape &shA, vec3 hB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA, posB = ctx.posB; const quat ornA = ctx.ornA, ornB = ctx.ornB;
    const vec3 box_axes[3] = {quaternion_x(ornB), quaternion_y(ornB), quaternion_z(ornB)};
    const vec3 cyl_axis = rotate(ornA, coordinate_axis_vector(shA.axis));
    const vec3 cyl_vertices[2] = {posA + cyl_axis * shA.half_length, posA - cyl_axis * shA.half_length};
    vec3 sep_axis{0, 0, 0};
    float distance = -kScalarMax;
    for (size_t i = 0; i < 3; ++i) {   // box faces
        vec3 dir = box_axes[i];
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = dot(posB, dir) + hB[i];
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // cylinder cap faces
        vec3 dir = cyl_axis;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.half_length);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (size_t i = 0; i < 3; ++i) {   // box edges vs cylinder side edges
        vec3 dir = cross(box_axes[i], cyl_axis);
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 8; ++i) {   // box vertices vs cylinder side edges
        const vec3 vertex = to_world(box_vertex(hB, i), posB, ornB);
        vec3 closest; float t;
        closest_point_line(posA, cyl_axis, vertex, t, closest);
        vec3 dir = closest - vertex;
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -(dot(posA, -dir) + shA.radius);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (size_t i = 0; i < 2; ++i) {   // cylinder cap edges vs box edges
        const vec3 circle_position = cyl_vertices[i];
        for (int j = 0; j < 12; ++j) {
            vec3 edge_vertices[2];
            box_edge_world(hB, j, posB, ornB, edge_vertices);
            size_t num_points; float s[2]; vec3 closest_circle[2], closest_line[2], dir;
            closest_point_circle_line(circle_position, ornA, shA.radius, shA.axis, edge_vertices[0], edge_vertices[1], num_points, s[0],
                                      closest_circle[0], closest_line[0], s[1], closest_circle[1], closest_line[1], dir, kSupportFeatureTolerance);
            if (num_points == 2) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -cylinder_support_projection(shA, posA, ornA, -dir);
            const float projB = box_support_projection(hB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    if (distance > ctx.threshold) return;
    int featureA; size_t feature_indexA = 0;
    cylinder_support_feature(shA, posA, ornA, -sep_axis, featureA, feature_indexA, kSupportFeatureTolerance);
    int featureB, fiB; float projB_unused;
    box_support_feature(hB, posB, ornB, vec3{0, 0, 0}, sep_axis, featureB, fiB, projB_unused, kSupportFeatureTolerance);
    const size_t feature_indexB = (size_t)fiB;
    coll_point point{};
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    const int ai = shA.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureA == CF_FACE && featureB == BF_FACE) {
        const float sign_faceA = to_sign(feature_indexA == 0);
        vec3 verticesB_local[4], verticesB_world[4];
        for (int i = 0; i < 4; ++i) {
            verticesB_local[i] = box_vertex(hB, kBoxFaceIndices[feature_indexB * 4 + i]);
            verticesB_world[i] = to_world(verticesB_local[i], posB, ornB);
        }
        point.attachment = NA_ON_B;
        size_t num_edge_intersections = 0;
        vec3 last_edge[2] = {{0, 0, 0}, {0, 0, 0}};
        for (size_t vertex_idx = 0; vertex_idx < 4; ++vertex_idx) {
            const size_t next_vertex_idx = (vertex_idx + 1) % 4;
            const vec3 v0w = verticesB_world[vertex_idx], v1w = verticesB_world[next_vertex_idx];
            const vec3 v0A = to_object(v0w, posA, ornA), v1A = to_object(v1w, posA, ornA);
            const vec2 v0A_proj{v0A[o0], v0A[o1]}, v1A_proj{v1A[o0], v1A[o1]};
            float s[2];
            const size_t num_points = intersect_line_circle(v0A_proj, v1A_proj, shA.radius, s[0], s[1]);
            if (num_points == 0) continue;
            if (num_points == 1 && (s[0] < 0 || s[0] > 1)) continue;
            if (num_points == 2 && ((s[0] < 0 && s[1] < 0) || (s[0] > 1 && s[1] > 1))) continue;
            ++num_edge_intersections;
            last_edge[0] = v0w; last_edge[1] = v1w;
            const vec3 v0B = verticesB_local[vertex_idx], v1B = verticesB_local[next_vertex_idx];
            const float pivotA_axis = shA.half_length * sign_faceA;
            for (size_t pt_idx = 0; pt_idx < num_points; ++pt_idx) {
                const float t = s[pt_idx];
                if (!(t < 1)) continue;
                const float u = clamp_unit(t);
                point.pivotA = lerp(v0A, v1A, u);
                point.pivotB = lerp(v0B, v1B, u);
                point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
                point.pivotA[ai] = pivotA_axis;
                result.maybe_add_point(point);
            }
        }
        const vec3 posA_in_B = to_object(posA, posB, ornB);
        const quat ornA_in_B = conjugate(ornB) * ornA;
        const vec3 face_normal_local = box_face_normal((int)feature_indexB);
        if (num_edge_intersections == 0) {
            if (point_in_quad_prism(verticesB_local, face_normal_local, posA_in_B)) {
                const float multipliers[4] = {0, 1, 0, -1};
                for (int i = 0; i < 4; ++i) {
                    const int j = (i + 1) % 4;
                    point.pivotA[ai] = shA.half_length * sign_faceA;
                    point.pivotA[o0] = shA.radius * multipliers[i];
                    point.pivotA[o1] = shA.radius * multipliers[j];
                    const vec3 pivotA_in_B = to_world(point.pivotA, posA_in_B, ornA_in_B);
                    point.distance = dot(pivotA_in_B - verticesB_local[0], face_normal_local);
                    point.pivotB = project_plane(pivotA_in_B, verticesB_local[0], face_normal_local);
                    result.maybe_add_point(point);
                }
            }
        } else if (num_edge_intersections == 1) {
            vec2 edge_in_A[2];
            for (size_t i = 0; i < 2; ++i) {
                const vec3 l = to_object(last_edge[i], posA, ornA);
                edge_in_A[i] = vec2{l[o0], l[o1]};
            }
            const vec2 edge_dir = edge_in_A[1] - edge_in_A[0];
            vec2 tangent = normalize(orthogonal(edge_dir));
            const vec3 posB_in_A = to_object(posB, posA, ornA);
            const vec2 box_face_center{posB_in_A[o0], posB_in_A[o1]};
            if (dot(tangent, box_face_center) < 0) tangent = tangent * -1.0f;
            point.pivotA[ai] = shA.half_length * to_sign(feature_indexA == 0);
            point.pivotA[o0] = tangent.x * shA.radius;
            point.pivotA[o1] = tangent.y * shA.radius;
            const vec3 pivotA_in_B = to_world(point.pivotA, posA_in_B, ornA_in_B);
            point.pivotB = project_plane(pivotA_in_B, verticesB_local[0], face_normal_local);
            point.distance = dot(pivotA_in_B - verticesB_local[0], face_normal_local);
            result.maybe_add_point(point);
        }
    } else if (featureA == CF_FACE && featureB == BF_EDGE) {
        const vec3 verticesB_local[2] = {box_vertex(hB, kBoxEdgeIndices[feature_indexB * 2]), box_vertex(hB, kBoxEdgeIndices[feature_indexB * 2 + 1])};
        const vec3 verticesB_world[2] = {to_world(verticesB_local[0], posB, ornB), to_world(verticesB_local[1], posB, ornB)};
        point.attachment = NA_ON_A;
        const vec3 v0A = to_object(verticesB_world[0], posA, ornA), v1A = to_object(verticesB_world[1], posA, ornA);
        const vec2 v0A_proj{v0A[o0], v0A[o1]}, v1A_proj{v1A[o0], v1A[o1]};
        float s[2];
        const size_t num_points = intersect_line_circle(v0A_proj, v1A_proj, shA.radius, s[0], s[1]);
        const float sign_faceA = to_sign(feature_indexA == 0);
        const float pivotA_axis = shA.half_length * sign_faceA;
        for (size_t pt_idx = 0; pt_idx < num_points; ++pt_idx) {
            const float t = clamp_unit(s[pt_idx]);
            point.pivotA = lerp(v0A, v1A, t);
            point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
            point.pivotA[ai] = pivotA_axis;
            point.pivotB = lerp(verticesB_local[0], verticesB_local[1], t);
            result.maybe_add_point(point);
        }
    } else if (featureA == CF_FACE && featureB == BF_VERTEX) {
        const float sign_faceA = to_sign(feature_indexA == 0);
        point.pivotB = box_vertex(hB, (int)feature_indexB);
        const vec3 pivotB_world = to_world(point.pivotB, posB, ornB);
        if (!(distance_sqr_line(posA, cyl_axis, pivotB_world) > square(shA.radius))) {
            const float pivotA_axis = shA.half_length * sign_faceA;
            point.pivotA = to_object(pivotB_world, posA, ornA);
            point.distance = (point.pivotA[ai] - pivotA_axis) * sign_faceA;
            point.pivotA[ai] = pivotA_axis;
            point.attachment = NA_ON_A;
            result.maybe_add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_FACE) {
        const vec3 face_normal = box_face_normal_world((int)feature_indexB, ornB);
        vec3 face_vertices[4];
        box_face_world(hB, (int)feature_indexB, posB, ornB, face_vertices);
        point.attachment = NA_ON_B;
        const vec3 edge_vertices[2] = {cyl_vertices[0] - sep_axis * shA.radius, cyl_vertices[1] - sep_axis * shA.radius};
        const vec3 face_center = box_face_center(hB, (int)feature_indexB, posB, ornB);
        const mat3 face_basis = box_face_basis((int)feature_indexB, ornB);
        const vec2 half_extents = box_face_half_extents(hB, (int)feature_indexB);
        const vec3 e0 = to_object(edge_vertices[0], face_center, face_basis), e1 = to_object(edge_vertices[1], face_center, face_basis);
        const vec2 p0{e0.x, e0.z}, p1{e1.x, e1.z};
        float s[2];
        const size_t num_points = intersect_line_aabb(p0, p1, -half_extents, half_extents, s[0], s[1]);
        for (size_t i = 0; i < num_points; ++i) {
            const float t = clamp_unit(s[i]);
            const vec3 edge_pivot = lerp(edge_vertices[0], edge_vertices[1], t);
            point.distance = dot(edge_pivot - face_vertices[0], face_normal);
            const vec3 pivot_on_face = edge_pivot - face_normal * point.distance;
            point.pivotA = to_object(edge_pivot, posA, ornA);
            point.pivotB = to_object(pivot_on_face, posB, ornB);
            result.add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_EDGE) {
        point.attachment = NA_NONE;
        vec3 box_edge[2];
        box_edge_world(hB, (int)feature_indexB, posB, ornB, box_edge);
        float s[2], t[2]; vec3 closestA[2], closestB[2]; size_t num_points = 0;
        closest_point_segment_segment(cyl_vertices[0], cyl_vertices[1], box_edge[0], box_edge[1], s[0], t[0], closestA[0], closestB[0], &num_points,
                                      &s[1], &t[1], &closestA[1], &closestB[1]);
        for (size_t i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closestA[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closestB[i], posB, ornB);
            result.add_point(point);
        }
    } else if (featureA == CF_SIDE_EDGE && featureB == BF_VERTEX) {
        point.pivotB = box_vertex(hB, (int)feature_indexB);
        const vec3 pivotB_world = to_world(point.pivotB, posB, ornB);
        vec3 closest; float t;
        closest_point_segment(cyl_vertices[0], cyl_vertices[1], pivotB_world, t, closest);
        point.pivotA = to_object(closest - sep_axis * shA.radius, posA, ornA);
        point.attachment = NA_NONE;
        result.add_point(point);
    } else if (featureA == CF_CAP_EDGE) {
        const vec3 supportA = cylinder_support_point(shA, posA, ornA, -sep_axis);
        point.pivotA = to_object(supportA, posA, ornA);
        point.pivotB = to_object(supportA - sep_axis * distance, posB, ornB);
        point.attachment = featureB == BF_FACE ? NA_ON_B : NA_NONE;
        result.maybe_add_point(point);
    }
}

// ---- collide(capsule, cylinder)   collide_capsule_cylinder.cpp:10-247
inline void collide_capsule_cylinder(const shape &shA, const shape &shB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA{0, 0, 0}; const quat ornA = ctx.ornA;
    const vec3 posB = ctx.posB - ctx.posA; const quat ornB = ctx.ornB;
    vec3 capsule_vertices_[2], cylinder_vertices_[2];
    capsule_vertices(shA, posA, ornA, capsule_vertices_);
    cylinder_vertices(shB, posB, ornB, cylinder_vertices_);
    const vec3 cap_axis = normalize(capsule_vertices_[1] - capsule_vertices_[0]);
    const vec3 cyl_axis = normalize(cylinder_vertices_[1] - cylinder_vertices_[0]);
    float distance = -kScalarMax;
    vec3 sep_axis{0, 0, 0};
    {   // cylinder cap faces
        vec3 dir = cyl_axis;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    {   // cylinder edge vs capsule edge
        vec3 dir = cross(cyl_axis, cap_axis);
        if (try_normalize(dir)) {
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = dot(posA, dir) - shA.radius;
            const float projB = dot(posB, dir) + shB.radius;
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    }
    for (int k = 0; k < 2; ++k) {   // cylinder edge vs capsule vertices
        const vec3 vertex = capsule_vertices_[k];
        vec3 closest; float t;
        closest_point_line(posB, cyl_axis, vertex, t, closest);
        vec3 dir = vertex - closest;
        if (!try_normalize(dir)) continue;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (size_t i = 0; i < 2; ++i) {   // cylinder caps vs capsule edge
        float s[2]; size_t num_points; vec3 closest_circle[2], closest_line[2], dir;
        closest_point_circle_line(cylinder_vertices_[i], ornB, shB.radius, shB.axis, capsule_vertices_[0], capsule_vertices_[1], num_points, s[0],
                                  closest_circle[0], closest_line[0], s[1], closest_circle[1], closest_line[1], dir);
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (size_t i = 0; i < 2; ++i)   // cylinder caps vs capsule vertices
        for (size_t j = 0; j < 2; ++j) {
            const vec3 vertex = capsule_vertices_[j];
            vec3 closest;
            closest_point_disc(cylinder_vertices_[i], ornB, shB.radius, shB.axis, vertex, closest);
            vec3 dir = closest - vertex;
            if (!try_normalize(dir)) continue;
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            const float projA = -capsule_support_projection(capsule_vertices_, shA.radius, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    if (distance > ctx.threshold) return;
    const float proj_capsule_vertices[2] = {dot(capsule_vertices_[0], sep_axis), dot(capsule_vertices_[1], sep_axis)};
    const bool is_capsule_edge = std::fabs(proj_capsule_vertices[0] - proj_capsule_vertices[1]) < kSupportFeatureTolerance;
    int featureB; size_t feature_indexB = 0;
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportFeatureTolerance);
    coll_point point{};
    point.normal = sep_axis; point.distance = distance; point.attachment = NA_NONE;
    if (featureB == CF_FACE) {
        point.attachment = NA_ON_B;
        if (is_capsule_edge) {
            const vec3 v0 = to_object(capsule_vertices_[0], posB, ornB), v1 = to_object(capsule_vertices_[1], posB, ornB);
            vec2 v0_proj, v1_proj;
            if (shB.axis == 0) { v0_proj = {v0.z, v0.y}; v1_proj = {v1.z, v1.y}; }
            else if (shB.axis == 1) { v0_proj = {v0.z, v0.x}; v1_proj = {v1.z, v1.x}; }
            else { v0_proj = {v0.y, v0.x}; v1_proj = {v1.y, v1.x}; }
            float s[2];
            const size_t num_points = intersect_line_circle(v0_proj, v1_proj, shB.radius, s[0], s[1]);
            for (size_t i = 0; i < num_points; ++i) {
                const float t = clamp_unit(s[i]);
                const vec3 pivotA_world = lerp(capsule_vertices_[0], capsule_vertices_[1], t) - sep_axis * shA.radius;
                const vec3 pivotB_world = project_plane(pivotA_world, cylinder_vertices_[feature_indexB], sep_axis);
                point.pivotA = to_object(pivotA_world, posA, ornA);
                point.pivotB = to_object(pivotB_world, posB, ornB);
                point.distance = dot(pivotA_world - pivotB_world, sep_axis);
                result.add_point(point);
            }
        } else {
            const vec3 closest_capsule_vertex = proj_capsule_vertices[0] < proj_capsule_vertices[1] ? capsule_vertices_[0] : capsule_vertices_[1];
            const vec3 pivotA_world = closest_capsule_vertex - sep_axis * shA.radius;
            const vec3 pivotB_world = project_plane(closest_capsule_vertex, cylinder_vertices_[feature_indexB], sep_axis);
            point.pivotA = to_object(pivotA_world, posA, ornA);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            result.add_point(point);
        }
    } else if (featureB == CF_SIDE_EDGE) {
        point.attachment = NA_NONE;
        float s[2], t[2]; vec3 closest_capsule[2], closest_cylinder[2]; size_t num_points = 0;
        closest_point_segment_segment(capsule_vertices_[0], capsule_vertices_[1], cylinder_vertices_[0], cylinder_vertices_[1], s[0], t[0],
                                      closest_capsule[0], closest_cylinder[0], &num_points, &s[1], &t[1], &closest_capsule[1], &closest_cylinder[1]);
        for (size_t i = 0; i < num_points; ++i) {
            point.pivotA = to_object(closest_capsule[i] - sep_axis * shA.radius, posA, ornA);
            point.pivotB = to_object(closest_cylinder[i] + sep_axis * shB.radius, posB, ornB);
            result.add_point(point);
        }
    } else {
        point.attachment = NA_NONE;
        const vec3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotB = to_object(supportB, posB, ornB);
        point.pivotA = to_object(supportB + sep_axis * distance, posA, ornA);
        result.add_point(point);
    }
}

}  // namespace orc
