// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// Closest-feature collision routines for the shape pairs on the hot path, restated from
//   /root/reference/src/edyn/collision/collide/collide_box_box.cpp:14-266
//   /root/reference/src/edyn/collision/collide/collide_box_plane.cpp:7-56
//   /root/reference/src/edyn/collision/collide/collide_sphere_sphere.cpp:5-27
//   /root/reference/src/edyn/collision/collide/collide_sphere_plane.cpp:5-20
//   /root/reference/src/edyn/collision/collide/collide_sphere_box.cpp:7-55
//   /root/reference/include/edyn/collision/collide.hpp:369-374 (swap_collide)
//   /root/reference/src/edyn/collision/collision_result.cpp:6-33 (add_point / maybe_add_point)
// Collision features (featureA/B) are not carried: they only feed per-vertex mesh materials, which
// are outside the hot-path scope.
#pragma once
#include "oshapes.hpp"

namespace orc {

enum normal_attachment : int { NA_NONE = 0, NA_ON_A = 1, NA_ON_B = 2 };

struct coll_point {
    vec3 pivotA, pivotB, normal;
    float distance;
    int attachment;
    void swap() {   // collision_result.hpp:23-35
        std::swap(pivotA, pivotB);
        normal *= -1.0f;
        if (attachment == NA_ON_A) attachment = NA_ON_B;
        else if (attachment == NA_ON_B) attachment = NA_ON_A;
    }
};

struct coll_result {
    size_t num_points = 0;
    coll_point point[kMaxContacts];
    void add_point(const coll_point &p) { point[num_points++] = p; }
    void maybe_add_point(const coll_point &np) {
        vec3 piv[kMaxContacts];
        for (size_t i = 0; i < num_points; ++i) piv[i] = point[i].pivotA;
        insert_result res = insertion_point_index(piv, kMaxContacts, num_points, np.pivotA);
        if (res.type == insert_type::none) {
            for (size_t i = 0; i < num_points; ++i) piv[i] = point[i].pivotB;
            res = insertion_point_index(piv, kMaxContacts, num_points, np.pivotB);
        }
        if (res.type != insert_type::none) point[res.index] = np;
    }
    void swap() { for (size_t i = 0; i < num_points; ++i) point[i].swap(); }
};

struct coll_ctx {
    vec3 posA; quat ornA;
    vec3 posB; quat ornB;
    float threshold;
    coll_ctx swapped() const { return {posB, ornB, posA, ornA, threshold}; }
};

inline void collide_box_box(vec3 hA, vec3 hB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA, posB = ctx.posB;
    const quat ornA = ctx.ornA, ornB = ctx.ornB;
    const float threshold = ctx.threshold;
    vec3 axesA[3] = {rotate(ornA, {1, 0, 0}), rotate(ornA, {0, 1, 0}), rotate(ornA, {0, 0, 1})};
    vec3 axesB[3] = {rotate(ornB, {1, 0, 0}), rotate(ornB, {0, 1, 0}), rotate(ornB, {0, 0, 1})};
    float distance = -kScalarMax;
    vec3 sep_axis{0, 0, 0};

    for (int i = 0; i < 3; ++i) {            // A's faces
        vec3 dir = axesA[i];
        if (dot(posA - posB, dir) < 0) dir = -dir;
        float projA = dot(posA, dir) - hA[i];
        float projB = box_support_projection(hB, posB, ornB, dir);
        float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 3; ++i) {            // B's faces
        vec3 dir = axesB[i];
        if (dot(posA - posB, dir) < 0) dir = -dir;
        float projA = -box_support_projection(hA, posA, ornA, -dir);
        float projB = dot(posB, dir) + hB[i];
        float dist = projA - projB;
        if (dist > distance) { distance = dist; sep_axis = dir; }
    }
    for (int i = 0; i < 3; ++i)              // edge x edge
        for (int j = 0; j < 3; ++j) {
            vec3 dir = cross(axesA[i], axesB[j]);
            float l2 = length_sqr(dir);
            if (!(l2 > kEps)) continue;
            dir /= std::sqrt(l2);
            if (dot(posA - posB, dir) < 0) dir *= -1.0f;
            float projA = -box_support_projection(hA, posA, ornA, -dir);
            float projB = box_support_projection(hB, posB, ornB, dir);
            float dist = projA - projB;
            if (dist > distance) { distance = dist; sep_axis = dir; }
        }
    if (distance > threshold) return;

    int featA, featB, idxA, idxB;
    float projA, projB;
    box_support_feature(hA, posA, ornA, {0, 0, 0}, -sep_axis, featA, idxA, projA, kSupportFeatureTolerance);
    box_support_feature(hB, posB, ornB, {0, 0, 0}, sep_axis, featB, idxB, projB, kSupportFeatureTolerance);

    coll_point point{};
    point.normal = sep_axis;
    point.distance = distance;
    point.attachment = NA_NONE;

    if (featA == BF_FACE && featB == BF_FACE) {
        vec3 fvA[4], fvB[4];
        box_face_world(hA, idxA, posA, ornA, fvA);
        vec3 fnA = box_face_normal_world(idxA, ornA);
        box_face_world(hB, idxB, posB, ornB, fvB);
        vec3 fnB = box_face_normal_world(idxB, ornB);
        point.attachment = NA_ON_B;
        for (int i = 0; i < 4; ++i)
            if (point_in_quad_prism(fvA, fnA, fvB[i])) {
                vec3 pf = project_plane(fvB[i], fvA[0], fnA);
                point.pivotA = to_object(pf, posA, ornA);
                point.pivotB = to_object(fvB[i], posB, ornB);
                result.maybe_add_point(point);
            }
        for (int i = 0; i < 4; ++i)
            if (point_in_quad_prism(fvB, fnB, fvA[i])) {
                vec3 pf = project_plane(fvA[i], fvB[0], fnB);
                point.pivotA = to_object(fvA[i], posA, ornA);
                point.pivotB = to_object(pf, posB, ornB);
                result.maybe_add_point(point);
            }
        if (result.num_points < 4) {
            vec3 fc = box_face_center(hA, idxA, posA, ornA);
            mat3 fb = box_face_basis(idxA, ornA);
            vec2 he = box_face_half_extents(hA, idxA);
            for (int j = 0; j < 4; ++j) {
                vec3 b0w = fvB[j], b1w = fvB[(j + 1) % 4];
                vec3 b0 = to_object(b0w, fc, fb), b1 = to_object(b1w, fc, fb);
                vec2 p0{b0.x, b0.z}, p1{b1.x, b1.z};
                float s[2];
                size_t n = intersect_line_aabb(p0, p1, -he, he, s[0], s[1]);
                for (size_t k = 0; k < n; ++k) {
                    if (s[k] < 0 || s[k] > 1) continue;
                    vec3 q1 = lerp(b0w, b1w, s[k]);
                    vec3 q0 = project_plane(q1, fc, fnA);
                    point.pivotA = to_object(q0, posA, ornA);
                    point.pivotB = to_object(q1, posB, ornB);
                    result.maybe_add_point(point);
                }
            }
        }
    } else if ((featA == BF_FACE && featB == BF_EDGE) || (featB == BF_FACE && featA == BF_EDGE)) {
        const bool faceA = featA == BF_FACE;
        vec3 fn = faceA ? box_face_normal_world(idxA, ornA) : box_face_normal_world(idxB, ornB);
        vec3 fv[4], ev[2];
        if (faceA) { box_face_world(hA, idxA, posA, ornA, fv); box_edge_world(hB, idxB, posB, ornB, ev); }
        else { box_face_world(hB, idxB, posB, ornB, fv); box_edge_world(hA, idxA, posA, ornA, ev); }
        point.attachment = faceA ? NA_ON_A : NA_ON_B;
        for (int i = 0; i < 2; ++i)
            if (point_in_quad_prism(fv, fn, ev[i])) {
                vec3 pf = project_plane(ev[i], fv[0], fn);
                point.pivotA = faceA ? to_object(pf, posA, ornA) : to_object(ev[i], posA, ornA);
                point.pivotB = faceA ? to_object(ev[i], posB, ornB) : to_object(pf, posB, ornB);
                result.add_point(point);
            }
        if (result.num_points < 2) {
            vec3 fc = faceA ? box_face_center(hA, idxA, posA, ornA) : box_face_center(hB, idxB, posB, ornB);
            mat3 fb = faceA ? box_face_basis(idxA, ornA) : box_face_basis(idxB, ornB);
            vec2 he = faceA ? box_face_half_extents(hA, idxA) : box_face_half_extents(hB, idxB);
            vec3 e0 = to_object(ev[0], fc, fb), e1 = to_object(ev[1], fc, fb);
            vec2 p0{e0.x, e0.z}, p1{e1.x, e1.z};
            float s[2];
            size_t n = intersect_line_aabb(p0, p1, -he, he, s[0], s[1]);
            for (size_t i = 0; i < n; ++i) {
                if (s[i] < 0 || s[i] > 1) continue;
                vec3 ep = lerp(ev[0], ev[1], s[i]);
                vec3 fp = project_plane(ep, fc, sep_axis);
                point.pivotA = to_object(faceA ? fp : ep, posA, ornA);
                point.pivotB = to_object(faceA ? ep : fp, posB, ornB);
                result.add_point(point);
            }
        }
    } else if (featA == BF_EDGE && featB == BF_EDGE) {
        float s[2], t[2];
        vec3 p0[2], p1[2];
        size_t n = 0;
        vec3 eA[2], eB[2];
        box_edge_world(hA, idxA, posA, ornA, eA);
        box_edge_world(hB, idxB, posB, ornB, eB);
        closest_point_segment_segment(eA[0], eA[1], eB[0], eB[1], s[0], t[0], p0[0], p1[0], &n, &s[1], &t[1],
                                      &p0[1], &p1[1]);
        point.attachment = NA_NONE;
        for (size_t i = 0; i < n; ++i) {
            point.pivotA = to_object(p0[i], posA, ornA);
            point.pivotB = to_object(p1[i], posB, ornB);
            result.add_point(point);
        }
    } else if (featA == BF_FACE && featB == BF_VERTEX) {
        point.pivotB = box_vertex(hB, idxB);
        point.pivotA = to_world(point.pivotB, posB, ornB) + sep_axis * distance;
        point.pivotA = to_object(point.pivotA, posA, ornA);
        point.attachment = NA_ON_A;
        result.add_point(point);
    } else if (featB == BF_FACE && featA == BF_VERTEX) {
        point.pivotA = box_vertex(hA, idxA);
        point.pivotB = to_world(point.pivotA, posA, ornA) - sep_axis * distance;
        point.pivotB = to_object(point.pivotB, posB, ornB);
        point.attachment = NA_ON_B;
        result.add_point(point);
    }
}

inline void collide_box_plane(vec3 hA, vec3 pn, float pc, const coll_ctx &ctx, coll_result &result) {
    vec3 center = pn * pc;
    int featA, idxA;
    float projA;
    box_support_feature(hA, ctx.posA, ctx.ornA, center, -pn, featA, idxA, projA, kSupportFeatureTolerance);
    float distance = -projA;
    if (distance > ctx.threshold) return;
    vec3 verts[4];
    int nv = 0;
    if (featA == BF_VERTEX) { verts[0] = box_vertex(hA, idxA); nv = 1; }
    else if (featA == BF_EDGE) {
        verts[0] = box_vertex(hA, kBoxEdgeIndices[idxA * 2]);
        verts[1] = box_vertex(hA, kBoxEdgeIndices[idxA * 2 + 1]);
        nv = 2;
    } else {
        for (int i = 0; i < 4; ++i) verts[i] = box_vertex(hA, kBoxFaceIndices[idxA * 4 + i]);
        nv = 4;
    }
    coll_point point{};
    point.normal = pn;
    point.distance = distance;
    point.attachment = NA_ON_B;
    for (int i = 0; i < nv; ++i) {
        point.pivotA = verts[i];
        vec3 pAw = to_world(point.pivotA, ctx.posA, ctx.ornA);
        vec3 pBw = project_plane(pAw, center, pn);
        point.pivotB = to_object(pBw, ctx.posB, ctx.ornB);
        point.distance = dot(pAw - pBw, pn);
        result.add_point(point);
    }
}

inline void collide_sphere_sphere(float rAr, float rBr, const coll_ctx &ctx, coll_result &result) {
    vec3 d = ctx.posA - ctx.posB;
    float d2 = length_sqr(d);
    float r = rAr + rBr + ctx.threshold;
    if (d2 > r * r) return;
    float dist = std::sqrt(d2);
    vec3 dn = dist > kEps ? d / dist : vec3{1, 0, 0};
    vec3 rA = -dn * rAr;
    rA = rotate(conjugate(ctx.ornA), rA);
    vec3 rB = dn * rBr;
    rB = rotate(conjugate(ctx.ornB), rB);
    result.add_point({rA, rB, dn, dist - rAr - rBr, NA_NONE});
}

inline void collide_sphere_plane(float radius, vec3 pn, float pc, const coll_ctx &ctx, coll_result &result) {
    vec3 center = pn * pc;
    vec3 d = ctx.posA - center;
    float l = dot(pn, d);
    if (l > radius) return;
    vec3 pivotA = rotate(conjugate(ctx.ornA), -pn * radius);
    vec3 pivotB = rotate(conjugate(ctx.ornB), d - pn * l - center);
    result.add_point({pivotA, pivotB, pn, l - radius, NA_ON_B});
}

inline void collide_sphere_box(float radius, vec3 hB, const coll_ctx &ctx, coll_result &result) {
    const quat ornB_conj = conjugate(ctx.ornB);
    const vec3 posA_in_B = rotate(ornB_conj, ctx.posA - ctx.posB);
    const quat ornA_in_B = ornB_conj * ctx.ornA;
    vec3 closest = closest_point_box_outside(hB, posA_in_B);
    vec3 normalB = posA_in_B - closest;
    float d2 = length_sqr(normalB);
    float min_dist = radius + ctx.threshold;
    if (d2 > min_dist * min_dist) return;
    float center_distance;
    int attach = NA_NONE;
    if (d2 <= kEps) {
        center_distance = -closest_point_box_inside(hB, posA_in_B, closest, normalB);
        attach = NA_ON_B;
    } else {
        center_distance = std::sqrt(d2);
        normalB /= center_distance;
        if (std::fabs(normalB.x) > 1.0f - kEps || std::fabs(normalB.y) > 1.0f - kEps ||
            std::fabs(normalB.z) > 1.0f - kEps)
            attach = NA_ON_B;
    }
    vec3 pivotA_in_B = posA_in_B - normalB * radius;
    vec3 pivotA = to_object(pivotA_in_B, posA_in_B, ornA_in_B);
    vec3 pivotB = closest;
    vec3 normal = rotate(ctx.ornB, normalB);
    result.add_point({pivotA, pivotB, normal, center_distance - radius, attach});
}

// ---- capsule pairs
inline float closest_point_segment(vec3 q0, vec3 q1, vec3 p, float &t, vec3 &q) {   // geom.cpp:12-22
    const vec3 v = q1 - q0, w = p - q0;
    const float a = dot(w, v), b = dot(v, v);
    t = clamp_unit(a / b);
    q = q0 + v * t;
    return length_sqr(p - q);
}
inline float closest_point_line(vec3 q0, vec3 dir, vec3 p, float &t, vec3 &r) {   // geom.cpp:35-44
    const vec3 w = p - q0;
    const float a = dot(w, dir), b = dot(dir, dir);
    t = a / b;
    r = q0 + dir * t;
    return length_sqr(p - r);
}
// collide_capsule_plane.cpp:6-38
inline void collide_capsule_plane(const shape &shA, vec3 pn, float pc, const coll_ctx &ctx, coll_result &result) {
    const vec3 center = pn * pc;
    vec3 cv[2];
    capsule_vertices(shA, ctx.posA, ctx.ornA, cv);
    const float proj[2] = {dot(cv[0] - center, pn), dot(cv[1] - center, pn)};
    for (int i = 0; i < 2; ++i) {
        const float distance = proj[i] - shA.radius;
        if (distance > ctx.threshold) continue;
        const vec3 vertex = cv[i];
        const vec3 pivotA_world = vertex - pn * shA.radius;
        const vec3 pivotA = to_object(pivotA_world, ctx.posA, ctx.ornA);
        const vec3 pivotB = project_plane(vertex, center, pn);
        result.add_point({pivotA, pivotB, pn, distance, NA_ON_B});
    }
}
// collide_capsule_sphere.cpp:10-51
inline void collide_capsule_sphere(const shape &shA, float rB, const coll_ctx &ctx, coll_result &result) {
    vec3 cv[2];
    capsule_vertices(shA, ctx.posA, ctx.ornA, cv);
    vec3 closest; float t;
    const float dist_sqr = closest_point_segment(cv[0], cv[1], ctx.posB, t, closest);
    const float min_dist = shA.radius + rB + ctx.threshold;
    if (dist_sqr > min_dist * min_dist) return;
    vec3 normal = closest - ctx.posB;
    const float nl2 = length_sqr(normal);
    float distance;
    if (nl2 > kEps) {
        const float nl = std::sqrt(nl2);
        normal /= nl;
        distance = nl - shA.radius - rB;
    } else {
        normal = rotate(ctx.ornA, vec3{0, 0, 1});   // quaternion_z
        distance = -(shA.radius + rB);
    }
    const vec3 normalB = rotate(conjugate(ctx.ornB), normal);
    const vec3 pivotA_world = closest - normal * shA.radius;
    result.add_point({to_object(pivotA_world, ctx.posA, ctx.ornA), normalB * rB, normal, distance, NA_NONE});
}
// collide_capsule_capsule.cpp:7-80
inline void collide_capsule_capsule(const shape &shA, const shape &shB, const coll_ctx &ctx, coll_result &result) {
    vec3 vA[2], vB[2];
    capsule_vertices(shA, ctx.posA, ctx.ornA, vA);
    capsule_vertices(shB, ctx.posB, ctx.ornB, vB);
    float s[2], t[2];
    vec3 cA[2], cB[2];
    size_t num_points = 0;
    const float dist_sqr = closest_point_segment_segment(vA[0], vA[1], vB[0], vB[1], s[0], t[0], cA[0], cB[0], &num_points,
                                                         &s[1], &t[1], &cA[1], &cB[1]);
    const float min_dist = shA.radius + shB.radius + ctx.threshold;
    if (dist_sqr > min_dist * min_dist) return;
    vec3 normal;
    float distance;
    if (dist_sqr > kEps) {
        const float dist = std::sqrt(dist_sqr);
        normal = (cA[0] - cB[0]) / dist;
        distance = dist - shA.radius - shB.radius;
    } else {
        const vec3 axisA = vA[1] - vA[0], axisB = vB[1] - vB[0];
        normal = cross(axisA, axisB);
        if (dot(ctx.posA - ctx.posB, normal) < 0) normal *= -1.0f;
        if (!try_normalize(normal)) normal = vec3{0, 1, 0};
        distance = -(shA.radius + shB.radius);
    }
    for (size_t i = 0; i < num_points; ++i) {
        const vec3 pA = cA[i] - normal * shA.radius, pB = cB[i] + normal * shB.radius;
        result.add_point({to_object(pA, ctx.posA, ctx.ornA), to_object(pB, ctx.posB, ctx.ornB), normal, distance, NA_NONE});
    }
}
// collide_capsule_box.cpp:14-213
inline void collide_capsule_box(const shape &shA, vec3 hB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA{0, 0, 0};
    const quat ornA = ctx.ornA, ornB = ctx.ornB;
    const vec3 posB = ctx.posB - ctx.posA;
    vec3 cv[2];
    capsule_vertices(shA, posA, ornA, cv);
    const vec3 box_axes[3] = {rotate(ornB, vec3{1, 0, 0}), rotate(ornB, vec3{0, 1, 0}), rotate(ornB, vec3{0, 0, 1})};
    float distance = -kScalarMax, projection_box = -kScalarMax;
    vec3 sep{0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        vec3 dir = box_axes[i];
        if (dot(posA - posB, dir) < 0) dir = -dir;
        const float projA = -capsule_support_projection(cv, shA.radius, -dir);
        const float projB = dot(posB, dir) + hB[i];
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_box = projB; sep = dir; }
    }
    for (int i = 0; i < 12; ++i) {
        vec3 ev[2];
        box_edge_world(hB, i, posB, ornB, ev);
        float s, t;
        vec3 cA, cB;
        closest_point_segment_segment(ev[0], ev[1], cv[0], cv[1], s, t, cA, cB, nullptr, nullptr, nullptr, nullptr, nullptr);
        vec3 dir = cA - cB;
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(cv, shA.radius, -dir);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_box = projB; sep = dir; }
    }
    if (distance > ctx.threshold) return;
    const float proj[2] = {dot(cv[0], sep), dot(cv[1], sep)};
    const bool is_capsule_edge = std::fabs(proj[0] - proj[1]) < kSupportFeatureTolerance;
    const vec3 contact_origin_box = sep * projection_box;
    int featB, idxB;
    float fdB;
    box_support_feature(hB, posB, ornB, contact_origin_box, sep, featB, idxB, fdB, kSupportFeatureTolerance);
    coll_point point;
    point.normal = sep; point.distance = distance;
    point.pivotA = {0, 0, 0}; point.pivotB = {0, 0, 0};
    if (featB == BF_FACE) {
        vec3 fv[4];
        box_face_world(hB, idxB, posB, ornB, fv);
        point.attachment = NA_ON_B;
        if (is_capsule_edge) {
            for (int k = 0; k < 2; ++k) {
                const vec3 pointA = cv[k];
                if (point_in_quad_prism(fv, sep, pointA)) {
                    point.pivotA = to_object(pointA - sep * shA.radius, posA, ornA);
                    point.pivotB = to_object(project_plane(pointA, contact_origin_box, sep), posB, ornB);
                    result.add_point(point);
                }
            }
            if (result.num_points == 2) return;
            const vec3 fc = box_face_center(hB, idxB, posB, ornB);
            const mat3 fb = box_face_basis(idxB, ornB);
            const vec2 he = box_face_half_extents(hB, idxB);
            const vec3 q0 = to_object(cv[0], fc, fb), q1 = to_object(cv[1], fc, fb);
            float ss[2];
            const size_t n = intersect_line_aabb({q0.x, q0.z}, {q1.x, q1.z}, -he, he, ss[0], ss[1]);
            for (size_t i = 0; i < n; ++i) {
                if (ss[i] < 0 || ss[i] > 1) continue;
                const vec3 edge_pivot = lerp(cv[0], cv[1], ss[i]);
                const vec3 face_pivot = project_plane(edge_pivot, fc, sep);
                point.pivotA = to_object(edge_pivot - sep * shA.radius, posA, ornA);
                point.pivotB = to_object(face_pivot, posB, ornB);
                result.add_point(point);
            }
        } else {
            const vec3 cvx = proj[0] < proj[1] ? cv[0] : cv[1];
            const vec3 pA = cvx - sep * shA.radius;
            const vec3 pB = project_plane(pA, contact_origin_box, sep);
            point.pivotA = to_object(pA, posA, ornA);
            point.pivotB = to_object(pB, posB, ornB);
            result.add_point(point);
        }
    } else if (featB == BF_EDGE) {
        vec3 ev[2];
        box_edge_world(hB, idxB, posB, ornB, ev);
        point.attachment = NA_NONE;
        if (is_capsule_edge) {
            float s[2], t[2];
            vec3 cA[2], cB[2];
            size_t n = 0;
            closest_point_segment_segment(cv[0], cv[1], ev[0], ev[1], s[0], t[0], cA[0], cB[0], &n, &s[1], &t[1], &cA[1], &cB[1]);
            for (size_t i = 0; i < n; ++i) {
                point.pivotA = to_object(cA[i] - sep * shA.radius, posA, ornA);
                point.pivotB = to_object(cB[i], posB, ornB);
                result.add_point(point);
            }
        } else {
            const vec3 cvx = proj[0] < proj[1] ? cv[0] : cv[1];
            const vec3 edge_dir = ev[1] - ev[0];
            vec3 pB; float t;
            closest_point_line(ev[0], edge_dir, cvx, t, pB);
            point.pivotB = to_object(pB, posB, ornB);
            point.pivotA = to_object(cvx - sep * shA.radius, posA, ornA);
            result.add_point(point);
        }
    } else {
        point.pivotB = box_vertex(hB, idxB);
        const vec3 pB = to_world(point.pivotB, posB, ornB);
        const vec3 pA = pB + sep * distance;
        point.pivotA = to_object(pA, posA, ornA);
        point.attachment = NA_NONE;
        result.add_point(point);
    }
}

}  // namespace orc
#include "ocylinder.hpp"
#include "opolyhedron.hpp"
namespace orc {

// Dispatch on the shape pair; mirrored overloads go through swap_collide (collide.hpp:369-374).
inline void collide(const shape &shA, const shape &shB, const coll_ctx &ctx, coll_result &r) {
    const int a = shA.type, b = shB.type;
    if (a == SHAPE_BOX && b == SHAPE_BOX) collide_box_box(shA.half_extents, shB.half_extents, ctx, r);
    else if (a == SHAPE_BOX && b == SHAPE_PLANE) collide_box_plane(shA.half_extents, shB.normal, shB.constant, ctx, r);
    else if (a == SHAPE_PLANE && b == SHAPE_BOX) {
        collide_box_plane(shB.half_extents, shA.normal, shA.constant, ctx.swapped(), r); r.swap();
    } else if (a == SHAPE_SPHERE && b == SHAPE_SPHERE) collide_sphere_sphere(shA.radius, shB.radius, ctx, r);
    else if (a == SHAPE_SPHERE && b == SHAPE_PLANE) collide_sphere_plane(shA.radius, shB.normal, shB.constant, ctx, r);
    else if (a == SHAPE_PLANE && b == SHAPE_SPHERE) {
        collide_sphere_plane(shB.radius, shA.normal, shA.constant, ctx.swapped(), r); r.swap();
    } else if (a == SHAPE_SPHERE && b == SHAPE_BOX) collide_sphere_box(shA.radius, shB.half_extents, ctx, r);
    else if (a == SHAPE_BOX && b == SHAPE_SPHERE) {
        collide_sphere_box(shB.radius, shA.half_extents, ctx.swapped(), r); r.swap();
    } else if (a == SHAPE_CAPSULE && b == SHAPE_PLANE) collide_capsule_plane(shA, shB.normal, shB.constant, ctx, r);
    else if (a == SHAPE_PLANE && b == SHAPE_CAPSULE) { collide_capsule_plane(shB, shA.normal, shA.constant, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CAPSULE && b == SHAPE_SPHERE) collide_capsule_sphere(shA, shB.radius, ctx, r);
    else if (a == SHAPE_SPHERE && b == SHAPE_CAPSULE) { collide_capsule_sphere(shB, shA.radius, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CAPSULE && b == SHAPE_CAPSULE) collide_capsule_capsule(shA, shB, ctx, r);
    else if (a == SHAPE_CAPSULE && b == SHAPE_BOX) collide_capsule_box(shA, shB.half_extents, ctx, r);
    else if (a == SHAPE_BOX && b == SHAPE_CAPSULE) { collide_capsule_box(shB, shA.half_extents, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CYLINDER && b == SHAPE_PLANE) collide_cylinder_plane(shA, shB.normal, shB.constant, ctx, r);
    else if (a == SHAPE_PLANE && b == SHAPE_CYLINDER) { collide_cylinder_plane(shB, shA.normal, shA.constant, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CYLINDER && b == SHAPE_SPHERE) collide_cylinder_sphere(shA, shB.radius, ctx, r);
    else if (a == SHAPE_SPHERE && b == SHAPE_CYLINDER) { collide_cylinder_sphere(shB, shA.radius, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CYLINDER && b == SHAPE_CYLINDER) collide_cylinder_cylinder(shA, shB, ctx, r);
    else if (a == SHAPE_CYLINDER && b == SHAPE_BOX) collide_cylinder_box(shA, shB.half_extents, ctx, r);
    else if (a == SHAPE_BOX && b == SHAPE_CYLINDER) { collide_cylinder_box(shB, shA.half_extents, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_CAPSULE && b == SHAPE_CYLINDER) collide_capsule_cylinder(shA, shB, ctx, r);
    else if (a == SHAPE_CYLINDER && b == SHAPE_CAPSULE) { collide_capsule_cylinder(shB, shA, ctx.swapped(), r); r.swap(); }
    else if (a == SHAPE_POLYHEDRON || b == SHAPE_POLYHEDRON) {
        // the rotated mesh is what update_rotated_meshes left after the last integration: rotate(orn * rotated_mesh_list::orientation)
        // with that orientation the identity for a polyhedron that is not part of a compound (update_rotated_meshes.cpp:62)
        const bool first = a == SHAPE_POLYHEDRON;
        const shape &P = first ? shA : shB, &Q = first ? shB : shA;
        const coll_ctx c = first ? ctx : ctx.swapped();
        const ConvexMesh &mP = *mesh_registry()[P.mesh];
        RotatedMesh rP, rQ;
        PolySh pP{MeshView{&mP}, RotView{&rP}};
        const int q = Q.type;
        if (q == SHAPE_PLANE) { rP.update(mP, c.ornA * quat{0, 0, 0, 1}); collide_polyhedron_plane(pP, Q.normal, Q.constant, c, r); }
        else if (q == SHAPE_SPHERE) collide_polyhedron_sphere(pP, Q.radius, c, r);
        else if (q == SHAPE_BOX) collide_polyhedron_box(pP, Q.half_extents, c, r);
        else if (q == SHAPE_CAPSULE) collide_polyhedron_capsule(pP, Q, c, r);
        else if (q == SHAPE_CYLINDER) collide_polyhedron_cylinder(pP, Q, c, r);
        else if (q == SHAPE_POLYHEDRON) {
            const ConvexMesh &mQ = *mesh_registry()[Q.mesh];
            rP.update(mP, c.ornA * quat{0, 0, 0, 1}); rQ.update(mQ, c.ornB * quat{0, 0, 0, 1});
            collide_polyhedron_polyhedron(pP, PolySh{MeshView{&mQ}, RotView{&rQ}}, c, r);
        }
        if (!first) r.swap();
    }
    // plane-plane: both static, never paired (only procedural bodies query the broadphase).
}

}  // namespace orc
