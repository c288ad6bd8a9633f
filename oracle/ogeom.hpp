// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// Geometry helpers on the narrowphase / manifold path, restated from
//   /root/reference/src/edyn/math/geom.cpp:73-170   closest_point_segment_segment
//   /root/reference/src/edyn/math/geom.cpp:730-754  plane_space
//   /root/reference/src/edyn/math/geom.cpp:846-985  manifold_score + insertion_point_index
//   /root/reference/src/edyn/math/geom.cpp:987-1042 closest_point_box_outside/inside
//   /root/reference/src/edyn/math/geom.cpp:1044-1138 intersect_line_aabb
//   /root/reference/include/edyn/math/geom.hpp:331-348 point_in_polygonal_prism
//   /root/reference/src/edyn/math/triangle.cpp:7-26 point_in_triangle
#pragma once
#include "omath.hpp"

namespace orc {

constexpr int kMaxContacts = 4;                       // config/constants.hpp:9
constexpr float kCollisionThreshold = 0.01f;          // :15
constexpr float kContactBreakingThreshold = 0.02f;    // :21
constexpr float kContactMergingThreshold = 0.01f;     // :27
constexpr float kContactCachingThreshold = 0.04f;     // :34
constexpr float kSupportFeatureTolerance = 0.005f;    // :56
constexpr float kContactPositionCorrectionRate = 0.2f;// :61 (== position_solver::error_correction_rate)

inline void plane_space(vec3 n, vec3 &p, vec3 &q) {
    if (std::fabs(n.z) > kHalfSqrt2) {
        float a = n.y * n.y + n.z * n.z;
        float k = 1.0f / std::sqrt(a);
        p.x = 0; p.y = -n.z * k; p.z = n.y * k;
        q.x = a * k; q.y = -n.x * p.z; q.z = n.x * p.y;
    } else {
        float a = n.x * n.x + n.y * n.y;
        float k = 1.0f / std::sqrt(a);
        p.x = -n.y * k; p.y = n.x * k; p.z = 0;
        q.x = -n.z * p.y; q.y = n.z * p.x; q.z = a * k;
    }
}

// Convex quad prism containment (face of a box extruded along its normal).
inline bool point_in_quad_prism(const vec3 v[4], vec3 normal, vec3 point) {
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) % 4;
        vec3 d = v[j] - v[i];
        vec3 t = cross(d, normal);
        if (dot(point - v[i], t) > kEps) return false;
    }
    return true;
}

inline bool point_in_triangle(const vec3 v[3], vec3 normal, vec3 p) {
    vec3 e0 = v[1] - v[0], e1 = v[2] - v[1], e2 = v[0] - v[2];
    vec3 q0 = p - v[0], q1 = p - v[1], q2 = p - v[2];
    vec3 en0 = cross(e0, normal), en1 = cross(e1, normal), en2 = cross(e2, normal);
    float d0 = dot(en0, q0), d1 = dot(en1, q1), d2 = dot(en2, q2);
    return (d0 > -kEps && d1 > -kEps && d2 > -kEps) || (d0 < kEps && d1 < kEps && d2 < kEps);
}

inline size_t intersect_line_aabb(vec2 p0, vec2 p1, vec2 bmin, vec2 bmax, float &s0, float &s1) {
    size_t n = 0;
    vec2 d = p1 - p0, e = bmin - p0, f = bmax - p0;
    if (std::fabs(d.x) < kEps) {            // vertical line
        if (e.x <= 0 && f.x >= 0) { s0 = e.y / d.y; s1 = f.y / d.y; n = 2; }
        return n;
    }
    if (std::fabs(d.y) < kEps) {            // horizontal line
        if (e.y <= 0 && f.y >= 0) { s0 = e.x / d.x; s1 = f.x / d.x; n = 2; }
        return n;
    }
    {   // left edge
        float t = e.x / d.x, qy = p0.y + d.y * t;
        if (qy >= bmin.y && qy < bmax.y) { s0 = t; ++n; }
    }
    {   // right edge
        float t = f.x / d.x, qy = p0.y + d.y * t;
        if (qy > bmin.y && qy <= bmax.y) {
            if (n == 0) { s0 = t; ++n; }
            else if (std::fabs(t - s0) > kEps) { s1 = t; ++n; }
        }
    }
    if (n == 2) return n;
    {   // bottom edge
        float t = e.y / d.y, qx = p0.x + d.x * t;
        if (qx >= bmin.x && qx < bmax.x) {
            if (n == 0) { s0 = t; ++n; }
            else if (std::fabs(t - s0) > kEps) { s1 = t; ++n; }
        }
    }
    if (n == 2) return n;
    {   // top edge
        float t = f.y / d.y, qx = p0.x + d.x * t;
        if (qx > bmin.x && qx <= bmax.x) {
            if (n == 0) { s0 = t; ++n; }
            else if (std::fabs(t - s0) > kEps) { s1 = t; ++n; }
        }
    }
    return n;
}

// Returns squared distance; may yield two closest pairs for parallel segments.
inline float closest_point_segment_segment(vec3 p1, vec3 q1, vec3 p2, vec3 q2, float &s, float &t, vec3 &c1,
                                           vec3 &c2, size_t *num_points, float *sp, float *tp, vec3 *c1p,
                                           vec3 *c2p) {
    const vec3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    if (a <= kEps && e <= kEps) {
        s = t = 0; c1 = p1; c2 = p2;
        return length_sqr(c1 - c2);
    }
    if (a <= kEps) {
        s = 0; t = f / e; t = clamp_unit(t);
    } else {
        float c = dot(d1, r);
        if (e <= kEps) {
            t = 0; s = clamp_unit(-c / a);
        } else {
            const float b = dot(d1, d2);
            const float denom = a * e - b * b;
            if (denom > kEps) {
                s = clamp_unit((b * f - c * e) / denom);
                if (num_points) *num_points = 1;
            } else if (num_points) {
                vec3 r1 = p1 - q2;
                float f1 = dot(d1, r1);
                float a_inv = 1 / a;
                s = clamp_unit(std::min(-c * a_inv, -f1 * a_inv));
                *sp = clamp_unit(std::max(-c * a_inv, -f1 * a_inv));
                vec3 r2 = p2 - q1;
                float f2 = dot(d2, r2);
                float e_inv = 1 / e;
                t = clamp_unit(std::min(-f * e_inv, -f2 * e_inv));
                *tp = clamp_unit(std::max(-f * e_inv, -f2 * e_inv));
                if (std::fabs(s - *sp) > kEps) {
                    *num_points = 2;
                    *c1p = p1 + d1 * *sp;
                    *c2p = p2 + d2 * *tp;
                } else {
                    *num_points = 1;
                }
            } else {
                s = 0;
            }
            const float tnom = b * s + f;
            if (tnom < 0) { t = 0; s = clamp_unit(-c / a); }
            else if (tnom > e) { t = 1; s = clamp_unit((b - c) / a); }
            else { t = tnom / e; }
        }
    }
    c1 = p1 + d1 * s;
    c2 = p2 + d2 * t;
    return length_sqr(c1 - c2);
}

inline vec3 closest_point_box_outside(vec3 h, vec3 p) {
    vec3 c = p;
    c.x = std::min(h.x, c.x); c.x = std::max(-h.x, c.x);
    c.y = std::min(h.y, c.y); c.y = std::max(-h.y, c.y);
    c.z = std::min(h.z, c.z); c.z = std::max(-h.z, c.z);
    return c;
}

// NB: returns the LAST computed `dist` (half_extent.z + p.z), exactly as the reference does
// (geom.cpp:1041 `return dist;` not `min_dist`).
inline float closest_point_box_inside(vec3 h, vec3 p, vec3 &closest, vec3 &normal) {
    float dist = h.x - p.x;
    float min_dist = dist;
    closest = {h.x, p.y, p.z}; normal = {1, 0, 0};
    dist = h.x + p.x;
    if (dist < min_dist) { min_dist = dist; closest = {-h.x, p.y, p.z}; normal = {-1, 0, 0}; }
    dist = h.y - p.y;
    if (dist < min_dist) { min_dist = dist; closest = {p.x, h.y, p.z}; normal = {0, 1, 0}; }
    dist = h.y + p.y;
    if (dist < min_dist) { min_dist = dist; closest = {p.x, -h.y, p.z}; normal = {0, -1, 0}; }
    dist = h.z - p.z;
    if (dist < min_dist) { min_dist = dist; closest = {p.x, p.y, h.z}; normal = {0, 0, 1}; }
    dist = h.z + p.z;
    if (dist < min_dist) { min_dist = dist; closest = {p.x, p.y, -h.z}; normal = {0, 0, -1}; }
    return dist;
}

enum class insert_type : int { none = 0, append = 1, similar = 2, replace = 3 };
struct insert_result { insert_type type; size_t index; };

inline float manifold_score(vec3 p0, vec3 p1, vec3 p2, vec3 p3) {
    vec3 c0 = cross(p0 - p1, p0 - p2);
    vec3 c1 = cross(p0 - p2, p0 - p3);
    vec3 c2 = cross(p0 - p3, p0 - p1);
    vec3 c3 = cross(p1 - p2, p2 - p3);
    return length_sqr(c0) + length_sqr(c1) + length_sqr(c2) + length_sqr(c3);
}

// `count` is the capacity (always kMaxContacts on this path); `num_points` is incremented on append.
inline insert_result insertion_point_index(const vec3 *pts, size_t count, size_t &num_points, vec3 np) {
    const float sim2 = kContactMergingThreshold * kContactMergingThreshold;
    if (num_points == 0) return {insert_type::append, num_points++};
    if (num_points == 1) {
        if (distance_sqr(np, pts[0]) > sim2) return {insert_type::append, num_points++};
        return {insert_type::similar, 0};
    }
    if (num_points == 2) {
        if (length_sqr(cross(np - pts[0], np - pts[1])) > kEps) return {insert_type::append, num_points++};
        float d0 = distance_sqr(np, pts[0]);
        float d1 = distance_sqr(np, pts[1]);
        float cur = distance_sqr(pts[0], pts[1]);
        if (d0 > cur && d0 > d1) return {d1 < sim2 ? insert_type::similar : insert_type::replace, 1};
        if (d1 > cur && d1 > d0) return {d0 < sim2 ? insert_type::similar : insert_type::replace, 0};
        return {insert_type::none, count};
    }
    if (num_points == 3) {
        vec3 verts[3] = {pts[0], pts[1], pts[2]};
        vec3 normal = cross(pts[0] - pts[1], pts[1] - pts[2]);
        if (try_normalize(normal)) {
            if (std::fabs(dot(np - pts[0], normal)) < kEps && point_in_triangle(verts, normal, np))
                return {insert_type::none, count};
            return {insert_type::append, num_points++};
        }
        float d0 = dot(pts[1] - pts[0], pts[2] - pts[0]);
        if (d0 > 0 && d0 < 1) return {insert_type::replace, 1};
        float d1 = dot(pts[0] - pts[1], pts[2] - pts[1]);
        if (d1 > 0 && d1 < 1) return {insert_type::replace, 0};
        float d2 = dot(pts[2] - pts[0], pts[1] - pts[0]);
        if (d2 > 0 && d2 < 1) return {insert_type::replace, 2};
        float ds[3] = {distance_sqr(pts[0], pts[1]), distance_sqr(pts[1], pts[2]), distance_sqr(pts[2], pts[0])};
        size_t mi = SIZE_MAX;
        float md = kScalarMax;
        for (size_t i = 0; i < 3; ++i) if (ds[i] < md) { md = ds[i]; mi = i; }
        return {insert_type::replace, mi};
    }
    float scores[4];
    scores[0] = manifold_score(np, pts[1], pts[2], pts[3]);
    scores[1] = manifold_score(np, pts[0], pts[2], pts[3]);
    scores[2] = manifold_score(np, pts[0], pts[1], pts[3]);
    scores[3] = manifold_score(np, pts[0], pts[1], pts[2]);
    float best = manifold_score(pts[0], pts[1], pts[2], pts[3]);
    size_t bi = SIZE_MAX;
    for (size_t i = 0; i < 4; ++i) if (scores[i] > best) { best = scores[i]; bi = i; }
    if (bi < (size_t)kMaxContacts)
        return {distance_sqr(pts[bi], np) < sim2 ? insert_type::similar : insert_type::replace, bi};
    return {insert_type::none, count};
}

}  // namespace orc
