// Device-side closest-feature collision for the shape pairs on the hot path and the ≤4-point
// manifold maintenance, executed one manifold per lane by k_narrowphase (narrowphase.hip).
// Behaviour follows the reference routines named at each function; arithmetic order is kept so that
// results are bit-identical to a scalar fp32 evaluation.
#pragma once
#include "dmath.hpp"

namespace dc {
using namespace dm;

constexpr int kMaxContacts = 4;                     // config/constants.hpp:9
constexpr float kCollisionThreshold = 0.01f;        // :15
constexpr float kBreakingThreshold = 0.02f;         // :21
constexpr float kMergingThreshold = 0.01f;          // :27
constexpr float kCachingThreshold = 0.04f;          // :34
constexpr float kSupportTolerance = 0.005f;         // :56

enum { SHAPE_NONE = 0, SHAPE_BOX = 1, SHAPE_SPHERE = 2, SHAPE_PLANE = 3, SHAPE_CAPSULE = 4, SHAPE_CYLINDER = 5, SHAPE_POLYHEDRON = 6 };
enum { BF_VERTEX = 0, BF_EDGE = 1, BF_FACE = 2 };
enum { NA_NONE = 0, NA_ON_A = 1, NA_ON_B = 2 };
enum { INS_NONE = 0, INS_APPEND = 1, INS_SIMILAR = 2, INS_REPLACE = 3 };

// include/edyn/shapes/box_shape.hpp:22-47 and src/edyn/shapes/box_shape.cpp:115-168 (feature tables)
__device__ static const signed char kVertSign[8][3] = {{1, 1, 1}, {1, -1, 1}, {1, -1, -1}, {1, 1, -1},
                                                        {-1, 1, 1}, {-1, 1, -1}, {-1, -1, -1}, {-1, -1, 1}};
__device__ static const unsigned char kEdgeIdx[24] = {0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 0, 4, 1, 7, 2, 6, 3, 5};
__device__ static const unsigned char kFaceIdx[24] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 3, 5, 4, 1, 7, 6, 2, 0, 4, 7, 1, 3, 2, 6, 5};
__device__ static const signed char kFaceNormal[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
__device__ static const signed char kFaceTangent[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};

DI f3 box_vertex(f3 h, int i) {
    return h * mk3((float)kVertSign[i][0], (float)kVertSign[i][1], (float)kVertSign[i][2]);
}
DI f3 face_normal(int f) { return mk3((float)kFaceNormal[f][0], (float)kFaceNormal[f][1], (float)kFaceNormal[f][2]); }
DI f3 face_tangent(int f) { return mk3((float)kFaceTangent[f][0], (float)kFaceTangent[f][1], (float)kFaceTangent[f][2]); }
DI f3 support_point_box(f3 h, f3 d) { return {d.x > 0 ? h.x : -h.x, d.y > 0 ? h.y : -h.y, d.z > 0 ? h.z : -h.z}; }
DI float box_support_projection(f3 h, f3 pos, q4 orn, f3 dir) {   // box_shape.cpp:24-28
    f3 ld = rotate(conjugate(orn), dir);
    f3 pt = support_point_box(h, ld);
    return dot(pos, dir) + dot(pt, ld);
}
DI int support_face_index(f3 d) {   // box_shape.cpp:227-235 via max_index_abs
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float m = ax; int i = 0;
    if (ay > m) { m = ay; i = 1; }
    if (az > m) { i = 2; }
    return comp(d, i) < 0 ? i * 2 + 1 : i * 2;
}
// Value selects: `d = c ? v : d` per scalar. The per-lane point lists (4 entries) are indexed with compile-time constants
// only - unrolled loops with guards, these select chains for the few data-dependent indices, array rotation inside rolled
// loops - so that they stay in registers; a dynamically indexed local array is scratch memory on this target.
DI void sel(int &d, bool c, int v) { d = c ? v : d; }
DI void sel(float &d, bool c, float v) { d = c ? v : d; }
DI void sel(f3 &d, bool c, const f3 &v) { sel(d.x, c, v.x); sel(d.y, c, v.y); sel(d.z, c, v.z); }
template <class T> DI T pick4(const T (&a)[4], int i) {
    T r = a[0];
    sel(r, i == 1, a[1]); sel(r, i == 2, a[2]); sel(r, i == 3, a[3]);
    return r;
}
template <class T> DI void put4(T (&a)[4], int i, const T &v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sel(a[k], k == i, v);
}
template <class T> DI void rotate4(T (&a)[4]) { T t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t; }

DI int edge_index(int v0, int v1) {
    for (int i = 0; i < 12; ++i) {
        int a = kEdgeIdx[i * 2], b = kEdgeIdx[i * 2 + 1];
        if ((a == v0 && b == v1) || (b == v0 && a == v1)) return i;
    }
    return 0;
}
// box_shape.cpp:30-96, object-space direction
DI void support_feature_local(f3 h, f3 dir, int &feature, int &findex, float &projection, float threshold) {
    const int face = support_face_index(dir);
    float proj[4]; int vidx[4];
    int i0 = 0, i1 = 0, i2 = 0;   // idx[0..2] of the reference's list of vertices within the threshold
    int count = 1, maxi = 0;
    projection = -kScalarMax;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int vi = kFaceIdx[face * 4 + i];
        vidx[i] = vi;
        float p = dot(box_vertex(h, vi), dir);
        proj[i] = p;
        if (p > projection) { projection = p; i0 = i; maxi = i; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i != maxi && proj[i] > projection - threshold) { sel(i1, count == 1, i); sel(i2, count == 2, i); ++count; }
    const int v0 = pick4(vidx, i0), v1 = pick4(vidx, i1), v2 = pick4(vidx, i2);
    if (count == 1) { feature = BF_VERTEX; findex = v0; }
    else if (count == 2) { feature = BF_EDGE; findex = edge_index(v0, v1); }
    else if (count == 3) {
        feature = BF_EDGE;
        float p0 = pick4(proj, i0), p1 = pick4(proj, i1), p2 = pick4(proj, i2);
        if (p0 <= p1 && p0 <= p2) findex = edge_index(v1, v2);
        else if (p1 <= p0 && p1 <= p2) findex = edge_index(v0, v2);
        else findex = edge_index(v0, v1);
    } else { feature = BF_FACE; findex = face; }
}
DI void support_feature(f3 h, f3 pos, q4 orn, f3 axis_pos, f3 axis_dir, int &feature, int &findex, float &projection,
                        float threshold) {
    f3 ld = rotate(conjugate(orn), axis_dir);
    support_feature_local(h, ld, feature, findex, projection, threshold);
    projection += dot(pos - axis_pos, axis_dir);
}
DI void face_world(f3 h, int f, f3 pos, q4 orn, f3 (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = to_world(box_vertex(h, kFaceIdx[f * 4 + i]), pos, orn);
}
DI void edge_world(f3 h, int e, f3 pos, q4 orn, f3 (&out)[2]) {
    out[0] = to_world(box_vertex(h, kEdgeIdx[e * 2]), pos, orn);
    out[1] = to_world(box_vertex(h, kEdgeIdx[e * 2 + 1]), pos, orn);
}
DI f3 face_normal_world(int f, q4 orn) { return rotate(orn, face_normal(f)); }
DI f3 face_center(f3 h, int f, f3 pos, q4 orn) { return pos + face_normal_world(f, orn) * comp(h, f / 2); }
DI m3 face_basis(int f, q4 orn) {
    f3 y = face_normal(f), x = face_tangent(f), z = cross(x, y);
    return m3_columns(rotate(orn, x), rotate(orn, y), rotate(orn, z));
}
DI f2 face_half_extents(f3 h, int f) {
    if (f == 0 || f == 1) return {h.z, h.y};
    if (f == 2 || f == 3) return {h.x, h.z};
    return {h.y, h.x};
}

// include/edyn/math/geom.hpp:331-348 (N = 4)
DI bool point_in_quad_prism(const f3 (&v)[4], f3 normal, f3 point) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        f3 t = cross(v[j] - v[i], normal);
        if (dot(point - v[i], t) > kEps) return false;
    }
    return true;
}
DI bool point_in_triangle(f3 v0, f3 v1, f3 v2, f3 normal, f3 p) {   // src/edyn/math/triangle.cpp:7-26
    f3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
    f3 q0 = p - v0, q1 = p - v1, q2 = p - v2;
    float d0 = dot(cross(e0, normal), q0), d1 = dot(cross(e1, normal), q1), d2 = dot(cross(e2, normal), q2);
    return (d0 > -kEps && d1 > -kEps && d2 > -kEps) || (d0 < kEps && d1 < kEps && d2 < kEps);
}
// src/edyn/math/geom.cpp:1044-1138
DI int intersect_line_aabb(f2 p0, f2 p1, f2 bmin, f2 bmax, float &s0, float &s1) {
    int n = 0;
    f2 d = p1 - p0, e = bmin - p0, f = bmax - p0;
    if (fabsf(d.x) < kEps) {
        if (e.x <= 0 && f.x >= 0) { s0 = e.y / d.y; s1 = f.y / d.y; n = 2; }
        return n;
    }
    if (fabsf(d.y) < kEps) {
        if (e.y <= 0 && f.y >= 0) { s0 = e.x / d.x; s1 = f.x / d.x; n = 2; }
        return n;
    }
    { float t = e.x / d.x, qy = p0.y + d.y * t;
      if (qy >= bmin.y && qy < bmax.y) { s0 = t; ++n; } }
    { float t = f.x / d.x, qy = p0.y + d.y * t;
      if (qy > bmin.y && qy <= bmax.y) {
          if (n == 0) { s0 = t; ++n; } else if (fabsf(t - s0) > kEps) { s1 = t; ++n; } } }
    if (n == 2) return n;
    { float t = e.y / d.y, qx = p0.x + d.x * t;
      if (qx >= bmin.x && qx < bmax.x) {
          if (n == 0) { s0 = t; ++n; } else if (fabsf(t - s0) > kEps) { s1 = t; ++n; } } }
    if (n == 2) return n;
    { float t = f.y / d.y, qx = p0.x + d.x * t;
      if (qx > bmin.x && qx <= bmax.x) {
          if (n == 0) { s0 = t; ++n; } else if (fabsf(t - s0) > kEps) { s1 = t; ++n; } } }
    return n;
}
// src/edyn/math/geom.cpp:73-170. MULTI = called with num_points != nullptr (parallel segments may yield two closest pairs);
// without it the parallel case takes s = 0, as the reference does when the optional outputs are absent.
template <bool MULTI = true>
DI void closest_segment_segment(f3 p1, f3 q1, f3 p2, f3 q2, f3 &c1, f3 &c2, int &num, f3 &c1p, f3 &c2p) {
    const f3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s, t;
    if (a <= kEps && e <= kEps) { c1 = p1; c2 = p2; return; }
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
                num = 1;
            } else if (!MULTI) {
                s = 0;
            } else {
                f3 r1 = p1 - q2;
                float f1 = dot(d1, r1);
                float a_inv = 1 / a;
                s = clamp_unit(fminf(-c * a_inv, -f1 * a_inv));
                float sp = clamp_unit(fmaxf(-c * a_inv, -f1 * a_inv));
                f3 r2 = p2 - q1;
                float f2v = dot(d2, r2);
                float e_inv = 1 / e;
                t = clamp_unit(fminf(-f * e_inv, -f2v * e_inv));
                float tp = clamp_unit(fmaxf(-f * e_inv, -f2v * e_inv));
                if (fabsf(s - sp) > kEps) { num = 2; c1p = p1 + d1 * sp; c2p = p2 + d2 * tp; }
                else num = 1;
            }
            const float tnom = b * s + f;
            if (tnom < 0) { t = 0; s = clamp_unit(-c / a); }
            else if (tnom > e) { t = 1; s = clamp_unit((b - c) / a); }
            else t = tnom / e;
        }
    }
    c1 = p1 + d1 * s;
    c2 = p2 + d2 * t;
}
DI f3 closest_point_box_outside(f3 h, f3 p) {
    f3 c = p;
    c.x = fminf(h.x, c.x); c.x = fmaxf(-h.x, c.x);
    c.y = fminf(h.y, c.y); c.y = fmaxf(-h.y, c.y);
    c.z = fminf(h.z, c.z); c.z = fmaxf(-h.z, c.z);
    return c;
}
DI float closest_point_box_inside(f3 h, f3 p, f3 &closest, f3 &normal) {   // geom.cpp:998-1042 (returns last dist)
    float dist = h.x - p.x, md = dist;
    closest = {h.x, p.y, p.z}; normal = {1, 0, 0};
    dist = h.x + p.x; if (dist < md) { md = dist; closest = {-h.x, p.y, p.z}; normal = {-1, 0, 0}; }
    dist = h.y - p.y; if (dist < md) { md = dist; closest = {p.x, h.y, p.z}; normal = {0, 1, 0}; }
    dist = h.y + p.y; if (dist < md) { md = dist; closest = {p.x, -h.y, p.z}; normal = {0, -1, 0}; }
    dist = h.z - p.z; if (dist < md) { md = dist; closest = {p.x, p.y, h.z}; normal = {0, 0, 1}; }
    dist = h.z + p.z; if (dist < md) { md = dist; closest = {p.x, p.y, -h.z}; normal = {0, 0, -1}; }
    return dist;
}
DI float manifold_score(f3 p0, f3 p1, f3 p2, f3 p3) {   // geom.cpp:846-855
    f3 c0 = cross(p0 - p1, p0 - p2), c1 = cross(p0 - p2, p0 - p3), c2 = cross(p0 - p3, p0 - p1), c3 = cross(p1 - p2, p2 - p3);
    return length_sqr(c0) + length_sqr(c1) + length_sqr(c2) + length_sqr(c3);
}
// geom.cpp:857-985. Returns type | index << 8; increments num_points on append.
DI int insertion_point_index(const f3 (&p)[4], int &num_points, f3 np) {
    const float sim2 = kMergingThreshold * kMergingThreshold;
    if (num_points == 0) { int i = num_points++; return INS_APPEND | i << 8; }
    if (num_points == 1) {
        if (distance_sqr(np, p[0]) > sim2) { int i = num_points++; return INS_APPEND | i << 8; }
        return INS_SIMILAR;
    }
    if (num_points == 2) {
        if (length_sqr(cross(np - p[0], np - p[1])) > kEps) { int i = num_points++; return INS_APPEND | i << 8; }
        float d0 = distance_sqr(np, p[0]), d1 = distance_sqr(np, p[1]), cur = distance_sqr(p[0], p[1]);
        if (d0 > cur && d0 > d1) return (d1 < sim2 ? INS_SIMILAR : INS_REPLACE) | 1 << 8;
        if (d1 > cur && d1 > d0) return (d0 < sim2 ? INS_SIMILAR : INS_REPLACE) | 0 << 8;
        return INS_NONE;
    }
    if (num_points == 3) {
        f3 normal = cross(p[0] - p[1], p[1] - p[2]);
        if (try_normalize(normal)) {
            if (fabsf(dot(np - p[0], normal)) < kEps && point_in_triangle(p[0], p[1], p[2], normal, np)) return INS_NONE;
            int i = num_points++;
            return INS_APPEND | i << 8;
        }
        float d0 = dot(p[1] - p[0], p[2] - p[0]);
        if (d0 > 0 && d0 < 1) return INS_REPLACE | 1 << 8;
        float d1 = dot(p[0] - p[1], p[2] - p[1]);
        if (d1 > 0 && d1 < 1) return INS_REPLACE | 0 << 8;
        float d2 = dot(p[2] - p[0], p[1] - p[0]);
        if (d2 > 0 && d2 < 1) return INS_REPLACE | 2 << 8;
        float ds0 = distance_sqr(p[0], p[1]), ds1 = distance_sqr(p[1], p[2]), ds2 = distance_sqr(p[2], p[0]);
        int mi = 0xFF; float md = kScalarMax;
        if (ds0 < md) { md = ds0; mi = 0; }
        if (ds1 < md) { md = ds1; mi = 1; }
        if (ds2 < md) { md = ds2; mi = 2; }
        return INS_REPLACE | mi << 8;
    }
    float s0 = manifold_score(np, p[1], p[2], p[3]);
    float s1 = manifold_score(np, p[0], p[2], p[3]);
    float s2 = manifold_score(np, p[0], p[1], p[3]);
    float s3 = manifold_score(np, p[0], p[1], p[2]);
    float best = manifold_score(p[0], p[1], p[2], p[3]);
    int bi = -1;
    if (s0 > best) { best = s0; bi = 0; }
    if (s1 > best) { best = s1; bi = 1; }
    if (s2 > best) { best = s2; bi = 2; }
    if (s3 > best) { best = s3; bi = 3; }
    if (bi >= 0) {
        const f3 pb = pick4(p, bi);
        return (distance_sqr(pb, np) < sim2 ? INS_SIMILAR : INS_REPLACE) | bi << 8;
    }
    return INS_NONE;
}

struct CPoint { f3 pivotA, pivotB, normal; float distance; int attachment; };
struct CResult {
    int num;
    CPoint pt[kMaxContacts];
};
DI void cp_swap(CPoint &p) {   // collision_result.hpp:23-35
    f3 t = p.pivotA; p.pivotA = p.pivotB; p.pivotB = t;
    p.normal *= -1.0f;
    if (p.attachment == NA_ON_A) p.attachment = NA_ON_B;
    else if (p.attachment == NA_ON_B) p.attachment = NA_ON_A;
}
DI void sel(CPoint &d, bool c, const CPoint &v) {
    sel(d.pivotA, c, v.pivotA); sel(d.pivotB, c, v.pivotB); sel(d.normal, c, v.normal);
    sel(d.distance, c, v.distance); sel(d.attachment, c, v.attachment);
}
DI void res_add(CResult &r, const CPoint &p) { put4(r.pt, r.num, p); ++r.num; }
DI void res_maybe_add(CResult &r, const CPoint &np) {   // collision_result.cpp:12-33
    f3 piv[kMaxContacts];
#pragma unroll
    for (int i = 0; i < kMaxContacts; ++i) piv[i] = r.pt[i].pivotA;
    int res = insertion_point_index(piv, r.num, np.pivotA);
    if ((res & 0xFF) == INS_NONE) {
#pragma unroll
        for (int i = 0; i < kMaxContacts; ++i) piv[i] = r.pt[i].pivotB;
        res = insertion_point_index(piv, r.num, np.pivotB);
    }
    put4(r.pt, (res & 0xFF) != INS_NONE ? res >> 8 : -1, np);
}

struct Ctx { f3 posA; q4 ornA; f3 posB; q4 ornB; float threshold; };

// src/edyn/collision/collide/collide_box_box.cpp:14-266
DI void collide_box_box(f3 hA, f3 hB, const Ctx &c, CResult &result) {
    const f3 posA = c.posA, posB = c.posB;
    const q4 ornA = c.ornA, ornB = c.ornB;
    f3 axA[3] = {rotate(ornA, mk3(1, 0, 0)), rotate(ornA, mk3(0, 1, 0)), rotate(ornA, mk3(0, 0, 1))};
    f3 axB[3] = {rotate(ornB, mk3(1, 0, 0)), rotate(ornB, mk3(0, 1, 0)), rotate(ornB, mk3(0, 0, 1))};
    float distance = -kScalarMax;
    f3 sep = mk3(0, 0, 0);
    const f3 dAB = posA - posB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f3 dir = axA[i];
        if (dot(dAB, dir) < 0) dir = -dir;
        float projA = dot(posA, dir) - comp(hA, i);
        float projB = box_support_projection(hB, posB, ornB, dir);
        float dist = projA - projB;
        if (dist > distance) { distance = dist; sep = dir; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f3 dir = axB[i];
        if (dot(dAB, dir) < 0) dir = -dir;
        float projA = -box_support_projection(hA, posA, ornA, -dir);
        float projB = dot(posB, dir) + comp(hB, i);
        float dist = projA - projB;
        if (dist > distance) { distance = dist; sep = dir; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f3 dir = cross(axA[i], axB[j]);
            float l2 = length_sqr(dir);
            if (!(l2 > kEps)) continue;
            dir = div_recip(dir, sqrtf(l2));
            if (dot(dAB, dir) < 0) dir *= -1.0f;
            float projA = -box_support_projection(hA, posA, ornA, -dir);
            float projB = box_support_projection(hB, posB, ornB, dir);
            float dist = projA - projB;
            if (dist > distance) { distance = dist; sep = dir; }
        }
    if (distance > c.threshold) return;

    int featA, featB, idxA, idxB;
    float prA, prB;
    support_feature(hA, posA, ornA, mk3(0, 0, 0), -sep, featA, idxA, prA, kSupportTolerance);
    support_feature(hB, posB, ornB, mk3(0, 0, 0), sep, featB, idxB, prB, kSupportTolerance);

    CPoint point;
    point.normal = sep; point.distance = distance; point.attachment = NA_NONE;
    point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);

    if (featA == BF_FACE && featB == BF_FACE) {
        f3 fvA[4], fvB[4];
        face_world(hA, idxA, posA, ornA, fvA);
        f3 fnA = face_normal_world(idxA, ornA);
        face_world(hB, idxB, posB, ornB, fvB);
        f3 fnB = face_normal_world(idxB, ornB);
        point.attachment = NA_ON_B;
        // rolled loops (res_maybe_add is large); the vertex list is rotated by one each pass so that the current vertex is
        // always element 0, and is back in its original order after the fourth
        const f3 fvA0 = fvA[0], fvB0 = fvB[0];
#pragma nounroll
        for (int i = 0; i < 4; ++i) {
            const f3 vb = fvB[0];
            rotate4(fvB);
            if (point_in_quad_prism(fvA, fnA, vb)) {
                f3 pf = project_plane(vb, fvA0, fnA);
                point.pivotA = to_object(pf, posA, ornA);
                point.pivotB = to_object(vb, posB, ornB);
                res_maybe_add(result, point);
            }
        }
#pragma nounroll
        for (int i = 0; i < 4; ++i) {
            const f3 va = fvA[0];
            rotate4(fvA);
            if (point_in_quad_prism(fvB, fnB, va)) {
                f3 pf = project_plane(va, fvB0, fnB);
                point.pivotA = to_object(va, posA, ornA);
                point.pivotB = to_object(pf, posB, ornB);
                res_maybe_add(result, point);
            }
        }
        if (result.num < 4) {
            f3 fc = face_center(hA, idxA, posA, ornA);
            m3 fb = face_basis(idxA, ornA);
            f2 he = face_half_extents(hA, idxA);
#pragma nounroll
            for (int j = 0; j < 4; ++j) {
                const f3 b0w = fvB[0], b1w = fvB[1];
                rotate4(fvB);
                f3 b0 = to_object(b0w, fc, fb), b1 = to_object(b1w, fc, fb);
                float s0, s1;
                int n = intersect_line_aabb({b0.x, b0.z}, {b1.x, b1.z}, -he, he, s0, s1);
#pragma nounroll
                for (int k = 0; k < n; ++k) {
                    const float sk = k ? s1 : s0;
                    if (sk < 0 || sk > 1) continue;
                    f3 q1 = lerp(b0w, b1w, sk);
                    f3 q0 = project_plane(q1, fc, fnA);
                    point.pivotA = to_object(q0, posA, ornA);
                    point.pivotB = to_object(q1, posB, ornB);
                    res_maybe_add(result, point);
                }
            }
        }
    } else if ((featA == BF_FACE && featB == BF_EDGE) || (featB == BF_FACE && featA == BF_EDGE)) {
        const bool fA = featA == BF_FACE;
        f3 fn = fA ? face_normal_world(idxA, ornA) : face_normal_world(idxB, ornB);
        f3 fv[4], ev[2];
        if (fA) { face_world(hA, idxA, posA, ornA, fv); edge_world(hB, idxB, posB, ornB, ev); }
        else { face_world(hB, idxB, posB, ornB, fv); edge_world(hA, idxA, posA, ornA, ev); }
        point.attachment = fA ? NA_ON_A : NA_ON_B;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (point_in_quad_prism(fv, fn, ev[i])) {
                f3 pf = project_plane(ev[i], fv[0], fn);
                point.pivotA = to_object(fA ? pf : ev[i], posA, ornA);
                point.pivotB = to_object(fA ? ev[i] : pf, posB, ornB);
                res_add(result, point);
            }
        if (result.num < 2) {
            f3 fc = fA ? face_center(hA, idxA, posA, ornA) : face_center(hB, idxB, posB, ornB);
            m3 fb = fA ? face_basis(idxA, ornA) : face_basis(idxB, ornB);
            f2 he = fA ? face_half_extents(hA, idxA) : face_half_extents(hB, idxB);
            f3 e0 = to_object(ev[0], fc, fb), e1 = to_object(ev[1], fc, fb);
            float s0, s1;
            int n = intersect_line_aabb({e0.x, e0.z}, {e1.x, e1.z}, -he, he, s0, s1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float si = i ? s1 : s0;
                if (i >= n || si < 0 || si > 1) continue;
                f3 ep = lerp(ev[0], ev[1], si);
                f3 fp = project_plane(ep, fc, sep);
                point.pivotA = to_object(fA ? fp : ep, posA, ornA);
                point.pivotB = to_object(fA ? ep : fp, posB, ornB);
                res_add(result, point);
            }
        }
    } else if (featA == BF_EDGE && featB == BF_EDGE) {
        f3 eA[2], eB[2], c1, c2, c1p = mk3(0, 0, 0), c2p = mk3(0, 0, 0);
        int n = 0;
        edge_world(hA, idxA, posA, ornA, eA);
        edge_world(hB, idxB, posB, ornB, eB);
        closest_segment_segment(eA[0], eA[1], eB[0], eB[1], c1, c2, n, c1p, c2p);
        point.attachment = NA_NONE;
        if (n >= 1) { point.pivotA = to_object(c1, posA, ornA); point.pivotB = to_object(c2, posB, ornB); res_add(result, point); }
        if (n >= 2) { point.pivotA = to_object(c1p, posA, ornA); point.pivotB = to_object(c2p, posB, ornB); res_add(result, point); }
    } else if (featA == BF_FACE && featB == BF_VERTEX) {
        point.pivotB = box_vertex(hB, idxB);
        point.pivotA = to_world(point.pivotB, posB, ornB) + sep * distance;
        point.pivotA = to_object(point.pivotA, posA, ornA);
        point.attachment = NA_ON_A;
        res_add(result, point);
    } else if (featB == BF_FACE && featA == BF_VERTEX) {
        point.pivotA = box_vertex(hA, idxA);
        point.pivotB = to_world(point.pivotA, posA, ornA) - sep * distance;
        point.pivotB = to_object(point.pivotB, posB, ornB);
        point.attachment = NA_ON_B;
        res_add(result, point);
    }
}

// collide_box_plane.cpp:7-56
DI void collide_box_plane(f3 hA, f3 pn, float pc, const Ctx &c, CResult &result) {
    f3 center = pn * pc;
    int featA, idxA; float prA;
    support_feature(hA, c.posA, c.ornA, center, -pn, featA, idxA, prA, kSupportTolerance);
    float distance = -prA;
    if (distance > c.threshold) return;
    f3 verts[4]; int nv;
    if (featA == BF_VERTEX) { verts[0] = box_vertex(hA, idxA); nv = 1; }
    else if (featA == BF_EDGE) { verts[0] = box_vertex(hA, kEdgeIdx[idxA * 2]); verts[1] = box_vertex(hA, kEdgeIdx[idxA * 2 + 1]); nv = 2; }
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) verts[i] = box_vertex(hA, kFaceIdx[idxA * 4 + i]);
        nv = 4;
    }
    CPoint point;
    point.normal = pn; point.attachment = NA_ON_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= nv) break;
        point.pivotA = verts[i];
        f3 pAw = to_world(point.pivotA, c.posA, c.ornA);
        f3 pBw = project_plane(pAw, center, pn);
        point.pivotB = to_object(pBw, c.posB, c.ornB);
        point.distance = dot(pAw - pBw, pn);
        res_add(result, point);
    }
}
// collide_sphere_sphere.cpp:5-27
DI void collide_sphere_sphere(float ra, float rb, const Ctx &c, CResult &result) {
    f3 d = c.posA - c.posB;
    float d2 = length_sqr(d);
    float r = ra + rb + c.threshold;
    if (d2 > r * r) return;
    float dist = sqrtf(d2);
    f3 dn = dist > kEps ? d / dist : mk3(1, 0, 0);
    f3 rA = rotate(conjugate(c.ornA), -dn * ra);
    f3 rB = rotate(conjugate(c.ornB), dn * rb);
    res_add(result, CPoint{rA, rB, dn, dist - ra - rb, NA_NONE});
}
// collide_sphere_plane.cpp:5-20
DI void collide_sphere_plane(float radius, f3 pn, float pc, const Ctx &c, CResult &result) {
    f3 center = pn * pc;
    f3 d = c.posA - center;
    float l = dot(pn, d);
    if (l > radius) return;
    f3 pivotA = rotate(conjugate(c.ornA), -pn * radius);
    f3 pivotB = rotate(conjugate(c.ornB), d - pn * l - center);
    res_add(result, CPoint{pivotA, pivotB, pn, l - radius, NA_ON_B});
}
// collide_sphere_box.cpp:7-55
DI void collide_sphere_box(float radius, f3 hB, const Ctx &c, CResult &result) {
    const q4 ornBc = conjugate(c.ornB);
    const f3 pAB = rotate(ornBc, c.posA - c.posB);
    const q4 oAB = ornBc * c.ornA;
    f3 closest = closest_point_box_outside(hB, pAB);
    f3 nB = pAB - closest;
    float d2 = length_sqr(nB);
    float min_dist = radius + c.threshold;
    if (d2 > min_dist * min_dist) return;
    float cd; int attach = NA_NONE;
    if (d2 <= kEps) {
        cd = -closest_point_box_inside(hB, pAB, closest, nB);
        attach = NA_ON_B;
    } else {
        cd = sqrtf(d2);
        nB = div_recip(nB, cd);
        if (fabsf(nB.x) > 1.0f - kEps || fabsf(nB.y) > 1.0f - kEps || fabsf(nB.z) > 1.0f - kEps) attach = NA_ON_B;
    }
    f3 pivotA_in_B = pAB - nB * radius;
    f3 pivotA = to_object(pivotA_in_B, pAB, oAB);
    res_add(result, CPoint{pivotA, closest, rotate(c.ornB, nB), cd - radius, attach});
}

// ---- capsule pairs. Shape param float4 of a capsule: radius, half_length, axis (0 x, 1 y, 2 z) - shapes/capsule_shape.hpp:17-30.
DI f3 axis_vector(float axis) { return axis == 0.0f ? mk3(1, 0, 0) : (axis == 1.0f ? mk3(0, 1, 0) : mk3(0, 0, 1)); }
DI void capsule_vertices(float4 sh, f3 pos, q4 orn, f3 &v0, f3 &v1) {   // capsule_shape::get_vertices
    const f3 dir = rotate(orn, axis_vector(sh.z));
    v0 = pos + dir * sh.y;
    v1 = pos - dir * sh.y;
}
DI float capsule_support_projection(f3 v0, f3 v1, float radius, f3 dir) { return fmaxf(dot(v0, dir), dot(v1, dir)) + radius; }   // shape_util.cpp:297-300
// collide_capsule_plane.cpp:6-38
DI void collide_capsule_plane(float4 shA, f3 pn, float pc, const Ctx &c, CResult &result) {
    const f3 center = pn * pc;
    f3 cv[2];
    capsule_vertices(shA, c.posA, c.ornA, cv[0], cv[1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float distance = dot(cv[i] - center, pn) - shA.x;
        if (distance > c.threshold) continue;
        const f3 pivotA_world = cv[i] - pn * shA.x;
        res_add(result, CPoint{to_object(pivotA_world, c.posA, c.ornA), project_plane(cv[i], center, pn), pn, distance, NA_ON_B});
    }
}
// collide_capsule_sphere.cpp:10-51
DI void collide_capsule_sphere(float4 shA, float rB, const Ctx &c, CResult &result) {
    f3 v0, v1;
    capsule_vertices(shA, c.posA, c.ornA, v0, v1);
    const f3 v = v1 - v0, w = c.posB - v0;   // closest_point_segment, geom.cpp:12-22
    const float t = clamp_unit(dot(w, v) / dot(v, v));
    const f3 closest = v0 + v * t;
    const float dist_sqr = length_sqr(c.posB - closest);
    const float min_dist = shA.x + rB + c.threshold;
    if (dist_sqr > min_dist * min_dist) return;
    f3 normal = closest - c.posB;
    const float nl2 = length_sqr(normal);
    float distance;
    if (nl2 > kEps) {
        const float nl = sqrtf(nl2);
        normal = div_recip(normal, nl);   // vector3::operator/= multiplies by the reciprocal
        distance = nl - shA.x - rB;
    } else {
        normal = rotate(c.ornA, mk3(0, 0, 1));
        distance = -(shA.x + rB);
    }
    const f3 normalB = rotate(conjugate(c.ornB), normal);
    const f3 pivotA_world = closest - normal * shA.x;
    res_add(result, CPoint{to_object(pivotA_world, c.posA, c.ornA), normalB * rB, normal, distance, NA_NONE});
}
// collide_capsule_capsule.cpp:7-80
DI void collide_capsule_capsule(float4 shA, float4 shB, const Ctx &c, CResult &result) {
    f3 a0, a1, b0, b1;
    capsule_vertices(shA, c.posA, c.ornA, a0, a1);
    capsule_vertices(shB, c.posB, c.ornB, b0, b1);
    f3 cA0, cB0, cA1 = mk3(0, 0, 0), cB1 = mk3(0, 0, 0);
    int n = 0;
    closest_segment_segment(a0, a1, b0, b1, cA0, cB0, n, cA1, cB1);
    const float dist_sqr = length_sqr(cA0 - cB0);
    const float min_dist = shA.x + shB.x + c.threshold;
    if (dist_sqr > min_dist * min_dist) return;
    f3 normal;
    float distance;
    if (dist_sqr > kEps) {
        const float dist = sqrtf(dist_sqr);
        normal = (cA0 - cB0) / dist;
        distance = dist - shA.x - shB.x;
    } else {
        normal = cross(a1 - a0, b1 - b0);
        if (dot(c.posA - c.posB, normal) < 0) normal *= -1.0f;
        if (!try_normalize(normal)) normal = mk3(0, 1, 0);
        distance = -(shA.x + shB.x);
    }
    if (n >= 1) res_add(result, CPoint{to_object(cA0 - normal * shA.x, c.posA, c.ornA), to_object(cB0 + normal * shB.x, c.posB, c.ornB), normal, distance, NA_NONE});
    if (n >= 2) res_add(result, CPoint{to_object(cA1 - normal * shA.x, c.posA, c.ornA), to_object(cB1 + normal * shB.x, c.posB, c.ornB), normal, distance, NA_NONE});
}
// collide_capsule_box.cpp:14-213
DI void collide_capsule_box(float4 shA, f3 hB, const Ctx &c, CResult &result) {
    const f3 posA = mk3(0, 0, 0), posB = c.posB - c.posA;
    const q4 ornA = c.ornA, ornB = c.ornB;
    f3 cv0, cv1;
    capsule_vertices(shA, posA, ornA, cv0, cv1);
    const f3 axB[3] = {rotate(ornB, mk3(1, 0, 0)), rotate(ornB, mk3(0, 1, 0)), rotate(ornB, mk3(0, 0, 1))};
    float distance = -kScalarMax, projection_box = -kScalarMax;
    f3 sep = mk3(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f3 dir = axB[i];
        if (dot(posA - posB, dir) < 0) dir = -dir;
        const float projA = -capsule_support_projection(cv0, cv1, shA.x, -dir);
        const float projB = dot(posB, dir) + comp(hB, i);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_box = projB; sep = dir; }
    }
#pragma nounroll
    for (int i = 0; i < 12; ++i) {
        f3 ev[2];
        edge_world(hB, i, posB, ornB, ev);
        f3 cA, cB, u0, u1;
        int nn = 0;
        closest_segment_segment<false>(ev[0], ev[1], cv0, cv1, cA, cB, nn, u0, u1);
        f3 dir = cA - cB;
        if (!try_normalize(dir)) continue;
        if (dot(posA - posB, dir) < 0) dir *= -1.0f;
        const float projA = -capsule_support_projection(cv0, cv1, shA.x, -dir);
        const float projB = box_support_projection(hB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_box = projB; sep = dir; }
    }
    if (distance > c.threshold) return;
    const float pr0 = dot(cv0, sep), pr1 = dot(cv1, sep);
    const bool is_capsule_edge = fabsf(pr0 - pr1) < kSupportTolerance;
    const f3 contact_origin_box = sep * projection_box;
    int featB, idxB; float fdB;
    support_feature(hB, posB, ornB, contact_origin_box, sep, featB, idxB, fdB, kSupportTolerance);
    const f3 cvx = pr0 < pr1 ? cv0 : cv1;   // the capsule vertex nearer to the box
    CPoint point;
    point.normal = sep; point.distance = distance; point.attachment = NA_NONE;
    point.pivotA = mk3(0, 0, 0); point.pivotB = mk3(0, 0, 0);
    if (featB == BF_FACE) {
        f3 fv[4];
        face_world(hB, idxB, posB, ornB, fv);
        point.attachment = NA_ON_B;
        if (is_capsule_edge) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f3 pointA = k == 0 ? cv0 : cv1;
                if (point_in_quad_prism(fv, sep, pointA)) {
                    point.pivotA = to_object(pointA - sep * shA.x, posA, ornA);
                    point.pivotB = to_object(project_plane(pointA, contact_origin_box, sep), posB, ornB);
                    res_add(result, point);
                }
            }
            if (result.num == 2) return;
            const f3 fc = face_center(hB, idxB, posB, ornB);
            const m3 fb = face_basis(idxB, ornB);
            const f2 he = face_half_extents(hB, idxB);
            const f3 q0 = to_object(cv0, fc, fb), q1 = to_object(cv1, fc, fb);
            float s0, s1;
            const int n = intersect_line_aabb({q0.x, q0.z}, {q1.x, q1.z}, -he, he, s0, s1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float si = i ? s1 : s0;
                if (i >= n || si < 0 || si > 1) continue;
                const f3 edge_pivot = lerp(cv0, cv1, si);
                const f3 face_pivot = project_plane(edge_pivot, fc, sep);
                point.pivotA = to_object(edge_pivot - sep * shA.x, posA, ornA);
                point.pivotB = to_object(face_pivot, posB, ornB);
                res_add(result, point);
            }
        } else {
            const f3 pA = cvx - sep * shA.x;
            const f3 pB = project_plane(pA, contact_origin_box, sep);
            point.pivotA = to_object(pA, posA, ornA);
            point.pivotB = to_object(pB, posB, ornB);
            res_add(result, point);
        }
    } else if (featB == BF_EDGE) {
        f3 ev[2];
        edge_world(hB, idxB, posB, ornB, ev);
        if (is_capsule_edge) {
            f3 cA0, cB0, cA1 = mk3(0, 0, 0), cB1 = mk3(0, 0, 0);
            int n = 0;
            closest_segment_segment(cv0, cv1, ev[0], ev[1], cA0, cB0, n, cA1, cB1);
            if (n >= 1) { point.pivotA = to_object(cA0 - sep * shA.x, posA, ornA); point.pivotB = to_object(cB0, posB, ornB); res_add(result, point); }
            if (n >= 2) { point.pivotA = to_object(cA1 - sep * shA.x, posA, ornA); point.pivotB = to_object(cB1, posB, ornB); res_add(result, point); }
        } else {
            const f3 edge_dir = ev[1] - ev[0], w = cvx - ev[0];   // closest_point_line, geom.cpp:35-44
            const float t = dot(w, edge_dir) / dot(edge_dir, edge_dir);
            const f3 pB = ev[0] + edge_dir * t;
            point.pivotB = to_object(pB, posB, ornB);
            point.pivotA = to_object(cvx - sep * shA.x, posA, ornA);
            res_add(result, point);
        }
    } else {
        point.pivotB = box_vertex(hB, idxB);
        const f3 pB = to_world(point.pivotB, posB, ornB);
        point.pivotA = to_object(pB + sep * distance, posA, ornA);
        res_add(result, point);
    }
}

// Shape pair dispatch incl. swap_collide (include/edyn/collision/collide.hpp:369-374).
// shape param float4: box = half extents xyz; sphere = radius in x; plane = normal xyz, constant w.
DI void collide(int tA, float4 sA, int tB, float4 sB, const Ctx &c, CResult &r) {
    r.num = 0;
    const Ctx sw{c.posB, c.ornB, c.posA, c.ornA, c.threshold};
    bool swapped = false;
    if (tA == SHAPE_BOX && tB == SHAPE_BOX) collide_box_box(from4(sA), from4(sB), c, r);
    else if (tA == SHAPE_BOX && tB == SHAPE_PLANE) collide_box_plane(from4(sA), from4(sB), sB.w, c, r);
    else if (tA == SHAPE_PLANE && tB == SHAPE_BOX) { collide_box_plane(from4(sB), from4(sA), sA.w, sw, r); swapped = true; }
    else if (tA == SHAPE_SPHERE && tB == SHAPE_SPHERE) collide_sphere_sphere(sA.x, sB.x, c, r);
    else if (tA == SHAPE_SPHERE && tB == SHAPE_PLANE) collide_sphere_plane(sA.x, from4(sB), sB.w, c, r);
    else if (tA == SHAPE_PLANE && tB == SHAPE_SPHERE) { collide_sphere_plane(sB.x, from4(sA), sA.w, sw, r); swapped = true; }
    else if (tA == SHAPE_SPHERE && tB == SHAPE_BOX) collide_sphere_box(sA.x, from4(sB), c, r);
    else if (tA == SHAPE_BOX && tB == SHAPE_SPHERE) { collide_sphere_box(sB.x, from4(sA), sw, r); swapped = true; }
    else if (tA == SHAPE_CAPSULE && tB == SHAPE_PLANE) collide_capsule_plane(sA, from4(sB), sB.w, c, r);
    else if (tA == SHAPE_PLANE && tB == SHAPE_CAPSULE) { collide_capsule_plane(sB, from4(sA), sA.w, sw, r); swapped = true; }
    else if (tA == SHAPE_CAPSULE && tB == SHAPE_SPHERE) collide_capsule_sphere(sA, sB.x, c, r);
    else if (tA == SHAPE_SPHERE && tB == SHAPE_CAPSULE) { collide_capsule_sphere(sB, sA.x, sw, r); swapped = true; }
    else if (tA == SHAPE_CAPSULE && tB == SHAPE_CAPSULE) collide_capsule_capsule(sA, sB, c, r);
    else if (tA == SHAPE_CAPSULE && tB == SHAPE_BOX) collide_capsule_box(sA, from4(sB), c, r);
    else if (tA == SHAPE_BOX && tB == SHAPE_CAPSULE) { collide_capsule_box(sB, from4(sA), sw, r); swapped = true; }
    if (swapped) {
#pragma unroll
        for (int i = 0; i < kMaxContacts; ++i) if (i < r.num) cp_swap(r.pt[i]);
    }
}

}  // namespace dc
