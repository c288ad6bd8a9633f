// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// Shapes on the hot path (box, sphere, plane) and their derived quantities, restated from
//   /root/reference/src/edyn/shapes/box_shape.cpp:24-225 (support_projection/feature, vertex/edge/face tables)
//   /root/reference/include/edyn/shapes/box_shape.hpp:22-47 (edge_indices, face_indices)
//   /root/reference/src/edyn/util/aabb_util.cpp:9-70 (plane_aabb, box_aabb, sphere_aabb)
//   /root/reference/src/edyn/dynamics/moment_of_inertia.cpp:11-21,163-181 (box, sphere inertia)
#pragma once
#include "ogeom.hpp"

namespace orc {

enum shape_type : int { SHAPE_NONE = 0, SHAPE_BOX = 1, SHAPE_SPHERE = 2, SHAPE_PLANE = 3, SHAPE_CAPSULE = 4, SHAPE_CYLINDER = 5, SHAPE_POLYHEDRON = 6 };
enum box_feature : int { BF_VERTEX = 0, BF_EDGE = 1, BF_FACE = 2 };

struct shape {
    int type = SHAPE_NONE;
    vec3 half_extents{0, 0, 0};   // box
    float radius = 0;             // sphere
    vec3 normal{0, 1, 0};         // plane
    float constant = 0;           // plane
    float half_length = 0;        // capsule, cylinder (radius above); shapes/capsule_shape.hpp:17-30, cylinder_shape.hpp:22-25
    int axis = 0;                 // capsule, cylinder: coordinate_axis x, y, z
    int mesh = -1;                // polyhedron: index into mesh_registry() (opolyhedron.hpp); shapes/polyhedron_shape.hpp:11-43
};
inline vec3 coordinate_axis_vector(int axis) { return axis == 0 ? vec3{1, 0, 0} : (axis == 1 ? vec3{0, 1, 0} : vec3{0, 0, 1}); }   // math/coordinate_axis.hpp:23-44
inline void capsule_vertices(const shape &s, vec3 pos, quat orn, vec3 out[2]) {   // capsule_shape::get_vertices
    const vec3 dir = rotate(orn, coordinate_axis_vector(s.axis));
    out[0] = pos + dir * s.half_length;
    out[1] = pos - dir * s.half_length;
}
inline float capsule_support_projection(const vec3 v[2], float radius, vec3 dir) {   // shape_util.cpp:297-305
    return std::max(dot(v[0], dir), dot(v[1], dir)) + radius;
}

static const int kBoxEdgeIndices[24] = {0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 0, 4, 1, 7, 2, 6, 3, 5};
static const int kBoxFaceIndices[24] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 3, 5, 4, 1, 7, 6, 2, 0, 4, 7, 1, 3, 2, 6, 5};

inline vec3 box_vertex(vec3 h, int i) {
    static const vec3 mult[8] = {{1, 1, 1}, {1, -1, 1}, {1, -1, -1}, {1, 1, -1},
                                 {-1, 1, 1}, {-1, 1, -1}, {-1, -1, -1}, {-1, -1, 1}};
    return h * mult[i];
}
inline vec3 box_face_normal(int f) {
    static const vec3 n[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    return n[f];
}
inline vec3 box_face_tangent(int f) {
    static const vec3 t[6] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
    return t[f];
}
inline vec3 support_point_box(vec3 h, vec3 dir) {   // shape_util.cpp:40-46
    return {dir.x > 0 ? h.x : -h.x, dir.y > 0 ? h.y : -h.y, dir.z > 0 ? h.z : -h.z};
}
inline float box_support_projection(vec3 h, vec3 pos, quat orn, vec3 dir) {   // box_shape.cpp:24-28
    vec3 ld = rotate(conjugate(orn), dir);
    vec3 pt = support_point_box(h, ld);
    return dot(pos, dir) + dot(pt, ld);
}
inline int box_support_face_index(vec3 dir) {
    size_t mi = max_index_abs(dir);
    return dir[mi] < 0 ? (int)mi * 2 + 1 : (int)mi * 2;
}
inline int box_edge_index(int v0, int v1) {
    for (int i = 0; i < 12; ++i) {
        int a = kBoxEdgeIndices[i * 2], b = kBoxEdgeIndices[i * 2 + 1];
        if ((a == v0 && b == v1) || (b == v0 && a == v1)) return i;
    }
    return -1;
}
// box_shape.cpp:30-96 (object-space direction)
inline void box_support_feature_local(vec3 h, vec3 dir, int &feature, int &feature_index, float &projection,
                                      float threshold) {
    int face = box_support_face_index(dir);
    float proj[4];
    int vidx[4];
    int idx[4] = {0, 0, 0, 0};
    int count = 1, maxi = 0;
    projection = -kScalarMax;
    for (int i = 0; i < 4; ++i) {
        int vi = kBoxFaceIndices[face * 4 + i];
        vidx[i] = vi;
        float p = dot(box_vertex(h, vi), dir);
        proj[i] = p;
        if (p > projection) { projection = p; idx[0] = i; maxi = i; }
    }
    for (int i = 0; i < 4; ++i)
        if (i != maxi && proj[i] > projection - threshold) idx[count++] = i;
    if (count == 1) {
        feature = BF_VERTEX; feature_index = vidx[idx[0]];
    } else if (count == 2) {
        feature = BF_EDGE; feature_index = box_edge_index(vidx[idx[0]], vidx[idx[1]]);
    } else if (count == 3) {
        feature = BF_EDGE;
        float p0 = proj[idx[0]], p1 = proj[idx[1]], p2 = proj[idx[2]];
        if (p0 <= p1 && p0 <= p2) feature_index = box_edge_index(vidx[idx[1]], vidx[idx[2]]);
        else if (p1 <= p0 && p1 <= p2) feature_index = box_edge_index(vidx[idx[0]], vidx[idx[2]]);
        else feature_index = box_edge_index(vidx[idx[0]], vidx[idx[1]]);
    } else {
        feature = BF_FACE; feature_index = face;
    }
}
// box_shape.cpp:98-105 (world-space axis)
inline void box_support_feature(vec3 h, vec3 pos, quat orn, vec3 axis_pos, vec3 axis_dir, int &feature,
                                int &feature_index, float &projection, float threshold) {
    vec3 ld = rotate(conjugate(orn), axis_dir);
    box_support_feature_local(h, ld, feature, feature_index, projection, threshold);
    projection += dot(pos - axis_pos, axis_dir);
}
inline void box_face_world(vec3 h, int f, vec3 pos, quat orn, vec3 out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = to_world(box_vertex(h, kBoxFaceIndices[f * 4 + i]), pos, orn);
}
inline void box_edge_world(vec3 h, int e, vec3 pos, quat orn, vec3 out[2]) {
    out[0] = to_world(box_vertex(h, kBoxEdgeIndices[e * 2]), pos, orn);
    out[1] = to_world(box_vertex(h, kBoxEdgeIndices[e * 2 + 1]), pos, orn);
}
inline vec3 box_face_normal_world(int f, quat orn) { return rotate(orn, box_face_normal(f)); }
inline vec3 box_face_center(vec3 h, int f, vec3 pos, quat orn) {
    vec3 n = box_face_normal_world(f, orn);
    float e = h[f / 2];
    return pos + n * e;
}
inline mat3 box_face_basis(int f, quat orn) {   // box_shape.cpp:208-213
    vec3 y = box_face_normal(f), x = box_face_tangent(f), z = cross(x, y);
    return mat3_columns(rotate(orn, x), rotate(orn, y), rotate(orn, z));
}
inline vec2 box_face_half_extents(vec3 h, int f) {   // box_shape.cpp:215-225
    if (f == 0 || f == 1) return {h.z, h.y};
    if (f == 2 || f == 3) return {h.x, h.z};
    return {h.y, h.x};
}

constexpr float kPlaneAabbHalfExtent = 99999.0f;
inline aabb plane_aabb(vec3 n, float c) {
    vec3 umin{-1, -1, -1}, umax{1, 1, 1};
    if (n == vec3{1, 0, 0}) umax = {0, 1, 1};
    else if (n == vec3{-1, 0, 0}) umin = {0, -1, -1};
    else if (n == vec3{0, 1, 0}) umax = {1, 0, 1};
    else if (n == vec3{0, -1, 0}) umin = {-1, 0, -1};
    else if (n == vec3{0, 0, 1}) umax = {1, 1, 0};
    else if (n == vec3{0, 0, -1}) umin = {-1, -1, 0};
    vec3 pw = n * c;
    return {umin * kPlaneAabbHalfExtent + pw, umax * kPlaneAabbHalfExtent + pw};
}
inline aabb box_aabb(vec3 h, vec3 pos, quat orn) {
    aabb r{pos, pos};
    mat3 basis = to_mat3(orn);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float e = basis[i][j] * -h[j];
            float f = -e;
            if (e < f) { r.min[i] += e; r.max[i] += f; }
            else { r.min[i] += f; r.max[i] += e; }
        }
    return r;
}
inline aabb sphere_aabb(float radius, vec3 pos) {
    return {{pos.x - radius, pos.y - radius, pos.z - radius}, {pos.x + radius, pos.y + radius, pos.z + radius}};
}
inline aabb cylinder_aabb(const shape &s, vec3 pos, quat orn);   // ocylinder.hpp
inline mat3 cylinder_inertia(const shape &s, float mass);
inline aabb polyhedron_aabb(const shape &s, vec3 pos, quat orn);   // opolyhedron.hpp
inline mat3 polyhedron_inertia(const shape &s, float mass);
inline aabb shape_aabb(const shape &s, vec3 pos, quat orn) {
    switch (s.type) {
    case SHAPE_CYLINDER: return cylinder_aabb(s, pos, orn);
    case SHAPE_POLYHEDRON: return polyhedron_aabb(s, pos, orn);
    case SHAPE_BOX: return box_aabb(s.half_extents, pos, orn);
    case SHAPE_SPHERE: return sphere_aabb(s.radius, pos);
    case SHAPE_PLANE: return plane_aabb(s.normal, s.constant);
    case SHAPE_CAPSULE: {   // aabb_util.cpp:81-88
        const vec3 v = rotate(orn, coordinate_axis_vector(s.axis)) * s.half_length;
        const vec3 p0 = pos - v, p1 = pos + v, off{s.radius, s.radius, s.radius};
        return {vec3{std::min(p0.x, p1.x), std::min(p0.y, p1.y), std::min(p0.z, p1.z)} - off,
                vec3{std::max(p0.x, p1.x), std::max(p0.y, p1.y), std::max(p0.z, p1.z)} + off};
    }
    default: return {pos, pos};
    }
}

inline mat3 moment_of_inertia(const shape &s, float mass) {
    if (s.type == SHAPE_CYLINDER) return cylinder_inertia(s, mass);
    if (s.type == SHAPE_POLYHEDRON) return polyhedron_inertia(s, mass);
    if (s.type == SHAPE_BOX) {
        vec3 ext = s.half_extents * 2.0f;
        vec3 d = 1.0f / 12.0f * mass * vec3{ext.y * ext.y + ext.z * ext.z, ext.z * ext.z + ext.x * ext.x,
                                            ext.x * ext.x + ext.y * ext.y};
        return diagonal(d);
    }
    if (s.type == SHAPE_SPHERE) {
        float i = 0.4f * mass * s.radius * s.radius;
        return {{{1 * i, 0 * i, 0 * i}, {0 * i, 1 * i, 0 * i}, {0 * i, 0 * i, 1 * i}}};
    }
    if (s.type == SHAPE_CAPSULE) {   // moment_of_inertia.cpp:65-90,171-173, shape_volume.cpp:10-16
        const float kPiF = 3.1415926535897932384626433832795029f;
        const float len = s.half_length * 2, radius = s.radius;
        const float cyl_vol = kPiF * radius * radius * len;
        const float sph_vol = kPiF * radius * radius * radius * 4.0f / 3.0f;
        const float total_vol = cyl_vol + sph_vol;
        const float cyl_mass = mass * cyl_vol / total_vol, sph_mass = mass * sph_vol / total_vol;
        const float cyl_xx = 0.5f * cyl_mass * radius * radius;   // moment_of_inertia_solid_cylinder :28-46
        const float cyl_yy = 1.0f / 12.0f * cyl_mass * (3.0f * radius * radius + len * len);
        const float sph_inertia = 0.4f * sph_mass * radius * radius;
        // moment_of_inertia_solid_cylinder returns its vector already permuted for the axis, and the capsule formula reads
        // .x as the axial and .y as the transverse term whatever the axis is (:77-81) - so for axis y / z the cylinder's
        // two terms arrive swapped / equal. Reproduced as is: this is the inertia the reference simulates with.
        const vec3 cyl = s.axis == 0 ? vec3{cyl_xx, cyl_yy, cyl_yy} : (s.axis == 1 ? vec3{cyl_yy, cyl_xx, cyl_yy} : vec3{cyl_yy, cyl_yy, cyl_xx});
        const float xx = sph_inertia + cyl.x;
        const float yy_zz = sph_inertia + sph_mass * square(4.0f * len + 3.0f * radius) / 64.0f + cyl.y;
        return diagonal(s.axis == 0 ? vec3{xx, yy_zz, yy_zz} : (s.axis == 1 ? vec3{yy_zz, xx, yy_zz} : vec3{yy_zz, yy_zz, xx}));
    }
    return diagonal({kScalarMax, kScalarMax, kScalarMax});
}

}  // namespace orc
