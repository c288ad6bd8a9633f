// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// polyhedron_shape: the convex mesh and its derived data, the per-body rotated mesh, support polygons and the
// closest-feature routines of its pairs, restated from
//   /root/reference/include/edyn/shapes/convex_mesh.hpp:17-198, src/edyn/shapes/convex_mesh.cpp:10-278
//   /root/reference/src/edyn/sys/update_rotated_meshes.cpp:12-52 (update_rotated_mesh)
//   /root/reference/src/edyn/util/shape_util.cpp:48-79 (polyhedron_support_projection), :89-190 (split_hull_edge, calculate_convex_hull),
//       :208-212 (is_triangle_ccw), :223-282 (closest_point_convex_polygon), :351-391 (mesh_centroid)
//   /root/reference/include/edyn/util/shape_util.hpp:201-273 (support_polygon, point_cloud_support_polygon)
//   /root/reference/src/edyn/math/geom.cpp:756-760 (make_tangent_basis), :800-845 (intersect_segments), :1140-1161 (point_in_polygonal_prism),
//       :1345-1352 (edges_generate_minkowski_face)
//   /root/reference/src/edyn/dynamics/moment_of_inertia.cpp:93-157 (moment_of_inertia_polyhedron), src/edyn/util/aabb_util.cpp:141-151
//   /root/reference/src/edyn/collision/collide/collide_polyhedron_{plane,sphere,box,polyhedron,capsule,cylinder}.cpp
// Support polygons are held in fixed arrays of kPolyMax vertices (the reference uses std::vector): a face with more vertices than
// that within the support tolerance is outside what the drop-in supports (edynhip_create_convex_mesh rejects such meshes).
#pragma once
// (included by ocollide.hpp after ocylinder.hpp)
#include <memory>
#include <vector>

namespace orc {

constexpr int kPolyMax = 32;
constexpr float kRelevantDirectionTolerance = 0.0006f;   // convex_mesh_relevant_direction_tolerance, config/constants.hpp

struct ConvexMesh {
    std::vector<vec3> vertices, normals, relevant_normals, edge_vertices, edge_normals;
    std::vector<uint32_t> indices, edges, faces, edge_faces, relevant_faces, relevant_edges, neighbors_start, neighbor_indices;
    size_t num_edges() const { return edges.size() / 2; }
    size_t num_faces() const { return faces.size() / 2; }
    static vec3 centroid(const std::vector<vec3> &vertices, const std::vector<uint32_t> &indices, const std::vector<uint32_t> &faces) {   // shape_util.cpp:351-391
        vec3 center{0, 0, 0};
        float volume = 0;
        for (size_t i = 0; i < faces.size(); i += 2) {
            const uint32_t first = faces[i], count = faces[i + 1];
            const vec3 v0 = vertices[indices[first]];
            for (size_t j = 1; j < size_t(count - 1); ++j) {
                const vec3 v1 = vertices[indices[first + j]], v2 = vertices[indices[first + j + 1]];
                const vec3 normal = cross(v1 - v0, v2 - v1);
                const float tet_vol = dot(v0, normal);
                volume += tet_vol;
                const vec3 vx{v0.x + v1.x, v1.x + v2.x, v2.x + v0.x}, vy{v0.y + v1.y, v1.y + v2.y, v2.y + v0.y}, vz{v0.z + v1.z, v1.z + v2.z, v2.z + v0.z};
                const vec3 w{length_sqr(vx), length_sqr(vy), length_sqr(vz)};
                center += normal * w;
            }
        }
        volume /= 6;
        center /= 24 * 2 * volume;
        return center;
    }
    void initialize() {   // convex_mesh.cpp:10-30
        const vec3 c = centroid(vertices, indices, faces);
        for (auto &v : vertices) v -= c;
        calculate_normals(); calculate_edges(); calculate_neighbors(); calculate_relevant_faces(); calculate_relevant_edges();
    }
    void calculate_normals() {   // :86-121
        normals.clear();
        for (size_t i = 0; i < num_faces(); ++i) {
            const uint32_t first = faces[i * 2], count = faces[i * 2 + 1];
            const vec3 v0 = vertices[indices[first]], v1 = vertices[indices[first + 1]];
            vec3 normal{0, 0, 0};
            for (size_t j = 1; j < count; ++j) {
                const vec3 v2 = vertices[indices[first + j]], v3 = vertices[indices[first + (j + 1) % count]];
                vec3 n = cross(v1 - v0, v3 - v2);
                if (try_normalize(n)) { normal = n; break; }
            }
            if (normal == vec3{0, 0, 0}) normal = vec3{0, 1, 0};
            normals.push_back(normal);
        }
    }
    void calculate_edges() {   // :123-173
        edges.clear(); edge_faces.clear(); edge_vertices.clear(); edge_normals.clear();
        for (size_t face_idx = 0; face_idx < num_faces(); ++face_idx) {
            const uint32_t first = faces[face_idx * 2], count = faces[face_idx * 2 + 1];
            for (size_t k = 0; k < count; ++k) {
                const uint32_t i0 = indices[first + k], i1 = indices[first + (k + 1) % count];
                bool contains = false;
                for (size_t e = 0; e < num_edges(); ++e) {
                    if ((edges[2 * e] == i0 && edges[2 * e + 1] == i1) || (edges[2 * e] == i1 && edges[2 * e + 1] == i0)) {
                        contains = true;
                        edge_faces[e * 2 + 1] = (uint32_t)face_idx;
                        edge_normals[e * 2 + 1] = normals[face_idx];
                        break;
                    }
                }
                if (!contains) {
                    edges.push_back(i0); edges.push_back(i1);
                    edge_faces.push_back((uint32_t)face_idx); edge_faces.push_back(0xFFFFFFFFu);
                    edge_vertices.push_back(vertices[i0]); edge_vertices.push_back(vertices[i1]);
                    edge_normals.push_back(normals[face_idx]); edge_normals.push_back(vec3{0, 0, 0});
                }
            }
        }
    }
    void calculate_neighbors() {   // :175-194
        neighbors_start.clear(); neighbor_indices.clear();
        neighbors_start.push_back(0);
        uint32_t count = 0;
        for (size_t v = 0; v < vertices.size(); ++v) {
            for (size_t e = 0; e < num_edges(); ++e)
                if (edges[2 * e] == v || edges[2 * e + 1] == v) { neighbor_indices.push_back(edges[2 * e] == v ? edges[2 * e + 1] : edges[2 * e]); ++count; }
            neighbors_start.push_back(count);
        }
    }
    void calculate_relevant_faces() {   // :196-211
        relevant_faces.clear(); relevant_normals.clear();
        for (size_t f = 0; f < num_faces(); ++f) {
            bool found = false;
            for (uint32_t other : relevant_faces)
                if (!(dot(normals[f], normals[other]) < 1.0f - kRelevantDirectionTolerance)) { found = true; break; }
            if (!found) { relevant_faces.push_back((uint32_t)f); relevant_normals.push_back(normals[f]); }
        }
    }
    vec3 edge_direction(size_t e) const { return edge_vertices[2 * e + 1] - edge_vertices[2 * e]; }
    void calculate_relevant_edges() {   // :213-230
        relevant_edges.clear();
        for (size_t e = 0; e < num_edges(); ++e) {
            const vec3 edge = normalize(edge_direction(e));
            bool found = false;
            for (uint32_t other : relevant_edges)
                if (!(std::fabs(dot(edge, normalize(edge_direction(other)))) < 1.0f - kRelevantDirectionTolerance)) { found = true; break; }
            if (!found) relevant_edges.push_back((uint32_t)e);
        }
    }
    mat3 inertia(float mass) const {   // moment_of_inertia.cpp:93-157
        float volume = 0, xx = 0, yy = 0, zz = 0, yz = 0, zx = 0, xy = 0;
        for (size_t i = 0; i < num_faces(); ++i) {
            const uint32_t first = faces[i * 2], count = faces[i * 2 + 1];
            const vec3 v0 = vertices[indices[first]];
            for (size_t j = 1; j < size_t(count - 1); ++j) {
                const vec3 v1 = vertices[indices[first + j]], v2 = vertices[indices[first + j + 1]];
                const float pd_vol = dot(v0, cross(v1, v2));   // triple_product
                volume += pd_vol;
                const vec3 v3 = v0 + v1 + v2;
                xx += pd_vol * (v0.x * v0.x + v1.x * v1.x + v2.x * v2.x + v3.x * v3.x);
                yy += pd_vol * (v0.y * v0.y + v1.y * v1.y + v2.y * v2.y + v3.y * v3.y);
                zz += pd_vol * (v0.z * v0.z + v1.z * v1.z + v2.z * v2.z + v3.z * v3.z);
                yz += pd_vol * (v0.y * v0.z + v1.y * v1.z + v2.y * v2.z + v3.y * v3.z);
                zx += pd_vol * (v0.z * v0.x + v1.z * v1.x + v2.z * v2.x + v3.z * v3.x);
                xy += pd_vol * (v0.x * v0.y + v1.x * v1.y + v2.x * v2.y + v3.x * v3.y);
            }
        }
        const float density = mass / (volume / 6.0f);
        const float r = density / 120.0f;
        const float Iyz = yz * r, Izx = zx * r, Ixy = xy * r, Ixx = (yy + zz) * r, Iyy = (zz + xx) * r, Izz = (xx + yy) * r;
        return {{{Ixx, Ixy, Izx}, {Ixy, Iyy, Iyz}, {Izx, Iyz, Izz}}};
    }
};
struct RotatedMesh {
    std::vector<vec3> vertices, normals, relevant_normals, edge_vertices, edge_normals;
    void update(const ConvexMesh &m, quat orn) {   // update_rotated_meshes.cpp:12-52
        vertices.resize(m.vertices.size()); normals.resize(m.normals.size()); relevant_normals.resize(m.relevant_normals.size());
        edge_vertices.resize(m.edge_vertices.size()); edge_normals.resize(m.edge_normals.size());
        for (size_t i = 0; i < m.vertices.size(); ++i) vertices[i] = rotate(orn, m.vertices[i]);
        for (size_t i = 0; i < m.edge_vertices.size(); ++i) edge_vertices[i] = rotate(orn, m.edge_vertices[i]);
        for (size_t i = 0; i < m.normals.size(); ++i) normals[i] = rotate(orn, m.normals[i]);
        for (size_t i = 0; i < m.relevant_normals.size(); ++i) relevant_normals[i] = rotate(orn, m.relevant_normals[i]);
        for (size_t i = 0; i < m.edge_normals.size(); ++i) edge_normals[i] = rotate(orn, m.edge_normals[i]);
    }
};
inline int g_poly_flags = 0;   // test bookkeeping: bit 0 set by collide_polyhedron_polyhedron when the reference's behaviour is undefined (see there)
inline std::vector<std::shared_ptr<ConvexMesh>> &mesh_registry() { static std::vector<std::shared_ptr<ConvexMesh>> r; return r; }
inline aabb polyhedron_aabb(const shape &s, vec3 pos, quat orn) {   // aabb_util.cpp:141-164,195-197; update_aabbs.cpp:22-32 gives the same box (x + pos is monotonic)
    aabb box{vec3{kScalarMax, kScalarMax, kScalarMax}, vec3{-kScalarMax, -kScalarMax, -kScalarMax}};
    for (const vec3 &p : mesh_registry()[s.mesh]->vertices) { const vec3 w = to_world(p, pos, orn); box.min = vmin(box.min, w); box.max = vmax(box.max, w); }
    return box;
}
inline mat3 polyhedron_inertia(const shape &s, float mass) { return mesh_registry()[s.mesh]->inertia(mass); }

// ---- views used by the routines (the same accessors exist on the device over its flat arrays)
struct MeshView {
    const ConvexMesh *m;
    int nv() const { return (int)m->vertices.size(); }
    int ne() const { return (int)m->num_edges(); }
    int nrf() const { return (int)m->relevant_faces.size(); }
    int nre() const { return (int)m->relevant_edges.size(); }
    vec3 vertex(int i) const { return m->vertices[i]; }
    vec3 normal(int f) const { return m->normals[f]; }
    int relevant_face(int k) const { return (int)m->relevant_faces[k]; }
    int relevant_edge(int k) const { return (int)m->relevant_edges[k]; }
    int first_vertex_index(int f) const { return (int)m->indices[m->faces[2 * f]]; }
    vec3 edge_vertex(int k) const { return m->edge_vertices[k]; }     // k = 2 * edge + {0, 1}
    int edge_face(int k) const { return (int)m->edge_faces[k]; }
    int edge_vertex_index(int k) const { return (int)m->edges[k]; }
    int neighbors_start(int v) const { return (int)m->neighbors_start[v]; }
    int neighbor(int k) const { return (int)m->neighbor_indices[k]; }
};
struct RotView {
    const RotatedMesh *r;
    vec3 vertex(int i) const { return r->vertices[i]; }
    vec3 relevant_normal(int k) const { return r->relevant_normals[k]; }
    vec3 edge_vertex(int k) const { return r->edge_vertices[k]; }
    vec3 edge_normal(int k) const { return r->edge_normals[k]; }
};
struct PolySh { MeshView mesh; RotView rot; };

// ---- geometry
inline vec2 to_vector2_xz(vec3 v) { return {v.x, v.z}; }
inline vec3 to_vector3_xz(vec2 v) { return {v.x, 0, v.y}; }
inline float perp_product(vec2 v, vec2 w) { return v.x * w.y - v.y * w.x; }
inline vec2 lerp(vec2 a, vec2 b, float s) { return a * (1.0f - s) + b * s; }
inline mat3 make_tangent_basis(vec3 n) { vec3 t, u; plane_space(n, t, u); return mat3_columns(t, n, u); }   // geom.cpp:756-760
inline bool is_triangle_ccw(vec2 v0, vec2 v1, vec2 v2) { return dot(v2 - v0, orthogonal(v1 - v0)) > 0; }   // shape_util.cpp:208-212
inline int intersect_segments(vec2 p0, vec2 p1, vec2 q0, vec2 q1, float &s0, float &t0, float &s1, float &t1) {   // geom.cpp:804-845
    const vec2 dp = p1 - p0, dq = q1 - q0, e = q0 - p0;
    const float denom = perp_product(dp, dq);
    if (std::fabs(denom) > kEps) {
        const float denom_inv = 1.0f / denom;
        s0 = perp_product(e, dq) * denom_inv;
        t0 = perp_product(e, dp) * denom_inv;
        return s0 < 0 || s0 > 1 || t0 < 0 || t0 > 1 ? 0 : 1;
    }
    if (std::fabs(perp_product(e, dp)) < kEps) {
        const float denom_p = 1.0f / dot(dp, dp), denom_q = 1.0f / dot(dq, dq);
        s0 = dot(q0 - p0, dp) * denom_p;
        s1 = dot(q1 - p0, dp) * denom_p;
        if ((s0 < 0 && s1 < 0) || (s0 > 1 && s1 > 1)) return 0;
        s0 = clamp_unit(s0); s1 = clamp_unit(s1);
        t0 = clamp_unit(dot(p0 - q0, dq) * denom_q);
        t1 = clamp_unit(dot(p1 - q0, dq) * denom_q);
        return std::fabs(s1 - s0) < kEps ? 1 : 2;
    }
    return 0;
}
inline bool edges_generate_minkowski_face(vec3 A, vec3 B, vec3 C_neg, vec3 D_neg, vec3 B_x_A, vec3 D_x_C) {   // geom.cpp:1345-1352
    const float CBA = -dot(C_neg, B_x_A), DBA = -dot(D_neg, B_x_A), ADC = dot(A, D_x_C), BDC = dot(B, D_x_C);
    return CBA * DBA < 0 && ADC * BDC < 0 && CBA * BDC > 0;
}
template <class V>
inline float polyhedron_support_projection(const V &verts, const MeshView &mesh, vec3 dir) {   // shape_util.cpp:48-79 (hill climbing over the vertex adjacency)
    int v_idx = 0;
    float max_proj = dot(verts.vertex(0), dir);
    for (;;) {
        const int n0 = mesh.neighbors_start(v_idx), n1 = mesh.neighbors_start(v_idx + 1);
        bool done = true;
        for (int i = n0; i < n1; ++i) {
            const int nv_idx = mesh.neighbor(i);
            const float proj = dot(verts.vertex(nv_idx), dir);
            if (proj > max_proj) { max_proj = proj; v_idx = nv_idx; done = false; }
        }
        if (done) break;
    }
    return max_proj;
}

// support polygon (shape_util.hpp:201-273) in fixed storage
struct SupportPolygon {
    vec3 vertices[kPolyMax];
    vec2 plane_vertices[kPolyMax];
    int hull[kPolyMax + 1];
    int nverts = 0, nhull = 0;
    vec3 origin;
    mat3 basis;
};
// split_hull_edge (shape_util.cpp:89-124), the recursion unrolled over an explicit stack: same insertions in the same order
inline void hull_insert(SupportPolygon &p, int at, int idx) {
    if (p.nhull > kPolyMax) return;
    for (int k = p.nhull; k > at; --k) p.hull[k] = p.hull[k - 1];
    p.hull[at] = idx; ++p.nhull;
}
inline int split_hull_edge(SupportPolygon &p, int i0_in, int i1_in, float tolerance) {
    struct Frame { int i0, i1, stage, n1; };
    Frame stack[kPolyMax + 2];
    int sp = 0, ret = 0;
    stack[sp++] = Frame{i0_in, i1_in, 0, 0};
    while (sp > 0) {
        Frame &f = stack[sp - 1];
        if (f.stage == 0) {
            const vec2 v0 = p.plane_vertices[p.hull[f.i0]], v1 = p.plane_vertices[p.hull[f.i1]];
            const vec2 dir = -orthogonal(v1 - v0);
            float max_proj = -kScalarMax; int idx = 0;
            for (int i = 0; i < p.nverts; ++i) {
                const float proj = dot(p.plane_vertices[i], dir);
                if (proj > max_proj) { max_proj = proj; idx = i; }
            }
            if (dot(p.plane_vertices[idx] - v0, dir) > tolerance && p.nhull <= kPolyMax && sp < kPolyMax + 1) {
                hull_insert(p, f.i1, idx);
                f.stage = 1;
                stack[sp++] = Frame{f.i0, f.i1, 0, 0};
            } else { ret = 0; --sp; }
        } else if (f.stage == 1) {
            f.n1 = ret;
            f.i1 += f.n1;
            f.stage = 2;
            const int a = f.i1, b = f.i1 + 1;
            stack[sp++] = Frame{a, b, 0, 0};
        } else {
            ret = 1 + f.n1 + ret;
            --sp;
        }
    }
    return ret;
}
inline void calculate_convex_hull(SupportPolygon &p, float tolerance) {   // shape_util.cpp:126-190
    const int n = p.nverts;
    p.nhull = 0;
    if (n <= 3) {
        if (n == 3) {
            if (is_triangle_ccw(p.plane_vertices[0], p.plane_vertices[1], p.plane_vertices[2])) { p.hull[0] = 0; p.hull[1] = 1; p.hull[2] = 2; }
            else { p.hull[0] = 2; p.hull[1] = 1; p.hull[2] = 0; }
            p.nhull = 3;
        } else if (n == 2) { p.hull[0] = 0; p.hull[1] = 1; p.nhull = 2; }
        else { p.hull[0] = 0; p.nhull = 1; }
        return;
    }
    vec2 pt_min{kScalarMax, kScalarMax}, pt_max{-kScalarMax, -kScalarMax};
    int pt_min_idx = 0, pt_max_idx = 0;
    for (int i = 0; i < n; ++i) {
        const vec2 q = p.plane_vertices[i];
        if (q.x < pt_min.x) { pt_min = q; pt_min_idx = i; }
        if (q.x > pt_max.x) { pt_max = q; pt_max_idx = i; }
    }
    if (pt_max.x - pt_min.x < tolerance) {   // a vertical sliver
        pt_min = vec2{kScalarMax, kScalarMax}; pt_max = vec2{-kScalarMax, -kScalarMax};
        for (int i = 0; i < n; ++i) {
            const vec2 q = p.plane_vertices[i];
            if (q.y < pt_min.y) { pt_min = q; pt_min_idx = i; }
            if (q.y > pt_max.y) { pt_max = q; pt_max_idx = i; }
        }
        p.hull[0] = pt_max_idx; p.hull[1] = pt_min_idx; p.nhull = 2;
        return;
    }
    p.hull[0] = pt_max_idx; p.hull[1] = pt_min_idx; p.hull[2] = pt_max_idx; p.nhull = 3;
    int i1 = 1;
    const int num_splits = split_hull_edge(p, 0, i1, tolerance);
    i1 += num_splits;
    split_hull_edge(p, i1, i1 + 1, tolerance);
    --p.nhull;   // hull.pop_back()
}
template <class V>
inline void point_cloud_support_polygon(SupportPolygon &polygon, const V &verts, int count, vec3 offset, vec3 dir, float projection, bool positive_side, float tolerance) {
    polygon.origin = dir * projection;
    polygon.basis = make_tangent_basis(dir);
    polygon.nverts = 0;
    const bool zero_offset = offset == vec3{0, 0, 0};
    for (int i = 0; i < count; ++i) {
        const vec3 vertex_world = zero_offset ? verts.vertex(i) : verts.vertex(i) + offset;
        const bool in_boundary = positive_side ? dot(vertex_world, dir) < projection + tolerance : dot(vertex_world, dir) > projection - tolerance;
        if (!in_boundary || polygon.nverts >= kPolyMax) continue;
        polygon.vertices[polygon.nverts] = vertex_world;
        polygon.plane_vertices[polygon.nverts] = to_vector2_xz(to_object(vertex_world, polygon.origin, polygon.basis));
        ++polygon.nverts;
    }
    calculate_convex_hull(polygon, 0.001f);
}
inline bool point_in_polygonal_prism(const SupportPolygon &p, vec3 normal, vec3 point) {   // geom.cpp:1140-1161 (vertices + hull indices)
    for (int i = 0; i < p.nhull; ++i) {
        const int j = (i + 1) % p.nhull;
        const vec3 v0 = p.vertices[p.hull[i]], v1 = p.vertices[p.hull[j]];
        const vec3 t = cross(v1 - v0, normal);
        if (dot(point - v0, t) > kEps) return false;
    }
    return true;
}
inline bool closest_point_polygon(const SupportPolygon &p, vec2 q, vec2 &closest) {   // shape_util.cpp:223-282
    for (int i = 0; i < p.nhull; ++i) {
        const int j = (i + 1) % p.nhull;
        const vec2 v0 = p.plane_vertices[p.hull[i]], v1 = p.plane_vertices[p.hull[j]];
        const vec2 e0 = v1 - v0;
        const vec2 n0 = -orthogonal(e0);
        if (dot(q - v0, n0) < 0) continue;
        if (dot(q - v0, e0) > 0) {
            if (dot(q - v1, e0) < 0) {
                const float t = dot(q - v0, e0) / dot(e0, e0);
                closest = lerp(v0, v1, t);
                return true;
            } else {
                const int k = (i + 2) % p.nhull;
                const vec2 v2 = p.plane_vertices[p.hull[k]];
                const vec2 e1 = v2 - v1;
                if (dot(q - v1, e1) < 0) { closest = v1; return true; }
            }
        }
    }
    return false;
}
// collision_result::add_point asserts room; a release build of the reference would write past the array - here the point is dropped
inline void poly_add(coll_result &result, const coll_point &p) { if (result.num_points < kMaxContacts) result.add_point(p); }

// ---- collide(polyhedron, plane)   collide_polyhedron_plane.cpp:10-37
inline void collide_polyhedron_plane(const PolySh &shA, vec3 pn, float pc, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA = ctx.posA;
    const vec3 normal = pn;
    const vec3 center = pn * pc - posA;
    const float proj_poly = -polyhedron_support_projection(shA.rot, shA.mesh, -normal);
    const float proj_plane = dot(center, normal);
    const float distance = proj_poly - proj_plane;
    if (distance > ctx.threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, shA.rot, shA.mesh.nv(), vec3{0, 0, 0}, normal, proj_poly, true, kSupportFeatureTolerance);
    for (int h = 0; h < polygon.nhull; ++h) {
        const vec3 pointA = polygon.vertices[polygon.hull[h]];
        const vec3 pivotA = rotate(conjugate(ctx.ornA), pointA);
        const float local_distance = dot(pointA - center, normal);
        const vec3 pivotB = pointA - normal * local_distance + posA;
        result.maybe_add_point({pivotA, pivotB, normal, local_distance, NA_ON_B});
    }
}

// ---- collide(polyhedron, sphere)   collide_polyhedron_sphere.cpp:9-86 (in the polyhedron's space)
inline void collide_polyhedron_sphere(const PolySh &shA, float radiusB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const quat ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    float distance = -kScalarMax, projection_poly = kScalarMax;
    vec3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const vec3 normalA = -meshA.normal(face_idx);
        const vec3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = dot(posB, normalA) + radiusB;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    if (distance > threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), vec3{0, 0, 0}, sep_axis, projection_poly, true, kSupportFeatureTolerance);
    const vec2 posB_plane = to_vector2_xz(to_object(posB, polygon.origin, polygon.basis));
    vec2 closest{0, 0};
    const bool inside_face = !closest_point_polygon(polygon, posB_plane, closest);
    if (inside_face) {
        const vec3 pivotA = project_plane(posB, polygon.origin, sep_axis);
        const vec3 normalB = rotate(conjugate(ornB), sep_axis);
        const vec3 pivotB = normalB * radiusB;
        const vec3 normal = rotate(ctx.ornA, sep_axis);
        poly_add(result, {pivotA, pivotB, normal, distance, NA_ON_A});
        return;
    }
    vec3 pivotA = to_world(to_vector3_xz(closest), polygon.origin, polygon.basis);
    vec3 new_sep_axis = pivotA - posB;
    const float new_sep_axis_len_sqr = length_sqr(new_sep_axis);
    if (new_sep_axis_len_sqr > kEps) {
        const float new_sep_axis_len = std::sqrt(new_sep_axis_len_sqr);
        new_sep_axis /= new_sep_axis_len;
        distance = new_sep_axis_len - radiusB;
        if (distance > threshold) return;
    } else {
        new_sep_axis = sep_axis;
        pivotA = project_plane(posB, polygon.origin, new_sep_axis);
    }
    const vec3 normalB = rotate(conjugate(ornB), new_sep_axis);
    const vec3 pivotB = normalB * radiusB;
    const vec3 normal = rotate(ctx.ornA, new_sep_axis);
    poly_add(result, {pivotA, pivotB, normal, distance, NA_NONE});
}

// ---- collide(polyhedron, polyhedron)   collide_polyhedron_polyhedron.cpp:13-241 (A at the origin, rotated meshes)
inline void poly_max_support_direction(const PolySh &shA, vec3 posA, const PolySh &shB, vec3 posB, vec3 &dir, float &distance, float &projectionA, float &projectionB) {
    float max_proj_A = kScalarMax, max_proj_B = -kScalarMax, max_distance = -kScalarMax;
    vec3 best_dir{0, 0, 0};
    for (int idx = 0; idx < shA.mesh.nrf(); ++idx) {
        const vec3 normal_world = -shA.rot.relevant_normal(idx);
        const int face_idx = shA.mesh.relevant_face(idx);
        const vec3 vertexA = shA.rot.vertex(shA.mesh.first_vertex_index(face_idx));
        const vec3 vertex_world = vertexA + posA;
        const float projA = dot(vertex_world, normal_world);
        const float projB = polyhedron_support_projection(shB.rot, shB.mesh, normal_world) + dot(posB, normal_world);
        const float dist = projA - projB;
        if (dist > max_distance) { max_distance = dist; max_proj_A = projA; max_proj_B = projB; best_dir = normal_world; }
    }
    dir = best_dir; distance = max_distance; projectionA = max_proj_A; projectionB = max_proj_B;
}
inline void collide_polyhedron_polyhedron(const PolySh &shA, const PolySh &shB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posA{0, 0, 0}; const quat ornA = ctx.ornA;
    const vec3 posB = ctx.posB - ctx.posA; const quat ornB = ctx.ornB;
    const float threshold = ctx.threshold;
    float distance = -kScalarMax, projectionA = kScalarMax, projectionB = -kScalarMax;
    vec3 sep_axis{0, 0, 0};
    poly_max_support_direction(shA, posA, shB, posB, sep_axis, distance, projectionA, projectionB);
    {
        float dist, projA, projB; vec3 dir;
        poly_max_support_direction(shB, posB, shA, posA, dir, dist, projB, projA);
        if (dist > distance) {
            dir *= -1.0f; projA *= -1.0f; projB *= -1.0f;
            distance = dist; projectionA = projA; projectionB = projB; sep_axis = dir;
        }
    }
    float min_edge_dist = -kScalarMax;
    // The reference declares edge_projectionA / edge_projectionB / edge_dir without initialisers (:98-100) and reads them even when no
    // edge pair spanned a Minkowski face (parallel edges only: axis-aligned prisms and boxes) - undefined behaviour that, with the
    // wrong stack garbage, runs its quickhull recursion off the stack. Defined here: such a pair has no edge axis.
    float edge_projectionA = -kScalarMax, edge_projectionB = 0;
    vec3 edge_dir{0, 0, 0};
    for (int eA = 0; eA < shA.mesh.ne(); ++eA) {
        const vec3 normalsA[2] = {shA.rot.edge_normal(2 * eA), shA.rot.edge_normal(2 * eA + 1)};
        vec3 verticesA[2] = {shA.rot.edge_vertex(2 * eA), shA.rot.edge_vertex(2 * eA + 1)};
        verticesA[0] += posA; verticesA[1] += posA;
        const vec3 edge_dirA = verticesA[0] - verticesA[1];
        for (int eB = 0; eB < shB.mesh.ne(); ++eB) {
            const vec3 normalsB[2] = {shB.rot.edge_normal(2 * eB), shB.rot.edge_normal(2 * eB + 1)};
            vec3 verticesB[2] = {shB.rot.edge_vertex(2 * eB), shB.rot.edge_vertex(2 * eB + 1)};
            verticesB[0] += posB; verticesB[1] += posB;
            const vec3 edge_dirB = verticesB[0] - verticesB[1];
            if (edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) {
                vec3 dir = cross(edge_dirA, edge_dirB);
                if (try_normalize(dir)) {
                    if (dot(verticesA[0] - posA, dir) < 0) dir *= -1.0f;
                    const float edge_dist = dot(verticesB[0] - verticesA[0], dir);
                    if (edge_dist > min_edge_dist) {
                        min_edge_dist = edge_dist;
                        dir *= -1.0f;
                        edge_projectionA = dot(verticesA[0], dir);
                        edge_projectionB = dot(verticesB[0], dir);
                        edge_dir = dir;
                    }
                }
            }
        }
    }
    if (min_edge_dist == -kScalarMax) g_poly_flags |= 1;
    const float edge_distance = edge_projectionA - edge_projectionB;
    if (edge_distance > distance) { distance = edge_distance; projectionA = edge_projectionA; projectionB = edge_projectionB; sep_axis = edge_dir; }
    if (distance > threshold) return;
    SupportPolygon polygonA, polygonB;
    point_cloud_support_polygon(polygonA, shA.rot, shA.mesh.nv(), posA, sep_axis, projectionA, true, kSupportFeatureTolerance);
    point_cloud_support_polygon(polygonB, shB.rot, shB.mesh.nv(), posB, sep_axis, projectionB, false, kSupportFeatureTolerance);
    int normal_attachment = NA_NONE;
    if (polygonB.nhull > 2) normal_attachment = NA_ON_B;
    else if (polygonA.nhull > 2) normal_attachment = NA_ON_A;
    if (polygonB.nhull > 2)
        for (int h = 0; h < polygonA.nhull; ++h) {
            const vec3 pointA = polygonA.vertices[polygonA.hull[h]];
            if (point_in_polygonal_prism(polygonB, sep_axis, pointA)) {
                const vec3 pivotA = to_object(pointA, posA, ornA);
                const vec3 pivotB = to_object(project_plane(pointA, polygonB.origin, sep_axis), posB, ornB);
                result.maybe_add_point({pivotA, pivotB, sep_axis, distance, normal_attachment});
            }
        }
    if (polygonA.nhull > 2)
        for (int h = 0; h < polygonB.nhull; ++h) {
            const vec3 pointB = polygonB.vertices[polygonB.hull[h]];
            if (point_in_polygonal_prism(polygonA, sep_axis, pointB)) {
                const vec3 pivotB = to_object(pointB, posB, ornB);
                const vec3 pivotA = to_object(project_plane(pointB, polygonA.origin, sep_axis), posA, ornA);
                result.maybe_add_point({pivotA, pivotB, sep_axis, distance, normal_attachment});
            }
        }
    if (polygonA.nhull > 1 && polygonB.nhull > 1) {
        const int sizeA = polygonA.nhull, sizeB = polygonB.nhull;
        const int limitA = sizeA == 2 ? 1 : sizeA, limitB = sizeB == 2 ? 1 : sizeB;
        float s[2], t[2];
        for (int i = 0; i < limitA; ++i) {
            const int idx0A = polygonA.hull[i], idx1A = polygonA.hull[(i + 1) % sizeA];
            const vec2 v0A = polygonA.plane_vertices[idx0A], v1A = polygonA.plane_vertices[idx1A];
            for (int j = 0; j < limitB; ++j) {
                const int idx0B = polygonB.hull[j], idx1B = polygonB.hull[(j + 1) % sizeB];
                const vec2 v0B = polygonB.plane_vertices[idx0B], v1B = polygonB.plane_vertices[idx1B];
                const int num_points = intersect_segments(v0A, v1A, v0B, v1B, s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    const vec3 pivotA_world = lerp(polygonA.vertices[idx0A], polygonA.vertices[idx1A], s[k]);
                    const vec3 pivotB_world = lerp(polygonB.vertices[idx0B], polygonB.vertices[idx1B], t[k]);
                    result.maybe_add_point({to_object(pivotA_world, posA, ornA), to_object(pivotB_world, posB, ornB), sep_axis, distance, normal_attachment});
                }
            }
        }
    }
}

// ---- collide(polyhedron, box)   collide_polyhedron_box.cpp:14-290 (in the polyhedron's space)
static const int kBoxEdgeFaces[24] = {0, 4, 0, 3, 0, 5, 0, 2, 1, 2, 1, 5, 1, 3, 1, 4, 4, 2, 3, 4, 5, 3, 2, 5};   // box_shape.cpp:388-405
inline bool point_in_quad(const vec3 v[4], vec3 normal, vec3 point) { return point_in_quad_prism(v, normal, point); }
inline void collide_polyhedron_box(const PolySh &shA, vec3 hB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const quat ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    const vec3 box_axes[3] = {quaternion_x(ornB), quaternion_y(ornB), quaternion_z(ornB)};
    float distance = -kScalarMax, projection_poly = 0;
    vec3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const vec3 normalA = -meshA.normal(face_idx);
        const vec3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = box_support_projection(hB, posB, ornB, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    for (int i = 0; i < 3; ++i) {
        vec3 dir = box_axes[i];
        if (dot(posB, dir) > 0) dir = -dir;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = dot(posB, dir) + hB[i];
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    float min_edge_dist = -kScalarMax, edge_projectionA = 0, edge_projectionB = 0;
    vec3 edge_dir{0, 0, 0};
    for (int eA = 0; eA < meshA.ne(); ++eA) {
        const vec3 normalsA[2] = {meshA.normal(meshA.edge_face(2 * eA)), meshA.normal(meshA.edge_face(2 * eA + 1))};
        const vec3 verticesA[2] = {meshA.vertex(meshA.edge_vertex_index(2 * eA)), meshA.vertex(meshA.edge_vertex_index(2 * eA + 1))};
        const vec3 edge_dirA = verticesA[0] - verticesA[1];
        for (int eB = 0; eB < 12; ++eB) {
            const vec3 normalsB[2] = {rotate(ornB, box_face_normal(kBoxEdgeFaces[2 * eB])), rotate(ornB, box_face_normal(kBoxEdgeFaces[2 * eB + 1]))};
            vec3 verticesB[2];
            box_edge_world(hB, eB, posB, ornB, verticesB);
            const vec3 edge_dirB = verticesB[0] - verticesB[1];
            if (edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) {
                vec3 dir = cross(edge_dirA, edge_dirB);
                if (try_normalize(dir)) {
                    if (dot(verticesA[0], dir) < 0) dir *= -1.0f;
                    const float edge_dist = dot(verticesB[0] - verticesA[0], dir);
                    if (edge_dist > min_edge_dist) {
                        min_edge_dist = edge_dist;
                        dir *= -1.0f;
                        edge_projectionA = dot(verticesA[0], dir);
                        edge_projectionB = dot(verticesB[0], dir);
                        edge_dir = dir;
                    }
                }
            }
        }
    }
    if (edge_dir != vec3{0, 0, 0}) {
        const float edge_distance = edge_projectionA - edge_projectionB;
        if (edge_distance > distance) { distance = edge_distance; projection_poly = edge_projectionA; sep_axis = edge_dir; }
    }
    if (distance > threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), vec3{0, 0, 0}, sep_axis, projection_poly, true, kSupportFeatureTolerance);
    int featureB, fiB; float proj_unused;
    box_support_feature(hB, posB, ornB, vec3{0, 0, 0}, sep_axis, featureB, fiB, proj_unused, kSupportFeatureTolerance);
    const int feature_indexB = fiB;
    coll_point point{};
    point.normal = rotate(ctx.ornA, sep_axis);
    point.distance = distance; point.attachment = NA_NONE;
    if (featureB == BF_FACE) {
        vec3 face_verticesB[4];
        box_face_world(hB, feature_indexB, posB, ornB, face_verticesB);
        point.attachment = NA_ON_B;
        for (int h = 0; h < polygon.nhull; ++h) {
            const vec3 pointA = polygon.vertices[polygon.hull[h]];
            if (point_in_quad(face_verticesB, sep_axis, pointA)) {
                point.distance = dot(pointA - face_verticesB[0], sep_axis);
                const vec3 pivotB_world = pointA - sep_axis * point.distance;
                point.pivotA = pointA;
                point.pivotB = to_object(pivotB_world, posB, ornB);
                result.maybe_add_point(point);
            }
        }
        if (polygon.nhull > 2)
            for (int i = 0; i < 4; ++i) {
                const vec3 pointB = face_verticesB[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.distance = dot(polygon.origin - pointB, sep_axis);
                    point.pivotA = pointB + sep_axis * point.distance;
                    point.pivotB = to_object(pointB, posB, ornB);
                    result.maybe_add_point(point);
                }
            }
        if (polygon.nhull > 1) {
            vec2 plane_vertices_box[4];
            for (int i = 0; i < 4; ++i) plane_vertices_box[i] = to_vector2_xz(to_object(face_verticesB[i], polygon.origin, polygon.basis));
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            float s[2], t[2];
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const vec2 v0A = polygon.plane_vertices[idx0A], v1A = polygon.plane_vertices[idx1A];
                for (int j = 0; j < 4; ++j) {
                    const int idx0B = j, idx1B = (j + 1) % 4;
                    const int num_points = intersect_segments(v0A, v1A, plane_vertices_box[idx0B], plane_vertices_box[idx1B], s[0], t[0], s[1], t[1]);
                    for (int k = 0; k < num_points; ++k) {
                        point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                        const vec3 pivotB_world = lerp(face_verticesB[idx0B], face_verticesB[idx1B], t[k]);
                        point.pivotB = to_object(pivotB_world, posB, ornB);
                        result.maybe_add_point(point);
                    }
                }
            }
        }
    } else if (featureB == BF_EDGE) {
        const vec3 edge_vertices_local[2] = {box_vertex(hB, kBoxEdgeIndices[feature_indexB * 2]), box_vertex(hB, kBoxEdgeIndices[feature_indexB * 2 + 1])};
        vec3 edge_vertices[2];
        box_edge_world(hB, feature_indexB, posB, ornB, edge_vertices);
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const vec3 pointB = edge_vertices[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB, posB, ornB);
                    poly_add(result, point);
                }
            }
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            const vec2 v0B = to_vector2_xz(to_object(edge_vertices[0], polygon.origin, polygon.basis));
            const vec2 v1B = to_vector2_xz(to_object(edge_vertices[1], polygon.origin, polygon.basis));
            float s[2], t[2];
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const int num_points = intersect_segments(polygon.plane_vertices[idx0A], polygon.plane_vertices[idx1A], v0B, v1B, s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                    point.pivotB = lerp(edge_vertices_local[0], edge_vertices_local[1], t[k]);
                    poly_add(result, point);
                }
            }
        } else {
            point.pivotA = polygon.vertices[polygon.hull[0]];
            const vec3 edge_dir2 = edge_vertices[1] - edge_vertices[0];
            vec3 pivotB_world; float t;
            closest_point_line(edge_vertices[0], edge_dir2, point.pivotA, t, pivotB_world);
            point.pivotB = lerp(edge_vertices_local[0], edge_vertices_local[1], t);
            poly_add(result, point);
        }
    } else {
        point.pivotB = box_vertex(hB, feature_indexB);
        const vec3 pivotB_world = to_world(point.pivotB, posB, ornB);
        point.pivotA = pivotB_world + sep_axis * distance;
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        poly_add(result, point);
    }
}

// ---- collide(polyhedron, capsule)   collide_polyhedron_capsule.cpp:10-164 (in the polyhedron's space)
inline void collide_polyhedron_capsule(const PolySh &shA, const shape &shB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const quat ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    vec3 capsule_vertices_[2];
    capsule_vertices(shB, posB, ornB, capsule_vertices_);
    float distance = -kScalarMax, projection_poly = kScalarMax;
    vec3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const vec3 normalA = -meshA.normal(face_idx);
        const vec3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = capsule_support_projection(capsule_vertices_, shB.radius, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    for (int i = 0; i < meshA.ne(); ++i) {
        const vec3 vertexA0 = meshA.edge_vertex(2 * i), vertexA1 = meshA.edge_vertex(2 * i + 1);
        float s, t; vec3 closestA, closestB;
        closest_point_segment_segment(vertexA0, vertexA1, capsule_vertices_[0], capsule_vertices_[1], s, t, closestA, closestB, nullptr, nullptr, nullptr, nullptr, nullptr);
        vec3 dir = closestA - closestB;
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = capsule_support_projection(capsule_vertices_, shB.radius, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    if (distance > threshold) return;
    const float proj_capsule_vertices[2] = {dot(capsule_vertices_[0], sep_axis), dot(capsule_vertices_[1], sep_axis)};
    const bool is_capsule_edge = std::fabs(proj_capsule_vertices[0] - proj_capsule_vertices[1]) < kSupportFeatureTolerance;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), vec3{0, 0, 0}, sep_axis, projection_poly, true, kSupportFeatureTolerance);
    coll_point point{};
    point.normal = rotate(ctx.ornA, sep_axis);
    point.distance = distance;
    point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
    if (is_capsule_edge) {
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const vec3 pointB = capsule_vertices_[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB + sep_axis * shB.radius, posB, ornB);
                    poly_add(result, point);
                }
            }
        if (result.num_points == 2) return;
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            float s[2], t[2];
            const vec2 plane_capsule_vertices[2] = {to_vector2_xz(to_object(capsule_vertices_[0], polygon.origin, polygon.basis)),
                                                    to_vector2_xz(to_object(capsule_vertices_[1], polygon.origin, polygon.basis))};
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const int num_points = intersect_segments(polygon.plane_vertices[idx0A], polygon.plane_vertices[idx1A], plane_capsule_vertices[0],
                                                          plane_capsule_vertices[1], s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                    const vec3 pivotB_world = lerp(capsule_vertices_[0], capsule_vertices_[1], t[k]) + sep_axis * shB.radius;
                    point.pivotB = to_object(pivotB_world, posB, ornB);
                    result.maybe_add_point(point);
                }
            }
        } else {
            point.pivotA = polygon.vertices[polygon.hull[0]];
            const vec3 edge_dir = capsule_vertices_[1] - capsule_vertices_[0];
            vec3 pivotB_world; float t;
            closest_point_line(capsule_vertices_[0], edge_dir, point.pivotA, t, pivotB_world);
            const vec3 normalB = rotate(conjugate(ornB), sep_axis);
            point.pivotB = to_object(pivotB_world, posB, ornB) + normalB * shB.radius;
            poly_add(result, point);
        }
    } else {
        const int closest_capsule_vertex_index = proj_capsule_vertices[0] > proj_capsule_vertices[1] ? 0 : 1;
        const vec3 pivotB_world = capsule_vertices_[closest_capsule_vertex_index] + sep_axis * shB.radius;
        point.pivotB = to_object(pivotB_world, posB, ornB);
        point.pivotA = pivotB_world + sep_axis * distance;
        poly_add(result, point);
    }
}

// ---- collide(polyhedron, cylinder)   collide_polyhedron_cylinder.cpp:12-336 (in the polyhedron's space)
inline void collide_polyhedron_cylinder(const PolySh &shA, const shape &shB, const coll_ctx &ctx, coll_result &result) {
    const vec3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const quat ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    const vec3 cyl_axis = rotate(ornB, coordinate_axis_vector(shB.axis));
    const vec3 face_center_pos = posB + cyl_axis * shB.half_length, face_center_neg = posB - cyl_axis * shB.half_length;
    float distance = -kScalarMax, projection_poly = kScalarMax;
    vec3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const vec3 normalA = -meshA.normal(face_idx);
        const vec3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = cylinder_support_projection(shB, posB, ornB, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    for (int i = 0; i < 2; ++i) {
        const vec3 dir = i == 0 ? cyl_axis : -cyl_axis;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    for (int k = 0; k < meshA.nre(); ++k) {
        const int edge_idx = meshA.relevant_edge(k);
        const vec3 poly_edge = meshA.edge_vertex(2 * edge_idx + 1) - meshA.edge_vertex(2 * edge_idx);
        vec3 dir = cross(poly_edge, cyl_axis);
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    for (int v = 0; v < meshA.nv(); ++v) {
        const vec3 rvertex = meshA.vertex(v);
        vec3 closest; float t;
        closest_point_line(face_center_neg, cyl_axis, rvertex, t, closest);
        vec3 dir = rvertex - closest;
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    for (int i = 0; i < meshA.ne(); ++i) {
        const vec3 vertexA0 = meshA.edge_vertex(2 * i), vertexA1 = meshA.edge_vertex(2 * i + 1);
        for (int j = 0; j < 2; ++j) {
            const vec3 face_center = j == 0 ? face_center_neg : face_center_pos;
            size_t num_points; float s0, s1; vec3 cc0, cl0, cc1, cl1, dir;
            closest_point_circle_line(face_center, ornB, shB.radius, shB.axis, vertexA0, vertexA1, num_points, s0, cc0, cl0, s1, cc1, cl1, dir, kSupportFeatureTolerance);
            if (num_points == 2) continue;
            if (!(s0 > 0 && s0 < 1)) continue;
            if (dot(posB, dir) > 0) dir *= -1.0f;
            const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
            const float projB = cylinder_support_projection(shB, posB, ornB, dir);
            const float dist = projA - projB;
            if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
        }
    }
    if (distance > threshold) return;
    const vec3 normal = rotate(ctx.ornA, sep_axis);
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), vec3{0, 0, 0}, sep_axis, projection_poly, true, kSupportFeatureTolerance);
    int featureB; size_t feature_indexB = 0;
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportFeatureTolerance);
    coll_point point{};
    point.normal = normal; point.distance = distance; point.attachment = NA_NONE;
    const int ai = shB.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureB == CF_FACE) {
        int num_vertices_in_face = 0;
        const float sign_faceB = to_sign(feature_indexB == 0);
        const float pivotB_axis = shB.half_length * sign_faceB;
        point.attachment = NA_ON_B;
        for (int h = 0; h < polygon.nhull; ++h) {
            const vec3 pointA = polygon.vertices[polygon.hull[h]];
            vec3 closest; float t;
            const float dist_sqr = closest_point_line(posB, cyl_axis, pointA, t, closest);
            if (dist_sqr > shB.radius * shB.radius) continue;
            point.pivotA = pointA;
            point.pivotB = to_object(pointA, posB, ornB);
            point.pivotB[ai] = pivotB_axis;
            result.maybe_add_point(point);
            ++num_vertices_in_face;
        }
        int num_edge_intersections = 0;
        const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
        for (int i = 0; i < limitA; ++i) {
            const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
            const vec3 v0A = polygon.vertices[idx0A], v1A = polygon.vertices[idx1A];
            const vec3 v0B = to_object(v0A, posB, ornB), v1B = to_object(v1A, posB, ornB);
            float s[2];
            const size_t num_points = intersect_line_circle(vec2{v0B.z, v0B.y}, vec2{v1B.z, v1B.y}, shB.radius, s[0], s[1]);   // (to_vector2_zy whatever the axis: as the reference)
            for (size_t j = 0; j < num_points; ++j) {
                const float t = s[j];
                if (t < 0 || t > 1) continue;
                point.pivotA = lerp(v0A, v1A, t);
                point.pivotB = lerp(v0B, v1B, t);
                point.pivotB[ai] = pivotB_axis;
                result.maybe_add_point(point);
                ++num_edge_intersections;
            }
        }
        if (polygon.nhull > 2 && num_vertices_in_face == 0 && num_edge_intersections == 0) {
            if (point_in_polygonal_prism(polygon, sep_axis, posB)) {
                const float multipliers[4] = {0, 1, 0, -1};
                for (int i = 0; i < 4; ++i) {
                    point.pivotB[ai] = pivotB_axis;
                    point.pivotB[o0] = shB.radius * multipliers[i];
                    point.pivotB[o1] = shB.radius * multipliers[(i + 1) % 4];
                    point.pivotA = to_world(point.pivotB, posB, ornB);
                    point.pivotA = project_plane(point.pivotA, polygon.origin, sep_axis);
                    result.maybe_add_point(point);
                }
            }
        }
    } else if (featureB == CF_SIDE_EDGE) {
        const vec3 edge_vertices[2] = {face_center_neg + sep_axis * shB.radius, face_center_pos + sep_axis * shB.radius};
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const vec3 pointB = edge_vertices[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB, posB, ornB);
                    result.maybe_add_point(point);
                }
            }
        if (result.num_points == 2) return;
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            const vec2 v0B = to_vector2_xz(to_object(edge_vertices[0], polygon.origin, polygon.basis));
            const vec2 v1B = to_vector2_xz(to_object(edge_vertices[1], polygon.origin, polygon.basis));
            float s[2], t[2];
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const int num_points = intersect_segments(polygon.plane_vertices[idx0A], polygon.plane_vertices[idx1A], v0B, v1B, s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                    const vec3 pivotB_world = lerp(edge_vertices[0], edge_vertices[1], t[k]);
                    point.pivotB = to_object(pivotB_world, posB, ornB);
                    result.maybe_add_point(point);
                }
            }
        } else {
            point.pivotA = polygon.vertices[polygon.hull[0]];
            const vec3 edge_dir = edge_vertices[1] - edge_vertices[0];
            vec3 pivotB_world; float t;
            closest_point_line(edge_vertices[0], edge_dir, point.pivotA, t, pivotB_world);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            poly_add(result, point);
        }
    } else {
        const vec3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotA = supportB + sep_axis * distance;
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        poly_add(result, point);
    }
}

}  // namespace orc
