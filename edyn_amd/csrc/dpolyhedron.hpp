// polyhedron_shape on the device: convex mesh tables, the per-body rotated mesh, support polygons and the
// closest-feature routines of its pairs.
//   /root/reference/include/edyn/shapes/convex_mesh.hpp:17-198, src/edyn/sys/update_rotated_meshes.cpp:12-52
//   /root/reference/src/edyn/util/shape_util.cpp:48-79 (polyhedron_support_projection), :89-190 (quickhull), :223-282 (closest_point_convex_polygon)
//   /root/reference/include/edyn/util/shape_util.hpp:201-273 (support_polygon), src/edyn/math/geom.cpp:756-760,800-845,1140-1161,1345-1352
//   /root/reference/src/edyn/collision/collide/collide_polyhedron_{plane,sphere,box,polyhedron,capsule,cylinder}.cpp
// A mesh is created once per context (edynhip_create_convex_mesh, mesh.hip: centroid shift, normals, edges, adjacency, relevant
// faces / edges computed on the host in the reference's order) and lives in flat tables shared by every body that uses it; a
// polyhedron body's shape record holds the mesh id. Its rotated vertices / relevant normals / edge vertices / edge normals are
// refreshed from the current orientation before every narrowphase (k_update_rotated), the role of update_rotated_meshes.
// The routines run in their own kernel (narrowphase.hip k_np_detect_poly), launched only for worlds that hold a polyhedron.
// Support polygons live in fixed arrays of kPolyMax vertices (std::vector in the reference): edynhip_create_convex_mesh rejects a
// mesh that could put more vertices than that within the support tolerance of one plane (a face with more than kPolyMax vertices).
// Where the reference reads uninitialised variables (collide_polyhedron_polyhedron.cpp:98-100,150: two polyhedra none of whose
// edge pairs spans a Minkowski face) the pair simply has no edge axis.
#pragma once
#include "dcylinder.hpp"
#include "dmesh.hpp"

namespace dc {

struct MeshView {
    const Meshes *t; MeshDesc d;
    DI int nv() const { return (int)d.nv; }
    DI int ne() const { return (int)d.ne; }
    DI int nrf() const { return (int)d.nrf; }
    DI int nre() const { return (int)d.nre; }
    DI f3 vertex(int i) const { return from4(t->vertices[d.v_off + i]); }
    DI f3 normal(int f) const { return from4(t->normals[d.f_off + f]); }
    DI int relevant_face(int k) const { return (int)t->relevant_faces[d.rf_off + k]; }
    DI int relevant_edge(int k) const { return (int)t->relevant_edges[d.re_off + k]; }
    DI int first_vertex_index(int f) const { return (int)t->face_first[d.f_off + f]; }
    DI f3 edge_vertex(int k) const { return from4(t->edge_vertices[2 * d.e_off + k]); }
    DI int edge_face(int k) const { return (int)t->edge_faces[2 * d.e_off + k]; }
    DI int edge_vertex_index(int k) const { return (int)t->edge_vidx[2 * d.e_off + k]; }
    DI int neighbors_start(int v) const { return (int)t->nb_start[d.nb_off + v]; }
    DI int neighbor(int k) const { return (int)t->nb_idx[d.ni_off + k]; }
};
struct RotView {   // one body's rotated mesh: [nv vertices][nrf relevant normals][2 ne edge vertices][2 ne edge normals]
    const float4 *base; uint32_t nv, nrf, ne;
    DI f3 vertex(int i) const { return from4(base[i]); }
    DI f3 relevant_normal(int k) const { return from4(base[nv + k]); }
    DI f3 edge_vertex(int k) const { return from4(base[nv + nrf + k]); }
    DI f3 edge_normal(int k) const { return from4(base[nv + nrf + 2 * ne + k]); }
};
struct PolySh { MeshView mesh; RotView rot; };
DI PolySh poly_of(const Meshes &t, float4 shape, const float4 *rot_base) {
    const MeshDesc d = t.desc[(uint32_t)shape.x];
    return PolySh{MeshView{&t, d}, RotView{rot_base, d.nv, d.nrf, d.ne}};
}
// update_rotated_mesh (update_rotated_meshes.cpp:12-52) for item k of a body's rotated mesh; orn = orientation * rotated_mesh_list::orientation
DI void rotate_mesh_item(const Meshes &t, const MeshDesc &d, q4 orn, float4 *base, uint32_t k) {
    f3 v;
    if (k < d.nv) v = from4(t.vertices[d.v_off + k]);
    else if (k < d.nv + d.nrf) v = from4(t.relevant_normals[d.rf_off + (k - d.nv)]);
    else if (k < d.nv + d.nrf + 2 * d.ne) v = from4(t.edge_vertices[2 * d.e_off + (k - d.nv - d.nrf)]);
    else v = from4(t.edge_normals[2 * d.e_off + (k - d.nv - d.nrf - 2 * d.ne)]);
    base[k] = to4(rotate(orn, v), 0.0f);
}
DI box3 polyhedron_aabb(const Meshes &t, float4 shape, f3 pos, q4 orn) {   // aabb_util.cpp:141-164,195-197 (update_aabbs.cpp:22-32 gives the same box)
    const MeshDesc d = t.desc[(uint32_t)shape.x];
    box3 b{mk3(kScalarMax, kScalarMax, kScalarMax), mk3(-kScalarMax, -kScalarMax, -kScalarMax)};
    for (uint32_t i = 0; i < d.nv; ++i) {
        const f3 w = to_world(from4(t.vertices[d.v_off + i]), pos, orn);
        b.mn = mk3(fminf(b.mn.x, w.x), fminf(b.mn.y, w.y), fminf(b.mn.z, w.z));
        b.mx = mk3(fmaxf(b.mx.x, w.x), fmaxf(b.mx.y, w.y), fmaxf(b.mx.z, w.z));
    }
    return b;
}
DI m3 polyhedron_inertia(const Meshes &t, float4 shape, float mass) {   // moment_of_inertia.cpp:143-157 (the sums are the mesh's, mesh.hip)
    const MeshDesc d = t.desc[(uint32_t)shape.x];
    const float density = mass / (d.isum[0] / 6.0f);
    const float r = density / 120.0f;
    const float Iyz = d.isum[4] * r, Izx = d.isum[5] * r, Ixy = d.isum[6] * r;
    const float Ixx = (d.isum[2] + d.isum[3]) * r, Iyy = (d.isum[3] + d.isum[1]) * r, Izz = (d.isum[1] + d.isum[2]) * r;
    return {{Ixx, Ixy, Izx}, {Ixy, Iyy, Iyz}, {Izx, Iyz, Izz}};
}

DI f2 to_vector2_xz(f3 v) { return {v.x, v.z}; }
DI f3 to_vector3_xz(f2 v) { return {v.x, 0, v.y}; }
DI float perp_product(f2 v, f2 w) { return v.x * w.y - v.y * w.x; }
DI f2 lerp(f2 a, f2 b, float s) { return a * (1.0f - s) + b * s; }
DI f3 to_world(f3 p, f3 pos, const m3 &basis) { return pos + mul(basis, p); }   // transform.hpp:33-35
DI m3 make_tangent_basis(f3 n) { f3 t, u; plane_space(n, t, u); return m3_columns(t, n, u); }   // geom.cpp:756-760
// collision_result::add_point asserts room; a release build of the reference would write past the array - here the point is dropped
DI void poly_add(CResult &result, const CPoint &p) { if (result.num < kMaxContacts) res_add(result, p); }

// ---- geometry
DI bool is_triangle_ccw(f2 v0, f2 v1, f2 v2) { return dot(v2 - v0, orthogonal(v1 - v0)) > 0; }   // shape_util.cpp:208-212
DI int intersect_segments(f2 p0, f2 p1, f2 q0, f2 q1, float &s0, float &t0, float &s1, float &t1) {   // geom.cpp:804-845
    const f2 dp = p1 - p0, dq = q1 - q0, e = q0 - p0;
    const float denom = perp_product(dp, dq);
    if (fabsf(denom) > kEps) {
        const float denom_inv = 1.0f / denom;
        s0 = perp_product(e, dq) * denom_inv;
        t0 = perp_product(e, dp) * denom_inv;
        return s0 < 0 || s0 > 1 || t0 < 0 || t0 > 1 ? 0 : 1;
    }
    if (fabsf(perp_product(e, dp)) < kEps) {
        const float denom_p = 1.0f / dot(dp, dp), denom_q = 1.0f / dot(dq, dq);
        s0 = dot(q0 - p0, dp) * denom_p;
        s1 = dot(q1 - p0, dp) * denom_p;
        if ((s0 < 0 && s1 < 0) || (s0 > 1 && s1 > 1)) return 0;
        s0 = clamp_unit(s0); s1 = clamp_unit(s1);
        t0 = clamp_unit(dot(p0 - q0, dq) * denom_q);
        t1 = clamp_unit(dot(p1 - q0, dq) * denom_q);
        return fabsf(s1 - s0) < kEps ? 1 : 2;
    }
    return 0;
}
DI bool edges_generate_minkowski_face(f3 A, f3 B, f3 C_neg, f3 D_neg, f3 B_x_A, f3 D_x_C) {   // geom.cpp:1345-1352
    const float CBA = -dot(C_neg, B_x_A), DBA = -dot(D_neg, B_x_A), ADC = dot(A, D_x_C), BDC = dot(B, D_x_C);
    return CBA * DBA < 0 && ADC * BDC < 0 && CBA * BDC > 0;
}
template <class V>
DI float polyhedron_support_projection(const V &verts, const MeshView &mesh, f3 dir) {   // shape_util.cpp:48-79 (hill climbing over the vertex adjacency)
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
    f3 vertices[kPolyMax];
    f2 plane_vertices[kPolyMax];
    int hull[kPolyMax + 1];
    int nverts = 0, nhull = 0;
    f3 origin;
    m3 basis;
};
// split_hull_edge (shape_util.cpp:89-124), the recursion unrolled over an explicit stack: same insertions in the same order
DI void hull_insert(SupportPolygon &p, int at, int idx) {
    if (p.nhull > kPolyMax) return;
    for (int k = p.nhull; k > at; --k) p.hull[k] = p.hull[k - 1];
    p.hull[at] = idx; ++p.nhull;
}
struct HullFrame { unsigned char i0, i1, stage, n1; };   // (indices and counts <= kPolyMax + 1)
DI HullFrame hull_frame(int i0, int i1) { return HullFrame{(unsigned char)i0, (unsigned char)i1, 0, 0}; }
DI int split_hull_edge(SupportPolygon &p, int i0_in, int i1_in, float tolerance, HullFrame *stack) {   // stack: kPolyMax + 2 frames
    int sp = 0, ret = 0;
    stack[sp++] = hull_frame(i0_in, i1_in);
    while (sp > 0) {
        HullFrame &f = stack[sp - 1];
        if (f.stage == 0) {
            const f2 v0 = p.plane_vertices[p.hull[f.i0]], v1 = p.plane_vertices[p.hull[f.i1]];
            const f2 dir = -orthogonal(v1 - v0);
            float max_proj = -kScalarMax; int idx = 0;
            for (int i = 0; i < p.nverts; ++i) {
                const float proj = dot(p.plane_vertices[i], dir);
                if (proj > max_proj) { max_proj = proj; idx = i; }
            }
            if (dot(p.plane_vertices[idx] - v0, dir) > tolerance && p.nhull <= kPolyMax && sp < kPolyMax + 1) {
                hull_insert(p, f.i1, idx);
                f.stage = 1;
                stack[sp] = hull_frame(f.i0, f.i1); ++sp;
            } else { ret = 0; --sp; }
        } else if (f.stage == 1) {
            f.n1 = (unsigned char)ret;
            f.i1 = (unsigned char)(f.i1 + f.n1);
            f.stage = 2;
            stack[sp] = hull_frame(f.i1, f.i1 + 1); ++sp;
        } else {
            ret = 1 + f.n1 + ret;
            --sp;
        }
    }
    return ret;
}
DI void calculate_convex_hull(SupportPolygon &p, float tolerance, HullFrame *stack) {   // shape_util.cpp:126-190
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
    f2 pt_min{kScalarMax, kScalarMax}, pt_max{-kScalarMax, -kScalarMax};
    int pt_min_idx = 0, pt_max_idx = 0;
    for (int i = 0; i < n; ++i) {
        const f2 q = p.plane_vertices[i];
        if (q.x < pt_min.x) { pt_min = q; pt_min_idx = i; }
        if (q.x > pt_max.x) { pt_max = q; pt_max_idx = i; }
    }
    if (pt_max.x - pt_min.x < tolerance) {   // a vertical sliver
        pt_min = f2{kScalarMax, kScalarMax}; pt_max = f2{-kScalarMax, -kScalarMax};
        for (int i = 0; i < n; ++i) {
            const f2 q = p.plane_vertices[i];
            if (q.y < pt_min.y) { pt_min = q; pt_min_idx = i; }
            if (q.y > pt_max.y) { pt_max = q; pt_max_idx = i; }
        }
        p.hull[0] = pt_max_idx; p.hull[1] = pt_min_idx; p.nhull = 2;
        return;
    }
    p.hull[0] = pt_max_idx; p.hull[1] = pt_min_idx; p.hull[2] = pt_max_idx; p.nhull = 3;
    int i1 = 1;
    const int num_splits = split_hull_edge(p, 0, i1, tolerance, stack);
    i1 += num_splits;
    split_hull_edge(p, i1, i1 + 1, tolerance, stack);
    --p.nhull;   // hull.pop_back()
}
DI void calculate_convex_hull(SupportPolygon &p, float tolerance) { HullFrame stack[kPolyMax + 2]; calculate_convex_hull(p, tolerance, stack); }
template <class V>
DI void point_cloud_support_polygon(SupportPolygon &polygon, const V &verts, int count, f3 offset, f3 dir, float projection, bool positive_side, float tolerance) {
    polygon.origin = dir * projection;
    polygon.basis = make_tangent_basis(dir);
    polygon.nverts = 0;
    const bool zero_offset = eq(offset, mk3(0, 0, 0));
    for (int i = 0; i < count; ++i) {
        const f3 vertex_world = zero_offset ? verts.vertex(i) : verts.vertex(i) + offset;
        const bool in_boundary = positive_side ? dot(vertex_world, dir) < projection + tolerance : dot(vertex_world, dir) > projection - tolerance;
        if (!in_boundary || polygon.nverts >= kPolyMax) continue;
        polygon.vertices[polygon.nverts] = vertex_world;
        polygon.plane_vertices[polygon.nverts] = to_vector2_xz(to_object(vertex_world, polygon.origin, polygon.basis));
        ++polygon.nverts;
    }
    calculate_convex_hull(polygon, 0.001f);
}
DI bool point_in_polygonal_prism(const SupportPolygon &p, f3 normal, f3 point) {   // geom.cpp:1140-1161 (vertices + hull indices)
    for (int i = 0; i < p.nhull; ++i) {
        const int j = (i + 1) % p.nhull;
        const f3 v0 = p.vertices[p.hull[i]], v1 = p.vertices[p.hull[j]];
        const f3 t = cross(v1 - v0, normal);
        if (dot(point - v0, t) > kEps) return false;
    }
    return true;
}
DI bool closest_point_polygon(const SupportPolygon &p, f2 q, f2 &closest) {   // shape_util.cpp:223-282
    for (int i = 0; i < p.nhull; ++i) {
        const int j = (i + 1) % p.nhull;
        const f2 v0 = p.plane_vertices[p.hull[i]], v1 = p.plane_vertices[p.hull[j]];
        const f2 e0 = v1 - v0;
        const f2 n0 = -orthogonal(e0);
        if (dot(q - v0, n0) < 0) continue;
        if (dot(q - v0, e0) > 0) {
            if (dot(q - v1, e0) < 0) {
                const float t = dot(q - v0, e0) / dot(e0, e0);
                closest = lerp(v0, v1, t);
                return true;
            } else {
                const int k = (i + 2) % p.nhull;
                const f2 v2 = p.plane_vertices[p.hull[k]];
                const f2 e1 = v2 - v1;
                if (dot(q - v1, e1) < 0) { closest = v1; return true; }
            }
        }
    }
    return false;
}

// ---- collide(polyhedron, plane)   collide_polyhedron_plane.cpp:10-37
DI void collide_polyhedron_plane(const PolySh &shA, f3 pn, float pc, const Ctx &ctx, CResult &result) {
    const f3 posA = ctx.posA;
    const f3 normal = pn;
    const f3 center = pn * pc - posA;
    const float proj_poly = -polyhedron_support_projection(shA.rot, shA.mesh, -normal);
    const float proj_plane = dot(center, normal);
    const float distance = proj_poly - proj_plane;
    if (distance > ctx.threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, shA.rot, shA.mesh.nv(), mk3(0, 0, 0), normal, proj_poly, true, kSupportTolerance);
    for (int h = 0; h < polygon.nhull; ++h) {
        const f3 pointA = polygon.vertices[polygon.hull[h]];
        const f3 pivotA = rotate(conjugate(ctx.ornA), pointA);
        const float local_distance = dot(pointA - center, normal);
        const f3 pivotB = pointA - normal * local_distance + posA;
        res_maybe_add(result, {pivotA, pivotB, normal, local_distance, NA_ON_B});
    }
}

// ---- collide(polyhedron, sphere)   collide_polyhedron_sphere.cpp:9-86 (in the polyhedron's space)
DI void collide_polyhedron_sphere(const PolySh &shA, float radiusB, const Ctx &ctx, CResult &result) {
    const f3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const q4 ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    float distance = -kScalarMax, projection_poly = kScalarMax;
    f3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const f3 normalA = -meshA.normal(face_idx);
        const f3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = dot(posB, normalA) + radiusB;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    if (distance > threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), mk3(0, 0, 0), sep_axis, projection_poly, true, kSupportTolerance);
    const f2 posB_plane = to_vector2_xz(to_object(posB, polygon.origin, polygon.basis));
    f2 closest{0, 0};
    const bool inside_face = !closest_point_polygon(polygon, posB_plane, closest);
    if (inside_face) {
        const f3 pivotA = project_plane(posB, polygon.origin, sep_axis);
        const f3 normalB = rotate(conjugate(ornB), sep_axis);
        const f3 pivotB = normalB * radiusB;
        const f3 normal = rotate(ctx.ornA, sep_axis);
        poly_add(result, {pivotA, pivotB, normal, distance, NA_ON_A});
        return;
    }
    f3 pivotA = to_world(to_vector3_xz(closest), polygon.origin, polygon.basis);
    f3 new_sep_axis = pivotA - posB;
    const float new_sep_axis_len_sqr = length_sqr(new_sep_axis);
    if (new_sep_axis_len_sqr > kEps) {
        const float new_sep_axis_len = sqrtf(new_sep_axis_len_sqr);
        new_sep_axis /= new_sep_axis_len;
        distance = new_sep_axis_len - radiusB;
        if (distance > threshold) return;
    } else {
        new_sep_axis = sep_axis;
        pivotA = project_plane(posB, polygon.origin, new_sep_axis);
    }
    const f3 normalB = rotate(conjugate(ornB), new_sep_axis);
    const f3 pivotB = normalB * radiusB;
    const f3 normal = rotate(ctx.ornA, new_sep_axis);
    poly_add(result, {pivotA, pivotB, normal, distance, NA_NONE});
}

// ---- collide(polyhedron, polyhedron)   collide_polyhedron_polyhedron.cpp:13-241 (A at the origin, rotated meshes)
// the contact points of two support polygons along the separating axis (collide_polyhedron_polyhedron.cpp:152-241)
DI void polygon_polygon_contacts(const SupportPolygon &polygonA, const SupportPolygon &polygonB, f3 posA, q4 ornA, f3 posB, q4 ornB, f3 sep_axis, float distance, CResult &result) {
    int normal_attachment = NA_NONE;
    if (polygonB.nhull > 2) normal_attachment = NA_ON_B;
    else if (polygonA.nhull > 2) normal_attachment = NA_ON_A;
    if (polygonB.nhull > 2)
        for (int h = 0; h < polygonA.nhull; ++h) {
            const f3 pointA = polygonA.vertices[polygonA.hull[h]];
            if (point_in_polygonal_prism(polygonB, sep_axis, pointA)) {
                const f3 pivotA = to_object(pointA, posA, ornA);
                const f3 pivotB = to_object(project_plane(pointA, polygonB.origin, sep_axis), posB, ornB);
                res_maybe_add(result, {pivotA, pivotB, sep_axis, distance, normal_attachment});
            }
        }
    if (polygonA.nhull > 2)
        for (int h = 0; h < polygonB.nhull; ++h) {
            const f3 pointB = polygonB.vertices[polygonB.hull[h]];
            if (point_in_polygonal_prism(polygonA, sep_axis, pointB)) {
                const f3 pivotB = to_object(pointB, posB, ornB);
                const f3 pivotA = to_object(project_plane(pointB, polygonA.origin, sep_axis), posA, ornA);
                res_maybe_add(result, {pivotA, pivotB, sep_axis, distance, normal_attachment});
            }
        }
    if (polygonA.nhull > 1 && polygonB.nhull > 1) {
        const int sizeA = polygonA.nhull, sizeB = polygonB.nhull;
        const int limitA = sizeA == 2 ? 1 : sizeA, limitB = sizeB == 2 ? 1 : sizeB;
        float s[2], t[2];
        for (int i = 0; i < limitA; ++i) {
            const int idx0A = polygonA.hull[i], idx1A = polygonA.hull[(i + 1) % sizeA];
            const f2 v0A = polygonA.plane_vertices[idx0A], v1A = polygonA.plane_vertices[idx1A];
            for (int j = 0; j < limitB; ++j) {
                const int idx0B = polygonB.hull[j], idx1B = polygonB.hull[(j + 1) % sizeB];
                const f2 v0B = polygonB.plane_vertices[idx0B], v1B = polygonB.plane_vertices[idx1B];
                const int num_points = intersect_segments(v0A, v1A, v0B, v1B, s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    const f3 pivotA_world = lerp(polygonA.vertices[idx0A], polygonA.vertices[idx1A], s[k]);
                    const f3 pivotB_world = lerp(polygonB.vertices[idx0B], polygonB.vertices[idx1B], t[k]);
                    res_maybe_add(result, {to_object(pivotA_world, posA, ornA), to_object(pivotB_world, posB, ornB), sep_axis, distance, normal_attachment});
                }
            }
        }
    }
}
DI void poly_max_support_direction(const PolySh &shA, f3 posA, const PolySh &shB, f3 posB, f3 &dir, float &distance, float &projectionA, float &projectionB) {
    float max_proj_A = kScalarMax, max_proj_B = -kScalarMax, max_distance = -kScalarMax;
    f3 best_dir{0, 0, 0};
    for (int idx = 0; idx < shA.mesh.nrf(); ++idx) {
        const f3 normal_world = -shA.rot.relevant_normal(idx);
        const int face_idx = shA.mesh.relevant_face(idx);
        const f3 vertexA = shA.rot.vertex(shA.mesh.first_vertex_index(face_idx));
        const f3 vertex_world = vertexA + posA;
        const float projA = dot(vertex_world, normal_world);
        const float projB = polyhedron_support_projection(shB.rot, shB.mesh, normal_world) + dot(posB, normal_world);
        const float dist = projA - projB;
        if (dist > max_distance) { max_distance = dist; max_proj_A = projA; max_proj_B = projB; best_dir = normal_world; }
    }
    dir = best_dir; distance = max_distance; projectionA = max_proj_A; projectionB = max_proj_B;
}
DI void collide_polyhedron_polyhedron(const PolySh &shA, const PolySh &shB, const Ctx &ctx, CResult &result) {
    const f3 posA{0, 0, 0}; const q4 ornA = ctx.ornA;
    const f3 posB = ctx.posB - ctx.posA; const q4 ornB = ctx.ornB;
    const float threshold = ctx.threshold;
    float distance = -kScalarMax, projectionA = kScalarMax, projectionB = -kScalarMax;
    f3 sep_axis{0, 0, 0};
    poly_max_support_direction(shA, posA, shB, posB, sep_axis, distance, projectionA, projectionB);
    {
        float dist, projA, projB; f3 dir;
        poly_max_support_direction(shB, posB, shA, posA, dir, dist, projB, projA);
        if (dist > distance) {
            dir *= -1.0f; projA *= -1.0f; projB *= -1.0f;
            distance = dist; projectionA = projA; projectionB = projB; sep_axis = dir;
        }
    }
    // (device only: the separation found so far can only grow with further axes, so a pair already beyond the threshold ends up with no
    // points whatever the remaining - expensive - axes say: same result as the reference, which tests once after all of them)
    if (distance > threshold) return;
    float min_edge_dist = -kScalarMax;
    // The reference declares edge_projectionA / edge_projectionB / edge_dir without initialisers (:98-100) and reads them even when no
    // edge pair spanned a Minkowski face (parallel edges only: axis-aligned prisms and boxes) - undefined behaviour that, with the
    // wrong stack garbage, runs its quickhull recursion off the stack. Defined here: such a pair has no edge axis.
    float edge_projectionA = -kScalarMax, edge_projectionB = 0;
    f3 edge_dir{0, 0, 0};
    for (int eA = 0; eA < shA.mesh.ne(); ++eA) {
        const f3 normalsA[2] = {shA.rot.edge_normal(2 * eA), shA.rot.edge_normal(2 * eA + 1)};
        f3 verticesA[2] = {shA.rot.edge_vertex(2 * eA), shA.rot.edge_vertex(2 * eA + 1)};
        verticesA[0] += posA; verticesA[1] += posA;
        const f3 edge_dirA = verticesA[0] - verticesA[1];
        for (int eB = 0; eB < shB.mesh.ne(); ++eB) {
            const f3 normalsB[2] = {shB.rot.edge_normal(2 * eB), shB.rot.edge_normal(2 * eB + 1)};
            f3 verticesB[2] = {shB.rot.edge_vertex(2 * eB), shB.rot.edge_vertex(2 * eB + 1)};
            verticesB[0] += posB; verticesB[1] += posB;
            const f3 edge_dirB = verticesB[0] - verticesB[1];
            if (edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) {
                f3 dir = cross(edge_dirA, edge_dirB);
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
    const float edge_distance = edge_projectionA - edge_projectionB;
    if (edge_distance > distance) { distance = edge_distance; projectionA = edge_projectionA; projectionB = edge_projectionB; sep_axis = edge_dir; }
    if (distance > threshold) return;
    SupportPolygon polygonA, polygonB;
    point_cloud_support_polygon(polygonA, shA.rot, shA.mesh.nv(), posA, sep_axis, projectionA, true, kSupportTolerance);
    point_cloud_support_polygon(polygonB, shB.rot, shB.mesh.nv(), posB, sep_axis, projectionB, false, kSupportTolerance);
    polygon_polygon_contacts(polygonA, polygonB, posA, ornA, posB, ornB, sep_axis, distance, result);
}

// ---- collide(polyhedron, polyhedron) by a GROUP of G lanes (narrowphase.hip k_np_pp_axes, k_np_pp_contacts) -------------------------------------------
// The routine above is one lane's serial walk: (faces of A + faces of B) hill climbs, edges of A x edges of B Minkowski tests, two support
// polygons, their hulls, the clipping - ~10k instructions of dependent loads with one wave per SIMD (245 VGPRs, 2.4 KB of scratch per lane
// for the polygons and the hull's stack). Here G lanes share one pair:
//   * every candidate separating axis is one lane's work - the face axes of both polyhedra in one round, the edge pairs G at a time - and
//     the winner is found by a group reduction on (distance, index in the serial order): "the first axis with the largest distance", which
//     is what the serial loops' strict `>` keeps;
//   * the support polygons are collected G vertices at a time (the in-boundary vertices take consecutive slots in vertex order: a ballot
//     and a prefix count) into LDS, lanes 0 and 1 run the two quickhulls side by side on LDS (the hull stack is LDS too), lane 0 clips.
// Same operations on the same values in the same order wherever order matters: bit-identical to collide_polyhedron_polyhedron
// (tests/test_gpu_parity.py: 100k random pairs per mesh combination through this routine, the heaps through the kernel).
constexpr int kNoAxis = 0x7fffffff;
// What the group routines read of one polyhedron: the body's rotated mesh (k_update_rotated ran) and the offsets into the shared tables.
// (Measured and dropped: rotating the table entries where they are used instead - rotate(orn, entry), what k_update_rotated stores, so the
// same bits - to keep the 0.8 GB per step of rotated-mesh reads of the 32k heap out of the L2s: the kernel is bound by instruction issue,
// not by those reads, and the rotations made it 12 % slower.)
struct PPSide {
    const float4 *rot; int nv, nrf, ne; uint32_t rf_off, f_off, nb_off, ni_off;
    DI f3 vertex(int i) const { return from4(rot[i]); }
    DI f3 relevant_normal(int k) const { return from4(rot[nv + k]); }
    DI f3 edge_vertex(int k) const { return from4(rot[nv + nrf + k]); }
    DI f3 edge_normal(int k) const { return from4(rot[nv + nrf + 2 * ne + k]); }
};
DI PPSide pp_side(const Meshes &t, float4 shape, const float4 *rot) {
    const MeshDesc d = t.desc[(uint32_t)shape.x];
    return PPSide{rot, (int)d.nv, (int)d.nrf, (int)d.ne, d.rf_off, d.f_off, d.nb_off, d.ni_off};
}
DI float pp_support_projection(const Meshes &t, const PPSide &Y, f3 dir) {   // polyhedron_support_projection
    int v_idx = 0;
    float max_proj = dot(Y.vertex(0), dir);
    for (;;) {
        const int n0 = (int)t.nb_start[Y.nb_off + v_idx], n1 = (int)t.nb_start[Y.nb_off + v_idx + 1];
        bool done = true;
        for (int i = n0; i < n1; ++i) {
            const int nv_idx = (int)t.nb_idx[Y.ni_off + i];
            const float proj = dot(Y.vertex(nv_idx), dir);
            if (proj > max_proj) { max_proj = proj; v_idx = nv_idx; done = false; }
        }
        if (done) break;
    }
    return max_proj;
}
struct PPAxis { float dist, projX, projY; f3 dir; int idx; };
DI PPAxis pp_no_axis(float projX, float projY) { return PPAxis{-kScalarMax, projX, projY, mk3(0, 0, 0), kNoAxis}; }
// one iteration of poly_max_support_direction: face axis `idx` of X against Y
DI void pp_face_axis(const Meshes &t, const PPSide &X, f3 posX, const PPSide &Y, f3 posY, int idx, PPAxis &best) {
    const f3 normal_world = -X.relevant_normal(idx);
    const int face_idx = (int)t.relevant_faces[X.rf_off + idx];
    const f3 vertexX = X.vertex((int)t.face_first[X.f_off + face_idx]);
    const f3 vertex_world = vertexX + posX;
    const float projX = dot(vertex_world, normal_world);
    const float projY = pp_support_projection(t, Y, normal_world) + dot(posY, normal_world);
    const float dist = projX - projY;
    if (dist > best.dist) best = PPAxis{dist, projX, projY, normal_world, idx};
}
template <int G> DI float gshfl(float v, int src) { return __shfl(v, src, G); }
template <int G> DI PPAxis pp_group_best(PPAxis a) {   // the largest distance, the lowest index among equals: every lane gets the winner
    int who = (int)(threadIdx.x % G);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const float od = __shfl_xor(a.dist, off, G);
        const int oi = __shfl_xor(a.idx, off, G), ow = __shfl_xor(who, off, G);
        if (od > a.dist || (od == a.dist && oi < a.idx)) { a.dist = od; a.idx = oi; who = ow; }
    }
    a.projX = gshfl<G>(a.projX, who); a.projY = gshfl<G>(a.projY, who);
    a.dir = mk3(gshfl<G>(a.dir.x, who), gshfl<G>(a.dir.y, who), gshfl<G>(a.dir.z, who));
    return a;
}
template <int G> DI uint32_t pp_group_ballot(bool pred) {
    const uint64_t b = __ballot(pred);
    return (uint32_t)(b >> (((threadIdx.x & 63u) / G) * G)) & (uint32_t)((1ull << G) - 1ull);
}
DI void pp_group_sync() {   // LDS written by one lane of the wave, read by another
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
struct PPLds { SupportPolygon poly[2]; HullFrame stack[2][kPolyMax + 2]; };   // one per group, in LDS
// point_cloud_support_polygon without its hull, G vertices at a time
template <int G>
DI void pp_group_polygon(SupportPolygon &polygon, const PPSide &S, f3 offset, f3 dir, float projection, bool positive_side, float tolerance) {
    const int j = (int)(threadIdx.x % G);
    const f3 origin = dir * projection;
    const m3 basis = make_tangent_basis(dir);
    const bool zero_offset = eq(offset, mk3(0, 0, 0));
    int n = 0;
    for (int base = 0; base < S.nv; base += G) {
        const int i = base + j;
        bool in_boundary = false;
        f3 vertex_world = mk3(0, 0, 0);
        if (i < S.nv) {
            vertex_world = zero_offset ? S.vertex(i) : S.vertex(i) + offset;
            in_boundary = positive_side ? dot(vertex_world, dir) < projection + tolerance : dot(vertex_world, dir) > projection - tolerance;
        }
        const uint32_t mask = pp_group_ballot<G>(in_boundary);
        const int slot = n + __popc(mask & ((1u << j) - 1u));
        if (in_boundary && slot < kPolyMax) {
            polygon.vertices[slot] = vertex_world;
            polygon.plane_vertices[slot] = to_vector2_xz(to_object(vertex_world, origin, basis));
        }
        n = min(n + (int)__popc(mask), kPolyMax);
    }
    if (j == 0) { polygon.nverts = n; polygon.origin = origin; polygon.basis = basis; }
}
// All G lanes of the group call this together (group-uniform arguments); lane 0 of the group holds the result.
// developer profile (EDYNHIP_PP_PROF=1): every phase boundary adds the 100 MHz clock to its sum; differences of the sums = ticks per phase
template <bool ON> struct PPProf { uint64_t t[8]; DI void stamp(int k) { if (ON) t[k] += wall_clock64(); } };
struct PPSeparation { f3 axis; float distance, projectionA, projectionB; uint32_t hint; };
// The axis that decided a pair, as a hint for the next step: 0 = none, else kind << 28 | index (kind 1 = face axis `index` of A, 2 = of B,
// 3 = edge pair eA * ne(B) + eB). pp_hint_separates tries that one axis before anything else: the final distance of the routine is the
// maximum over its axes, so ONE axis of the routine's own set beyond the threshold means "no points" whatever the others say - the same
// argument as the routine's early return after the face axes. (An axis from outside the set would not do: between a vertex and a vertex
// the set's maximum is smaller than the true distance.) A face axis is evaluated exactly as the loop evaluates it; an edge pair must span
// a Minkowski face now, like in the loop, and - its final projection difference is rounded differently from the distance that selects
// it - must clear the threshold by 1e-3 (rounding: ~1e-6). A stale or meaningless hint costs one wasted axis, never a result.
constexpr uint32_t kHintFaceA = 1u << 28, kHintFaceB = 2u << 28, kHintEdge = 3u << 28, kHintIndex = (1u << 28) - 1u;
DI bool pp_hint_separates(const Meshes &t, const PPSide &A, const PPSide &B, f3 posB, float threshold, uint32_t hint) {
    const f3 posA{0, 0, 0};
    const uint32_t kind = hint & ~kHintIndex;
    const int idx = (int)(hint & kHintIndex);
    if (kind == kHintFaceA || kind == kHintFaceB) {
        const bool second = kind == kHintFaceB;
        if (idx >= (second ? B.nrf : A.nrf)) return false;
        PPAxis a = pp_no_axis(0, 0);
        if (second) pp_face_axis(t, B, posB, A, posA, idx, a); else pp_face_axis(t, A, posA, B, posB, idx, a);
        return a.dist > threshold;
    }
    if (kind != kHintEdge || B.ne <= 0 || idx >= A.ne * B.ne) return false;
    const int eA = idx / B.ne, eB = idx % B.ne;
    const f3 normalsA[2] = {A.edge_normal(2 * eA), A.edge_normal(2 * eA + 1)};
    f3 verticesA[2] = {A.edge_vertex(2 * eA), A.edge_vertex(2 * eA + 1)};
    verticesA[0] += posA; verticesA[1] += posA;
    const f3 edge_dirA = verticesA[0] - verticesA[1];
    const f3 normalsB[2] = {B.edge_normal(2 * eB), B.edge_normal(2 * eB + 1)};
    f3 verticesB[2] = {B.edge_vertex(2 * eB), B.edge_vertex(2 * eB + 1)};
    verticesB[0] += posB; verticesB[1] += posB;
    const f3 edge_dirB = verticesB[0] - verticesB[1];
    if (!edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) return false;
    f3 dir = cross(edge_dirA, edge_dirB);
    if (!try_normalize(dir)) return false;
    if (dot(verticesA[0] - posA, dir) < 0) dir *= -1.0f;
    return dot(verticesB[0] - verticesA[0], dir) > threshold + 1e-3f;
}
// The separating-axis half: all G lanes of the group call this together (group-uniform arguments) and get the same answer - false when the
// polyhedra are further apart than the threshold along some axis (no contact points, as the serial routine's early returns).
template <int G, class Prof>
DI bool pp_group_axes(const Meshes &t, const PPSide &A, const PPSide &B, f3 posB, float threshold, PPSeparation &sep, Prof &prof) {
    const int j = (int)(threadIdx.x % G);
    const f3 posA{0, 0, 0};
    // face axes of A (poly_max_support_direction(A, B)) and of B ((B, A)): one lane each
    PPAxis bestA = pp_no_axis(kScalarMax, -kScalarMax), bestB = pp_no_axis(kScalarMax, -kScalarMax);
    for (int k = j; k < A.nrf + B.nrf; k += G) {
        if (k < A.nrf) pp_face_axis(t, A, posA, B, posB, k, bestA);
        else pp_face_axis(t, B, posB, A, posA, k - A.nrf, bestB);
    }
    bestA = pp_group_best<G>(bestA);
    bestB = pp_group_best<G>(bestB);
    prof.stamp(1);
    float distance = bestA.dist, projectionA = bestA.projX, projectionB = bestA.projY;
    f3 sep_axis = bestA.dir;
    uint32_t hint = bestA.idx != kNoAxis ? kHintFaceA | (uint32_t)bestA.idx : 0u;
    if (bestB.dist > distance) {   // (the second call's X is B: projX is B's projection)
        distance = bestB.dist; projectionA = bestB.projY * -1.0f; projectionB = bestB.projX * -1.0f; sep_axis = bestB.dir * -1.0f;
        hint = kHintFaceB | (uint32_t)bestB.idx;
    }
    sep.hint = hint;
    if (distance > threshold) { prof.stamp(2); return false; }
    // edge pairs in the serial order eA * ne(B) + eB, G at a time (ascending per lane: the strict `>` keeps a lane's first; the group
    // reduction keeps the lowest index among equal distances)
    PPAxis edge = pp_no_axis(-kScalarMax, 0);
    if (A.ne > 0 && B.ne > 0) {
        const int pairs = A.ne * B.ne;
        int eA = j / B.ne, eB = j % B.ne;
        for (int k = j; k < pairs; k += G) {
            const f3 normalsA[2] = {A.edge_normal(2 * eA), A.edge_normal(2 * eA + 1)};
            f3 verticesA[2] = {A.edge_vertex(2 * eA), A.edge_vertex(2 * eA + 1)};
            verticesA[0] += posA; verticesA[1] += posA;
            const f3 edge_dirA = verticesA[0] - verticesA[1];
            const f3 normalsB[2] = {B.edge_normal(2 * eB), B.edge_normal(2 * eB + 1)};
            f3 verticesB[2] = {B.edge_vertex(2 * eB), B.edge_vertex(2 * eB + 1)};
            verticesB[0] += posB; verticesB[1] += posB;
            const f3 edge_dirB = verticesB[0] - verticesB[1];
            if (edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) {
                f3 dir = cross(edge_dirA, edge_dirB);
                if (try_normalize(dir)) {
                    if (dot(verticesA[0] - posA, dir) < 0) dir *= -1.0f;
                    const float edge_dist = dot(verticesB[0] - verticesA[0], dir);
                    if (edge_dist > edge.dist) {
                        dir *= -1.0f;
                        edge = PPAxis{edge_dist, dot(verticesA[0], dir), dot(verticesB[0], dir), dir, k};
                    }
                }
            }
            eB += G;
            while (eB >= B.ne) { eB -= B.ne; ++eA; }
        }
    }
    edge = pp_group_best<G>(edge);
    const float edge_distance = edge.projX - edge.projY;
    if (edge_distance > distance) {
        distance = edge_distance; projectionA = edge.projX; projectionB = edge.projY; sep_axis = edge.dir;
        if (edge.idx != kNoAxis && (uint32_t)edge.idx <= kHintIndex) hint = kHintEdge | (uint32_t)edge.idx;
    }
    prof.stamp(2);
    sep = PPSeparation{sep_axis, distance, projectionA, projectionB, hint};
    return !(distance > threshold);
}
// The contact half, for a pair pp_group_axes let through: the two support polygons along the axis, their hulls, the clipping. Lane 0 of the
// group holds the result.
template <int G, class Prof>
DI void pp_group_contacts(const PPSide &A, const PPSide &B, const Ctx &ctx, const PPSeparation &sep, PPLds &lds, CResult &result, Prof &prof) {
    const int j = (int)(threadIdx.x % G);
    const f3 posA{0, 0, 0};
    const f3 posB = ctx.posB - ctx.posA;
    pp_group_polygon<G>(lds.poly[0], A, posA, sep.axis, sep.projectionA, true, kSupportTolerance);
    pp_group_polygon<G>(lds.poly[1], B, posB, sep.axis, sep.projectionB, false, kSupportTolerance);
    pp_group_sync();
    prof.stamp(3);
    if (j < 2) calculate_convex_hull(lds.poly[j], 0.001f, lds.stack[j]);
    pp_group_sync();
    prof.stamp(4);
    if (j == 0) polygon_polygon_contacts(lds.poly[0], lds.poly[1], posA, ctx.ornA, posB, ctx.ornB, sep.axis, sep.distance, result);
    prof.stamp(5);
}

// ---- collide(polyhedron, box)   collide_polyhedron_box.cpp:14-290 (in the polyhedron's space)
__device__ static const unsigned char kBoxEdgeFaces[24] = {0, 4, 0, 3, 0, 5, 0, 2, 1, 2, 1, 5, 1, 3, 1, 4, 4, 2, 3, 4, 5, 3, 2, 5};   // box_shape.cpp:388-405
DI void collide_polyhedron_box(const PolySh &shA, f3 hB, const Ctx &ctx, CResult &result) {
    const f3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const q4 ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    const f3 box_axes[3] = {quaternion_x(ornB), quaternion_y(ornB), quaternion_z(ornB)};
    float distance = -kScalarMax, projection_poly = 0;
    f3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const f3 normalA = -meshA.normal(face_idx);
        const f3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = box_support_projection(hB, posB, ornB, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    for (int i = 0; i < 3; ++i) {
        f3 dir = box_axes[i];
        if (dot(posB, dir) > 0) dir = -dir;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = dot(posB, dir) + hB[i];
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    // (device only: the separation found so far can only grow with further axes, so a pair already beyond the threshold ends up with no
    // points whatever the remaining - expensive - axes say: same result as the reference, which tests once after all of them)
    if (distance > threshold) return;
    float min_edge_dist = -kScalarMax, edge_projectionA = 0, edge_projectionB = 0;
    f3 edge_dir{0, 0, 0};
    for (int eA = 0; eA < meshA.ne(); ++eA) {
        const f3 normalsA[2] = {meshA.normal(meshA.edge_face(2 * eA)), meshA.normal(meshA.edge_face(2 * eA + 1))};
        const f3 verticesA[2] = {meshA.vertex(meshA.edge_vertex_index(2 * eA)), meshA.vertex(meshA.edge_vertex_index(2 * eA + 1))};
        const f3 edge_dirA = verticesA[0] - verticesA[1];
        for (int eB = 0; eB < 12; ++eB) {
            const f3 normalsB[2] = {rotate(ornB, face_normal(kBoxEdgeFaces[2 * eB])), rotate(ornB, face_normal(kBoxEdgeFaces[2 * eB + 1]))};
            f3 verticesB[2];
            edge_world(hB, eB, posB, ornB, verticesB);
            const f3 edge_dirB = verticesB[0] - verticesB[1];
            if (edges_generate_minkowski_face(normalsA[0], normalsA[1], normalsB[0], normalsB[1], edge_dirA, edge_dirB)) {
                f3 dir = cross(edge_dirA, edge_dirB);
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
    if (!eq(edge_dir, mk3(0, 0, 0))) {
        const float edge_distance = edge_projectionA - edge_projectionB;
        if (edge_distance > distance) { distance = edge_distance; projection_poly = edge_projectionA; sep_axis = edge_dir; }
    }
    if (distance > threshold) return;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), mk3(0, 0, 0), sep_axis, projection_poly, true, kSupportTolerance);
    int featureB, fiB; float proj_unused;
    support_feature(hB, posB, ornB, mk3(0, 0, 0), sep_axis, featureB, fiB, proj_unused, kSupportTolerance);
    const int feature_indexB = fiB;
    CPoint point{};
    point.normal = rotate(ctx.ornA, sep_axis);
    point.distance = distance; point.attachment = NA_NONE;
    if (featureB == BF_FACE) {
        f3 face_verticesB[4];
        face_world(hB, feature_indexB, posB, ornB, face_verticesB);
        point.attachment = NA_ON_B;
        for (int h = 0; h < polygon.nhull; ++h) {
            const f3 pointA = polygon.vertices[polygon.hull[h]];
            if (point_in_quad_prism(face_verticesB, sep_axis, pointA)) {
                point.distance = dot(pointA - face_verticesB[0], sep_axis);
                const f3 pivotB_world = pointA - sep_axis * point.distance;
                point.pivotA = pointA;
                point.pivotB = to_object(pivotB_world, posB, ornB);
                res_maybe_add(result, point);
            }
        }
        if (polygon.nhull > 2)
            for (int i = 0; i < 4; ++i) {
                const f3 pointB = face_verticesB[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.distance = dot(polygon.origin - pointB, sep_axis);
                    point.pivotA = pointB + sep_axis * point.distance;
                    point.pivotB = to_object(pointB, posB, ornB);
                    res_maybe_add(result, point);
                }
            }
        if (polygon.nhull > 1) {
            f2 plane_vertices_box[4];
            for (int i = 0; i < 4; ++i) plane_vertices_box[i] = to_vector2_xz(to_object(face_verticesB[i], polygon.origin, polygon.basis));
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            float s[2], t[2];
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const f2 v0A = polygon.plane_vertices[idx0A], v1A = polygon.plane_vertices[idx1A];
                for (int j = 0; j < 4; ++j) {
                    const int idx0B = j, idx1B = (j + 1) % 4;
                    const int num_points = intersect_segments(v0A, v1A, plane_vertices_box[idx0B], plane_vertices_box[idx1B], s[0], t[0], s[1], t[1]);
                    for (int k = 0; k < num_points; ++k) {
                        point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                        const f3 pivotB_world = lerp(face_verticesB[idx0B], face_verticesB[idx1B], t[k]);
                        point.pivotB = to_object(pivotB_world, posB, ornB);
                        res_maybe_add(result, point);
                    }
                }
            }
        }
    } else if (featureB == BF_EDGE) {
        const f3 edge_vertices_local[2] = {box_vertex(hB, kEdgeIdx[feature_indexB * 2]), box_vertex(hB, kEdgeIdx[feature_indexB * 2 + 1])};
        f3 edge_vertices[2];
        edge_world(hB, feature_indexB, posB, ornB, edge_vertices);
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const f3 pointB = edge_vertices[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB, posB, ornB);
                    poly_add(result, point);
                }
            }
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            const f2 v0B = to_vector2_xz(to_object(edge_vertices[0], polygon.origin, polygon.basis));
            const f2 v1B = to_vector2_xz(to_object(edge_vertices[1], polygon.origin, polygon.basis));
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
            const f3 edge_dir2 = edge_vertices[1] - edge_vertices[0];
            f3 pivotB_world; float t;
            closest_point_line(edge_vertices[0], edge_dir2, point.pivotA, t, pivotB_world);
            point.pivotB = lerp(edge_vertices_local[0], edge_vertices_local[1], t);
            poly_add(result, point);
        }
    } else {
        point.pivotB = box_vertex(hB, feature_indexB);
        const f3 pivotB_world = to_world(point.pivotB, posB, ornB);
        point.pivotA = pivotB_world + sep_axis * distance;
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        poly_add(result, point);
    }
}

// ---- collide(polyhedron, capsule)   collide_polyhedron_capsule.cpp:10-164 (in the polyhedron's space)
DI void collide_polyhedron_capsule(const PolySh &shA, const CylSh &shB, const Ctx &ctx, CResult &result) {
    const f3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const q4 ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    f3 capsule_vertices_[2];
    capsule_vertices(shB, posB, ornB, capsule_vertices_);
    float distance = -kScalarMax, projection_poly = kScalarMax;
    f3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const f3 normalA = -meshA.normal(face_idx);
        const f3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = capsule_support_projection(capsule_vertices_, shB.radius, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    if (distance > threshold) return;   // (device only, see collide_polyhedron_polyhedron)
    for (int i = 0; i < meshA.ne(); ++i) {
        const f3 vertexA0 = meshA.edge_vertex(2 * i), vertexA1 = meshA.edge_vertex(2 * i + 1);
        f3 closestA, closestB;
        { int n_; f3 a_, b_; closest_segment_segment<false>(vertexA0, vertexA1, capsule_vertices_[0], capsule_vertices_[1], closestA, closestB, n_, a_, b_); }
        f3 dir = closestA - closestB;
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = capsule_support_projection(capsule_vertices_, shB.radius, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    if (distance > threshold) return;
    const float proj_capsule_vertices[2] = {dot(capsule_vertices_[0], sep_axis), dot(capsule_vertices_[1], sep_axis)};
    const bool is_capsule_edge = fabsf(proj_capsule_vertices[0] - proj_capsule_vertices[1]) < kSupportTolerance;
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), mk3(0, 0, 0), sep_axis, projection_poly, true, kSupportTolerance);
    CPoint point{};
    point.normal = rotate(ctx.ornA, sep_axis);
    point.distance = distance;
    point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
    if (is_capsule_edge) {
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const f3 pointB = capsule_vertices_[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB + sep_axis * shB.radius, posB, ornB);
                    poly_add(result, point);
                }
            }
        if (result.num == 2) return;
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            float s[2], t[2];
            const f2 plane_capsule_vertices[2] = {to_vector2_xz(to_object(capsule_vertices_[0], polygon.origin, polygon.basis)),
                                                    to_vector2_xz(to_object(capsule_vertices_[1], polygon.origin, polygon.basis))};
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const int num_points = intersect_segments(polygon.plane_vertices[idx0A], polygon.plane_vertices[idx1A], plane_capsule_vertices[0],
                                                          plane_capsule_vertices[1], s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                    const f3 pivotB_world = lerp(capsule_vertices_[0], capsule_vertices_[1], t[k]) + sep_axis * shB.radius;
                    point.pivotB = to_object(pivotB_world, posB, ornB);
                    res_maybe_add(result, point);
                }
            }
        } else {
            point.pivotA = polygon.vertices[polygon.hull[0]];
            const f3 edge_dir = capsule_vertices_[1] - capsule_vertices_[0];
            f3 pivotB_world; float t;
            closest_point_line(capsule_vertices_[0], edge_dir, point.pivotA, t, pivotB_world);
            const f3 normalB = rotate(conjugate(ornB), sep_axis);
            point.pivotB = to_object(pivotB_world, posB, ornB) + normalB * shB.radius;
            poly_add(result, point);
        }
    } else {
        const int closest_capsule_vertex_index = proj_capsule_vertices[0] > proj_capsule_vertices[1] ? 0 : 1;
        const f3 pivotB_world = capsule_vertices_[closest_capsule_vertex_index] + sep_axis * shB.radius;
        point.pivotB = to_object(pivotB_world, posB, ornB);
        point.pivotA = pivotB_world + sep_axis * distance;
        poly_add(result, point);
    }
}

// ---- collide(polyhedron, cylinder)   collide_polyhedron_cylinder.cpp:12-336 (in the polyhedron's space)
DI void collide_polyhedron_cylinder(const PolySh &shA, const CylSh &shB, const Ctx &ctx, CResult &result) {
    const f3 posB = to_object(ctx.posB, ctx.posA, ctx.ornA);
    const q4 ornB = conjugate(ctx.ornA) * ctx.ornB;
    const float threshold = ctx.threshold;
    const MeshView &meshA = shA.mesh;
    const f3 cyl_axis = rotate(ornB, axis_vec(shB.axis));
    const f3 face_center_pos = posB + cyl_axis * shB.half_length, face_center_neg = posB - cyl_axis * shB.half_length;
    float distance = -kScalarMax, projection_poly = kScalarMax;
    f3 sep_axis{0, 0, 0};
    for (int k = 0; k < meshA.nrf(); ++k) {
        const int face_idx = meshA.relevant_face(k);
        const f3 normalA = -meshA.normal(face_idx);
        const f3 vertexA = meshA.vertex(meshA.first_vertex_index(face_idx));
        const float projA = dot(vertexA, normalA);
        const float projB = cylinder_support_projection(shB, posB, ornB, normalA);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = normalA; }
    }
    for (int i = 0; i < 2; ++i) {
        const f3 dir = i == 0 ? cyl_axis : -cyl_axis;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = dot(posB, dir) + shB.half_length;
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    if (distance > threshold) return;   // (device only, see collide_polyhedron_polyhedron)
    for (int k = 0; k < meshA.nre(); ++k) {
        const int edge_idx = meshA.relevant_edge(k);
        const f3 poly_edge = meshA.edge_vertex(2 * edge_idx + 1) - meshA.edge_vertex(2 * edge_idx);
        f3 dir = cross(poly_edge, cyl_axis);
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    for (int v = 0; v < meshA.nv(); ++v) {
        const f3 rvertex = meshA.vertex(v);
        f3 closest; float t;
        closest_point_line(face_center_neg, cyl_axis, rvertex, t, closest);
        f3 dir = rvertex - closest;
        if (!try_normalize(dir)) continue;
        if (dot(posB, dir) > 0) dir *= -1.0f;
        const float projA = -polyhedron_support_projection(meshA, meshA, -dir);
        const float projB = cylinder_support_projection(shB, posB, ornB, dir);
        const float dist = projA - projB;
        if (dist > distance) { distance = dist; projection_poly = projA; sep_axis = dir; }
    }
    if (distance > threshold) return;   // (device only, see collide_polyhedron_polyhedron)
    for (int i = 0; i < meshA.ne(); ++i) {
        const f3 vertexA0 = meshA.edge_vertex(2 * i), vertexA1 = meshA.edge_vertex(2 * i + 1);
        for (int j = 0; j < 2; ++j) {
            const f3 face_center = j == 0 ? face_center_neg : face_center_pos;
            int num_points; float s0, s1; f3 cc0, cl0, cc1, cl1, dir;
            closest_point_circle_line(face_center, ornB, shB.radius, shB.axis, vertexA0, vertexA1, num_points, s0, cc0, cl0, s1, cc1, cl1, dir, kSupportTolerance);
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
    const f3 normal = rotate(ctx.ornA, sep_axis);
    SupportPolygon polygon;
    point_cloud_support_polygon(polygon, meshA, meshA.nv(), mk3(0, 0, 0), sep_axis, projection_poly, true, kSupportTolerance);
    int featureB; int feature_indexB = 0;
    cylinder_support_feature(shB, posB, ornB, sep_axis, featureB, feature_indexB, kSupportTolerance);
    CPoint point{};
    point.normal = normal; point.distance = distance; point.attachment = NA_NONE;
    const int ai = shB.axis, o0 = (ai + 1) % 3, o1 = (ai + 2) % 3;
    if (featureB == CF_FACE) {
        int num_vertices_in_face = 0;
        const float sign_faceB = to_sign(feature_indexB == 0);
        const float pivotB_axis = shB.half_length * sign_faceB;
        point.attachment = NA_ON_B;
        for (int h = 0; h < polygon.nhull; ++h) {
            const f3 pointA = polygon.vertices[polygon.hull[h]];
            f3 closest; float t;
            const float dist_sqr = closest_point_line(posB, cyl_axis, pointA, t, closest);
            if (dist_sqr > shB.radius * shB.radius) continue;
            point.pivotA = pointA;
            point.pivotB = to_object(pointA, posB, ornB);
            point.pivotB[ai] = pivotB_axis;
            res_maybe_add(result, point);
            ++num_vertices_in_face;
        }
        int num_edge_intersections = 0;
        const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
        for (int i = 0; i < limitA; ++i) {
            const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
            const f3 v0A = polygon.vertices[idx0A], v1A = polygon.vertices[idx1A];
            const f3 v0B = to_object(v0A, posB, ornB), v1B = to_object(v1A, posB, ornB);
            float s[2];
            const int num_points = intersect_line_circle(f2{v0B.z, v0B.y}, f2{v1B.z, v1B.y}, shB.radius, s[0], s[1]);   // (to_vector2_zy whatever the axis: as the reference)
            for (int j = 0; j < num_points; ++j) {
                const float t = s[j];
                if (t < 0 || t > 1) continue;
                point.pivotA = lerp(v0A, v1A, t);
                point.pivotB = lerp(v0B, v1B, t);
                point.pivotB[ai] = pivotB_axis;
                res_maybe_add(result, point);
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
                    res_maybe_add(result, point);
                }
            }
        }
    } else if (featureB == CF_SIDE_EDGE) {
        const f3 edge_vertices[2] = {face_center_neg + sep_axis * shB.radius, face_center_pos + sep_axis * shB.radius};
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        if (polygon.nhull > 2)
            for (int i = 0; i < 2; ++i) {
                const f3 pointB = edge_vertices[i];
                if (point_in_polygonal_prism(polygon, sep_axis, pointB)) {
                    point.pivotA = project_plane(pointB, polygon.origin, sep_axis);
                    point.pivotB = to_object(pointB, posB, ornB);
                    res_maybe_add(result, point);
                }
            }
        if (result.num == 2) return;
        if (polygon.nhull > 1) {
            const int sizeA = polygon.nhull, limitA = sizeA == 2 ? 1 : sizeA;
            const f2 v0B = to_vector2_xz(to_object(edge_vertices[0], polygon.origin, polygon.basis));
            const f2 v1B = to_vector2_xz(to_object(edge_vertices[1], polygon.origin, polygon.basis));
            float s[2], t[2];
            for (int i = 0; i < limitA; ++i) {
                const int idx0A = polygon.hull[i], idx1A = polygon.hull[(i + 1) % sizeA];
                const int num_points = intersect_segments(polygon.plane_vertices[idx0A], polygon.plane_vertices[idx1A], v0B, v1B, s[0], t[0], s[1], t[1]);
                for (int k = 0; k < num_points; ++k) {
                    point.pivotA = lerp(polygon.vertices[idx0A], polygon.vertices[idx1A], s[k]);
                    const f3 pivotB_world = lerp(edge_vertices[0], edge_vertices[1], t[k]);
                    point.pivotB = to_object(pivotB_world, posB, ornB);
                    res_maybe_add(result, point);
                }
            }
        } else {
            point.pivotA = polygon.vertices[polygon.hull[0]];
            const f3 edge_dir = edge_vertices[1] - edge_vertices[0];
            f3 pivotB_world; float t;
            closest_point_line(edge_vertices[0], edge_dir, point.pivotA, t, pivotB_world);
            point.pivotB = to_object(pivotB_world, posB, ornB);
            poly_add(result, point);
        }
    } else {
        const f3 supportB = cylinder_support_point(shB, posB, ornB, sep_axis);
        point.pivotA = supportB + sep_axis * distance;
        point.pivotB = to_object(supportB, posB, ornB);
        point.attachment = polygon.nhull > 2 ? NA_ON_A : NA_NONE;
        poly_add(result, point);
    }
}


// Pairs that involve a polyhedron, incl. swap_collide (collide.hpp:369-374); other pairs leave the result untouched (returns false).
// rotA / rotB: the bodies' rotated meshes (only read for a polyhedron).
DI bool collide_poly(const Meshes &t, int tA, float4 sA, const float4 *rotA, int tB, float4 sB, const float4 *rotB, const Ctx &c, CResult &r) {
    if (tA != SHAPE_POLYHEDRON && tB != SHAPE_POLYHEDRON) return false;
    r.num = 0;
    const bool first = tA == SHAPE_POLYHEDRON;
    const Ctx cc = first ? c : Ctx{c.posB, c.ornB, c.posA, c.ornA, c.threshold};
    const PolySh P = poly_of(t, first ? sA : sB, first ? rotA : rotB);
    const int q = first ? tB : tA;
    const float4 sQ = first ? sB : sA;
    if (q == SHAPE_PLANE) collide_polyhedron_plane(P, from4(sQ), sQ.w, cc, r);
    else if (q == SHAPE_SPHERE) collide_polyhedron_sphere(P, sQ.x, cc, r);
    else if (q == SHAPE_BOX) collide_polyhedron_box(P, from4(sQ), cc, r);
    else if (q == SHAPE_CAPSULE) collide_polyhedron_capsule(P, cyl_of(sQ), cc, r);
    else if (q == SHAPE_CYLINDER) collide_polyhedron_cylinder(P, cyl_of(sQ), cc, r);
    else if (q == SHAPE_POLYHEDRON) collide_polyhedron_polyhedron(P, poly_of(t, sB, rotB), cc, r);
    if (!first)
        for (int i = 0; i < r.num; ++i) cp_swap(r.pt[i]);
    return true;
}

}  // namespace dc
