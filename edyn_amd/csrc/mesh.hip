// Convex meshes of a context (polyhedron_shape): creation, the device tables, the per-body rotated meshes.
//   /root/reference/src/edyn/shapes/convex_mesh.cpp:10-230 (initialize: shift_to_centroid, calculate_normals / edges / neighbors /
//       relevant_faces / relevant_edges), src/edyn/util/shape_util.cpp:351-391 (mesh_centroid),
//   /root/reference/src/edyn/dynamics/moment_of_inertia.cpp:93-157, src/edyn/sys/update_rotated_meshes.cpp:12-76
// A mesh is derived once, on the host, in the reference's operation order (single precision, no contraction: this file is compiled
// with -ffp-contract=off like the kernels), appended to the context's flat tables and uploaded; bodies refer to it by id
// (shape_param[0]). The rotated mesh of every polyhedron body is recomputed from its current orientation at the head of each
// narrowphase - what update_rotated_meshes leaves behind after an integration, and immune to edits of the state between steps.
#include "ctx.hpp"
#include "dpolyhedron.hpp"
#include <cmath>

namespace eh {
using namespace dm;
using namespace dc;

namespace {
struct H3 { float x, y, z; };
inline H3 operator+(H3 a, H3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 operator-(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 operator*(H3 a, H3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline H3 operator/(H3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float hlen2(H3 a) { return hdot(a, a); }
inline H3 hnormalize(H3 a) { return a / std::sqrt(hlen2(a)); }                        // vector3.hpp normalize: true division
inline bool htry_normalize(H3 &v) {                                                   // vector3.hpp:233-241: multiplies by the reciprocal
    const float l2 = hlen2(v);
    if ((double)l2 > 1e-18) { const float z = 1.0f / std::sqrt(l2); v = {v.x * z, v.y * z, v.z * z}; return true; }
    return false;
}
inline float4 h4(H3 v) { return make_float4(v.x, v.y, v.z, 0.0f); }
constexpr float kRelevantDirectionTolerance = 0.0006f;   // convex_mesh_relevant_direction_tolerance, config/constants.hpp
}  // namespace

// Appends one mesh to the host tables. Returns an error text or nullptr.
static const char *append_mesh(edynhip_ctx::HostMeshes &t, uint32_t nv, const float *verts, uint32_t nidx, const uint32_t *indices, uint32_t nfaces, const uint32_t *faces, bool centred) {
    if (nv < 4 || nfaces < 4 || !verts || !indices || !faces) return "a convex mesh needs at least 4 vertices and 4 faces";
    for (uint32_t f = 0; f < nfaces; ++f) {
        const uint32_t first = faces[2 * f], count = faces[2 * f + 1];
        if (count < 3 || (uint64_t)first + count > nidx) return "face index range out of bounds";
        if (count > (uint32_t)kPolyMax) return "a face has more vertices than a support polygon can hold (32)";
    }
    for (uint32_t i = 0; i < nidx; ++i) if (indices[i] >= nv) return "vertex index out of range";
    std::vector<H3> v(nv);
    for (uint32_t i = 0; i < nv; ++i) v[i] = {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
    auto face_vertex = [&](uint32_t f, uint32_t k) { return indices[faces[2 * f] + k]; };
    if (!centred) {   // shift_to_centroid: mesh_centroid (shape_util.cpp:351-391), then every vertex minus it
        H3 center{0, 0, 0};
        float volume = 0;
        for (uint32_t f = 0; f < nfaces; ++f) {
            const uint32_t count = faces[2 * f + 1];
            const H3 v0 = v[face_vertex(f, 0)];
            for (uint32_t j = 1; j + 1 < count; ++j) {
                const H3 v1 = v[face_vertex(f, j)], v2 = v[face_vertex(f, j + 1)];
                const H3 normal = hcross(v1 - v0, v2 - v1);
                volume += hdot(v0, normal);
                const H3 vx{v0.x + v1.x, v1.x + v2.x, v2.x + v0.x}, vy{v0.y + v1.y, v1.y + v2.y, v2.y + v0.y}, vz{v0.z + v1.z, v1.z + v2.z, v2.z + v0.z};
                center = center + normal * H3{hlen2(vx), hlen2(vy), hlen2(vz)};
            }
        }
        volume /= 6;
        const float z = 1.0f / (24 * 2 * volume);   // vector3 operator/=
        center = {center.x * z, center.y * z, center.z * z};
        for (H3 &p : v) p = p - center;
    }
    MeshDesc d{};
    d.v_off = (uint32_t)t.vertices.size(); d.nv = nv;
    d.f_off = (uint32_t)t.normals.size(); d.nf = nfaces;
    d.e_off = (uint32_t)t.edge_vidx.size() / 2;
    d.rf_off = (uint32_t)t.relevant_faces.size();
    d.re_off = (uint32_t)t.relevant_edges.size();
    d.nb_off = (uint32_t)t.nb_start.size(); d.ni_off = (uint32_t)t.nb_idx.size();
    for (const H3 &p : v) t.vertices.push_back(h4(p));
    // calculate_normals (:86-121)
    std::vector<H3> normals(nfaces);
    for (uint32_t f = 0; f < nfaces; ++f) {
        const uint32_t count = faces[2 * f + 1];
        const H3 v0 = v[face_vertex(f, 0)], v1 = v[face_vertex(f, 1)];
        H3 normal{0, 0, 0};
        for (uint32_t j = 1; j < count; ++j) {
            H3 n = hcross(v1 - v0, v[face_vertex(f, (j + 1) % count)] - v[face_vertex(f, j)]);
            if (htry_normalize(n)) { normal = n; break; }
        }
        if (normal.x == 0 && normal.y == 0 && normal.z == 0) normal = {0, 1, 0};
        normals[f] = normal;
        t.normals.push_back(h4(normal));
        t.face_first.push_back(face_vertex(f, 0));
    }
    // calculate_edges (:123-173): unique edges in order of first appearance, with the (up to) two faces that share them
    std::vector<uint32_t> e_v, e_f;
    std::vector<H3> e_n;
    for (uint32_t f = 0; f < nfaces; ++f) {
        const uint32_t count = faces[2 * f + 1];
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t i0 = face_vertex(f, k), i1 = face_vertex(f, (k + 1) % count);
            bool known = false;
            for (size_t e = 0; e < e_v.size() / 2 && !known; ++e)
                if ((e_v[2 * e] == i0 && e_v[2 * e + 1] == i1) || (e_v[2 * e] == i1 && e_v[2 * e + 1] == i0)) {
                    known = true; e_f[2 * e + 1] = f; e_n[2 * e + 1] = normals[f];
                }
            if (!known) {
                e_v.push_back(i0); e_v.push_back(i1);
                e_f.push_back(f); e_f.push_back(0xFFFFFFFFu);
                e_n.push_back(normals[f]); e_n.push_back(H3{0, 0, 0});
            }
        }
    }
    const uint32_t ne = (uint32_t)e_v.size() / 2;
    d.ne = ne;
    for (uint32_t k = 0; k < 2 * ne; ++k) {
        if (e_f[k] == 0xFFFFFFFFu) return "the mesh is not closed (an edge belongs to one face only)";
        t.edge_vidx.push_back(e_v[k]); t.edge_faces.push_back(e_f[k]);
        t.edge_vertices.push_back(h4(v[e_v[k]])); t.edge_normals.push_back(h4(e_n[k]));
    }
    // calculate_neighbors (:175-194)
    uint32_t count = 0;
    t.nb_start.push_back(0);
    for (uint32_t i = 0; i < nv; ++i) {
        for (uint32_t e = 0; e < ne; ++e)
            if (e_v[2 * e] == i || e_v[2 * e + 1] == i) { t.nb_idx.push_back(e_v[2 * e] == i ? e_v[2 * e + 1] : e_v[2 * e]); ++count; }
        t.nb_start.push_back(count);
    }
    // calculate_relevant_faces (:196-211): one face per direction
    std::vector<uint32_t> rf;
    for (uint32_t f = 0; f < nfaces; ++f) {
        bool found = false;
        for (uint32_t o : rf) if (!(hdot(normals[f], normals[o]) < 1.0f - kRelevantDirectionTolerance)) { found = true; break; }
        if (!found) { rf.push_back(f); t.relevant_faces.push_back(f); t.relevant_normals.push_back(h4(normals[f])); }
    }
    d.nrf = (uint32_t)rf.size();
    // calculate_relevant_edges (:213-230): one edge per direction (either sense)
    std::vector<uint32_t> re;
    auto edge_dir = [&](uint32_t e) { return v[e_v[2 * e + 1]] - v[e_v[2 * e]]; };
    for (uint32_t e = 0; e < ne; ++e) {
        const H3 edge = hnormalize(edge_dir(e));
        bool found = false;
        for (uint32_t o : re) if (!(std::fabs(hdot(edge, hnormalize(edge_dir(o)))) < 1.0f - kRelevantDirectionTolerance)) { found = true; break; }
        if (!found) { re.push_back(e); t.relevant_edges.push_back(e); }
    }
    d.nre = (uint32_t)re.size();
    {   // moment_of_inertia_polyhedron (moment_of_inertia.cpp:93-141): the sums; the mass enters per body (dpolyhedron.hpp polyhedron_inertia)
        float volume = 0, xx = 0, yy = 0, zz = 0, yz = 0, zx = 0, xy = 0;
        for (uint32_t f = 0; f < nfaces; ++f) {
            const uint32_t cnt = faces[2 * f + 1];
            const H3 v0 = v[face_vertex(f, 0)];
            for (uint32_t j = 1; j + 1 < cnt; ++j) {
                const H3 v1 = v[face_vertex(f, j)], v2 = v[face_vertex(f, j + 1)];
                const float pd = hdot(v0, hcross(v1, v2));
                volume += pd;
                const H3 v3 = v0 + v1 + v2;
                xx += pd * (v0.x * v0.x + v1.x * v1.x + v2.x * v2.x + v3.x * v3.x);
                yy += pd * (v0.y * v0.y + v1.y * v1.y + v2.y * v2.y + v3.y * v3.y);
                zz += pd * (v0.z * v0.z + v1.z * v1.z + v2.z * v2.z + v3.z * v3.z);
                yz += pd * (v0.y * v0.z + v1.y * v1.z + v2.y * v2.z + v3.y * v3.z);
                zx += pd * (v0.z * v0.x + v1.z * v1.x + v2.z * v2.x + v3.z * v3.x);
                xy += pd * (v0.x * v0.y + v1.x * v1.y + v2.x * v2.y + v3.x * v3.y);
            }
        }
        const float s[7] = {volume, xx, yy, zz, yz, zx, xy};
        for (int k = 0; k < 7; ++k) d.isum[k] = s[k];
        if (!(volume > 0)) return "the mesh has no positive volume (faces must wind counter-clockwise seen from outside)";
    }
    if (nv > (uint32_t)kPolyMax) {
        // A support polygon (dpolyhedron.hpp point_cloud_support_polygon) gathers EVERY vertex within support_feature_tolerance of
        // the supporting plane of a direction - across coplanar faces (a triangulated cap) and across facets narrower than the
        // tolerance. Its arrays hold kPolyMax vertices (std::vector in the reference), so a mesh that could put more than that
        // into one of them is refused here instead of being clipped silently: checked for every face normal and for the
        // direction between the two faces of every edge, with a margin on the tolerance.
        const float tol = 0.005f + 1e-4f;   // support_feature_tolerance, config/constants.hpp:56
        auto crowded = [&](H3 dir) {
            float top = -3.0e38f;
            for (const H3 &p : v) top = std::max(top, hdot(p, dir));
            uint32_t within = 0;
            for (const H3 &p : v) if (hdot(p, dir) > top - tol) ++within;
            return within > (uint32_t)kPolyMax;
        };
        for (uint32_t f = 0; f < nfaces; ++f)
            if (crowded(normals[f])) return "more than 32 vertices lie within the support tolerance of one face plane (a support polygon holds 32)";
        for (uint32_t e = 0; e < ne; ++e) {
            H3 mid = e_n[2 * e] + e_n[2 * e + 1];
            if (htry_normalize(mid) && crowded(mid)) return "more than 32 vertices lie within the support tolerance of one supporting plane (a support polygon holds 32)";
        }
    }
    d.rot_size = d.nv + d.nrf + 4 * d.ne;
    t.desc.push_back(d);
    return nullptr;
}

template <typename T>
static int put(edynhip_ctx *c, const std::vector<T> &h, const T *&dev, std::vector<void *> &allocs) {
    void *q = nullptr;
    EH_HIP(c, hipMalloc(&q, std::max<size_t>(1, h.size()) * sizeof(T)));
    allocs.push_back(q);
    if (!h.empty()) EH_HIP(c, hipMemcpyAsync(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    dev = (const T *)q;
    return EDYNHIP_OK;
}
// Uploads the tables `t` into FRESH allocations; only when every copy has succeeded do they replace the context's tables (and the
// old allocations go). A failure leaves the context exactly as it was - device tables, host tables, registered meshes.
static int upload_meshes(edynhip_ctx *c, edynhip_ctx::HostMeshes &t) {
    std::vector<void *> fresh;
    Meshes m = c->meshes;
    auto all = [&]() -> int {
        EH_TRY(put(c, t.desc, m.desc, fresh));
        EH_TRY(put(c, t.vertices, m.vertices, fresh)); EH_TRY(put(c, t.normals, m.normals, fresh)); EH_TRY(put(c, t.edge_vertices, m.edge_vertices, fresh));
        EH_TRY(put(c, t.edge_normals, m.edge_normals, fresh)); EH_TRY(put(c, t.relevant_normals, m.relevant_normals, fresh));
        EH_TRY(put(c, t.face_first, m.face_first, fresh)); EH_TRY(put(c, t.edge_vidx, m.edge_vidx, fresh)); EH_TRY(put(c, t.edge_faces, m.edge_faces, fresh));
        EH_TRY(put(c, t.relevant_faces, m.relevant_faces, fresh)); EH_TRY(put(c, t.relevant_edges, m.relevant_edges, fresh));
        EH_TRY(put(c, t.nb_start, m.nb_start, fresh)); EH_TRY(put(c, t.nb_idx, m.nb_idx, fresh));
        EH_HIP(c, hipStreamSynchronize(c->stream));   // the copies read t: done before anybody may touch it; also drains the steps that read the old tables
        return EDYNHIP_OK;
    };
    const int rc = all();
    if (rc != EDYNHIP_OK) {
        (void)hipStreamSynchronize(c->stream);
        for (void *p : fresh) (void)hipFree(p);
        return rc;
    }
    m.num = (uint32_t)t.desc.size();
    for (void *p : c->mesh_allocs) (void)hipFree(p);
    c->mesh_allocs.swap(fresh);
    c->meshes = m;
    c->host_meshes.swap(t);
    return EDYNHIP_OK;
}

// Called by load_bodies (capi.hip) before the bodies are initialised: checks the mesh ids of the polyhedra among bodies
// [first, first + n) and gives each its slice of the rotated-mesh buffer.
int mesh_bind_bodies(edynhip_ctx *c, uint32_t first, uint32_t n, const int32_t *shape_type, const float *shape_param) {
    if (first == 0) { c->has_polyhedron = false; c->rot_used = 0; c->host_rot_off.clear(); }
    c->host_rot_off.resize(first, 0xFFFFFFFFu);
    bool any = false;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t off = 0xFFFFFFFFu;
        if (shape_type[i] == EDYNHIP_SHAPE_POLYHEDRON) {
            const float id = shape_param[4 * i];
            if (!(id >= 0) || id != std::floor(id) || id >= (float)c->host_meshes.desc.size())
                return set_error(c, EDYNHIP_ERR_INVALID, "polyhedron: shape_param[0] is not the id of a mesh created with edynhip_create_convex_mesh");
            off = (uint32_t)c->rot_used;
            c->rot_used += c->host_meshes.desc[(size_t)id].rot_size;
            any = true;
        }
        c->host_rot_off.push_back(off);
    }
    if (!any && !c->has_polyhedron) return EDYNHIP_OK;
    c->has_polyhedron = c->has_polyhedron || any;
    if (!c->poly_work) {   // the polyhedron narrowphase's binning scratch (narrowphase.hip PolyBins), with the first polyhedron of the context
        const size_t cap = c->m[0].cap, words = 3 * 1024 + 1 + 4 * cap;   // bins, total, keys, list, two hint arrays
        EH_HIP(c, hipMalloc((void **)&c->poly_work, words * sizeof(uint32_t)));
        EH_HIP(c, hipMemsetAsync(c->poly_work, 0, (3 * 1024 + 1) * sizeof(uint32_t), c->stream));
        EH_HIP(c, hipMemsetAsync(c->poly_work + 3 * 1024 + 1 + 2 * cap, 0, 2 * cap * sizeof(uint32_t), c->stream));   // no hints yet
    }
    if (c->rot_used > c->rot_cap) {   // contents are recomputed before every use: nothing to carry over
        EH_HIP(c, hipStreamSynchronize(c->stream));
        if (c->rot) (void)hipFree(c->rot);
        c->rot = nullptr;
        c->rot_cap = std::max<size_t>(c->rot_used, c->rot_cap * 2);
        EH_HIP(c, hipMalloc((void **)&c->rot, c->rot_cap * sizeof(float4)));
    }
    EH_HIP(c, hipMemcpyAsync(c->rot_off + first, c->host_rot_off.data() + first, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    EH_HIP(c, hipStreamSynchronize(c->stream));
    c->meshes.rot = c->rot;
    c->meshes.rot_off = c->rot_off;
    return EDYNHIP_OK;
}

// update_rotated_meshes.cpp:54-76 for every polyhedron body: one workgroup per body, one item per lane and round.
__global__ void __launch_bounds__(64)
k_update_rotated(uint32_t n, Bodies b, Meshes t) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t fl = b.flags[i];
    if ((int)((fl & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT) != SHAPE_POLYHEDRON || (fl & BF_REMOVED)) return;
    const MeshDesc d = t.desc[(uint32_t)b.shape[i].x];
    const q4 orn = q_from4(B_ORN(b, i)) * q4{0, 0, 0, 1};   // orientation * rotated_mesh_list::orientation (identity outside a compound)
    float4 *base = t.rot + t.rot_off[i];
    for (uint32_t k = threadIdx.x; k < d.rot_size; k += 64) rotate_mesh_item(t, d, orn, base, k);
}
int update_rotated(edynhip_ctx *c) {
    if (!c->has_polyhedron || c->b.n == 0) return EDYNHIP_OK;
    hipLaunchKernelGGL(k_update_rotated, dim3(c->b.n), dim3(64), 0, c->stream, c->b.n, c->b, c->meshes);
    EH_HIP(c, hipGetLastError());
    return EDYNHIP_OK;
}

}  // namespace eh

using namespace eh;

extern "C" {

int edynhip_create_convex_mesh(edynhip_ctx *c, uint32_t num_vertices, const float *vertices, uint32_t num_indices, const uint32_t *indices,
                               uint32_t num_faces, const uint32_t *faces, uint32_t flags, uint32_t *mesh_id) {
    if (!c || !mesh_id || (flags & ~(uint32_t)EDYNHIP_MESH_INITIALIZED)) return EDYNHIP_ERR_INVALID;
    EH_HIP(c, hipSetDevice(c->device));
    edynhip_ctx::HostMeshes trial = c->host_meshes;   // a rejected mesh leaves the tables as they were
    if (const char *why = append_mesh(trial, num_vertices, vertices, num_indices, indices, num_faces, faces, (flags & EDYNHIP_MESH_INITIALIZED) != 0))
        return set_error(c, EDYNHIP_ERR_INVALID, (std::string("edynhip_create_convex_mesh: ") + why).c_str());
    EH_TRY(upload_meshes(c, trial));   // swaps the tables in on success only
    *mesh_id = (uint32_t)c->host_meshes.desc.size() - 1;
    return EDYNHIP_OK;
}

int edynhip_get_convex_mesh(edynhip_ctx *c, uint32_t mesh_id, int field, void *out, uint32_t capacity, uint32_t *count) {
    if (!c || !count || mesh_id >= c->host_meshes.desc.size()) return EDYNHIP_ERR_INVALID;
    const auto &t = c->host_meshes;
    const MeshDesc &d = t.desc[mesh_id];
    const float4 *fsrc = nullptr; const uint32_t *usrc = nullptr; uint32_t n = 0;
    switch (field) {
    case EDYNHIP_MESH_VERTICES: fsrc = t.vertices.data() + d.v_off; n = d.nv; break;
    case EDYNHIP_MESH_NORMALS: fsrc = t.normals.data() + d.f_off; n = d.nf; break;
    case EDYNHIP_MESH_RELEVANT_NORMALS: fsrc = t.relevant_normals.data() + d.rf_off; n = d.nrf; break;
    case EDYNHIP_MESH_EDGE_VERTICES: fsrc = t.edge_vertices.data() + 2 * d.e_off; n = 2 * d.ne; break;
    case EDYNHIP_MESH_EDGE_NORMALS: fsrc = t.edge_normals.data() + 2 * d.e_off; n = 2 * d.ne; break;
    case EDYNHIP_MESH_EDGES: usrc = t.edge_vidx.data() + 2 * d.e_off; n = 2 * d.ne; break;
    case EDYNHIP_MESH_EDGE_FACES: usrc = t.edge_faces.data() + 2 * d.e_off; n = 2 * d.ne; break;
    case EDYNHIP_MESH_RELEVANT_FACES: usrc = t.relevant_faces.data() + d.rf_off; n = d.nrf; break;
    case EDYNHIP_MESH_RELEVANT_EDGES: usrc = t.relevant_edges.data() + d.re_off; n = d.nre; break;
    case EDYNHIP_MESH_NEIGHBORS_START: usrc = t.nb_start.data() + d.nb_off; n = d.nv + 1; break;
    case EDYNHIP_MESH_NEIGHBOR_INDICES: usrc = t.nb_idx.data() + d.ni_off; n = t.nb_start[d.nb_off + d.nv]; break;
    case EDYNHIP_MESH_INERTIA_SUMS: n = 7; break;
    default: return set_error(c, EDYNHIP_ERR_INVALID, "edynhip_get_convex_mesh: unknown field");
    }
    *count = n;
    if (!out) return EDYNHIP_OK;
    if (capacity < n) return set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_convex_mesh: capacity");
    if (field == EDYNHIP_MESH_INERTIA_SUMS) for (int k = 0; k < 7; ++k) ((float *)out)[k] = d.isum[k];
    else if (fsrc) for (uint32_t k = 0; k < n; ++k) { ((float *)out)[3 * k] = fsrc[k].x; ((float *)out)[3 * k + 1] = fsrc[k].y; ((float *)out)[3 * k + 2] = fsrc[k].z; }
    else std::copy(usrc, usrc + n, (uint32_t *)out);
    return EDYNHIP_OK;
}

}  // extern "C"
