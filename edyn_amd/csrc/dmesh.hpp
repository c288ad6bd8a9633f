// Convex mesh tables of a context (polyhedron_shape): see dpolyhedron.hpp for the routines, mesh.hip for the host side.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dc {

constexpr int kPolyMax = 32;   // vertices of a support polygon (and so of a mesh face)

struct MeshDesc {
    uint32_t v_off, nv;      // vertices, nb_start (at v_off + mesh id, nv + 1 entries)
    uint32_t f_off, nf;      // normals, face_first
    uint32_t e_off, ne;      // edge_vertices / edge_normals / edge_vidx / edge_faces at 2 * (e_off + e) + {0, 1}
    uint32_t rf_off, nrf;    // relevant_faces, relevant_normals
    uint32_t re_off, nre;    // relevant_edges
    uint32_t nb_off, ni_off; // nb_start, nb_idx
    float isum[7];           // moment_of_inertia_polyhedron's sums over the faces: volume, xx, yy, zz, yz, zx, xy
    uint32_t rot_size;       // float4 per body for the rotated mesh: nv + nrf + 4 ne
};
struct Meshes {
    const MeshDesc *desc = nullptr;
    const float4 *vertices = nullptr, *normals = nullptr, *edge_vertices = nullptr, *edge_normals = nullptr, *relevant_normals = nullptr;
    const uint32_t *face_first = nullptr, *edge_vidx = nullptr, *edge_faces = nullptr, *relevant_faces = nullptr, *relevant_edges = nullptr,
                   *nb_start = nullptr, *nb_idx = nullptr;
    float4 *rot = nullptr;              // rotated meshes of the polyhedron bodies, each at rot_off[body]
    const uint32_t *rot_off = nullptr;  // per body (~0u: not a polyhedron)
    uint32_t num = 0;
};

}  // namespace dc
