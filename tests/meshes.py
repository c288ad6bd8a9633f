"""Convex meshes for the polyhedron parity tests: (vertices, indices, faces) as edyn::convex_mesh takes them
(convex_mesh.hpp:17-66: faces = (first index, vertex count) pairs, counter-clockwise seen from outside).
The same list, created in the same order, gives the same mesh ids in the oracle, the reference driver and the device."""
import numpy as np


def _mesh(verts, face_lists):
    verts = np.asarray(verts, np.float32)
    indices, faces = [], []
    for f in face_lists:
        faces += [len(indices), len(f)]
        indices += list(f)
    return dict(vertices=verts, indices=np.asarray(indices, np.uint32), faces=np.asarray(faces, np.uint32).reshape(-1, 2))


def box_mesh(he):   # the reference's make_box_mesh (shape_util.cpp:12-38)
    x, y, z = he
    v = [(-x, -y, -z), (x, -y, -z), (x, -y, z), (-x, -y, z), (-x, y, -z), (x, y, -z), (x, y, z), (-x, y, z)]
    return _mesh(v, [(0, 1, 2, 3), (7, 6, 5, 4), (4, 5, 1, 0), (6, 7, 3, 2), (7, 4, 0, 3), (5, 6, 2, 1)])


def _oriented(verts, faces):
    """Orient every face counter-clockwise seen from outside (the centroid of the vertex cloud is inside)."""
    v = np.asarray(verts, np.float64)
    c = v.mean(axis=0)
    out = []
    for f in faces:
        p = v[list(f)]
        n = np.cross(p[1] - p[0], p[2] - p[1])
        out.append(tuple(f) if np.dot(n, p[0] - c) > 0 else tuple(reversed(f)))
    return out


def tetrahedron(s=0.5):
    v = [(s, s, s), (s, -s, -s), (-s, s, -s), (-s, -s, s)]
    return _mesh(v, _oriented(v, [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)]))


def octahedron(a=0.5, b=0.35, c=0.6):
    v = [(a, 0, 0), (-a, 0, 0), (0, b, 0), (0, -b, 0), (0, 0, c), (0, 0, -c)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    return _mesh(v, _oriented(v, f))


def prism(n=6, r=0.4, h=0.3, offset=(0.2, -0.1, 0.05)):
    """Right prism over a regular n-gon: two n-vertex faces and n quads. Built off-centre: initialize() moves the centroid to the origin."""
    ang = np.arange(n) * 2 * np.pi / n
    top = [(r * np.cos(a) + offset[0], h + offset[1], r * np.sin(a) + offset[2]) for a in ang]
    bot = [(r * np.cos(a) + offset[0], -h + offset[1], r * np.sin(a) + offset[2]) for a in ang]
    v = top + bot
    f = [tuple(range(n)), tuple(range(n, 2 * n))] + [(i, (i + 1) % n, n + (i + 1) % n, n + i) for i in range(n)]
    return _mesh(v, _oriented(v, f))


def wedge():
    """A ramp: two triangles, three quads, one of them oblique."""
    v = [(-0.5, -0.25, -0.3), (0.5, -0.25, -0.3), (0.5, -0.25, 0.3), (-0.5, -0.25, 0.3), (-0.5, 0.25, -0.3), (-0.5, 0.25, 0.3)]
    return _mesh(v, _oriented(v, [(0, 1, 2, 3), (0, 4, 1), (3, 2, 5), (0, 3, 5, 4), (1, 4, 5, 2)]))


def random_hull(seed, n=14, scale=(0.5, 0.4, 0.45)):
    """Convex hull of random points: triangles only, no two coplanar."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)); pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    pts *= rng.uniform(0.7, 1.0, size=(n, 1)) * np.asarray(scale)
    hull = ConvexHull(pts)
    used = np.unique(hull.simplices)
    remap = {int(o): i for i, o in enumerate(used)}
    v = pts[used]
    return _mesh(v, _oriented(v, [tuple(remap[int(i)] for i in s) for s in hull.simplices]))


def library():
    return [box_mesh((0.5, 0.5, 0.5)), box_mesh((0.3, 0.15, 0.45)), tetrahedron(), octahedron(), prism(6), prism(5, 0.3, 0.45, (0, 0, 0)), wedge(),
            random_hull(1), random_hull(2, 20, (0.3, 0.5, 0.4)), prism(12, 0.45, 0.12)]


def radii(mesh):
    """(inner, outer) radius about the volume centroid, in float64 (placement of test pairs only)."""
    v = mesh["vertices"].astype(np.float64)
    idx, faces = mesh["indices"], mesh["faces"]
    vol, cen = 0.0, np.zeros(3)
    for first, count in faces:
        p = v[idx[first:first + count]]
        for j in range(1, count - 1):
            t = np.dot(p[0], np.cross(p[j], p[j + 1]))
            vol += t; cen += t * (p[0] + p[j] + p[j + 1]) / 4
    cen /= vol
    inner = np.inf
    for first, count in faces:
        p = v[idx[first:first + count]]
        n = np.cross(p[1] - p[0], p[2] - p[1]); n /= np.linalg.norm(n)
        inner = min(inner, np.dot(n, p[0] - cen))
    return float(inner), float(np.linalg.norm(v - cen, axis=1).max())


_registered = None


def registered():
    """The library registered once per process with the oracle (and the reference driver where built) under ids 0..len-1;
    returns (meshes, radii). A device context given the same list in the same order uses the same ids."""
    global _registered
    if _registered is None:
        from oracle import binding as ob
        lib = library()
        # (through the binding's content cache, so that scenes carrying these meshes map to the same ids)
        assert ob.scene_mesh_ids({"meshes": lib}, real=False) == list(range(len(lib))), "register the library before any other mesh"
        if ob.ref() is not None:
            assert ob.scene_mesh_ids({"meshes": lib}, real=True) == list(range(len(lib)))
        _registered = (lib, [radii(m) for m in lib])
    return _registered
