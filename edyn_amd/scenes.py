"""Synthetic benchmark scenes (BASELINE.json configs).

Every generator returns a dict of packed numpy arrays in the layout include/edynhip.h's
edynhip_bodies expects (body 0 is the static ground plane where there is one), plus an optional
"joints" list. Jitter comes from SplitMix64 with seed 0x9E3779B97F4A7C15 so every run (and any
checker fed the same arrays) sees identical inputs.
"""
import math
import numpy as np

KIND_DYNAMIC, KIND_KINEMATIC, KIND_STATIC = 0, 1, 2
SHAPE_NONE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_PLANE, SHAPE_CAPSULE, SHAPE_CYLINDER, SHAPE_POLYHEDRON = 0, 1, 2, 3, 4, 5, 6
JOINT_POINT, JOINT_HINGE, JOINT_DISTANCE, JOINT_SOFT_DISTANCE, JOINT_CONE, JOINT_CVJOINT, JOINT_GRAVITY, JOINT_GENERIC = 0, 1, 2, 3, 4, 5, 6, 7
JOINT_NULL = 8   # null_constraint: no rows, keeps two bodies in one island
ALL = np.uint64(2**64 - 1)
SEED = 0x9E3779B97F4A7C15


def splitmix64_uniform(count, seed=SEED, stream=0):
    """count uniform floats in [0,1) from SplitMix64 (counter mode: state_i = seed + (i+1)*gamma)."""
    gamma = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64) + np.uint64(stream) * np.uint64(0x100000000)
        z = np.uint64(seed) + idx * gamma
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def _empty(n):
    s = {
        "kind": np.zeros(n, np.int32),
        "pos": np.zeros((n, 3), np.float32),
        "orn": np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)),
        "linvel": np.zeros((n, 3), np.float32),
        "angvel": np.zeros((n, 3), np.float32),
        "mass": np.ones(n, np.float32),
        "shape_type": np.zeros(n, np.int32),
        "shape_param": np.zeros((n, 4), np.float32),
        "friction": np.full(n, 0.5, np.float32),
        "restitution": np.zeros(n, np.float32),
        "group": np.full(n, ALL, np.uint64),
        "mask": np.full(n, ALL, np.uint64),
        "inertia": np.zeros((n, 9), np.float32),
        "has_inertia": np.zeros(n, np.uint8),
        "joints": [],
    }
    return s


def _add_plane(s, i=0):
    s["kind"][i] = KIND_STATIC
    s["shape_type"][i] = SHAPE_PLANE
    s["shape_param"][i] = (0, 1, 0, 0)   # plane_shape{normal (0,1,0), constant 0}


def _yaw_quat(yaw):
    h = (yaw * np.float32(0.5)).astype(np.float32)
    q = np.zeros((len(yaw), 4), np.float32)
    q[:, 1] = np.sin(h)
    q[:, 3] = np.cos(h)
    return q


def _jitter(s, first, count, stream=0):
    u = splitmix64_uniform(3 * count, stream=stream).reshape(count, 3)
    s["pos"][first:first + count, 0] += (u[:, 0] * 2 - 1) * np.float32(0.005)
    s["pos"][first:first + count, 2] += (u[:, 1] * 2 - 1) * np.float32(0.005)
    s["orn"][first:first + count] = _yaw_quat((u[:, 2] * 2 - 1) * np.float32(0.02))


def _lattice(nx, ny, nz, origin=(0.0, 0.0, 0.0), pitch_h=1.02, pitch_v=1.005, y0=0.505, brick=True):
    k, i, j = np.meshgrid(np.arange(ny), np.arange(nx), np.arange(nz), indexing="ij")
    k = k.reshape(-1); i = i.reshape(-1); j = j.reshape(-1)
    off = 0.5 * (k & 1) if brick else 0.0
    x = origin[0] + (i - (nx - 1) / 2.0) * pitch_h + off
    z = origin[2] + (j - (nz - 1) / 2.0) * pitch_h + off
    y = origin[1] + y0 + k * pitch_v
    return np.stack([x, y, z], 1).astype(np.float32), (i + j + k)


def box_pile(nx, ny, nz, mixed=False):
    """Brick-offset lattice of unit boxes on a static plane (configs C2 / C3 / headline).
    mixed=True alternates boxes and radius-0.5 spheres by (i+j+k) parity (C3)."""
    n = nx * ny * nz + 1
    s = _empty(n)
    _add_plane(s)
    pos, parity = _lattice(nx, ny, nz)
    s["pos"][1:] = pos
    s["shape_type"][1:] = SHAPE_BOX
    s["shape_param"][1:, :3] = 0.5
    if mixed:
        sph = (parity & 1) == 1
        st = s["shape_type"][1:]
        st[sph] = SHAPE_SPHERE
        sp = s["shape_param"][1:]
        sp[sph] = (0.5, 0, 0, 0)
    _jitter(s, 1, n - 1)
    return s


def _convex_mesh(verts, face_lists):
    """(vertices, indices, faces) as edyn::convex_mesh takes them; every face is wound counter-clockwise seen from outside."""
    v = np.asarray(verts, np.float64)
    c = v.mean(axis=0)
    indices, faces = [], []
    for f in face_lists:
        p = v[list(f)]
        f = list(f) if np.dot(np.cross(p[1] - p[0], p[2] - p[1]), p[0] - c) > 0 else list(reversed(f))
        faces.append((len(indices), len(f))); indices += f
    return dict(vertices=np.asarray(verts, np.float32), indices=np.asarray(indices, np.uint32), faces=np.asarray(faces, np.uint32))


def convex_library():
    """Six convex meshes of roughly unit size for the polyhedron scenes: the unit cube (the reference's make_box_mesh,
    shape_util.cpp:12-38), a tetrahedron, an octahedron, a hexagonal and a pentagonal prism, a wedge."""
    def prism(n, r, h):
        ang = np.arange(n) * 2 * np.pi / n
        top = [(r * np.cos(a), h, r * np.sin(a)) for a in ang]; bot = [(r * np.cos(a), -h, r * np.sin(a)) for a in ang]
        return _convex_mesh(top + bot, [tuple(range(n)), tuple(range(n, 2 * n))] + [(i, (i + 1) % n, n + (i + 1) % n, n + i) for i in range(n)])
    h = 0.5
    cube = dict(vertices=np.float32([(-h, -h, -h), (h, -h, -h), (h, -h, h), (-h, -h, h), (-h, h, -h), (h, h, -h), (h, h, h), (-h, h, h)]),
                indices=np.uint32([0, 1, 2, 3, 7, 6, 5, 4, 4, 5, 1, 0, 6, 7, 3, 2, 7, 4, 0, 3, 5, 6, 2, 1]),
                faces=np.uint32([(4 * f, 4) for f in range(6)]))
    tet = _convex_mesh([(h, h, h), (h, -h, -h), (-h, h, -h), (-h, -h, h)], [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)])
    octa = _convex_mesh([(0.6, 0, 0), (-0.6, 0, 0), (0, 0.5, 0), (0, -0.5, 0), (0, 0, 0.6), (0, 0, -0.6)],
                        [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)])
    wedge = _convex_mesh([(-0.5, -0.3, -0.4), (0.5, -0.3, -0.4), (0.5, -0.3, 0.4), (-0.5, -0.3, 0.4), (-0.5, 0.3, -0.4), (-0.5, 0.3, 0.4)],
                         [(0, 1, 2, 3), (0, 4, 1), (3, 2, 5), (0, 3, 5, 4), (1, 4, 5, 2)])
    return [cube, tet, octa, prism(6, 0.5, 0.4), prism(5, 0.45, 0.5), wedge]


def polyhedron_heap(nx, ny, nz, pitch=1.3):
    """A lattice of convex polyhedra (convex_library, cycled through) with pseudo-random orientations, falling into a heap on a
    static plane: every polyhedron pair routine at work, plus polyhedron-plane."""
    n = nx * ny * nz + 1
    s = _empty(n)
    _add_plane(s)
    pos, _ = _lattice(nx, ny, nz, pitch_h=pitch, pitch_v=pitch, y0=0.9, brick=False)
    s["pos"][1:] = pos
    s["shape_type"][1:] = SHAPE_POLYHEDRON
    lib = convex_library()
    s["shape_param"][1:, 0] = np.arange(n - 1) % len(lib)
    u = splitmix64_uniform(4 * (n - 1), stream=7).reshape(n - 1, 4).astype(np.float64) * 2 - 1
    q = u / np.linalg.norm(u, axis=1, keepdims=True)
    s["orn"][1:] = q.astype(np.float32)
    s["meshes"] = lib
    return s


def pyramid(n):
    """Stable test scene: square pyramid, layer k has (n-k)^2 unit boxes, each resting on four below."""
    pts = []
    for k in range(n):
        m = n - k
        for i in range(m):
            for j in range(m):
                pts.append(((i - (m - 1) / 2.0) * 1.02, 0.505 + 1.005 * k, (j - (m - 1) / 2.0) * 1.02))
    cnt = len(pts)
    s = _empty(cnt + 1)
    _add_plane(s)
    s["pos"][1:] = np.array(pts, np.float32)
    s["shape_type"][1:] = SHAPE_BOX
    s["shape_param"][1:, :3] = 0.5
    _jitter(s, 1, cnt)
    return s


def c1_columns():
    """C1: 10x10x10 straight columns, pitch 1.05, lowest centre y=0.55 -> 100 independent islands."""
    n = 1001
    s = _empty(n)
    _add_plane(s)
    pos, _ = _lattice(10, 10, 10, pitch_h=1.05, pitch_v=1.05, y0=0.55, brick=False)
    s["pos"][1:] = pos
    s["shape_type"][1:] = SHAPE_BOX
    s["shape_param"][1:, :3] = 0.5
    _jitter(s, 1, n - 1)
    return s


def c2_pile():
    return box_pile(20, 20, 20)


def c3_mixed():
    return box_pile(32, 32, 32, mixed=True)


def headline_pile():
    """32k-box pile (32x32x32), the configuration BASELINE.json's metric is quoted on."""
    return box_pile(32, 32, 32)


def mini_piles(sites_x, sites_z, site_pitch=8.0, first_site=0, num_sites=None):
    """C4 building block: independent 4x4x4 brick-offset mini-piles on a grid of sites `site_pitch` apart.
    Returns the sites [first_site, first_site+num_sites) in row-major site order (for sharding across ranks)."""
    total = sites_x * sites_z
    if num_sites is None:
        num_sites = total - first_site
    n = 64 * num_sites + 1
    s = _empty(n)
    _add_plane(s)
    base, _ = _lattice(4, 4, 4)
    for t in range(num_sites):
        site = first_site + t
        sx, sz = site % sites_x, site // sites_x
        o = np.array([(sx - (sites_x - 1) / 2.0) * site_pitch, 0, (sz - (sites_z - 1) / 2.0) * site_pitch], np.float32)
        s["pos"][1 + 64 * t:1 + 64 * (t + 1)] = base + o
    s["shape_type"][1:] = SHAPE_BOX
    s["shape_param"][1:, :3] = 0.5
    # jitter is keyed by the GLOBAL body slot so a shard sees the same values as the full scene
    u = splitmix64_uniform(3 * 64 * total).reshape(64 * total, 3)[64 * first_site:64 * (first_site + num_sites)]
    s["pos"][1:, 0] += (u[:, 0] * 2 - 1) * np.float32(0.005)
    s["pos"][1:, 2] += (u[:, 1] * 2 - 1) * np.float32(0.005)
    s["orn"][1:] = _yaw_quat((u[:, 2] * 2 - 1) * np.float32(0.02))
    return s


def c4_islands(first_site=0, num_sites=None):
    """C4: 262 144 bodies = 4096 mini-piles on a 64x64 grid of sites."""
    return mini_piles(64, 64, first_site=first_site, num_sites=num_sites)


def c5_chains(num_chains=1024, links=16):
    """C5: amorphous bodies with inertia diag(0.01) in chains; joint k alternates hinge (axis z) / point;
    link 0 hangs from a static anchor by a point constraint."""
    nb = num_chains * (links + 1)
    s = _empty(nb)
    u = splitmix64_uniform(num_chains)
    side = int(math.ceil(math.sqrt(num_chains)))
    joints = []
    I = np.diag([0.01, 0.01, 0.01]).astype(np.float32).reshape(9)
    for c in range(num_chains):
        base = c * (links + 1)
        ax, az = (c % side) * 2.0, (c // side) * 2.0
        yaw = (float(u[c]) * 2 - 1) * 0.02
        s["kind"][base] = KIND_STATIC
        s["pos"][base] = (ax, 10.0, az)
        for k in range(links):
            b = base + 1 + k
            # chain laid out sideways (+x) from the anchor so that it swings down
            s["pos"][b] = (ax + 0.5 * (k + 0.5) * math.cos(yaw), 10.0, az + 0.5 * (k + 0.5) * math.sin(yaw))
            s["inertia"][b] = I
            s["has_inertia"][b] = 1
            prev = base + k
            pa = (0.0, 0.0, 0.0) if k == 0 else (0.25, 0.0, 0.0)
            pb = (-0.25, 0.0, 0.0)
            jt = JOINT_POINT if (k == 0 or k % 2 == 0) else JOINT_HINGE
            joints.append((jt, prev, b, pa, pb, (0.0, 0.0, 1.0), (0.0, 0.0, 1.0)))
    s["joints"] = joints
    return s


def _quat_mul(a, b):
    ax, ay, az, aw = (a[..., k] for k in range(4)); bx, by, bz, bw = (b[..., k] for k in range(4))
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], axis=-1).astype(np.float32)


def _quat_rotate(q, v):
    u = q[..., :3]; w = q[..., 3:4]
    t = np.float32(2) * np.cross(u, v)
    return (v + w * t + np.cross(u, t)).astype(np.float32)


def load_figure(path):
    """An articulated figure template (tests/golden/ragdoll_*.npz: the reference's make_ragdoll, exported from the real engine)."""
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def figures(template, nx, nz, pitch=2.2, hip_height=1.25, tumble=0.5, floor=True, ny=1, pitch_v=2.0):
    """nx x nz (x ny layers) copies of an articulated figure (load_figure) standing over a floor plane, each turned about the vertical by a
    seeded angle and given a small seeded spin so that they fall differently. Beside the body arrays and "joints" the scene carries
    what the constraints and the figure need after upload - apply_figure_settings(world, scene):
    "hinge_params" [(joint, 10 floats)], "joint_defs" [(joint, frameA, frameB, 16 floats)], "exclusions" [k, 2]."""
    nb, nj = len(template["kind"]), len(template["joint_type"])
    count = nx * nz * ny
    first = 1 if floor else 0
    s = _empty(first + count * nb)
    if floor:
        _add_plane(s)
    u = splitmix64_uniform(4 * count, stream=7).reshape(count, 4)
    joints, hinge_params, joint_defs, excl = [], [], [], []
    shapeless = template["shape_type"] == SHAPE_NONE
    for c in range(count):
        base = first + c * nb
        sl = slice(base, base + nb)
        yaw = _yaw_quat(np.full(nb, (u[c, 0] * 2 - 1) * np.float32(math.pi), np.float32))
        origin = np.array([(c % nx) * pitch, hip_height + (c // (nx * nz)) * pitch_v, ((c // nx) % nz) * pitch], np.float32)
        s["pos"][sl] = _quat_rotate(yaw, template["pos"]) + origin
        s["orn"][sl] = _quat_mul(yaw, template["orn"])
        s["angvel"][sl] = ((u[c, 1:4] * 2 - 1) * np.float32(tumble)).astype(np.float32)
        for k in ("kind", "mass", "shape_type", "shape_param", "friction", "restitution", "group", "mask"):
            s[k][sl] = template[k]
        s["inertia"][sl][shapeless] = template["inertia"][shapeless]
        s["has_inertia"][sl] = shapeless
        for j in range(nj):
            jt = int(template["joint_type"][j])
            a, b = (int(x) + base for x in template["joint_body"][j])
            joints.append((jt, a, b, tuple(template["pivotA"][j]), tuple(template["pivotB"][j]), tuple(template["axisA"][j]), tuple(template["axisB"][j])))
            if jt == JOINT_HINGE:
                hinge_params.append((len(joints) - 1, template["params10"][j]))
            elif jt in (JOINT_CONE, JOINT_CVJOINT):
                joint_defs.append((len(joints) - 1, template["frameA"][j].reshape(3, 3), template["frameB"][j].reshape(3, 3), template["params16"][j]))
        excl.append(template["exclusions"].astype(np.uint32) + np.uint32(base))
    s["joints"] = joints
    s["hinge_params"] = hinge_params
    s["joint_defs"] = joint_defs
    s["exclusions"] = np.concatenate(excl) if excl else np.zeros((0, 2), np.uint32)
    return s


def merge(first, second):
    """The bodies of `second` (e.g. figures(..., floor=False)) appended to `first`: joint, definition and exclusion indices shifted."""
    n = len(first["kind"])
    out = {}
    for k, v in first.items():
        if isinstance(v, np.ndarray) and len(v) == n and k in second:
            out[k] = np.concatenate([v, second[k]], axis=0)
    shift = lambda t: (t[0], t[1] + n, t[2] + n) + tuple(t[3:])
    nj = len(first.get("joints") or [])
    out["joints"] = list(first.get("joints") or []) + [shift(t) for t in second.get("joints", [])]
    out["hinge_params"] = list(first.get("hinge_params", [])) + [(j + nj, p) for j, p in second.get("hinge_params", [])]
    out["joint_defs"] = list(first.get("joint_defs", [])) + [(j + nj, fa, fb, p) for j, fa, fb, p in second.get("joint_defs", [])]
    ex = [first["exclusions"]] if "exclusions" in first else []
    out["exclusions"] = np.concatenate(ex + [second.get("exclusions", np.zeros((0, 2), np.uint32)) + np.uint32(n)])
    return out


def apply_figure_settings(world, scene):
    """What a figure scene needs after its bodies and joints are uploaded (any of the three world classes)."""
    for j, p in scene.get("hinge_params", []):
        world.set_joint_params(j, p)
    for j, fa, fb, p in scene.get("joint_defs", []):
        world.set_joint_definition(j, fa, fb, p)
    for a, b in scene.get("exclusions", []):
        world.exclude_collision(int(a), int(b))


def scene_from_defs(defs, joints):
    """Pack a list of edyn_amd.rigidbody_def (+ joint tuples) into the array layout."""
    n = len(defs)
    s = _empty(n)
    grav = np.zeros((n, 3), np.float32)
    any_grav = False
    for i, d in enumerate(defs):
        s["kind"][i] = d.kind
        s["pos"][i] = d.position
        s["orn"][i] = d.orientation
        s["linvel"][i] = d.linvel
        s["angvel"][i] = d.angvel
        s["mass"][i] = d.mass
        s["shape_type"][i] = d.shape_type
        s["shape_param"][i] = d.shape_param
        s["friction"][i] = d.friction
        s["restitution"][i] = d.restitution
        s["group"][i] = np.uint64(d.collision_group)
        s["mask"][i] = np.uint64(d.collision_mask)
        if d.inertia is not None:
            s["inertia"][i] = np.asarray(d.inertia, np.float32).reshape(9)
            s["has_inertia"][i] = 1
        if d.gravity is not None:
            any_grav = True
            grav[i] = d.gravity
    if any(getattr(d, "sleeping_disabled", False) for d in defs):
        s["sleeping_disabled"] = np.array([1 if getattr(d, "sleeping_disabled", False) else 0 for d in defs], np.uint8)
    if any_grav:
        raise NotImplementedError("per-body gravity via rigidbody_def: pass a full 'gravity' array with set_scene")
    s["joints"] = list(joints)
    return s


def subset(scene, idx):
    """Bodies `idx` (array of body ids, ascending) as a new scene; joints are dropped."""
    out = {}
    for k, v in scene.items():
        if k in ("joints", "hinge_params", "joint_defs", "exclusions"):
            out[k] = []
        elif k == "meshes":   # convex meshes are referred to by position (shape_param[0] of a polyhedron): every subset keeps the list
            out[k] = v
        else:
            out[k] = np.ascontiguousarray(v[idx])
    return out
