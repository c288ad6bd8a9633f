"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding for oracle/liboracle.so (the CPU restatement of the reference step loop) and, when
built, oracle/_ref/libedynref.so (the REAL reference engine: its own translation units compiled where
they lie against oracle/entt_min, driven by ref_world.cpp / ref_xcheck.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (edyn_amd/) never does.
"""
import ctypes as C
PAIR_FILTER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)   # int filter(void *user, uint32_t body, uint32_t other)
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POINT_DTYPE = np.dtype([
    ("pivotA", np.float32, 3), ("pivotB", np.float32, 3), ("normal", np.float32, 3),
    ("local_normal", np.float32, 3), ("distance", np.float32), ("friction", np.float32),
    ("restitution", np.float32), ("attachment", np.int32), ("lifetime", np.uint32),
    ("normal_impulse", np.float32), ("friction_impulse", np.float32, 2)])
MANIFOLD_DTYPE = np.dtype([
    ("body", np.uint32, 2), ("num_points", np.uint32), ("colour", np.uint32), ("pt", POINT_DTYPE, 4)])

SHAPE_NONE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_PLANE, SHAPE_CAPSULE, SHAPE_CYLINDER, SHAPE_POLYHEDRON = 0, 1, 2, 3, 4, 5, 6
KIND_DYNAMIC, KIND_KINEMATIC, KIND_STATIC = 0, 1, 2
JOINT_POINT, JOINT_HINGE, JOINT_DISTANCE, JOINT_SOFT_DISTANCE, JOINT_CONE, JOINT_CVJOINT, JOINT_GRAVITY, JOINT_GENERIC = 0, 1, 2, 3, 4, 5, 6, 7
ORDER_SEQUENTIAL, ORDER_COLOURED, ORDER_EXTERNAL = 0, 1, 2


def build(ref=True):
    """Compile liboracle.so (and _ref/libedynref.so when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/src/edyn"):
        subprocess.check_call(["make", "-s", "-j%d" % max(2, os.cpu_count() or 2), "-C", _HERE, "ref"])


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.orc_world_create.restype = C.c_void_p
        L.orc_world_create.argtypes = [C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
        L.orc_world_destroy.argtypes = [C.c_void_p]
        L.orc_add_body.restype = C.c_uint32
        L.orc_add_body.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_float)] * 4 + [
            C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int,
            C.c_uint64, C.c_uint64, C.POINTER(C.c_float)]
        L.orc_add_joint.restype = C.c_uint32
        L.orc_add_joint.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.POINTER(C.c_float)] * 4
        L.orc_set_sleeping.argtypes = [C.c_void_p, C.c_int]; L.orc_set_sleeping.restype = None
        L.orc_set_sleeping_disabled.argtypes = [C.c_void_p, C.c_uint32, C.c_int]; L.orc_set_sleeping_disabled.restype = None
        L.orc_wake_all.argtypes = [C.c_void_p]; L.orc_wake_all.restype = None
        L.orc_get_asleep.argtypes = [C.c_void_p, C.c_void_p]; L.orc_get_asleep.restype = None
        L.orc_set_asleep.argtypes = [C.c_void_p, C.c_void_p]; L.orc_set_asleep.restype = None
        L.orc_set_joint_warm_start.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]; L.orc_set_joint_warm_start.restype = None
        L.orc_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_run_stage.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_bodies.restype = C.c_uint32
        L.orc_num_bodies.argtypes = [C.c_void_p]
        L.orc_get_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
        L.orc_set_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
        L.orc_get_derived.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.orc_num_manifolds.restype = C.c_uint32
        L.orc_num_manifolds.argtypes = [C.c_void_p]
        L.orc_get_manifolds.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_manifolds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_get_joint_impulses.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_time_steps.restype = C.c_double
        L.orc_time_steps.argtypes = [C.c_void_p, C.c_int]
        L.orc_sizeof_manifold_rec.restype = C.c_uint32
        assert L.orc_sizeof_manifold_rec() == MANIFOLD_DTYPE.itemsize
        _lib = L
    return _lib


EVENT_DTYPE = np.dtype([("type", np.uint32), ("step", np.uint32), ("body", np.uint32, 2), ("point_id", np.uint64)])


def ref():
    """The real reference leaf functions (None if oracle/_ref was not built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libedynref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


class _Leaf:
    """Leaf functions exported under a prefix ('orc_' restatement, 'ref_' real reference)."""

    def __init__(self, L, prefix):
        self.L, self.p = L, prefix

    def _f(self, name):
        return getattr(self.L, self.p + name)

    def intersect_line_aabb(self, p0, p1, bmin, bmax):
        s = np.zeros(2, np.float32)
        f = self._f("intersect_line_aabb"); f.restype = C.c_int
        n = f(_fp(_f32(p0, 2)), _fp(_f32(p1, 2)), _fp(_f32(bmin, 2)), _fp(_f32(bmax, 2)), _fp(s))
        return n, s

    def plane_space(self, n):
        p = np.zeros(3, np.float32); q = np.zeros(3, np.float32)
        self._f("plane_space")(_fp(_f32(n, 3)), _fp(p), _fp(q))
        return p, q

    def integrate(self, q, w, dt):
        out = np.zeros(4, np.float32)
        f = self._f("integrate"); f.argtypes = [C.POINTER(C.c_float)] * 2 + [C.c_float, C.POINTER(C.c_float)]
        f(_fp(_f32(q, 4)), _fp(_f32(w, 3)), dt, _fp(out))
        return out

    def rotate(self, q, v):
        out = np.zeros(3, np.float32)
        self._f("rotate")(_fp(_f32(q, 4)), _fp(_f32(v, 3)), _fp(out))
        return out

    def insertion_point_index(self, pts, num_points, new_point):
        n = C.c_int(num_points)
        f = self._f("insertion_point_index"); f.restype = C.c_int
        r = f(_fp(_f32(pts, 12)), C.byref(n), _fp(_f32(new_point, 3)))
        return r & 0xFF, r >> 8, n.value

    def closest_segment_segment(self, p1, q1, p2, q2):
        st = np.zeros(4, np.float32); c = np.zeros(12, np.float32); num = C.c_int(0)
        f = self._f("closest_segment_segment"); f.restype = C.c_float
        d = f(_fp(_f32(p1, 3)), _fp(_f32(q1, 3)), _fp(_f32(p2, 3)), _fp(_f32(q2, 3)), _fp(st), _fp(c), C.byref(num))
        return np.float32(d), st, c, num.value

    def box_support_feature(self, h, direction, threshold):
        feat = C.c_int(); idx = C.c_int(); proj = C.c_float()
        f = self._f("box_support_feature")
        f.argtypes = [C.POINTER(C.c_float)] * 2 + [C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        f(_fp(_f32(h, 3)), _fp(_f32(direction, 3)), threshold, C.byref(feat), C.byref(idx), C.byref(proj))
        return feat.value, idx.value, np.float32(proj.value)

    def box_support_projection(self, h, pos, orn, direction):
        f = self._f("box_support_projection"); f.restype = C.c_float
        return np.float32(f(_fp(_f32(h, 3)), _fp(_f32(pos, 3)), _fp(_f32(orn, 4)), _fp(_f32(direction, 3))))

    def row_prepare_solve(self, rowdata, vel, delta):
        d = _f32(delta, 12).copy(); out = np.zeros(4, np.float32)
        self._f("row_prepare_solve")(_fp(_f32(rowdata, 38)), _fp(_f32(vel, 12)), _fp(d), _fp(out))
        return out, d


def set_libm_trig(on):
    """integrate(): C library sinf/cosf (what the reference calls) instead of the correctly rounded fp32 value."""
    f = lib().orc_set_libm_trig
    f.argtypes = [C.c_int]; f.restype = None
    f(int(bool(on)))


def set_fused_rows(on):
    """The coloured order's contact rows: its own fused arithmetic (default, what the device computes) or the reference's."""
    f = lib().orc_set_fused_rows
    f.argtypes = [C.c_int]; f.restype = None
    f(int(bool(on)))


ARITH_REFERENCE, ARITH_FUSED_VELOCITY, ARITH_BLOCK_POSITION, ARITH_TWO_PHASE = 0, 1, 2, 4   # TWO_PHASE: checker-only (island-wide normals, then friction: island_solver.cpp:94-111)


def set_arithmetic(mode):
    """Contact arithmetic of the coloured order: ARITH_REFERENCE (the reference's operations, default), | ARITH_FUSED_VELOCITY
    (fma velocity rows), | ARITH_BLOCK_POSITION (per-manifold block position correction). Process-wide."""
    f = lib().orc_set_arithmetic
    f.argtypes = [C.c_int]; f.restype = None
    f(int(mode))


def get_arithmetic():
    f = lib().orc_get_arithmetic
    f.argtypes = []; f.restype = C.c_int
    return int(f())


def leaf():
    return _Leaf(lib(), "orc_")


def ref_leaf():
    r = ref()
    return _Leaf(r, "ref_") if r is not None else None


def collide(typeA, paramA, posA, ornA, typeB, paramB, posB, ornB, threshold):
    out = np.zeros(44, np.float32)
    f = lib().orc_collide
    f.restype = C.c_int
    f.argtypes = [C.c_int] + [C.POINTER(C.c_float)] * 3 + [C.c_int] + [C.POINTER(C.c_float)] * 3 + [C.c_float, C.POINTER(C.c_float)]
    n = f(typeA, _fp(_f32(paramA, 4)), _fp(_f32(posA, 3)), _fp(_f32(ornA, 4)), typeB, _fp(_f32(paramB, 4)),
          _fp(_f32(posB, 3)), _fp(_f32(ornB, 4)), threshold, _fp(out))
    return out.reshape(4, 11)[:n].copy()


MESH_FIELDS = ("vertices", "normals", "relevant_normals", "edge_vertices", "edge_normals", "edges", "edge_faces", "relevant_faces",
               "relevant_edges", "neighbors_start", "neighbor_indices")


def create_mesh(mesh, real=False):
    """Register a convex mesh (tests/meshes.py dict) with the oracle (or the reference driver); returns its id = shape_param[0]."""
    L = ref() if real else lib()
    f = getattr(L, "ref_create_mesh" if real else "orc_create_mesh")
    f.restype = C.c_int; f.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    v = np.ascontiguousarray(mesh["vertices"], np.float32); i = np.ascontiguousarray(mesh["indices"], np.uint32)
    fc = np.ascontiguousarray(mesh["faces"], np.uint32)
    return f(len(v), v.ctypes.data, len(i), i.ctypes.data, len(fc), fc.ctypes.data)


_mesh_ids = ({}, {})   # content -> registry id, for the oracle and for the reference driver


def scene_mesh_ids(scene, real=False):
    """The registry ids of a scene's convex meshes (scene["meshes"], referred to by position in shape_param[0] of its polyhedra):
    each distinct mesh is registered once per process."""
    ids = []
    for m in scene.get("meshes") or []:
        key = (np.ascontiguousarray(m["vertices"], np.float32).tobytes(), np.ascontiguousarray(m["indices"], np.uint32).tobytes(),
               np.ascontiguousarray(m["faces"], np.uint32).tobytes())
        cache = _mesh_ids[1 if real else 0]
        if key not in cache:
            cache[key] = create_mesh(m, real)
        ids.append(cache[key])
    return ids


def _shape_param(scene, i, ids):
    sp = scene["shape_param"][i]
    if ids and int(scene["shape_type"][i]) == SHAPE_POLYHEDRON:
        sp = np.array(sp, np.float32); sp[0] = ids[int(sp[0])]
    return sp


def mesh_get(mesh_id, field, real=False):
    L = ref() if real else lib()
    f = getattr(L, "ref_mesh_get" if real else "orc_mesh_get")
    f.restype = C.c_uint32; f.argtypes = [C.c_int, C.c_int, C.c_void_p]
    what = MESH_FIELDS.index(field)
    n = f(mesh_id, what, None)
    out = np.zeros((n, 3), np.float32) if what < 5 else np.zeros(n, np.uint32)
    f(mesh_id, what, out.ctypes.data)
    return out


def mesh_inertia(mesh_id, mass, real=False):
    L = ref() if real else lib()
    f = getattr(L, "ref_mesh_inertia" if real else "orc_mesh_inertia")
    f.restype = None; f.argtypes = [C.c_int, C.c_float, C.c_void_p]
    out = np.zeros((3, 3), np.float32)
    f(mesh_id, mass, out.ctypes.data)
    return out


def should_collide(groupA, maskA, groupB, maskB):
    f = lib().orc_should_collide
    f.argtypes = [C.c_uint64] * 4
    f.restype = C.c_int
    return bool(f(groupA, maskA, groupB, maskB))


class World:
    """Mirror of edyn_amd.World's interface over the CPU oracle."""

    def __init__(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3, gravity=(0, -9.8, 0), order=ORDER_SEQUENTIAL):
        g = _f32(gravity, 3)
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_world_create(dt, vel_iters, pos_iters, _fp(g), order))
        self.n_joints = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_world_destroy(self.h)
            self.h = None

    def add_body(self, kind=KIND_DYNAMIC, pos=(0, 0, 0), orn=(0, 0, 0, 1), linvel=(0, 0, 0), angvel=(0, 0, 0),
                 mass=1.0, shape_type=SHAPE_NONE, shape_param=(0, 0, 0, 0), inertia=None, friction=0.5,
                 restitution=0.0, has_material=True, group=2**64 - 1, mask=2**64 - 1, gravity=None):
        I = _fp(_f32(inertia, 9)) if inertia is not None else None
        g = _fp(_f32(gravity, 3)) if gravity is not None else None
        return self.L.orc_add_body(self.h, kind, _fp(_f32(pos, 3)), _fp(_f32(orn, 4)), _fp(_f32(linvel, 3)),
                                   _fp(_f32(angvel, 3)), mass, shape_type, _fp(_f32(shape_param, 4)), I,
                                   friction, restitution, int(has_material), group, mask, g)

    def add_bodies(self, scene):
        """scene: dict of arrays as produced by edyn_amd.scenes (kind,pos,orn,linvel,angvel,mass,shape_type,shape_param,...)."""
        n = len(scene["kind"])
        inertia = scene.get("inertia")
        has_inertia = scene.get("has_inertia")
        mesh_ids = scene_mesh_ids(scene, real=False)
        for i in range(n):
            I = inertia[i] if (inertia is not None and has_inertia is not None and has_inertia[i]) else None
            grav = scene["gravity"][i] if scene.get("gravity") is not None else None
            b = self.add_body(int(scene["kind"][i]), scene["pos"][i], scene["orn"][i], scene["linvel"][i],
                              scene["angvel"][i], float(scene["mass"][i]), int(scene["shape_type"][i]),
                              _shape_param(scene, i, mesh_ids), I, float(scene["friction"][i]), float(scene["restitution"][i]),
                              True, int(scene["group"][i]), int(scene["mask"][i]), grav)
            if scene.get("com") is not None and np.any(scene["com"][i] != 0):   # rigidbody_def::center_of_mass: `pos` was the origin
                self.set_center_of_mass(b, scene["com"][i], float(scene["mass"][i]))
        joints = scene.get("joints")
        if joints is not None:
            for j in joints:
                self.add_joint(*j)

    def add_joint(self, jtype, a, b, pivotA, pivotB, axisA=(1, 0, 0), axisB=(1, 0, 0), params=None):
        self.n_joints += 1
        j = self.L.orc_add_joint(self.h, jtype, a, b, _fp(_f32(pivotA, 3)), _fp(_f32(pivotB, 3)),
                                 _fp(_f32(axisA, 3)), _fp(_f32(axisB, 3)))
        if params is not None:
            self.set_joint_params(j, params)
        return j

    def set_center_of_mass(self, body, com, mass):
        """rigidbody_def::center_of_mass, right after add_body (the position given there is the origin)."""
        f = self.L.orc_set_center_of_mass; f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.c_float]; f.restype = None
        f(self.h, body, _fp(_f32(com, 3)), mass)

    def move_center_of_mass(self, body, com):
        """edyn::set_center_of_mass on a running world: position / linear velocity follow the new centre of mass, the inertia stays."""
        f = self.L.orc_move_center_of_mass; f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, body, _fp(_f32(com, 3)))

    def exclude_collision(self, a, b):
        f = self.L.orc_exclude_collision; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = None
        f(self.h, a, b)

    def set_should_collide(self, func):
        """settings.should_collide_func: func(body, other) -> bool replaces should_collide_default (None restores it)."""
        self._filter_cb = PAIR_FILTER(lambda user, a, b: 1 if func(int(a), int(b)) else 0) if func else None
        f = self.L.orc_set_should_collide; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = None
        f(self.h, C.cast(self._filter_cb, C.c_void_p) if func else None, None)

    def default_should_collide(self, a, b):
        f = self.L.orc_default_should_collide; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = C.c_int
        return bool(f(self.h, a, b))

    def remove_collision_exclusion(self, a, b):
        f = self.L.orc_remove_collision_exclusion; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = None
        f(self.h, a, b)

    def set_restitution_iterations(self, iters, individual=3):
        f = self.L.orc_set_restitution_iterations; f.argtypes = [C.c_void_p, C.c_int, C.c_int]; f.restype = None
        f(self.h, iters, individual)

    def set_ext_order(self, contacts, joints=()):
        """ORDER_EXTERNAL: visiting order for the next step (RefWorld.get_solve_order() of the same step)."""
        c = np.ascontiguousarray(contacts, np.uint32).reshape(-1, 3); j = np.ascontiguousarray(joints, np.uint32)
        f = self.L.orc_set_ext_order
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, c.ctypes.data, len(c), j.ctypes.data, len(j))

    def set_ext_restitution_walk(self, manifolds, adjacency):
        """ORDER_EXTERNAL: the orders the real engine's restitution solver walks in (RefWorld.get_restitution_walk() of the same step)."""
        m = np.ascontiguousarray(manifolds, np.uint32).reshape(-1, 2); a = np.ascontiguousarray(adjacency, np.uint32)
        f = self.L.orc_set_ext_restitution_walk
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, m.ctypes.data, len(m), a.ctypes.data, len(a))

    def ext_order_mismatch(self):
        f = self.L.orc_ext_order_mismatch
        f.argtypes = [C.c_void_p]; f.restype = C.c_int
        return bool(f(self.h))

    def set_sleeping(self, enable=True):
        self.L.orc_set_sleeping(self.h, 1 if enable else 0)

    def set_sleeping_disabled(self, body, disabled=True):
        self.L.orc_set_sleeping_disabled(self.h, int(body), 1 if disabled else 0)

    def wake_all(self):
        self.L.orc_wake_all(self.h)

    def set_joint_warm_start(self, impulses24, angles=None):
        imp = np.ascontiguousarray(impulses24, np.float32)
        ang = None if angles is None else np.ascontiguousarray(angles, np.float32)
        self.L.orc_set_joint_warm_start(self.h, _fp(imp), _fp(ang) if ang is not None else None)

    def set_asleep(self, flags):
        f = np.ascontiguousarray(np.asarray(flags).astype(np.uint8))
        self.L.orc_set_asleep(self.h, f.ctypes.data_as(C.c_void_p))

    def get_asleep(self):
        out = np.zeros(self.num_bodies, np.uint8)
        self.L.orc_get_asleep(self.h, out.ctypes.data)
        return out.astype(bool)

    def step(self, n=1):
        self.L.orc_step(self.h, n)

    def run_stage(self, stage):
        self.L.orc_run_stage(self.h, stage)

    def time_steps(self, n):
        return self.L.orc_time_steps(self.h, n)

    @property
    def num_bodies(self):
        return self.L.orc_num_bodies(self.h)

    def get_state(self):
        n = self.num_bodies
        pos = np.zeros((n, 3), np.float32); orn = np.zeros((n, 4), np.float32)
        lv = np.zeros((n, 3), np.float32); av = np.zeros((n, 3), np.float32)
        self.L.orc_get_state(self.h, _fp(pos), _fp(orn), _fp(lv), _fp(av))
        return pos, orn, lv, av

    def set_state(self, pos, orn, lv, av):
        n = self.num_bodies
        self.L.orc_set_state(self.h, _fp(_f32(pos, 3 * n)), _fp(_f32(orn, 4 * n)), _fp(_f32(lv, 3 * n)), _fp(_f32(av, 3 * n)))

    def refresh_derived(self):
        f = self.L.orc_refresh_derived
        f.argtypes = [C.c_void_p]; f.restype = None
        f(self.h)

    def get_derived(self):
        n = self.num_bodies
        aabb = np.zeros((n, 6), np.float32); iw = np.zeros((n, 9), np.float32); isl = np.zeros(n, np.uint32)
        self.L.orc_get_derived(self.h, _fp(aabb), _fp(iw), isl.ctypes.data_as(C.POINTER(C.c_uint32)))
        return aabb, iw, isl

    def get_manifolds(self):
        m = self.L.orc_num_manifolds(self.h)
        out = np.zeros(m, MANIFOLD_DTYPE)
        if m:
            self.L.orc_get_manifolds(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def set_manifolds(self, recs):
        recs = np.ascontiguousarray(recs, dtype=MANIFOLD_DTYPE)
        self.L.orc_set_manifolds(self.h, recs.ctypes.data_as(C.c_void_p), len(recs))

    def set_material_id(self, body, mid):
        f = self.L.orc_set_material_id; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = None
        f(self.h, body, mid)

    def insert_material_mixing(self, id0, id1, restitution=0.0, friction=0.5, spin=0.0, roll=0.0, stiffness=1e18, damping=1e18):
        m = np.array([restitution, friction, spin, roll, stiffness, damping], np.float32)
        f = self.L.orc_insert_material_mixing; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, id0, id1, _fp(m))

    # contact_extras (rolling / spinning friction, soft contacts)
    def set_material_extras(self, body, spin=0.0, roll=0.0, stiffness=1e18, damping=1e18):
        f = self.L.orc_set_material_extras; f.argtypes = [C.c_void_p, C.c_uint32] + [C.c_float] * 4; f.restype = None
        f(self.h, body, spin, roll, stiffness, damping)

    def get_point_extras(self):
        """[num_manifolds, 4, 7]: rolling impulse 0/1, spin impulse, roll mu, spin mu, stiffness, damping (canonical manifold order)."""
        m = self.L.orc_num_manifolds(self.h)
        out = np.zeros((m, 4, 7), np.float32)
        if m:
            g = self.L.orc_get_point_extras; g.argtypes = [C.c_void_p, C.c_void_p]; g.restype = None
            g(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    # contact events (the test counterpart of edynhip_get_contact_events / edynhip_get_point_ids)
    def record_events(self, on=True):
        f = self.L.orc_record_events; f.argtypes = [C.c_void_p, C.c_int]; f.restype = None
        f(self.h, int(on))

    def clear_events(self):
        f = self.L.orc_clear_events; f.argtypes = [C.c_void_p]; f.restype = None
        f(self.h)

    def get_events(self):
        """Structured array (type, step, body[2], point_id) of the events since record_events / clear_events."""
        f = self.L.orc_num_events; f.argtypes = [C.c_void_p]; f.restype = C.c_uint32
        n = f(self.h)
        out = np.zeros(n, EVENT_DTYPE)
        if n:
            g = self.L.orc_get_events; g.argtypes = [C.c_void_p, C.c_void_p]; g.restype = None
            g(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def get_point_ids(self):
        m = self.L.orc_num_manifolds(self.h)
        out = np.zeros((m, 4), np.uint64)
        if m:
            g = self.L.orc_get_point_ids; g.argtypes = [C.c_void_p, C.c_void_p]; g.restype = None
            g(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def get_pairs(self):
        """Canonical (hi, lo) pairs sorted ascending — the representation bit-exact parity is checked on."""
        m = self.get_manifolds()
        b = m["body"].astype(np.uint64)
        hi = np.maximum(b[:, 0], b[:, 1]); lo = np.minimum(b[:, 0], b[:, 1])
        return np.sort((hi << np.uint64(32)) | lo)

    def get_joint_impulses(self):
        """[n, 10]: the 9 applied-impulse slots + the tracked hinge angle (see oracle_capi.cpp)."""
        out = np.zeros((self.n_joints, 10), np.float32)
        if self.n_joints:
            self.L.orc_get_joint_impulses(self.h, _fp(out))
        return out

    def set_joint_params(self, joint, params):
        p = np.zeros(10, np.float32); p[:len(params)] = params
        f = self.L.orc_set_joint_params; f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, joint, _fp(p))

    def set_generic_definition(self, joint, frameA, frameB, dofs):
        """dofs: [6][10] (linear x, y, z, angular x, y, z) - limit_enabled, min, max, restitution, bump length|angle, bump stiffness,
        friction, rest, spring stiffness, damping."""
        p = np.ascontiguousarray(np.asarray(dofs, np.float32).reshape(60))
        f = self.L.orc_set_generic_definition; f.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_float)] * 3; f.restype = None
        f(self.h, joint, _fp(_f32(np.asarray(frameA).ravel(), 9)), _fp(_f32(np.asarray(frameB).ravel(), 9)), _fp(p))

    def get_joint_impulses24(self):
        out = np.zeros((self.n_joints, 24), np.float32)
        if self.n_joints:
            f = self.L.orc_get_joint_impulses24; f.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; f.restype = None
            f(self.h, _fp(out))
        return out

    def set_joint_definition(self, joint, frameA, frameB, params):
        p = np.zeros(16, np.float32); p[:len(params)] = params
        f = self.L.orc_set_joint_definition; f.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_float)] * 3; f.restype = None
        f(self.h, joint, _fp(_f32(np.asarray(frameA).ravel(), 9)), _fp(_f32(np.asarray(frameB).ravel(), 9)), _fp(p))

    def remove_body(self, body):
        f = self.L.orc_remove_body; f.argtypes = [C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, body)

    def remove_joint(self, joint):
        f = self.L.orc_remove_joint; f.argtypes = [C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, joint)

    def set_params(self, dt, vel_iters, pos_iters, gravity=None):
        f = self.L.orc_set_params; f.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, dt, vel_iters, pos_iters, _fp(_f32(gravity, 3)) if gravity is not None else None)

    def step_timed(self, n, first_time, step_dt):
        f = self.L.orc_step_timed; f.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]; f.restype = None
        f(self.h, n, first_time, step_dt)

    def get_stats(self):
        s = np.zeros(7, np.uint32)
        self.L.orc_get_stats(self.h, s.ctypes.data_as(C.POINTER(C.c_uint32)))
        return dict(zip(["num_manifolds", "num_points", "num_rows", "num_islands", "num_colours",
                         "num_joint_colours", "colour_rounds"], [int(x) for x in s]))


def collide_batch(shape_type, shape_param, pos, orn, threshold=0.01):
    st = np.ascontiguousarray(shape_type, np.int32).reshape(-1, 2)
    n = len(st)
    sp = np.ascontiguousarray(shape_param, np.float32).reshape(n, 2, 4)
    ps = np.ascontiguousarray(pos, np.float32).reshape(n, 2, 3)
    qs = np.ascontiguousarray(orn, np.float32).reshape(n, 2, 4)
    out = np.zeros((n, 4, 11), np.float32); cnt = np.zeros(n, np.uint32)
    f = lib().orc_collide_batch
    f.argtypes = [C.c_uint32] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
    f.restype = None
    f(n, st.ctypes.data, sp.ctypes.data, ps.ctypes.data, qs.ctypes.data, threshold, out.ctypes.data, cnt.ctypes.data)
    return out, cnt


def last_batch_flags(n):
    """Per pair of the last collide_batch: bit 0 = a configuration on which the reference's routine has undefined behaviour."""
    out = np.zeros(n, np.uint8)
    f = lib().orc_last_batch_flags; f.argtypes = [C.c_void_p]; f.restype = None
    f(out.ctypes.data)
    return out


def _batch(fn, shape_type, shape_param, pos, orn, threshold):
    st = np.ascontiguousarray(shape_type, np.int32).reshape(-1, 2)
    n = len(st)
    sp = np.ascontiguousarray(shape_param, np.float32).reshape(n, 2, 4)
    ps = np.ascontiguousarray(pos, np.float32).reshape(n, 2, 3)
    qs = np.ascontiguousarray(orn, np.float32).reshape(n, 2, 4)
    out = np.zeros((n, 4, 11), np.float32); cnt = np.zeros(n, np.uint32)
    fn.argtypes = [C.c_uint32] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(n, st.ctypes.data, sp.ctypes.data, ps.ctypes.data, qs.ctypes.data, threshold, out.ctypes.data, cnt.ctypes.data)
    return out, cnt


def ref_collide_batch(shape_type, shape_param, pos, orn, threshold=0.01):
    """The real edyn::collide overloads (collide.hpp:43-330) on a batch of pairs."""
    return _batch(ref().ref_collide_batch, shape_type, shape_param, pos, orn, threshold)


def tree_run(ops, boxes, real=False, max_hits=1 << 22):
    """Scripted dynamic-tree session; see ref_xcheck.cpp ref_tree_run. Returns (hits, moved)."""
    ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 2)
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(len(ops), 6)
    hits = np.zeros(max_hits, np.uint32); moved = np.zeros(len(ops), np.uint8)
    fn = ref().ref_tree_run if real else lib().orc_tree_run
    fn.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    fn.restype = C.c_uint32
    n = fn(len(ops), ops.ctypes.data, boxes.ctypes.data, hits.ctypes.data, max_hits, moved.ctypes.data)
    assert n < max_hits
    return hits[:n].copy(), moved


def friction_solve(normal, fric, delta, warm=True, sweeps=1, real=False):
    """Friction pair vs its normal row; see ref_xcheck.cpp ref_friction_solve. Returns (impulses[2], delta[12])."""
    d = _f32(delta, 12).copy(); out = np.zeros(2, np.float32)
    fn = ref().ref_friction_solve if real else lib().orc_friction_solve
    fn.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_int, C.c_int, C.POINTER(C.c_float)]
    fn.restype = None
    fn(_fp(_f32(normal, 33)), _fp(_f32(fric, 31)), _fp(d), int(warm), int(sweeps), _fp(out))
    return out, d


class RefWorld:
    """The REAL reference engine (oracle/_ref/libedynref.so, ref_world.cpp) behind the same interface as World.

    mode 0 = execution_mode::sequential, 1 = sequential_multithreaded (workers=0 -> hardware_concurrency-1).
    """

    def __init__(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3, gravity=(0, -9.8, 0), mode=0, workers=0):
        L = ref()
        if L is None:
            raise RuntimeError("oracle/_ref/libedynref.so is not built (make -C oracle ref)")
        self.L = L
        L.refw_create.restype = C.c_void_p
        L.refw_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.refw_destroy.argtypes = [C.c_void_p]
        L.refw_add_body.restype = C.c_uint32
        L.refw_add_body.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_float)] * 4 + [
            C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int,
            C.c_uint64, C.c_uint64, C.POINTER(C.c_float), C.c_int]
        L.refw_add_joint.restype = C.c_uint32
        L.refw_add_joint.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.POINTER(C.c_float)] * 4
        L.refw_set_joint_params.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
        L.refw_exclude_collision.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.refw_set_restitution_iterations.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.refw_step.argtypes = [C.c_void_p, C.c_int]
        L.refw_time_steps.argtypes = [C.c_void_p, C.c_int]; L.refw_time_steps.restype = C.c_double
        L.refw_update.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.refw_set_max_steps_per_update.argtypes = [C.c_void_p, C.c_uint]
        L.refw_num_bodies.argtypes = [C.c_void_p]; L.refw_num_bodies.restype = C.c_uint32
        L.refw_get_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
        L.refw_set_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
        L.refw_get_derived.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.refw_num_islands.argtypes = [C.c_void_p]; L.refw_num_islands.restype = C.c_uint32
        L.refw_num_manifolds.argtypes = [C.c_void_p]; L.refw_num_manifolds.restype = C.c_uint32
        L.refw_get_manifolds.argtypes = [C.c_void_p, C.c_void_p]
        L.refw_get_joint_impulses.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.refw_sizeof_manifold_rec.restype = C.c_uint32
        assert L.refw_sizeof_manifold_rec() == MANIFOLD_DTYPE.itemsize
        self.h = C.c_void_p(L.refw_create(mode, workers, dt, vel_iters, pos_iters, _fp(_f32(gravity, 3))))
        self.n_joints = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.refw_destroy(self.h)
            self.h = None

    def add_body(self, kind=KIND_DYNAMIC, pos=(0, 0, 0), orn=(0, 0, 0, 1), linvel=(0, 0, 0), angvel=(0, 0, 0),
                 mass=1.0, shape_type=SHAPE_NONE, shape_param=(0, 0, 0, 0), inertia=None, friction=0.5,
                 restitution=0.0, has_material=True, group=2**64 - 1, mask=2**64 - 1, gravity=None,
                 sleeping_disabled=True):
        I = _fp(_f32(inertia, 9)) if inertia is not None else None
        g = _fp(_f32(gravity, 3)) if gravity is not None else None
        return self.L.refw_add_body(self.h, kind, _fp(_f32(pos, 3)), _fp(_f32(orn, 4)), _fp(_f32(linvel, 3)),
                                    _fp(_f32(angvel, 3)), mass, shape_type, _fp(_f32(shape_param, 4)), I,
                                    friction, restitution, int(has_material), group, mask, g, int(sleeping_disabled))

    def move_center_of_mass(self, body, com):
        f = self.L.refw_set_center_of_mass; f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, body, _fp(_f32(com, 3)))

    def next_center_of_mass(self, com):
        """rigidbody_def::center_of_mass of the next add_body (whose position is then the origin)."""
        f = self.L.refw_next_center_of_mass; f.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, _fp(_f32(com, 3)))

    def add_bodies(self, scene, sleeping_disabled=True):
        n = len(scene["kind"])
        inertia = scene.get("inertia")
        has_inertia = scene.get("has_inertia")
        mesh_ids = scene_mesh_ids(scene, real=True)
        for i in range(n):
            I = inertia[i] if (inertia is not None and has_inertia is not None and has_inertia[i]) else None
            grav = scene["gravity"][i] if scene.get("gravity") is not None else None
            if scene.get("com") is not None and np.any(scene["com"][i] != 0):
                self.next_center_of_mass(scene["com"][i])
            self.add_body(int(scene["kind"][i]), scene["pos"][i], scene["orn"][i], scene["linvel"][i],
                          scene["angvel"][i], float(scene["mass"][i]), int(scene["shape_type"][i]),
                          _shape_param(scene, i, mesh_ids), I, float(scene["friction"][i]), float(scene["restitution"][i]),
                          True, int(scene["group"][i]), int(scene["mask"][i]), grav, sleeping_disabled)
        for j in scene.get("joints") or []:
            self.add_joint(*j)

    def add_joint(self, jtype, a, b, pivotA, pivotB, axisA=(1, 0, 0), axisB=(1, 0, 0), params=None):
        self.n_joints += 1
        j = self.L.refw_add_joint(self.h, jtype, a, b, _fp(_f32(pivotA, 3)), _fp(_f32(pivotB, 3)),
                                  _fp(_f32(axisA, 3)), _fp(_f32(axisB, 3)))
        if params is not None:
            self.set_joint_params(j, params)
        return j

    def set_joint_params(self, joint, params):
        p = np.zeros(10, np.float32); p[:len(params)] = params
        self.L.refw_set_joint_params(self.h, joint, _fp(p))

    def exclude_collision(self, a, b):
        self.L.refw_exclude_collision(self.h, a, b)

    def set_should_collide(self, func):
        """edyn::set_should_collide on the real engine: func(body, other) -> bool (None restores should_collide_default)."""
        self._filter_cb = PAIR_FILTER(lambda user, a, b: 1 if func(int(a), int(b)) else 0) if func else None
        f = self.L.refw_set_should_collide; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = None
        f(self.h, C.cast(self._filter_cb, C.c_void_p) if func else None, None)

    def default_should_collide(self, a, b):
        f = self.L.refw_default_should_collide; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = C.c_int
        return bool(f(self.h, a, b))

    def remove_body(self, body):
        f = self.L.refw_remove_body; f.argtypes = [C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, body)

    def set_generic_definition(self, joint, frameA, frameB, dofs):
        p = np.ascontiguousarray(np.asarray(dofs, np.float32).reshape(60))
        f = self.L.refw_set_generic_definition; f.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_float)] * 3; f.restype = None
        f(self.h, joint, _fp(_f32(np.asarray(frameA).ravel(), 9)), _fp(_f32(np.asarray(frameB).ravel(), 9)), _fp(p))

    def get_joint_impulses24(self):
        out = np.zeros((self.n_joints, 24), np.float32)
        if self.n_joints:
            f = self.L.refw_get_joint_impulses24; f.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; f.restype = None
            f(self.h, _fp(out))
        return out

    def set_joint_definition(self, joint, frameA, frameB, params):
        p = np.zeros(16, np.float32); p[:len(params)] = params
        f = self.L.refw_set_joint_definition; f.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_float)] * 3; f.restype = None
        f(self.h, joint, _fp(_f32(np.asarray(frameA).ravel(), 9)), _fp(_f32(np.asarray(frameB).ravel(), 9)), _fp(p))

    def make_ragdoll(self, shape="capsule", pos=(0, 0, 0), orn=(0, 0, 0, 1), height=1.7, weight=72.0, friction=0.5, restitution=0.0):
        """edyn::make_ragdoll (util/ragdoll.cpp:65-914) run by the real engine. Returns (first body, first joint) of the figure."""
        f = self.L.refw_make_ragdoll; f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)] + [C.c_float] * 4
        g = self.L.refw_num_joints; g.argtypes = [C.c_void_p]; g.restype = C.c_uint32
        first_body, first_joint = self.num_bodies, self.n_joints
        f(self.h, {"box": 0, "capsule": 1, "cylinder": 2}[shape], _fp(_f32(pos, 3)), _fp(_f32(orn, 4)), height, weight, friction, restitution)
        self.n_joints = g(self.h)
        return first_body, first_joint

    def export_figure(self, first_body=0, first_joint=0):
        """Bodies [first_body:] and constraints [first_joint:] as plain arrays (indices relative to first_body): what
        World.add_bodies / set_joint_params / set_joint_definition / exclude_collision take - see tests/golden/make_ragdoll.py."""
        L = self.L
        fb = L.refw_export_body; fb.restype = None
        fb.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)] + [C.POINTER(C.c_float)] * 5 + [C.POINTER(C.c_int32)] + \
                      [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        fj = L.refw_export_joint; fj.restype = None
        fj.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)] + [C.POINTER(C.c_float)] * 8
        fx = L.refw_export_exclusions; fx.restype = C.c_uint32; fx.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        nb, nj = self.num_bodies - first_body, self.n_joints - first_joint
        out = dict(kind=np.zeros(nb, np.int32), pos=np.zeros((nb, 3), np.float32), orn=np.zeros((nb, 4), np.float32),
                   linvel=np.zeros((nb, 3), np.float32), angvel=np.zeros((nb, 3), np.float32), mass=np.zeros(nb, np.float32),
                   shape_type=np.zeros(nb, np.int32), shape_param=np.zeros((nb, 4), np.float32), inertia=np.zeros((nb, 9), np.float32),
                   friction=np.zeros(nb, np.float32), restitution=np.zeros(nb, np.float32), has_material=np.zeros(nb, np.int32),
                   group=np.zeros(nb, np.uint64), mask=np.zeros(nb, np.uint64))
        i32 = lambda a, i: a[i:i + 1].ctypes.data_as(C.POINTER(C.c_int32))
        u64 = lambda a, i: a[i:i + 1].ctypes.data_as(C.POINTER(C.c_uint64))
        for i in range(nb):
            fb(self.h, first_body + i, i32(out["kind"], i), _fp(out["pos"][i]), _fp(out["orn"][i]), _fp(out["linvel"][i]), _fp(out["angvel"][i]),
               _fp(out["mass"][i:i + 1]), i32(out["shape_type"], i), _fp(out["shape_param"][i]), _fp(out["inertia"][i]),
               _fp(out["friction"][i:i + 1]), _fp(out["restitution"][i:i + 1]), i32(out["has_material"], i), u64(out["group"], i), u64(out["mask"], i))
        j = dict(joint_type=np.zeros(nj, np.int32), joint_body=np.zeros((nj, 2), np.uint32), pivotA=np.zeros((nj, 3), np.float32),
                 pivotB=np.zeros((nj, 3), np.float32), axisA=np.zeros((nj, 3), np.float32), axisB=np.zeros((nj, 3), np.float32),
                 params10=np.zeros((nj, 10), np.float32), frameA=np.zeros((nj, 9), np.float32), frameB=np.zeros((nj, 9), np.float32),
                 params16=np.zeros((nj, 16), np.float32))
        for i in range(nj):
            fj(self.h, first_joint + i, i32(j["joint_type"], i), j["joint_body"][i].ctypes.data_as(C.POINTER(C.c_uint32)), _fp(j["pivotA"][i]),
               _fp(j["pivotB"][i]), _fp(j["axisA"][i]), _fp(j["axisB"][i]), _fp(j["params10"][i]), _fp(j["frameA"][i]), _fp(j["frameB"][i]), _fp(j["params16"][i]))
        j["joint_body"] -= np.uint32(first_body)
        ex = np.zeros((4096, 2), np.uint32)
        ne = fx(self.h, ex.ctypes.data_as(C.POINTER(C.c_uint32)), len(ex))
        ex = ex[:ne]
        ex = ex[(ex >= first_body).all(axis=1)] - np.uint32(first_body)
        out.update(j); out["exclusions"] = ex
        return out

    def remove_joint(self, joint):
        f = self.L.refw_remove_joint; f.argtypes = [C.c_void_p, C.c_uint32]; f.restype = None
        f(self.h, joint)

    def set_params(self, dt, vel_iters, pos_iters, gravity=None):
        f = self.L.refw_set_params; f.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, dt, vel_iters, pos_iters, _fp(_f32(gravity, 3)) if gravity is not None else None)

    def set_restitution_iterations(self, iters, individual=3):
        self.L.refw_set_restitution_iterations(self.h, iters, individual)

    def step(self, n=1):
        self.L.refw_step(self.h, n)

    def time_steps(self, n):
        return self.L.refw_time_steps(self.h, n)

    def update(self, time, paused=False):
        self.L.refw_update(self.h, float(time), int(paused))

    def set_max_steps_per_update(self, n):
        self.L.refw_set_max_steps_per_update(self.h, n)

    @property
    def num_bodies(self):
        return self.L.refw_num_bodies(self.h)

    def get_state(self):
        n = self.num_bodies
        pos = np.zeros((n, 3), np.float32); orn = np.zeros((n, 4), np.float32)
        lv = np.zeros((n, 3), np.float32); av = np.zeros((n, 3), np.float32)
        self.L.refw_get_state(self.h, _fp(pos), _fp(orn), _fp(lv), _fp(av))
        return pos, orn, lv, av

    def set_state(self, pos, orn, lv, av):
        n = self.num_bodies
        self.L.refw_set_state(self.h, _fp(_f32(pos, 3 * n)), _fp(_f32(orn, 4 * n)), _fp(_f32(lv, 3 * n)), _fp(_f32(av, 3 * n)))

    def get_derived(self):
        n = self.num_bodies
        aabb = np.zeros((n, 6), np.float32); iw = np.zeros((n, 9), np.float32)
        isl = np.zeros(n, np.uint32); asleep = np.zeros(n, np.uint8)
        self.L.refw_get_derived(self.h, _fp(aabb), _fp(iw), isl.ctypes.data, asleep.ctypes.data)
        return aabb, iw, isl, asleep.astype(bool)

    def get_asleep(self):
        return self.get_derived()[3]

    @property
    def num_islands(self):
        return self.L.refw_num_islands(self.h)

    def get_manifolds(self):
        m = self.L.refw_num_manifolds(self.h)
        out = np.zeros(m, MANIFOLD_DTYPE)
        if m:
            self.L.refw_get_manifolds(self.h, out.ctypes.data)
        return out

    def get_pairs(self):
        m = self.get_manifolds()
        b = m["body"].astype(np.uint64)
        hi = np.maximum(b[:, 0], b[:, 1]); lo = np.minimum(b[:, 0], b[:, 1])
        return np.sort((hi << np.uint64(32)) | lo)

    def set_material_id(self, body, mid):
        f = self.L.refw_set_material_id; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; f.restype = None
        f(self.h, body, mid)

    def insert_material_mixing(self, id0, id1, restitution=0.0, friction=0.5, spin=0.0, roll=0.0, stiffness=1e18, damping=1e18):
        m = np.array([restitution, friction, spin, roll, stiffness, damping], np.float32)
        f = self.L.refw_insert_material_mixing; f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]; f.restype = None
        f(self.h, id0, id1, _fp(m))

    def set_material_extras(self, body, spin=0.0, roll=0.0, stiffness=1e18, damping=1e18):
        f = self.L.refw_set_material_extras; f.argtypes = [C.c_void_p, C.c_uint32] + [C.c_float] * 4; f.restype = None
        f(self.h, body, spin, roll, stiffness, damping)

    def get_point_extras(self):
        m = self.L.refw_num_manifolds(self.h)
        out = np.zeros((m, 4, 7), np.float32)
        if m:
            g = self.L.refw_get_point_extras; g.argtypes = [C.c_void_p, C.c_void_p]; g.restype = None
            g(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def record_events(self, on=True):
        f = self.L.refw_record_events; f.argtypes = [C.c_void_p, C.c_int]; f.restype = None
        f(self.h, int(on))

    def clear_events(self):
        f = self.L.refw_clear_events; f.argtypes = [C.c_void_p]; f.restype = None
        f(self.h)

    def get_events(self):
        """[n, 3] (type, body A, body B): the engine's on_construct / on_destroy of contact_manifold (1, 2) and contact_point (3, 4)."""
        f = self.L.refw_num_events; f.argtypes = [C.c_void_p]; f.restype = C.c_uint32
        n = f(self.h)
        out = np.zeros((n, 3), np.uint32)
        if n:
            g = self.L.refw_get_events; g.argtypes = [C.c_void_p, C.c_void_p]; g.restype = None
            g(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def get_solve_order(self, max_entries=1 << 22):
        """(contacts[n,3], joints[m]) in the order the last step's island solvers visited them."""
        c = np.zeros((max_entries, 3), np.uint32); j = np.zeros(max(self.n_joints, 1), np.uint32)
        f = self.L.refw_get_contact_order; f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_uint32
        g = self.L.refw_get_joint_order; g.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; g.restype = C.c_uint32
        nc = f(self.h, c.ctypes.data, max_entries); nj = g(self.h, j.ctypes.data, len(j))
        return c[:nc].copy(), j[:nj].copy()

    def get_restitution_walk(self, max_manifolds=1 << 20, max_words=1 << 23):
        """(manifolds[n,2] in island edge order, adjacency words): what the last step's restitution solver walked by (ref_world.cpp)."""
        m = np.zeros((max_manifolds, 2), np.uint32); a = np.zeros(max_words, np.uint32); nm = C.c_uint32(0)
        f = self.L.refw_get_restitution_walk
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32]; f.restype = C.c_uint32
        na = f(self.h, m.ctypes.data, max_manifolds, C.byref(nm), a.ctypes.data, max_words)
        return m[:nm.value].copy(), a[:na].copy()

    def get_joint_impulses(self):
        out = np.zeros((self.n_joints, 10), np.float32)
        if self.n_joints:
            self.L.refw_get_joint_impulses(self.h, _fp(out))
        return out
