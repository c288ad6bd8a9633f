"""Host-side mirror of the reference's public stepping API over the C-ABI.

Names and argument meaning follow include/edyn/edyn.hpp:66-150 (attach / detach / update /
step_simulation / set_paused / get_fixed_dt ...), include/edyn/util/rigidbody.hpp:29-93
(rigidbody_def, make_rigidbody) and include/edyn/util/constraint_util.hpp:38-54 (make_constraint).
The EnTT registry itself is a C++ construct: the C++ drop-in shim lives in include/edyn/edyn.hpp of
this repo; this Python mirror exists so the parity tests and the bench read like the reference's
own integration tests (attach -> make_rigidbody -> update -> read components).
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence
import numpy as np
from . import _capi
from ._capi import EdynHipError, MANIFOLD_DTYPE

KIND_DYNAMIC, KIND_KINEMATIC, KIND_STATIC = 0, 1, 2          # rigidbody_kind
SHAPE_NONE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_PLANE, SHAPE_CAPSULE, SHAPE_CYLINDER, SHAPE_POLYHEDRON = 0, 1, 2, 3, 4, 5, 6
JOINT_POINT, JOINT_HINGE, JOINT_DISTANCE, JOINT_SOFT_DISTANCE, JOINT_CONE, JOINT_CVJOINT, JOINT_GRAVITY, JOINT_GENERIC = 0, 1, 2, 3, 4, 5, 6, 7
ALL_GROUPS = 2**64 - 1                                       # collision_filter::all_groups


@dataclass
class init_config:
    """edyn::init_config + the settings fields on the hot path (context/settings.hpp:21-57)."""
    fixed_dt: float = 1.0 / 60.0
    num_solver_velocity_iterations: int = 8
    num_solver_position_iterations: int = 3
    max_steps_per_update: int = 10
    gravity: Sequence[float] = (0.0, -9.8, 0.0)
    device: int = 0
    max_bodies: int = 0          # 0 = size from the first scene upload
    max_manifolds: int = 0
    max_joints: int = 0
    timing: bool = False         # HIP events around every stage (diagnostic: each event idles the GPU ~6 us)
    timing_solve: bool = False   # HIP events around the velocity solve only
    sleeping: bool = False       # island sleeping (off = every body sleeping_disabled, as the benchmark scenes are)
    contact_events: bool = False # record manifold / contact point creation and destruction (get_contact_events)
    exclusive_device: bool = False   # promise that this stepper is the only user of its GPU while it steps (see edynhip.h)
    # Contact arithmetic (edynhip.h EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / _BLOCK_POSITION). Default: every contact row and position
    # correction with the reference's operations in the reference's order; the two switches opt in to faster forms of the same equations.
    fused_velocity_rows: bool = False
    block_position: bool = False


@dataclass
class rigidbody_def:
    """edyn::rigidbody_def (include/edyn/util/rigidbody.hpp:29-81), hot-path fields."""
    kind: int = KIND_DYNAMIC
    position: Sequence[float] = (0.0, 0.0, 0.0)
    orientation: Sequence[float] = (0.0, 0.0, 0.0, 1.0)
    mass: float = 1.0
    inertia: Optional[Sequence[float]] = None    # 3x3, row-major
    linvel: Sequence[float] = (0.0, 0.0, 0.0)
    angvel: Sequence[float] = (0.0, 0.0, 0.0)
    gravity: Optional[Sequence[float]] = None
    shape_type: int = SHAPE_NONE
    shape_param: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    friction: float = 0.5
    restitution: float = 0.0
    collision_group: int = ALL_GROUPS
    collision_mask: int = ALL_GROUPS
    sleeping_disabled: bool = False


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fixed_step_plan(accumulated, elapsed, fixed_dt, max_steps):
    """stepper_sequential::update's accumulator (stepper_sequential.cpp:45-66): returns
    (effective_steps, new_accumulated, step_dt). num_steps = floor(acc / dt); the remainder stays accumulated; the
    number of steps actually run is clamped to max_steps_per_update, and then the time stamps of the steps that do run
    are stretched over the whole advance: step_dt = num_steps * dt / effective_steps (it feeds the island sleep
    timers only - the solver always integrates with fixed_dt, solver.cpp:390)."""
    accumulated += max(elapsed, 0.0)
    num_steps = int(math.floor(accumulated / fixed_dt))
    advance = num_steps * fixed_dt
    accumulated -= advance
    if num_steps > max_steps:
        return max_steps, accumulated, advance / max_steps
    return num_steps, accumulated, fixed_dt


class World:
    """One attached simulation: the analogue of a registry with edyn attached."""

    def __init__(self, config: Optional[init_config] = None):
        self.cfg = config or init_config()
        self._h = None
        self._defs = []
        self._joints = []
        self._dirty = True
        self._paused = False
        self._accum = 0.0
        self._last_time = 0.0
        self.n = 0
        self.nj = 0
        self.num_meshes = 0
        self._mesh_sig = []   # content hashes of the meshes created through set_scene, in id order
        self._L = _capi.lib()

    # ---- edyn::attach / detach
    def attach(self, max_bodies, max_joints=0):
        self.detach()
        cfg = _capi.Config()
        cfg.device = self.cfg.device
        cfg.max_bodies = max(int(max_bodies), 1)
        cfg.max_manifolds = int(self.cfg.max_manifolds)
        cfg.max_joints = max(int(max_joints), 0)
        cfg.fixed_dt = self.cfg.fixed_dt
        cfg.num_velocity_iterations = self.cfg.num_solver_velocity_iterations
        cfg.num_position_iterations = self.cfg.num_solver_position_iterations
        cfg.gravity = (C.c_float * 3)(*[float(x) for x in self.cfg.gravity])
        cfg.flags = ((_capi.FLAG_TIMING if self.cfg.timing else 0) | (_capi.FLAG_TIMING_SOLVE if self.cfg.timing_solve else 0)
                     | (_capi.FLAG_SLEEPING if self.cfg.sleeping else 0)
                     | (_capi.FLAG_CONTACT_EVENTS if self.cfg.contact_events else 0)
                     | (_capi.FLAG_EXCLUSIVE_DEVICE if self.cfg.exclusive_device else 0)
                     | (_capi.FLAG_FUSED_VELOCITY_ROWS if self.cfg.fused_velocity_rows else 0)
                     | (_capi.FLAG_BLOCK_POSITION if self.cfg.block_position else 0))
        st = C.c_int(0)
        h = self._L.edynhip_create(C.byref(cfg), C.byref(st))
        if not h:
            raise EdynHipError(st.value, self._L.edynhip_last_error(None).decode())
        self._h = C.c_void_p(h)
        self.num_meshes = 0   # meshes belong to the context
        self._mesh_sig = []
        if getattr(self, "_filter_cb", None) is not None:   # the user's should_collide predicate belongs to the world, not to one context
            self._check(self._L.edynhip_set_pair_filter(self._h, C.cast(self._filter_cb, C.c_void_p), None))

    def detach(self):
        if self._h:
            self._L.edynhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.detach()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EdynHipError(rc, self._L.edynhip_last_error(self._h).decode())

    # ---- edyn::make_rigidbody / make_constraint (deferred: uploaded on the next update)
    def make_rigidbody(self, d: rigidbody_def) -> int:
        self._defs.append(d)
        self._dirty = True
        return len(self._defs) - 1

    def make_constraint(self, jtype, body0, body1, pivot0, pivot1, axis0=(1, 0, 0), axis1=(1, 0, 0), params=None) -> int:
        """edyn::make_constraint<point_constraint|hinge_constraint>; params = the optional-row settings (see edynhip_joints)."""
        j = (jtype, body0, body1, tuple(pivot0), tuple(pivot1), tuple(axis0), tuple(axis1))
        self._joints.append(j + ((tuple(params),) if params is not None else ()))
        self._dirty = True
        return len(self._joints) - 1

    def _body_arrays(self, scene):
        n = len(scene["kind"])
        a = {}
        a["kind"] = np.ascontiguousarray(scene["kind"], np.int32)
        for k, w in (("pos", 3), ("orn", 4), ("linvel", 3), ("angvel", 3), ("shape_param", 4)):
            a[k] = np.ascontiguousarray(scene[k], np.float32).reshape(n, w)
        for k in ("mass", "friction", "restitution"):
            a[k] = np.ascontiguousarray(scene[k], np.float32)
        a["shape_type"] = np.ascontiguousarray(scene["shape_type"], np.int32)
        a["group"] = np.ascontiguousarray(scene["group"], np.uint64)
        a["mask"] = np.ascontiguousarray(scene["mask"], np.uint64)
        inertia = scene.get("inertia")
        has_inertia = scene.get("has_inertia")
        if inertia is not None and has_inertia is not None:
            a["inertia"] = np.ascontiguousarray(inertia, np.float32).reshape(n, 9)
            a["has_inertia"] = np.ascontiguousarray(has_inertia, np.uint8)
        grav = scene.get("gravity")
        if grav is not None:
            a["gravity"] = np.ascontiguousarray(grav, np.float32).reshape(n, 3)
        if scene.get("sleeping_disabled") is not None:
            a["sleeping_disabled"] = np.ascontiguousarray(scene["sleeping_disabled"], np.uint8)
        if scene.get("com") is not None:   # rigidbody_def::center_of_mass: `pos` is then the origin
            a["center_of_mass"] = np.ascontiguousarray(scene["com"], np.float32).reshape(n, 3)
        b = _capi.Bodies()
        for f, _ in _capi.Bodies._fields_:
            setattr(b, f, _ptr(a.get(f)))
        return n, a, b   # `a` keeps the arrays alive while `b` points into them

    @staticmethod
    def _joint_arrays(joints):
        jt = np.array([j[0] for j in joints], np.int32)
        jb = np.array([[j[1], j[2]] for j in joints], np.uint32)
        jp = np.array([[j[3], j[4]] for j in joints], np.float32).reshape(-1, 6)
        ja = np.array([[j[5], j[6]] for j in joints], np.float32).reshape(-1, 6)
        jq = np.zeros((len(joints), 10), np.float32)
        for i, j in enumerate(joints):
            if len(j) > 7:
                jq[i, :len(j[7])] = j[7]
        keep = (jt, jb, jp, ja, jq)
        return keep, _capi.Joints(_ptr(jt), _ptr(jb), _ptr(jp), _ptr(ja), _ptr(jq))

    def _upload_joints(self, joints):
        self.nj = len(joints)
        if joints:
            keep, js = self._joint_arrays(joints)
            self._check(self._L.edynhip_set_joints(self._h, len(joints), C.byref(js)))
        else:
            self._check(self._L.edynhip_set_joints(self._h, 0, None))

    def add_joints(self, joints):
        """Append joints to a running world (applied impulses of the existing ones are kept). Returns the first new index."""
        keep, js = self._joint_arrays(joints)
        first = C.c_uint32(0)
        self._check(self._L.edynhip_add_joints(self._h, len(joints), C.byref(js), C.byref(first)))
        self.nj += len(joints)
        return first.value

    def remove_joints(self, indices):
        """registry.destroy(constraint entity): the indices stay reserved."""
        self._flush_defs()
        idx = np.ascontiguousarray(indices, np.uint32)
        self._check(self._L.edynhip_remove_joints(self._h, len(idx), _ptr(idx)))

    def set_joint_params(self, joint, params):
        self._flush_defs()
        p = np.zeros(10, np.float32); p[:len(params)] = params
        self._check(self._L.edynhip_set_joint_params(self._h, int(joint), _ptr(p)))

    def set_generic_definition(self, joint, frameA, frameB, dofs):
        """generic_constraint: frames and [6][10] degree-of-freedom parameters (edynhip.h edynhip_set_generic_definition)."""
        self._flush_defs()
        p = np.ascontiguousarray(np.asarray(dofs, np.float32).reshape(60))
        fa = np.ascontiguousarray(np.asarray(frameA, np.float32).reshape(9)); fb = np.ascontiguousarray(np.asarray(frameB, np.float32).reshape(9))
        self._check(self._L.edynhip_set_generic_definition(self._h, int(joint), _ptr(fa), _ptr(fb), _ptr(p)))

    def get_joint_impulses24(self):
        self._flush_defs()
        out = np.zeros((self.nj, 24), np.float32)
        if self.nj:
            self._check(self._L.edynhip_get_joint_slot_impulses(self._h, _ptr(out)))
        return out

    def set_joint_definition(self, joint, frameA, frameB, params):
        """Frames (3x3, first column = cone direction / twist axis) and parameters of a cone or cvjoint constraint (edynhip.h)."""
        self._flush_defs()
        p = np.zeros(16, np.float32); p[:len(params)] = params
        fa = np.ascontiguousarray(np.asarray(frameA, np.float32).reshape(9)); fb = np.ascontiguousarray(np.asarray(frameB, np.float32).reshape(9))
        self._check(self._L.edynhip_set_joint_definition(self._h, int(joint), _ptr(fa), _ptr(fb), _ptr(p)))

    def remove_bodies(self, indices):
        """registry.destroy(rigid body) on a running world: manifolds and joints of the body go with it, its index stays reserved."""
        self._flush_defs()
        idx = np.ascontiguousarray(indices, np.uint32)
        self._check(self._L.edynhip_remove_bodies(self._h, len(idx), _ptr(idx)))

    def set_params(self, fixed_dt=None, velocity_iterations=None, position_iterations=None, gravity=None,
                   restitution_iterations=None, individual_restitution_iterations=None):
        """set_fixed_dt / set_solver_*_iterations / set_gravity on the running world: no contact state is lost."""
        self._flush_defs()
        p = _capi.Params()
        self._check(self._L.edynhip_get_params(self._h, C.byref(p)))
        if fixed_dt is not None:
            p.fixed_dt = fixed_dt; self.cfg.fixed_dt = fixed_dt
        if velocity_iterations is not None:
            p.num_velocity_iterations = velocity_iterations; self.cfg.num_solver_velocity_iterations = velocity_iterations
        if position_iterations is not None:
            p.num_position_iterations = position_iterations; self.cfg.num_solver_position_iterations = position_iterations
        if gravity is not None:
            p.gravity = (C.c_float * 3)(*[float(x) for x in gravity]); self.cfg.gravity = tuple(gravity)
        if restitution_iterations is not None:
            p.num_restitution_iterations = restitution_iterations
        if individual_restitution_iterations is not None:
            p.num_individual_restitution_iterations = individual_restitution_iterations
        self._check(self._L.edynhip_set_params(self._h, C.byref(p)))

    def step_timed(self, n, first_step_time, step_dt):
        self._flush_defs()
        self._check(self._L.edynhip_step_timed(self._h, n, float(first_step_time), float(step_dt)))

    def set_scene(self, scene):
        """Bulk scene upload from a dict of arrays (see edyn_amd.scenes). Replaces the whole world."""
        joints = scene.get("joints") or []
        n, keep, b = self._body_arrays(scene)
        if self._h is None:
            self.attach(self.cfg.max_bodies or n, self.cfg.max_joints or len(joints))
        # polyhedron bodies refer to the scene's meshes by position (shape_param[0]): the context must hold exactly that list as its
        # first meshes - a context that holds other meshes is replaced
        sig = [hash((np.ascontiguousarray(m["vertices"], np.float32).tobytes(), np.ascontiguousarray(m["indices"], np.uint32).tobytes(),
                     np.ascontiguousarray(m["faces"], np.uint32).tobytes())) for m in scene.get("meshes") or []]
        if sig and self._mesh_sig[:len(sig)] != sig[:len(self._mesh_sig)]:
            self.attach(self.cfg.max_bodies or n, self.cfg.max_joints or len(joints))
        for k, mesh in enumerate(scene.get("meshes") or []):
            if k >= self.num_meshes:
                mid = self.create_convex_mesh(mesh["vertices"], mesh["indices"], mesh["faces"])
                if mid != k:
                    raise RuntimeError(f"convex mesh {k} of the scene was registered as mesh {mid}")
                self._mesh_sig.append(sig[k])
        self._check(self._L.edynhip_set_bodies(self._h, n, C.byref(b)))
        self.n = n
        self._upload_joints(joints)
        self._uploaded = (n, len(joints))
        self._dirty = False

    def add_scene(self, scene):
        """Append the bodies of `scene` to a running world (contact manifolds and cached impulses of the existing bodies
        are kept). Returns the index of the first new body. Joints in `scene` are ignored here."""
        n, keep, b = self._body_arrays(scene)
        if self._h is None:
            raise EdynHipError(-1, "add_scene needs an attached world (call set_scene or attach first)")
        first = self.n
        self._check(self._L.edynhip_add_bodies(self._h, n, C.byref(b)))
        self.n += n
        return first

    def _flush_defs(self):
        if not self._dirty:
            return
        from .scenes import scene_from_defs
        up_b, up_j = getattr(self, "_uploaded", (0, 0))
        if self._h is not None and 0 < up_b <= len(self._defs) and self.n == up_b:
            # bodies were only appended since the last upload: keep the running contact state
            if len(self._defs) > up_b:
                self.add_scene(scene_from_defs(self._defs[up_b:], []))
            if len(self._joints) != up_j:
                self._upload_joints(self._joints)
            self._uploaded = (len(self._defs), len(self._joints))
            self._dirty = False
            return
        self.set_scene(scene_from_defs(self._defs, self._joints))

    # ---- stepping (stepper_sequential.cpp:28-147)
    def get_fixed_dt(self):
        return self.cfg.fixed_dt

    def set_paused(self, paused):
        self._paused = paused
        self._accum = 0.0

    def is_paused(self):
        return self._paused

    def update(self, time: float):
        """edyn::update(registry, time): fixed-dt accumulator with the max_steps_per_update clamp."""
        self._flush_defs()
        if self._paused:
            return 0
        sim_time = self._last_time - self._accum      # get_simulation_timestamp() before this update
        steps, self._accum, step_dt = fixed_step_plan(self._accum, time - self._last_time, float(np.float32(self.cfg.fixed_dt)),
                                                      self.cfg.max_steps_per_update)
        if steps:
            self._check(self._L.edynhip_step_timed(self._h, steps, sim_time, step_dt))
        self._last_time = time
        return steps

    def step_simulation(self, n=1):
        """edyn::step_simulation: exactly one fixed step per call (n calls)."""
        self._flush_defs()
        self._check(self._L.edynhip_step(self._h, n))

    def run_stages(self, mask):
        self._flush_defs()
        self._check(self._L.edynhip_run_stages(self._h, mask))

    def synchronize(self):
        self._check(self._L.edynhip_synchronize(self._h))

    def set_stream(self, stream_ptr):
        self._check(self._L.edynhip_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- component read-back
    def get_state(self):
        n = self.n
        pos = np.zeros((n, 3), np.float32); orn = np.zeros((n, 4), np.float32)
        lv = np.zeros((n, 3), np.float32); av = np.zeros((n, 3), np.float32)
        self._check(self._L.edynhip_get_state(self._h, _ptr(pos), _ptr(orn), _ptr(lv), _ptr(av)))
        return pos, orn, lv, av

    def set_state(self, pos, orn, lv, av):
        n = self.n
        arrs = [np.ascontiguousarray(x, np.float32).reshape(n, w) for x, w in ((pos, 3), (orn, 4), (lv, 3), (av, 3))]
        self._check(self._L.edynhip_set_state(self._h, *[_ptr(x) for x in arrs]))

    def exclude_collision(self, a, b):
        """edyn::exclude_collision(registry, first, second)."""
        self._flush_defs()
        self._check(self._L.edynhip_exclude_collision(self._h, int(a), int(b)))

    def remove_collision_exclusion(self, a, b):
        self._flush_defs()
        self._check(self._L.edynhip_remove_collision_exclusion(self._h, int(a), int(b)))

    def set_should_collide(self, func):
        """edyn::set_should_collide: func(body, other) -> bool replaces should_collide_default for NEW manifolds (None restores the device
        test). A host callback: steps with new candidate pairs take the slow path described in include/edynhip.h."""
        self._filter_cb = _capi.PAIR_FILTER(lambda user, a, b: 1 if func(int(a), int(b)) else 0) if func else None
        if self._h is not None:   # (before attach: installed when the context is created)
            self._check(self._L.edynhip_set_pair_filter(self._h, C.cast(self._filter_cb, C.c_void_p) if func else None, None))

    def default_should_collide(self, a, b):
        """should_collide_default (collision groups / masks, exclusion lists) - for predicates that extend it."""
        r = self._L.edynhip_default_should_collide(self._h, int(a), int(b))
        if r < 0:
            self._check(r)
        return bool(r)

    def set_material_extras(self, first, spin=None, roll=None, stiffness=None, damping=None):
        """material::{spin_friction, roll_friction, stiffness, damping} of bodies [first, first + n) - contact_extras_constraint
        (rolling / spinning friction, soft contacts). Arrays of equal length; None = the reference's default."""
        self._flush_defs()
        arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (spin, roll, stiffness, damping)]
        n = max(len(a) for a in arrs if a is not None)
        self._check(self._L.edynhip_set_material_extras(self._h, int(first), n, *[None if a is None else _ptr(a) for a in arrs]))

    def set_material_ids(self, first, ids):
        """material::id of bodies [first, first + len(ids)) (0xFFFF = unassigned): keys into the material mix table."""
        self._flush_defs()
        a = np.ascontiguousarray(ids, np.uint32)
        self._check(self._L.edynhip_set_material_ids(self._h, int(first), len(a), _ptr(a)))

    def insert_material_mixing(self, id0, id1, restitution=0.0, friction=0.5, spin=0.0, roll=0.0, stiffness=1e18, damping=1e18):
        """edyn::insert_material_mixing: the material of contact points between bodies with these material ids."""
        self._flush_defs()
        m = np.array([restitution, friction, spin, roll, stiffness, damping], np.float32)
        self._check(self._L.edynhip_insert_material_mixing(self._h, int(id0), int(id1), _ptr(m)))

    def get_point_extras(self):
        """[num_manifolds, 4, 7]: rolling impulse 0/1, spin impulse, roll mu, spin mu, stiffness, damping (get_manifolds order)."""
        m = C.c_uint32(0)
        self._check(self._L.edynhip_num_manifolds(self._h, C.byref(m)))
        out = np.zeros((m.value, 4, 7), np.float32)
        if m.value:
            self._check(self._L.edynhip_get_point_extras(self._h, _ptr(out), m.value, C.byref(m)))
        return out

    def refresh_derived(self):
        """update_aabbs + update_inertias from the current transforms (after set_state)."""
        self._check(self._L.edynhip_refresh_derived(self._h))

    def pack_state_device(self, dst_ptr, first=0, count=None):
        self._check(self._L.edynhip_pack_state_device(self._h, C.c_void_p(dst_ptr), first, self.n if count is None else count))

    def get_derived(self):
        n = self.n
        aabb = np.zeros((n, 6), np.float32); iw = np.zeros((n, 9), np.float32); isl = np.zeros(n, np.uint32)
        self._check(self._L.edynhip_get_derived(self._h, _ptr(aabb), _ptr(iw), _ptr(isl)))
        return aabb, iw, isl

    def get_manifolds(self):
        m = C.c_uint32(0)
        self._check(self._L.edynhip_num_manifolds(self._h, C.byref(m)))
        out = np.zeros(m.value, MANIFOLD_DTYPE)
        if m.value:
            self._check(self._L.edynhip_get_manifolds(self._h, _ptr(out), m.value, C.byref(m)))
        return out

    # ---- contact events: what registry.on_construct / on_destroy<contact_manifold | contact_point> deliver in the reference
    def get_contact_events(self):
        """Events of the steps of the last step() / update() call: structured array (type, step, body[2], point_id)."""
        n = C.c_uint32(0)
        self._check(self._L.edynhip_get_contact_events(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, _capi.EVENT_DTYPE)
        if n.value:
            self._check(self._L.edynhip_get_contact_events(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def get_point_ids(self):
        """[num_manifolds, 4] point ids in get_manifolds() order (0 = no point)."""
        m = C.c_uint32(0)
        self._check(self._L.edynhip_num_manifolds(self._h, C.byref(m)))
        out = np.zeros((m.value, 4), np.uint64)
        if m.value:
            self._check(self._L.edynhip_get_point_ids(self._h, _ptr(out), m.value, C.byref(m)))
        return out

    # ---- double-buffered read-back: step(); snapshot(); step(); snapshot_read() returns the first step's state while the second runs
    def snapshot(self):
        self._check(self._L.edynhip_snapshot(self._h))

    def snapshot_read(self):
        n = self.n
        pos = np.zeros((n, 3), np.float32); orn = np.zeros((n, 4), np.float32)
        lin = np.zeros((n, 3), np.float32); ang = np.zeros((n, 3), np.float32)
        step = C.c_uint32(0)
        self._check(self._L.edynhip_snapshot_read(self._h, _ptr(pos), _ptr(orn), _ptr(lin), _ptr(ang), C.byref(step)))
        return (pos, orn, lin, ang), step.value

    # ---- the registry write-back read in place: 96-byte records (state, presentation transforms, origin, flags) + the contact events
    def snapshot_records(self, present_dt=0.0, max_events=4096, direct=False):
        self._check(self._L.edynhip_snapshot_records(self._h, float(present_dt), int(max_events), 1 if direct else 0))

    def snapshot_map(self):
        """(records, events, total_events, step_index): structured COPIES of the pinned slot (the C caller reads it in place)."""
        v = _capi.RecordView()
        self._check(self._L.edynhip_snapshot_map(self._h, C.byref(v)))
        rec = np.ctypeslib.as_array((C.c_uint8 * (v.num_bodies * _capi.RECORD_DTYPE.itemsize)).from_address(v.records)).view(_capi.RECORD_DTYPE).copy() \
            if v.num_bodies else np.zeros(0, _capi.RECORD_DTYPE)
        ev = np.zeros(0, _capi.EVENT_DTYPE)
        if v.events and v.num_events:
            ev = np.ctypeslib.as_array((C.c_uint8 * (v.num_events * _capi.EVENT_DTYPE.itemsize)).from_address(v.events)).view(_capi.EVENT_DTYPE).copy()
        return rec, ev, int(v.total_events), int(v.step_index)

    def set_event_prefetch(self, max_events):
        self._check(self._L.edynhip_set_event_prefetch(self._h, int(max_events)))

    def prefetched_events(self):
        """(events, total): the event list of the last step call as copied right after its last narrowphase (a structured copy)."""
        ptr, num, total = C.c_void_p(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.edynhip_prefetched_events(self._h, C.byref(ptr), C.byref(num), C.byref(total)))
        ev = np.zeros(0, _capi.EVENT_DTYPE)
        if ptr.value and num.value:
            ev = np.ctypeslib.as_array((C.c_uint8 * (num.value * _capi.EVENT_DTYPE.itemsize)).from_address(ptr.value)).view(_capi.EVENT_DTYPE).copy()
        return ev, int(total.value)

    def set_manifolds(self, recs):
        recs = np.ascontiguousarray(recs, MANIFOLD_DTYPE)
        self._check(self._L.edynhip_set_manifolds(self._h, _ptr(recs), len(recs)))

    def get_pairs(self):
        m = C.c_uint32(0)
        self._check(self._L.edynhip_num_manifolds(self._h, C.byref(m)))
        keys = np.zeros(m.value, np.uint64)
        if m.value:
            self._check(self._L.edynhip_get_pairs(self._h, _ptr(keys), m.value, C.byref(m)))
        return keys

    def get_joint_impulses(self):
        """[n, 10] by caller index: the 9 applied-impulse slots + the tracked hinge angle (edynhip_get_joint_impulses)."""
        out = np.zeros((self.nj, 10), np.float32)
        if self.nj:
            self._check(self._L.edynhip_get_joint_impulses(self._h, _ptr(out)))
        return out

    def debug_collide(self, shape_type, shape_param, pos, orn, threshold=0.01):
        """Device collide() on n independent shape pairs: returns (points[n,4,11], count[n])."""
        st = np.ascontiguousarray(shape_type, np.int32).reshape(-1, 2)
        n = len(st)
        sp = np.ascontiguousarray(shape_param, np.float32).reshape(n, 2, 4)
        ps = np.ascontiguousarray(pos, np.float32).reshape(n, 2, 3)
        qs = np.ascontiguousarray(orn, np.float32).reshape(n, 2, 4)
        out = np.zeros((n, 4, 11), np.float32); cnt = np.zeros(n, np.uint32)
        if self._h is None:
            self.attach(1)
        self._check(self._L.edynhip_debug_collide(self._h, n, _ptr(st), _ptr(sp), _ptr(ps), _ptr(qs), threshold, _ptr(out), _ptr(cnt)))
        return out, cnt

    def set_joint_warm_start(self, impulses24, angles=None):
        """Applied impulses ([nj, 24], the layout of get_joint_impulses24) and tracked angles of the joints, by caller index: what a
        world carries into another (edynhip_set_joint_warm_start)."""
        self._flush_defs()
        imp = np.ascontiguousarray(impulses24, np.float32).reshape(self.nj, 24)
        ang = None if angles is None else np.ascontiguousarray(angles, np.float32).reshape(self.nj)
        self._check(self._L.edynhip_set_joint_warm_start(self._h, _ptr(imp), _ptr(ang) if ang is not None else None))

    def get_sleep_timers(self):
        """(island label per body, time stamp since which each island - by label - meets the sleep thresholds, clock of the last step):
        edynhip_get_sleep_timers."""
        lab = np.zeros(self.n, np.uint32); since = np.zeros(self.n, np.float64); clock = C.c_double(0)
        self._check(self._L.edynhip_get_sleep_timers(self._h, _ptr(lab), _ptr(since), C.byref(clock)))
        return lab, since, clock.value

    def set_sleep_timers(self, labels, since, clock):
        """Continue another world's island sleep timers (edynhip_set_sleep_timers): after set_state / set_manifolds / set_asleep."""
        lab = np.ascontiguousarray(labels, np.uint32); sn = np.ascontiguousarray(since, np.float64)
        assert len(lab) == self.n and len(sn) == self.n
        self._check(self._L.edynhip_set_sleep_timers(self._h, _ptr(lab), _ptr(sn), float(clock)))

    def set_asleep(self, flags):
        """Sleeping tags by body (edynhip_set_asleep): sleeping bodies get zero velocities."""
        f = np.ascontiguousarray(np.asarray(flags).astype(np.uint8))
        assert len(f) == self.n
        self._check(self._L.edynhip_set_asleep(self._h, _ptr(f)))

    def get_asleep(self):
        out = np.zeros(self.n, np.uint8)
        if self.n:
            self._check(self._L.edynhip_get_asleep(self._h, _ptr(out)))
        return out.astype(bool)

    def wake_all(self):
        self._check(self._L.edynhip_wake_all(self._h))

    def move_center_of_mass(self, body, com):
        """edyn::set_center_of_mass on a running world."""
        self._flush_defs()
        c3 = np.ascontiguousarray(com, np.float32)
        self._check(self._L.edynhip_set_center_of_mass(self._h, int(body), _ptr(c3)))

    def wake_bodies(self, indices):
        """edyn::wake_up_entity for the listed bodies: wakes their islands."""
        idx = np.ascontiguousarray(indices, np.uint32)
        self._check(self._L.edynhip_wake_bodies(self._h, len(idx), _ptr(idx)))

    def get_timings(self):
        t = _capi.Timings()
        self._check(self._L.edynhip_get_timings(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in _capi.Timings._fields_}

    MESH_FIELDS = ("vertices", "normals", "relevant_normals", "edge_vertices", "edge_normals", "edges", "edge_faces", "relevant_faces",
                   "relevant_edges", "neighbors_start", "neighbor_indices", "inertia_sums")

    def create_convex_mesh(self, vertices, indices, faces, initialized=False):
        """polyhedron_shape's convex_mesh (convex_mesh.hpp:17-70) + initialize(): vertices [nv, 3], the faces' vertex indices and
        faces [nf, 2] = (first index, vertex count); initialized = the vertices are already relative to the centroid (initialize() ran
        on them). Returns the mesh id a SHAPE_POLYHEDRON body puts in shape_param[0]."""
        if self._h is None:
            self.attach(self.cfg.max_bodies or 1, self.cfg.max_joints)
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        f = np.ascontiguousarray(faces, np.uint32).reshape(-1, 2)
        out = C.c_uint32(0)
        self._check(self._L.edynhip_create_convex_mesh(self._h, len(v), _ptr(v), len(i), _ptr(i), len(f), _ptr(f), 1 if initialized else 0, C.byref(out)))
        self.num_meshes = int(out.value) + 1
        if len(self._mesh_sig) < int(out.value):   # created directly, ahead of a scene's list: no signature to compare with
            self._mesh_sig += [None] * (int(out.value) - len(self._mesh_sig))
        return int(out.value)

    def get_convex_mesh(self, mesh_id, field):
        """One derived array of a mesh (MESH_FIELDS): float fields [count, 3], index fields uint32."""
        what = self.MESH_FIELDS.index(field)
        n = C.c_uint32(0)
        self._check(self._L.edynhip_get_convex_mesh(self._h, mesh_id, what, None, 0, C.byref(n)))
        out = np.zeros(7, np.float32) if what == 11 else (np.zeros((n.value, 3), np.float32) if what < 5 else np.zeros(n.value, np.uint32))
        self._check(self._L.edynhip_get_convex_mesh(self._h, mesh_id, what, _ptr(out), n.value, C.byref(n)))
        return out

    def measure_bandwidth(self, nbytes=1 << 30):
        """(read GB/s, copy GB/s) of this GPU: the measured ceilings bench.py prints beside the HBM spec peak."""
        r, c = C.c_float(0), C.c_float(0)
        self._check(self._L.edynhip_measure_bandwidth(self._h, int(nbytes), C.byref(r), C.byref(c)))
        return float(r.value), float(c.value)

    def get_stats(self):
        s = _capi.Stats()
        self._check(self._L.edynhip_get_stats(self._h, C.byref(s)))
        out = {f: getattr(s, f) for f, _ in _capi.Stats._fields_ if f != "colour_size"}
        out["colour_size"] = [int(x) for x in s.colour_size][: out["num_colours"]]
        return out


# Free functions with the reference's names.
def attach(world: World, max_bodies, max_joints=0):
    world.attach(max_bodies, max_joints)


def detach(world: World):
    world.detach()


def make_rigidbody(world: World, d: rigidbody_def) -> int:
    return world.make_rigidbody(d)


def update(world: World, time: float):
    return world.update(time)


def step_simulation(world: World):
    world.step_simulation(1)
