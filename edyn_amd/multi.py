"""MultiWorld: thin binding of the library's multi-GPU world (include/edynhip.h "Multi-GPU world", edyn_amd/csrc/multi.hip).

One simulation, several GPUs of one node, one process: every device steps the islands it owns (solver.cpp:408-428: the island is
the reference's own unit of parallelism), the library gathers the state after every step, watches island bounding boxes reduced on
the devices and re-partitions when islands of different shards meet. Everything happens behind the C-ABI; this class only
marshals arrays. (Processes that own one GPU each - torch.distributed ranks over RCCL - use edyn_amd.parallel.ShardedWorld, which
shares the library's partitioner and box sweep.)"""
import ctypes as C
import numpy as np
from . import _capi
from ._capi import EdynHipError, MANIFOLD_DTYPE
from .world import World, init_config, _ptr


class MultiWorld:
    def __init__(self, config=None, devices=(0,)):
        self.cfg = config or init_config()
        self._L = _capi.lib()
        cfg = _capi.Config()
        cfg.device = 0
        cfg.max_bodies = 0; cfg.max_manifolds = int(self.cfg.max_manifolds); cfg.max_joints = 0
        cfg.fixed_dt = self.cfg.fixed_dt
        cfg.num_velocity_iterations = self.cfg.num_solver_velocity_iterations
        cfg.num_position_iterations = self.cfg.num_solver_position_iterations
        cfg.gravity = (C.c_float * 3)(*[float(x) for x in self.cfg.gravity])
        cfg.flags = ((_capi.FLAG_SLEEPING if self.cfg.sleeping else 0) | (_capi.FLAG_EXCLUSIVE_DEVICE if self.cfg.exclusive_device else 0)
                     | (_capi.FLAG_TIMING_SOLVE if self.cfg.timing_solve else 0)
                     | (_capi.FLAG_FUSED_VELOCITY_ROWS if self.cfg.fused_velocity_rows else 0) | (_capi.FLAG_BLOCK_POSITION if self.cfg.block_position else 0))
        dev = np.ascontiguousarray(devices, np.int32)
        st = C.c_int(0)
        h = self._L.edynhip_world_create(C.byref(cfg), dev.ctypes.data, len(dev), C.byref(st))
        if not h:
            raise EdynHipError(st.value, self._L.edynhip_world_last_error(None).decode())
        self._h = C.c_void_p(h)
        self.num_shards = len(dev)
        self.n = 0

    def close(self):
        if self._h:
            self._L.edynhip_world_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EdynHipError(rc, self._L.edynhip_world_last_error(self._h).decode())

    def set_scene(self, scene):
        """The whole scene in global indices: meshes, bodies, joints (with their optional-row parameters), cone / cvjoint
        definitions, collision exclusions - what World.set_scene + scenes.apply_figure_settings upload to one context."""
        for mesh in scene.get("meshes") or []:
            v = np.ascontiguousarray(mesh["vertices"], np.float32).reshape(-1, 3)
            idx = np.ascontiguousarray(mesh["indices"], np.uint32); faces = np.ascontiguousarray(mesh["faces"], np.uint32).reshape(-1, 2)
            mid = C.c_uint32(0)
            self._check(self._L.edynhip_world_create_convex_mesh(self._h, len(v), _ptr(v), len(idx), _ptr(idx), len(faces), _ptr(faces), 0, C.byref(mid)))
        n, keep, b = World._body_arrays(None, scene)
        self._check(self._L.edynhip_world_set_bodies(self._h, n, C.byref(b)))
        self.n = n
        joints = scene.get("joints") or []
        self.nj = len(joints)
        if joints:
            (jt, jb, jp, ja, jq), js = World._joint_arrays(joints)
            for j, p in scene.get("hinge_params", []):
                jq[j, :len(p)] = p
            self._check(self._L.edynhip_world_set_joints(self._h, len(joints), C.byref(js)))
        for j, fa, fb, p in scene.get("joint_defs", []):
            q = np.zeros(16, np.float32); q[:len(p)] = p
            fa = np.ascontiguousarray(np.asarray(fa, np.float32).reshape(9)); fb = np.ascontiguousarray(np.asarray(fb, np.float32).reshape(9))
            self._check(self._L.edynhip_world_set_joint_definition(self._h, int(j), _ptr(fa), _ptr(fb), _ptr(q), 0))
        for a, bb in scene.get("exclusions", []):
            self._check(self._L.edynhip_world_exclude_collision(self._h, int(a), int(bb)))

    def set_should_collide(self, func):
        """edyn::set_should_collide on a world over several devices: func(body, other) -> bool with GLOBAL body indices replaces
        should_collide_default for new manifolds (None restores the device test); edynhip_world_set_pair_filter."""
        self._filter_cb = _capi.PAIR_FILTER(lambda user, a, b: 1 if func(int(a), int(b)) else 0) if func else None
        self._check(self._L.edynhip_world_set_pair_filter(self._h, C.cast(self._filter_cb, C.c_void_p) if func else None, None))

    def default_should_collide(self, a, b):
        return self._L.edynhip_world_default_should_collide(self._h, int(a), int(b)) == 1

    def step_simulation(self, n=1):
        self._check(self._L.edynhip_world_step(self._h, int(n)))

    def get_state(self):
        pos = np.zeros((self.n, 3), np.float32); orn = np.zeros((self.n, 4), np.float32)
        lv = np.zeros((self.n, 3), np.float32); av = np.zeros((self.n, 3), np.float32)
        self._check(self._L.edynhip_world_get_state(self._h, _ptr(pos), _ptr(orn), _ptr(lv), _ptr(av)))
        return pos, orn, lv, av

    def get_partition(self):
        out = np.zeros(self.n, np.int32)
        self._check(self._L.edynhip_world_get_partition(self._h, _ptr(out)))
        return out

    def repartition(self):
        self._check(self._L.edynhip_world_repartition(self._h))

    def get_manifolds(self):
        n = C.c_uint32(0)
        self._check(self._L.edynhip_world_get_manifolds(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), MANIFOLD_DTYPE)
        self._check(self._L.edynhip_world_get_manifolds(self._h, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def get_stats(self):
        st = _capi.WorldStats()
        self._check(self._L.edynhip_world_get_stats(self._h, C.byref(st)))
        d = {f: getattr(st, f) for f, _ in _capi.WorldStats._fields_ if f != "bodies_per_shard"}
        d["bodies_per_shard"] = list(st.bodies_per_shard)[:self.num_shards]
        return d
