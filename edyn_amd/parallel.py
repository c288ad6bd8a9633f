"""Multi-GPU sharding of independent islands.

Islands never exchange data inside a step, so the path shards with NO data-path collective: every rank
owns a contiguous block of islands (here: sites of a mini-pile grid, or whole replicas of a pile) plus
its own copy of the static bodies, and steps them independently. The only collective is the gather of
the integrated state (13 floats per body: pos3, orn4, linvel3, angvel3) that the host registry
write-back needs - one all_gather per step over RCCL ("nccl" backend on ROCm; "gloo" in CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

from .scenes import KIND_DYNAMIC, subset


def shard_range(total, rank, world_size):
    """Contiguous, balanced [first, first+count) of `total` items for `rank`."""
    base, rem = divmod(total, world_size)
    first = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return first, count


def gather_state(local_state: torch.Tensor, counts):
    """All-gather per-rank state blocks [n_r, 13] into one [sum n_r, 13] tensor on every rank.
    `counts` = bodies per rank (known on all ranks from shard_range); ragged blocks are padded to the max."""
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size == 1:
        return local_state
    nmax = max(counts)
    pad = local_state
    if local_state.shape[0] < nmax:
        pad = torch.zeros((nmax, local_state.shape[1]), dtype=local_state.dtype, device=local_state.device)
        pad[: local_state.shape[0]] = local_state
    out = torch.empty((world_size * nmax, local_state.shape[1]), dtype=local_state.dtype, device=local_state.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    return torch.cat([out[r * nmax: r * nmax + counts[r]] for r in range(world_size)], 0)


class StateGather:
    """The per-step state gather with persistent buffers (no allocation inside the timed loop): every rank contributes
    its [n_r, 13] block, ragged blocks padded to the largest; `gather()` returns the [world, nmax, 13] buffer, rows beyond
    counts[r] of block r are padding. On ROCm the "nccl" backend is RCCL (device buffers); "gloo" stages through the host."""

    def __init__(self, counts, device, backend):
        self.counts = [int(c) for c in counts]
        self.world = len(self.counts)
        self.nmax = max(self.counts)
        self.backend = backend
        self.local = torch.zeros((self.nmax, 13), dtype=torch.float32, device=device)
        self.out = torch.empty((self.world, self.nmax, 13), dtype=torch.float32, device=device if backend == "nccl" else "cpu")

    def gather(self):
        """`self.local` holds this rank's packed state (rows beyond its count stay zero)."""
        if self.world == 1 or not dist.is_initialized():
            self.out[0].copy_(self.local)
            return self.out
        src = self.local if self.backend == "nccl" else self.local.cpu()
        dist.all_gather_into_tensor(self.out.view(-1), src.view(-1))
        return self.out

    def split(self):
        """Per-rank [n_r, 13] views of the last gather."""
        return [self.out[r, : self.counts[r]] for r in range(self.world)]


def pack_state(pos, orn, linvel, angvel):
    return np.concatenate([pos, orn, linvel, angvel], axis=1).astype(np.float32)


# ------------------------------------------------------------------------------------------------ island partitioner
def partition_islands(labels, kind, weights, world_size):
    """Assign every island to a rank, balancing the summed weight ("balancing sum(contact rows)").

    labels  [n] island label per body (the device labels of edynhip_get_derived: lowest body index of the island)
    kind    [n] body kinds; non-dynamic bodies belong to no island and are replicated on every rank
    weights [n] per-body cost (e.g. contact points the body takes part in; 1 = balance body counts)
    Deterministic: islands by descending weight, ties by ascending label, each to the currently lightest rank (ties to
    the lowest rank) - longest-processing-time-first. Returns rank_of_body [n] (-1 = replicated).
    The partitioner itself is the library's (edynhip_partition_islands, edyn_amd/csrc/multi.hip - host code, no GPU needed):
    the single-process multi-GPU world (edynhip_world_*) and this multi-process one partition alike."""
    from . import _capi
    labels = np.ascontiguousarray(labels, np.uint32); kind = np.ascontiguousarray(kind, np.int32)
    weights = np.ascontiguousarray(weights, np.float64)
    rank_of = np.full(len(kind), -1, np.int32)
    rc = _capi.lib().edynhip_partition_islands(len(kind), labels.ctypes.data, kind.ctypes.data, weights.ctypes.data, int(world_size), rank_of.ctypes.data)
    if rc != 0:
        raise ValueError(f"edynhip_partition_islands: {rc}")
    return rank_of


def island_boxes(aabb, labels, kind):
    """(labels [k], boxes [k, 6]) - per island the union of its dynamic bodies' AABBs."""
    kind = np.asarray(kind); labels = np.asarray(labels)
    dyn = kind == KIND_DYNAMIC
    isl, inv = np.unique(labels[dyn], return_inverse=True)
    lo = np.full((len(isl), 3), np.inf); hi = np.full((len(isl), 3), -np.inf)
    if len(isl):
        np.minimum.at(lo, inv, aabb[dyn][:, :3]); np.maximum.at(hi, inv, aabb[dyn][:, 3:])
    return isl, np.concatenate([lo, hi], axis=1).astype(np.float32), inv


def island_boxes_overlap(aabb, labels, kind, rank_of, margin=0.026, any_owner=False):
    """Cross-shard merge detection ("re-partition only when an island merge crosses shards"): island bounding
    boxes (union of the body AABBs, grown by the manifold separation threshold) of islands owned by DIFFERENT ranks that
    overlap. Returns the list of (label_a, label_b) pairs; empty = the partition is still valid. Conservative: overlapping
    island boxes do not yet touch, but no contact between two shards can appear without it. any_owner = also the pairs that
    live on ONE rank: what a new partition has to keep together (two islands about to touch that were co-located by the last
    partition have no manifold between them yet - the partitioner would be free to split them again).
    The sweep is the library's (edynhip_island_boxes_overlap, host code)."""
    from . import _capi
    rank_of = np.asarray(rank_of)
    isl, boxes, inv = island_boxes(np.asarray(aabb, np.float64), labels, kind)
    if len(isl) < 2:
        return []
    dyn = np.asarray(kind) == KIND_DYNAMIC
    owner = np.zeros(len(isl), np.int32); owner[inv] = rank_of[dyn]
    lab = np.ascontiguousarray(isl, np.uint32); boxes = np.ascontiguousarray(boxes, np.float32)
    n = _capi.C.c_uint32(0)
    L = _capi.lib()
    L.edynhip_island_boxes_overlap(len(lab), boxes.ctypes.data, lab.ctypes.data, owner.ctypes.data, float(margin), int(any_owner), None, 0, _capi.C.byref(n))
    pairs = np.zeros((max(n.value, 1), 2), np.uint32)
    rc = L.edynhip_island_boxes_overlap(len(lab), boxes.ctypes.data, lab.ctypes.data, owner.ctypes.data, float(margin), int(any_owner), pairs.ctypes.data, len(pairs), _capi.C.byref(n))
    if rc != 0:
        raise ValueError(f"edynhip_island_boxes_overlap: {rc}")
    return [(int(a), int(b)) for a, b in pairs[:n.value]]


class ShardedWorld:
    """One simulation sharded over the ranks of a process group at island granularity (solver.cpp:408-428: the
    island is the reference's own unit of parallelism). Every rank owns the dynamic bodies of its islands plus a replica of
    all non-dynamic bodies and steps them with its own stepper - NO collective inside a step. Once per step the integrated
    state is gathered (StateGather: RCCL on GPUs, gloo on CPU) so that every rank - and the host registry - sees the whole
    world; the gathered AABBs also tell when islands of different ranks come close (island_boxes_overlap), and then the
    islands are re-partitioned, carrying to their new owner: the contact manifolds (warm-start impulses, colours), the joints'
    applied impulses and tracked angles, the sleeping tags, the collision exclusions and joint definitions of the scene.
    step() runs the approach check itself after every gather (auto_repartition=True) - from the gathered positions and a
    conservative reach per body, with the island labels of the last partition (islands that merged inside a rank since then
    are still owned by one rank; islands that split only make the test more conservative) - and re-partitions when it fires.
    The gather of THIS class goes through a host copy of the state (get_state -> numpy -> torch -> all_gather): it serves the
    registry write-back of a sharded world; bench.py's timed loop packs on the device (edynhip_pack_state_device) and gathers
    device buffers over RCCL.

    make_world(scene) -> a stepper with the World interface (edyn_amd.World on a GPU; the CPU tests pass the checker).
    Body order inside a shard is ascending global index, so canonical pair keys, island labels (lowest index) and the
    colouring keep their relative order: a shard computes exactly what the unsharded world computes for those islands."""

    def __init__(self, scene, make_world, rank, world_size, backend="nccl", device="cuda", labels=None, weights=None, auto_repartition=True):
        self.scene, self.make_world = scene, make_world
        self.auto_repartition = auto_repartition
        self.rank, self.world_size, self.backend, self.device = rank, world_size, backend, device
        self.kind = np.asarray(scene["kind"])
        self.n = len(self.kind)
        if labels is None:
            # the islands of the initial state, from the stepper itself: broadphase pairs (AABBs within the contact margin) and
            # joints connect dynamic bodies - exactly the graph the island manager partitions (island_manager.cpp:117-247)
            probe = make_world(scene)
            if hasattr(probe, "run_stages"):
                probe.run_stages(1 | 2 | 4)                      # EDYNHIP_STAGE_BROADPHASE | NARROWPHASE | ISLANDS
            else:
                for stage in (0, 1, 2):
                    probe.run_stage(stage)
            labels = np.asarray(probe.get_derived()[2]).copy()
            del probe
        if weights is None:
            weights = np.ones(self.n)
        self.repartitions = 0
        # conservative reach of every body around its position (no point of the shape is further away), for the per-step approach check
        sp, st = np.asarray(scene["shape_param"], np.float64), np.asarray(scene["shape_type"])
        self.reach = np.where(st == 1, np.linalg.norm(sp[:, :3], axis=1), np.where(st == 2, sp[:, 0], np.where(st == 4, sp[:, 0] + sp[:, 1],
                              np.where(st == 5, np.hypot(sp[:, 0], sp[:, 1]), 0.0))))
        if (st == 6).any():   # polyhedra: the mesh's diameter bounds the distance from its centroid to any vertex
            diam = []
            for m in scene["meshes"]:
                v = np.asarray(m["vertices"], np.float64)
                diam.append(float(np.linalg.norm(v[:, None, :] - v[None, :, :], axis=2).max()))
            self.reach = np.where(st == 6, np.asarray(diam)[np.clip(sp[:, 0].astype(np.int64), 0, len(diam) - 1)], self.reach)
        com = scene.get("com", scene.get("center_of_mass"))   # scenes carry the offsets as "com" (world.py)
        if com is not None:
            self.reach = self.reach + np.linalg.norm(np.asarray(com, np.float64).reshape(-1, 3), axis=1)
        self.part_labels = np.asarray(labels).copy()
        self._build(partition_islands(labels, self.kind, weights, world_size), scene, manifolds=None)

    # -- shard construction
    def _build(self, rank_of, scene, manifolds, carry=None):
        self.rank_of = rank_of
        mine = (rank_of == self.rank) | (rank_of < 0)
        self.local_ids = np.nonzero(mine)[0].astype(np.int64)            # ascending global indices
        self.to_local = np.full(self.n, -1, np.int64); self.to_local[self.local_ids] = np.arange(len(self.local_ids))
        self.owned = np.nonzero(rank_of == self.rank)[0]                   # what this rank contributes to the gather
        counts = [int((rank_of == r).sum()) for r in range(self.world_size)]
        self.owners = [np.nonzero(rank_of == r)[0] for r in range(self.world_size)]
        local_scene = subset(scene, self.local_ids)
        # joints whose bodies are here and of which this rank owns one (a joint to a replicated static anchor lives on one rank)
        self.local_joints = [g for g, j in enumerate(scene.get("joints") or []) if self.to_local[j[1]] >= 0 and self.to_local[j[2]] >= 0
                             and (rank_of[j[1]] == self.rank or rank_of[j[2]] == self.rank)]
        jl = {g: l for l, g in enumerate(self.local_joints)}
        local_scene["joints"] = [(scene["joints"][g][0], int(self.to_local[scene["joints"][g][1]]), int(self.to_local[scene["joints"][g][2]])) + tuple(scene["joints"][g][3:])
                                 for g in self.local_joints]
        # what apply_figure_settings needs, in local indices: joint parameter blocks / definitions and collision exclusions
        local_scene["hinge_params"] = [(jl[j], p) for j, p in scene.get("hinge_params", []) if j in jl]
        local_scene["joint_defs"] = [(jl[j], fa, fb, p) for j, fa, fb, p in scene.get("joint_defs", []) if j in jl]
        local_scene["exclusions"] = [(int(self.to_local[a]), int(self.to_local[b])) for a, b in scene.get("exclusions", [])
                                     if self.to_local[a] >= 0 and self.to_local[b] >= 0 and (rank_of[a] == self.rank or rank_of[b] == self.rank)]
        self.world = self.make_world(local_scene)
        from .scenes import apply_figure_settings
        apply_figure_settings(self.world, local_scene)
        if manifolds is not None and len(manifolds):
            a, b = manifolds["body"][:, 0], manifolds["body"][:, 1]
            keep = ((rank_of[a] == self.rank) | (rank_of[b] == self.rank))
            rec = manifolds[keep].copy()
            rec["body"] = self.to_local[rec["body"]]
            self.world.set_manifolds(rec)                                  # relative order = canonical order (monotone index map)
        if carry is not None:
            if self.local_joints and carry.get("joint_imp") is not None and hasattr(self.world, "set_joint_warm_start"):
                self.world.set_joint_warm_start(carry["joint_imp"][self.local_joints], carry["joint_angle"][self.local_joints])
            if carry.get("asleep") is not None and carry["asleep"].any() and hasattr(self.world, "set_asleep"):
                self.world.set_asleep(carry["asleep"][self.local_ids])
        self.gather = StateGather(counts, self.device, self.backend)
        self.state = np.zeros((self.n, 13), np.float32)

    # -- stepping
    def step(self, n=1):
        for _ in range(n):
            self.world.step_simulation(1) if hasattr(self.world, "step_simulation") else self.world.step(1)
            self._gather_state()
            if self.auto_repartition and self.world_size > 1 and self._islands_close():
                self.maybe_repartition()

    def _islands_close(self):
        """Conservative, collective-free approach test on the gathered state (identical on every rank, so every rank takes the
        same decision): body boxes = position +- reach, islands = those of the last partition."""
        pos = self.state[:, 0:3].astype(np.float64)
        r = self.reach[:, None]
        aabb = np.concatenate([pos - r, pos + r], axis=1)
        return bool(island_boxes_overlap(aabb, self.part_labels, self.kind, self.rank_of))

    def _gather_state(self):
        pos, orn, lv, av = self.world.get_state()
        loc = self.to_local[self.owned]
        local = pack_state(pos[loc], orn[loc], lv[loc], av[loc])
        self.gather.local.zero_()
        if len(local):
            self.gather.local[: len(local)].copy_(torch.from_numpy(local))
        self.gather.gather()
        blocks = self.gather.split()
        for r in range(self.world_size):
            if len(self.owners[r]):
                self.state[self.owners[r]] = blocks[r].cpu().numpy()
        rep = np.nonzero(self.rank_of < 0)[0]                              # replicated bodies: every rank has the same values
        if len(rep):
            l = self.to_local[rep]
            self.state[rep] = pack_state(pos[l], orn[l], lv[l], av[l])

    def get_state(self):
        """The whole world's (pos, orn, linvel, angvel), identical on every rank."""
        s = self.state
        return s[:, 0:3].copy(), s[:, 3:7].copy(), s[:, 7:10].copy(), s[:, 10:13].copy()

    # -- global views needed for (re)partitioning: rare, so they travel as Python objects over the group
    def _all_gather_object(self, obj):
        if self.world_size == 1 or not dist.is_initialized():
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj)
        return out

    def global_islands(self):
        """(labels [n], aabb [n,6], weights [n]) of the whole world from every rank's device labels."""
        aabb_l, _, isl_l = self.world.get_derived()[:3]
        m = self.world.get_manifolds()
        w_l = np.zeros(len(self.local_ids))
        if len(m):
            np.add.at(w_l, m["body"][:, 0], m["num_points"]); np.add.at(w_l, m["body"][:, 1], m["num_points"])
        loc = self.to_local[self.owned]
        parts = self._all_gather_object((self.owned, self.local_ids[isl_l[loc]], aabb_l[loc], w_l[loc]))
        labels = np.arange(self.n); aabb = np.zeros((self.n, 6), np.float32); weights = np.ones(self.n)
        for ids, lab, bb, w in parts:
            labels[ids] = lab; aabb[ids] = bb; weights[ids] = 1 + w
        return labels, aabb, weights

    def maybe_repartition(self, force=False):
        """Re-partition when islands owned by different ranks approach each other (or on request). The contact manifolds
        move with their islands. Returns True when the shards were rebuilt."""
        labels, aabb, weights = self.global_islands()
        close = island_boxes_overlap(aabb, labels, self.kind, self.rank_of)
        if not close and not force:
            return False
        # islands whose boxes overlap must end up on one rank: weld them for the partitioner - all of them, also the pairs an earlier
        # partition already co-located (they may still be separate islands)
        close = island_boxes_overlap(aabb, labels, self.kind, self.rank_of, any_owner=True)
        parent = {}
        def find(x):
            while parent.get(x, x) != x:
                x = parent[x]
            return x
        for a, b in close:
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
        welded = np.array([find(int(l)) for l in labels])
        rank_of = partition_islands(welded, self.kind, weights, self.world_size)
        m = self.world.get_manifolds()
        rec = m.copy()
        if len(rec):
            rec["body"] = self.local_ids[rec["body"]]
            a, b = rec["body"][:, 0], rec["body"][:, 1]
            rec = rec[(self.rank_of[a] == self.rank) | ((self.rank_of[a] < 0) & (self.rank_of[b] == self.rank))]   # each manifold once
        all_m = [x for x in self._all_gather_object(rec) if len(x)]
        manifolds = np.concatenate(all_m) if all_m else rec[:0]
        if len(manifolds):   # canonical order: ascending (owner << 32 | other), owner = the dynamic body, the higher index of two
            a, b = manifolds["body"][:, 0].astype(np.uint64), manifolds["body"][:, 1].astype(np.uint64)
            da, db = self.kind[manifolds["body"][:, 0]] == KIND_DYNAMIC, self.kind[manifolds["body"][:, 1]] == KIND_DYNAMIC
            owner = np.where(da & db, np.maximum(a, b), np.where(da, a, b)); other = np.where(owner == a, b, a)
            manifolds = manifolds[np.argsort((owner << np.uint64(32)) | other, kind="stable")]
        # joints (applied impulses of all 24 slots + tracked angles) and sleeping tags travel with their islands
        carry = {"joint_imp": None, "joint_angle": None, "asleep": None}
        nj = len(self.scene.get("joints") or [])
        if nj and hasattr(self.world, "get_joint_impulses24"):
            imp = self.world.get_joint_impulses24() if self.local_joints else np.zeros((0, 24), np.float32)
            ang = self.world.get_joint_impulses()[:, 9] if self.local_joints else np.zeros(0, np.float32)
            carry["joint_imp"] = np.zeros((nj, 24), np.float32); carry["joint_angle"] = np.zeros(nj, np.float32)
            for ids, i24, a in self._all_gather_object((np.asarray(self.local_joints, np.int64), imp, ang)):
                if len(ids):
                    carry["joint_imp"][ids] = i24; carry["joint_angle"][ids] = a
        if hasattr(self.world, "get_asleep"):
            asl = np.asarray(self.world.get_asleep(), bool)[self.to_local[self.owned]]
            carry["asleep"] = np.zeros(self.n, bool)
            for ids, a in self._all_gather_object((self.owned, asl)):
                carry["asleep"][ids] = a
        scene = dict(self.scene)
        pos, orn, lv, av = self.get_state()
        scene["pos"], scene["orn"], scene["linvel"], scene["angvel"] = pos, orn, lv, av
        self.part_labels = welded
        self._build(rank_of, scene, manifolds, carry)
        self.repartitions += 1
        self._gather_state_without_step()
        return True

    def _gather_state_without_step(self):
        self._gather_state()
