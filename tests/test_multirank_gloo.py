"""World-size-2 test of the N>1 path on CPU (gloo): islands are sharded across ranks with NO data-path
collective, every rank steps its own block, and the integrated state is all-gathered - the result must equal
the un-sharded world bit for bit. The CPU oracle stands in for the stepper here (no GPU in this container);
the sharding / gather logic (edyn_amd.parallel, edyn_amd.scenes) is exactly what bench.py runs over RCCL."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, steps, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from edyn_amd import scenes
    from edyn_amd.parallel import shard_range, gather_state, pack_state
    from oracle import binding as ob
    total_sites = 6
    first, count = shard_range(total_sites, rank, world_size)
    scene = scenes.mini_piles(3, 2, first_site=first, num_sites=count)
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
    w.add_bodies(scene)
    counts = [64 * shard_range(total_sites, r, world_size)[1] for r in range(world_size)]
    gathered = None
    for _ in range(steps):
        w.step(1)
        pos, orn, lv, av = w.get_state()
        local = torch.from_numpy(pack_state(pos[1:], orn[1:], lv[1:], av[1:]))   # dynamic bodies only (the plane is replicated)
        gathered = gather_state(local, counts)
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_islands_match_single_world(tmp_path):
    from edyn_amd import scenes
    from edyn_amd.parallel import pack_state
    from oracle import binding as ob
    steps = 8
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, steps, out), nprocs=2, join=True)
    got = np.load(out)
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
    w.add_bodies(scenes.mini_piles(3, 2))
    w.step(steps)
    pos, orn, lv, av = w.get_state()
    ref = pack_state(pos[1:], orn[1:], lv[1:], av[1:])
    assert got.shape == ref.shape == (6 * 64, 13)
    assert np.array_equal(got, ref)
    assert w.get_stats()["num_islands"] == 6


# ------------------------------------------------------------------ the product's ShardedWorld (edyn_amd.parallel)
def _bridge_scene(along="x"):
    """Six mini-piles (six islands) plus a sphere that rolls from the first site into the second: its island meets an
    island that may live on another rank, which forces a re-partition with the contact manifolds carried along.
    along = "z": the sphere rolls from site 0 into site 3 instead (the row behind it) - the library's multi-device world places islands
    along a space-filling curve, whose cuts run between the two rows of sites for 2 and 3 shards (tests/test_multirank_gpu.py)."""
    from edyn_amd import scenes
    sc = scenes.mini_piles(3, 2)
    n = len(sc["kind"])
    ext = scenes._empty(n + 1)
    for k, v in sc.items():
        if k != "joints":
            ext[k][:n] = v
    ext["kind"][n] = scenes.KIND_DYNAMIC
    if along == "x":
        ext["pos"][n] = (-3.4, 0.5, -4.0)            # between site 0 (x = -8) and site 1 (x = 0), rolling towards +x
        ext["linvel"][n] = (4.0, 0, 0)
    else:
        ext["pos"][n] = (-7.7, 0.5, -0.4)            # between site 0 (z = -4) and site 3 (z = +4), rolling towards +z
        ext["linvel"][n] = (0, 0, 4.0)
    ext["shape_type"][n] = scenes.SHAPE_SPHERE; ext["shape_param"][n] = (0.5, 0, 0, 0)
    return ext


def _poly_bridge_scene(along="x"):
    """The bridge scene with every second box a convex polyhedron (meshes of tests/meshes.py; their ids are positions in the scene's
    mesh list, which every shard keeps whole) and some cylinders: the shards create the meshes in their own worlds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import meshes
    from edyn_amd import scenes
    lib, _ = meshes.registered()
    sc = _bridge_scene(along)
    for i in range(1, len(sc["kind"]) - 1):
        if i % 2 == 0:
            sc["shape_type"][i] = scenes.SHAPE_POLYHEDRON; sc["shape_param"][i] = (float((i // 2) % 2 * 4), 0, 0, 0)   # the unit cube or the hexagonal prism
        elif i % 7 == 1:
            sc["shape_type"][i] = scenes.SHAPE_CYLINDER; sc["shape_param"][i] = (0.5, 0.5, 1.0, 0)
    sc["meshes"] = lib
    return sc


def _jointed_bridge_scene(along="x"):
    """The bridge scene plus two pendulum chains (point + hinge joints, one hinge with angle limits, a bump stop and friction
    torque: tracked angle and four optional row slots) hanging from static anchors, and an excluded pair of overlapping spheres:
    what a re-partition has to carry besides the manifolds."""
    from edyn_amd import scenes
    chains = scenes.c5_chains(2, 4)
    chains["pos"][:, 0] += 30.0; chains["pos"][:, 2] += 6.0
    # every hinge: limits (+-0.6 rad, restitution 0.3), a bump stop, friction torque and a soft spring
    chains["hinge_params"] = [(j, [-0.6, 0.6, 0.3, 0.2, 20.0, 0.05, 0.0, 0.1, 2.0, 0.01]) for j, t in enumerate(chains["joints"]) if t[0] == scenes.JOINT_HINGE]
    sc = scenes.merge(_bridge_scene(along), chains)
    n = len(sc["kind"])
    two = scenes._empty(2)
    two["pos"][:] = [(40.0, 0.5, 0.0), (40.3, 0.5, 0.0)]        # overlapping spheres that must keep ignoring each other
    two["shape_type"][:] = scenes.SHAPE_SPHERE; two["shape_param"][:, 0] = 0.5
    two["exclusions"] = np.array([[0, 1]], np.uint32)
    sc = scenes.merge(sc, two)
    assert len(sc["kind"]) == n + 2 and len(sc["exclusions"]) == 1 and len(sc["hinge_params"]) >= 2
    return sc


def _make_world_fn(use_gpu):
    from edyn_amd import scenes
    if use_gpu:
        import edyn_amd
        def make_world(sc):
            w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); w.set_scene(sc); return w
    else:
        from oracle import binding as ob
        def make_world(sc):
            w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); w.add_bodies(sc); return w
    return make_world


def _jointed_worker(rank, world_size, port, steps, out_path, use_gpu):
    """ShardedWorld.step() alone: the approach check runs inside it (auto_repartition)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from edyn_amd.parallel import ShardedWorld
    sw = ShardedWorld(_jointed_bridge_scene(), _make_world_fn(use_gpu), rank, world_size, backend="gloo", device="cpu")
    states = []
    for k in range(steps):
        sw.step(1)
        states.append(np.concatenate(sw.get_state(), axis=1))
        if k in (30, 60):
            sw.maybe_repartition(force=True)   # every shard is rebuilt: the chains swing with warm-started joints mid-motion
    if rank == 0:
        np.savez(out_path, states=np.stack(states), reparts=sw.repartitions)
    dist.barrier()
    dist.destroy_process_group()


def _sharded_worker(rank, world_size, port, steps, out_path, use_gpu, poly=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from edyn_amd.parallel import ShardedWorld
    scene = _poly_bridge_scene() if poly else _bridge_scene()
    if use_gpu:
        import edyn_amd
        def make_world(sc):
            w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); w.set_scene(sc); return w
    else:
        from oracle import binding as ob
        def make_world(sc):
            w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); w.add_bodies(sc); return w
    sw = ShardedWorld(scene, make_world, rank, world_size, backend="gloo", device="cpu")
    states = []
    for _ in range(steps):
        sw.step(1)     # the approach check (and the re-partition it asks for) happens inside
        states.append(np.concatenate(sw.get_state(), axis=1))
    reparts = sw.repartitions
    owners = np.bincount(sw.rank_of[sw.rank_of >= 0], minlength=world_size)
    if rank == 0:
        np.savez(out_path, states=np.stack(states), reparts=reparts, owners=owners)
    dist.barrier()
    dist.destroy_process_group()


def _unsharded_states(steps, use_gpu, poly=False):
    scene = _poly_bridge_scene() if poly else _bridge_scene()
    if use_gpu:
        import edyn_amd
        w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); w.set_scene(scene)
        step = lambda: w.step_simulation(1)
    else:
        from oracle import binding as ob
        w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); w.add_bodies(scene)
        step = lambda: w.step(1)
    out = []
    for _ in range(steps):
        step()
        out.append(np.concatenate(w.get_state(), axis=1))
    return np.stack(out)


def test_partitioner_balances_islands_and_detects_cross_shard_contact():
    from edyn_amd import scenes
    from edyn_amd.parallel import partition_islands, island_boxes_overlap
    kind = np.array([scenes.KIND_STATIC] + [scenes.KIND_DYNAMIC] * 10)
    labels = np.array([0, 1, 1, 1, 1, 5, 5, 7, 8, 8, 8])
    w = np.ones(11)
    r = partition_islands(labels, kind, w, 2)
    assert r[0] == -1                                  # static body: replicated
    assert len(set(r[1:5])) == 1 and len(set(r[5:7])) == 1 and len(set(r[8:])) == 1   # islands stay whole
    loads = np.bincount(r[r >= 0], minlength=2)
    assert abs(int(loads[0]) - int(loads[1])) <= 1     # 4+1 vs 3+2
    assert np.array_equal(r, partition_islands(labels, kind, w, 2))   # deterministic
    aabb = np.zeros((11, 6), np.float32)
    for i in range(1, 11):
        aabb[i] = (labels[i] * 3.0, 0, 0, labels[i] * 3.0 + 1, 1, 1)
    assert island_boxes_overlap(aabb, labels, kind, r) == []
    aabb[7] = (8 * 3.0 - 0.5, 0, 0, 8 * 3.0 + 0.2, 1, 1)          # island 7 reaches island 8
    hits = island_boxes_overlap(aabb, labels, kind, np.where(labels == 7, 0, np.where(labels == 8, 1, r)))
    assert hits == [(7, 8)]


def test_sharded_world_repartitions_and_matches_the_unsharded_world(tmp_path):
    """edyn_amd.parallel.ShardedWorld over 2 gloo ranks (the checker stands in for the stepper on this GPU-less box): islands
    balanced over the ranks, per-step state gather, a cross-shard approach detected from the gathered AABBs, the islands
    re-partitioned with their manifolds - and the whole trajectory equals the unsharded world bit for bit."""
    steps = 90
    out = str(tmp_path / "sharded.npz")
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_sharded_worker, args=(2, port, steps, out, False), nprocs=2, join=True)
    got = np.load(out)
    ref = _unsharded_states(steps, False)
    assert got["states"].shape == ref.shape
    assert int(got["reparts"]) >= 1, "the rolling sphere must have triggered a re-partition"
    assert got["owners"].min() > 0
    assert np.array_equal(got["states"], ref)


def test_sharded_world_with_polyhedra_matches_the_unsharded_world(tmp_path):
    """Polyhedra and cylinders in the sharded scene: every shard holds the scene's mesh list (ids are positions in it), the approach check
    knows their reach, and the re-partitioned trajectory equals the unsharded world bit for bit."""
    steps = 90
    out = str(tmp_path / "sharded_poly.npz")
    port = 29500 + (os.getpid() % 2000) + 23
    mp.spawn(_sharded_worker, args=(2, port, steps, out, False, True), nprocs=2, join=True)
    got = np.load(out)
    ref = _unsharded_states(steps, False, True)
    assert int(got["reparts"]) >= 1 and got["owners"].min() > 0
    assert np.array_equal(got["states"], ref)


def test_sharded_world_carries_joints_and_exclusions_across_a_repartition(tmp_path):
    """a re-partition rebuilds every shard; besides the manifolds it must carry the joints' applied impulses and
    tracked angles (warm start, angle-limit rows) and apply the scene's joint parameters and collision exclusions in the new
    local indices - and step() must notice the approaching islands by itself. Two pendulum chains with limited, sprung hinges
    and an excluded pair ride along with the bridge scene; the trajectory equals the unsharded world bit for bit."""
    from edyn_amd import scenes
    from oracle import binding as ob
    steps = 90
    out = str(tmp_path / "jointed.npz")
    port = 29500 + (os.getpid() % 2000) + 13
    mp.spawn(_jointed_worker, args=(2, port, steps, out, False), nprocs=2, join=True)
    got = np.load(out)
    sc = _jointed_bridge_scene()
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); w.add_bodies(sc); scenes.apply_figure_settings(w, sc)
    ref = []
    for _ in range(steps):
        w.step(1)
        ref.append(np.concatenate(w.get_state(), axis=1))
    assert int(got["reparts"]) >= 2
    assert np.array_equal(got["states"], np.stack(ref))


def test_sharded_world_over_eight_gloo_ranks(tmp_path):
    """world_size 8: the bridge scene has seven islands - one rank owns nothing but the replicated plane, the
    partition (the library's edynhip_partition_islands) is identical on every rank, the approach check and the re-partition run on
    all eight - and the trajectory equals the unsharded world bit for bit."""
    steps = 60
    out = str(tmp_path / "sharded8.npz")
    port = 29500 + (os.getpid() % 2000) + 31
    mp.spawn(_sharded_worker, args=(8, port, steps, out, False), nprocs=8, join=True)
    got = np.load(out)
    ref = _unsharded_states(steps, False)
    assert int(got["reparts"]) >= 1
    assert got["owners"].sum() == ref.shape[1] - 1 and (got["owners"] == 0).sum() >= 1    # more ranks than islands: some own nothing
    assert np.array_equal(got["states"], ref)
