"""The sharded N>1 path with the HIP stepper on every rank: two ranks share cuda:0 (the 8-GPU RCCL run is the driver's),
islands partitioned by edyn_amd.parallel.ShardedWorld, state gathered every step, one re-partition when an island crosses
shards - the trajectory must equal the unsharded GPU world bit for bit."""
import os
import numpy as np
import pytest
import torch.multiprocessing as mp

from test_multirank_gloo import _sharded_worker, _unsharded_states

pytestmark = pytest.mark.gpu


def test_sharded_hip_worlds_match_the_unsharded_gpu_world(tmp_path):
    steps = 90
    out = str(tmp_path / "sharded_gpu.npz")
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_sharded_worker, args=(2, port, steps, out, True), nprocs=2, join=True)
    got = np.load(out)
    ref = _unsharded_states(steps, True)
    assert int(got["reparts"]) >= 1
    assert np.array_equal(got["states"], ref)


# ---- the multi-GPU world BEHIND the C-ABI (edynhip_world_*, edyn_amd/csrc/multi.hip) ----------------------
# Two shards on the one physical GPU of the test box: one process, one host thread per shard, island boxes reduced on the device,
# re-partitions inside edynhip_world_step. The trajectory must equal ONE context stepping the whole scene, bit for bit.
def _single_world_states(scene, steps):
    import edyn_amd
    from edyn_amd import scenes
    w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); w.set_scene(scene); scenes.apply_figure_settings(w, scene)
    out = []
    for _ in range(steps):
        w.step_simulation(1)
        out.append(np.concatenate(w.get_state(), axis=1))
    return np.stack(out), w


@pytest.mark.parametrize("shards", [2, 3])
def test_multi_world_matches_one_context_through_an_approach_triggered_repartition(shards):
    import edyn_amd
    from test_multirank_gloo import _bridge_scene
    scene = _bridge_scene(along="z")   # the sphere crosses the world's cut (islands are placed along a space-filling curve)
    steps = 90
    ref, single = _single_world_states(scene, steps)
    mw = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0] * shards)
    mw.set_scene(scene)
    part0 = mw.get_partition()
    assert part0[0] == -1 and set(part0[1:]) == set(range(shards))
    for k in range(steps):
        mw.step_simulation(1)
        assert np.array_equal(np.concatenate(mw.get_state(), axis=1), ref[k]), k
    st = mw.get_stats()
    assert st["repartitions"] >= 1, "the rolling sphere must have forced a re-partition"
    assert st["approach_checks"] < steps, "a check every step means the growth budget is not working"
    assert sum(st["bodies_per_shard"]) == len(scene["kind"])
    gm, sm = mw.get_manifolds(), single.get_manifolds()
    assert len(gm) == len(sm) and gm.tobytes() == sm.tobytes()     # manifolds incl. impulses and colours, global indices, canonical order


def test_multi_world_carries_joints_exclusions_and_a_forced_repartition():
    """Several shards (4, else 3 or 5), the jointed bridge scene: (1) the APPROACH-TRIGGERED sticky path with joints in the world - the sphere rolls into a
    site that lives on another shard (the variant of the scene in which that is so is picked from the initial partition, so the
    crossing does not depend on where the chains happen to put the cuts: ADVICE r05), the two islands end up on one shard and every
    body of a shard that was not involved stays where it was; (2) two forced full re-partitions - every shard rebuilt: the chains
    swing on with warm-started joints and tracked angles. Bit-equal to one context at every step."""
    import edyn_amd
    from test_multirank_gloo import _jointed_bridge_scene
    sphere, per_site = 1 + 6 * 64, 64                      # _bridge_scene: six 4x4x4 sites, then the sphere
    target_site = {"z": 3, "x": 1}                          # the site the sphere rolls into
    picked = None
    for shards in (4, 3, 5):                                # (islands are placed along a space-filling curve: where its cuts fall depends on the count)
        for along in ("z", "x"):
            scene = _jointed_bridge_scene(along=along)
            probe = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0] * shards)
            probe.set_scene(scene)
            part = probe.get_partition()
            first_of_target = 1 + target_site[along] * per_site
            del probe
            if part[sphere] != part[first_of_target] and picked is None:
                picked = (shards, along, scene, part, first_of_target)
    assert picked is not None, "in no variant does the sphere cross a shard boundary"
    shards, along, scene, part0, first_of_target = picked
    steps = 90
    ref, _ = _single_world_states(scene, steps)
    mw = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0] * shards)
    mw.set_scene(scene)
    assert np.array_equal(mw.get_partition(), part0)
    involved = {int(part0[sphere]), int(part0[first_of_target])}
    sticky_seen = False
    for k in range(steps):
        before = mw.get_stats()["repartitions"]
        mw.step_simulation(1)
        assert np.array_equal(np.concatenate(mw.get_state(), axis=1), ref[k]), k
        if not sticky_seen and mw.get_stats()["repartitions"] > before:   # the approach-triggered one (no forced one has happened yet)
            sticky_seen = True
            assert k < 55
            part1 = mw.get_partition()
            assert part1[sphere] == part1[first_of_target], "the sphere's island and the site it rolls into share a shard now"
            bystanders = [i for i in range(len(part0)) if part0[i] >= 0 and int(part0[i]) not in involved]
            assert bystanders and all(part1[i] == part0[i] for i in bystanders), "a sticky re-partition moves nothing but the meeting islands"
        if k in (55, 75):   # (after the sphere has arrived: 14 steps along x, 35 along z)
            mw.repartition()
    assert sticky_seen and mw.get_stats()["repartitions"] >= 3   # one approach-triggered + the two forced ones


def test_multi_world_with_polyhedra_and_cylinders():
    import edyn_amd
    from test_multirank_gloo import _poly_bridge_scene
    scene = _poly_bridge_scene(along="z")
    steps = 90
    ref, _ = _single_world_states(scene, steps)
    mw = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0, 0])
    mw.set_scene(scene)
    mw.step_simulation(steps)
    assert np.array_equal(np.concatenate(mw.get_state(), axis=1), ref[-1])
    assert mw.get_stats()["repartitions"] >= 1


def test_island_boxes_reduced_on_the_device():
    """edynhip_get_island_boxes: per island the union of its bodies' AABBs - against numpy on the read-back AABBs and labels."""
    import ctypes as C
    import edyn_amd
    from edyn_amd import scenes, _capi
    from edyn_amd.parallel import island_boxes
    scene = scenes.mini_piles(3, 3)
    w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); w.set_scene(scene)
    w.step_simulation(40)
    aabb, _, labels = w.get_derived()[:3]
    isl, boxes, _ = island_boxes(np.asarray(aabb, np.float64), labels, scene["kind"])
    n = C.c_uint32(0)
    lab = np.zeros(len(scene["kind"]), np.uint32); bx = np.zeros((len(scene["kind"]), 6), np.float32)
    assert w._L.edynhip_get_island_boxes(w._h, lab.ctypes.data, bx.ctypes.data, len(lab), C.byref(n)) == 0
    order = np.argsort(lab[:n.value])
    assert np.array_equal(lab[:n.value][order], isl.astype(np.uint32))
    assert np.array_equal(bx[:n.value][order], boxes)


def test_a_process_that_stepped_a_multi_world_exits_cleanly():
    """Several shards of a world on ONE device step from one host thread each; their cooperative launches used to enter the runtime
    concurrently and the process then died in the runtime's exit handler (SIGSEGV in hsa_shut_down after the script's last line).
    solver.hip launch_resident serialises them per device: the process must end with status 0."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("multi2", "multi4_gloo"):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "exit_probe.py"), mode], cwd=root, capture_output=True, text=True, timeout=300)
        assert "end of script" in r.stdout, r.stdout + r.stderr
        assert r.returncode == 0, f"{mode}: exit status {r.returncode}\n{r.stdout[-500:]}\n{r.stderr[-500:]}"
