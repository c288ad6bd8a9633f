"""The restatement (oracle/liboracle.so) against the REAL reference engine.

oracle/_ref/libedynref.so is the reference's own source — every translation unit of its simulation path, compiled where
it lies under /root/reference against oracle/entt_min (a from-scratch implementation of the EnTT subset it uses) — behind
the small drivers oracle/ref_world.cpp (edyn::attach / make_rigidbody / make_constraint / step_simulation) and
oracle/ref_xcheck.cpp (leaf functions). These tests pin the oracle to it:

  * leaves, bit for bit on random inputs: the five collide() routines in all eight ordered shape combinations (200k pairs
    each), dynamic_tree (a 20k-operation create/move/destroy/query script, visit order included), friction rows;
  * whole steps, bit for bit over hundreds of steps: positions, orientations, velocities, pair sets, manifolds (points in
    list order, lifetimes, warm-start impulses), sleeping flags — on box piles, mixed box/sphere piles, pyramids, many-island
    scenes, collapsing towers, spinning drops and hinge/point chains.

Two switches make the bit-for-bit comparison possible, both test-only and both off for the GPU's specification:
  ORDER_EXTERNAL — the oracle visits an island's constraints in the order the reference did for that step (island.edges
                   iteration order, an artefact of EnTT pool history; exported by ref_world.cpp after the step);
  libm trig      — integrate() calls the C library's sinf/cosf like the reference (the default is the correctly rounded
                   value, which differs from this image's glibc by at most 2 ulp in <2 % of calls).
Everything else — broadphase, narrowphase, manifold persistence, row preparation, row arithmetic, integration, position
solver, sleeping — is therefore identical to the reference's, operation for operation.
Skipped where oracle/_ref was never built (it needs /root/reference at build time; the built library travels)."""
import os
import numpy as np
import pytest

from edyn_amd import scenes
from oracle import binding as ob
from pairgen import pair_batch
import meshes

pytestmark = pytest.mark.skipif(ob.ref() is None, reason="oracle/_ref/libedynref.so not built (needs /root/reference at build time)")


@pytest.fixture(autouse=True)
def _libm_trig():
    ob.set_libm_trig(True)
    yield
    ob.set_libm_trig(False)


# ------------------------------------------------------------------------------------------------ leaves
@pytest.mark.parametrize("tA,tB", [
    (scenes.SHAPE_BOX, scenes.SHAPE_BOX), (scenes.SHAPE_SPHERE, scenes.SHAPE_BOX), (scenes.SHAPE_BOX, scenes.SHAPE_SPHERE),
    (scenes.SHAPE_SPHERE, scenes.SHAPE_SPHERE), (scenes.SHAPE_BOX, scenes.SHAPE_PLANE), (scenes.SHAPE_PLANE, scenes.SHAPE_BOX),
    (scenes.SHAPE_SPHERE, scenes.SHAPE_PLANE), (scenes.SHAPE_PLANE, scenes.SHAPE_SPHERE),
    (scenes.SHAPE_CAPSULE, scenes.SHAPE_CAPSULE), (scenes.SHAPE_CAPSULE, scenes.SHAPE_BOX), (scenes.SHAPE_BOX, scenes.SHAPE_CAPSULE),
    (scenes.SHAPE_CAPSULE, scenes.SHAPE_SPHERE), (scenes.SHAPE_SPHERE, scenes.SHAPE_CAPSULE),
    (scenes.SHAPE_CAPSULE, scenes.SHAPE_PLANE), (scenes.SHAPE_PLANE, scenes.SHAPE_CAPSULE),
    (scenes.SHAPE_CYLINDER, scenes.SHAPE_PLANE), (scenes.SHAPE_PLANE, scenes.SHAPE_CYLINDER),
    (scenes.SHAPE_CYLINDER, scenes.SHAPE_SPHERE), (scenes.SHAPE_SPHERE, scenes.SHAPE_CYLINDER),
    (scenes.SHAPE_CYLINDER, scenes.SHAPE_CYLINDER), (scenes.SHAPE_CYLINDER, scenes.SHAPE_BOX), (scenes.SHAPE_BOX, scenes.SHAPE_CYLINDER),
    (scenes.SHAPE_CAPSULE, scenes.SHAPE_CYLINDER), (scenes.SHAPE_CYLINDER, scenes.SHAPE_CAPSULE)])
def test_collide_matches_the_reference_routines(tA, tB):
    """collide_box_box.cpp:14-266, collide_box_plane.cpp, collide_sphere_{sphere,plane,box}.cpp, swap_collide
    (collide.hpp:369-374), collision_result.cpp: counts, pivots, normals, distances, attachments — the same 200k random
    pairs the GPU test feeds the device routines (tests/pairgen.py, same seeds)."""
    rng = np.random.default_rng(1000 + 10 * tA + tB)
    st, sp, pos, orn = pair_batch(rng, 200_000, tA, tB)
    op, oc = ob.collide_batch(st, sp, pos, orn, threshold=0.02)
    rp, rc = ob.ref_collide_batch(st, sp, pos, orn, threshold=0.02)
    assert np.array_equal(oc, rc)
    assert (rc > 0).mean() > 0.15
    if tA == scenes.SHAPE_BOX and tB == scenes.SHAPE_BOX:
        assert set(np.unique(rc)) == {0, 1, 2, 3, 4}
    assert np.array_equal(op.view(np.uint32), rp.view(np.uint32))


def test_convex_mesh_initialisation_matches_the_reference():
    """convex_mesh::initialize (convex_mesh.cpp:10-230: centroid shift, face normals, unique edges with their two faces, vertex
    adjacency, relevant faces / edges) and moment_of_inertia_polyhedron, every derived array bit for bit."""
    lib, _ = meshes.registered()
    for k in range(len(lib)):
        for f in ob.MESH_FIELDS:
            x, y = ob.mesh_get(k, f, False), ob.mesh_get(k, f, True)
            assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), (k, f)
        assert np.array_equal(ob.mesh_inertia(k, 2.5).view(np.uint32), ob.mesh_inertia(k, 2.5, True).view(np.uint32))
    assert len(ob.mesh_get(9, "relevant_edges")) < len(ob.mesh_get(9, "edges")) // 2   # the 12-gon prism: parallel edges folded


@pytest.mark.parametrize("other", [scenes.SHAPE_PLANE, scenes.SHAPE_SPHERE, scenes.SHAPE_POLYHEDRON, scenes.SHAPE_BOX, scenes.SHAPE_CAPSULE,
                                   scenes.SHAPE_CYLINDER])
def test_polyhedron_collide_matches_the_reference_routines(other):
    """collide_polyhedron_{plane,sphere,polyhedron,box,capsule,cylinder}.cpp in both argument orders over ten convex meshes (boxes,
    tetrahedron, octahedron, prisms with 5/6/12-gon faces, a wedge, two random hulls): hill-climbing support projection, support
    polygons and their quickhull, Minkowski-face edge pruning, segment intersections - bit for bit.
    Pairs of polyhedra with only parallel edges are left out of the comparison: the reference reads uninitialised variables there
    (collide_polyhedron_polyhedron.cpp:98-100,150) - the oracle reports which pairs took that path."""
    _, rad = meshes.registered()
    P = scenes.SHAPE_POLYHEDRON
    for tA, tB in ((P, other), (other, P)) if other != P else ((P, P),):
        rng = np.random.default_rng(1000 + 10 * tA + tB)
        n = 100_000
        st, sp, pos, orn = pair_batch(rng, n, tA, tB, rad)
        op, oc = ob.collide_batch(st, sp, pos, orn, threshold=0.02)
        ok = ob.last_batch_flags(n) == 0
        assert (other == P and 0 < (~ok).sum() < n // 50) or ok.all()
        rp, rc = ob.ref_collide_batch(st[ok], sp[ok], pos[ok], orn[ok], threshold=0.02)
        assert np.array_equal(oc[ok], rc)
        assert (rc > 0).mean() > 0.5 and (other in (scenes.SHAPE_SPHERE, scenes.SHAPE_CAPSULE) or (rc == 4).sum() > 100)
        assert np.array_equal(op[ok].view(np.uint32), rp.view(np.uint32))


def test_dynamic_tree_matches_the_reference_tree():
    """dynamic_tree.cpp:41-339 + query_tree.hpp:9-42: same leaves visited in the same order by every query (i.e. the same
    tree shape after every insert / remove / rotation), same move() results."""
    rng = np.random.default_rng(7)
    ops, boxes, alive, cur, nxt = [], [], [], {}, 0

    def rbox(scale=10):
        c = rng.uniform(-scale, scale, 3); h = rng.uniform(0.1, 1.0, 3)
        return np.concatenate([c - h, c + h])

    for _ in range(20000):
        r = rng.random()
        if r < 0.35 or len(alive) < 5:
            b = rbox(); ops.append((0, nxt)); boxes.append(b); cur[nxt] = b; alive.append(nxt); nxt += 1
        elif r < 0.65:
            h = alive[rng.integers(len(alive))]; b = cur[h].copy()
            d = rng.normal(size=3) * rng.choice([0.02, 0.3]); b[:3] += d; b[3:] += d; cur[h] = b
            ops.append((1, h)); boxes.append(b)
        elif r < 0.75:
            h = alive.pop(rng.integers(len(alive))); ops.append((2, h)); boxes.append(np.zeros(6))
        else:
            ops.append((3, 0)); boxes.append(rbox())
    h_orc, m_orc = ob.tree_run(ops, boxes, real=False)
    h_ref, m_ref = ob.tree_run(ops, boxes, real=True)
    assert len(h_ref) > 20000 and m_ref.sum() > 1000
    assert np.array_equal(h_orc, h_ref) and np.array_equal(m_orc, m_ref)


def test_friction_rows_match_the_reference():
    """constraint_row_friction.cpp:11-66 warm_start + solve_friction (friction circle) on random rows."""
    rng = np.random.default_rng(11)
    for t in range(3000):
        n = rng.normal(size=33).astype(np.float32); n[12] = abs(n[12]); n[13] = abs(n[13])
        n[32] = abs(n[32]) * rng.choice([0, 1, 5])
        for o in (14, 23):
            A = rng.normal(size=(3, 3)); n[o:o + 9] = (A @ A.T).astype(np.float32).reshape(-1)
        f = rng.normal(size=31).astype(np.float32); f[12] = abs(f[12]); f[27] = abs(f[27]); f[30] = rng.choice([0, 0.5, 1.0])
        d = (rng.normal(size=12) * 0.1).astype(np.float32)
        a = ob.friction_solve(n, f, d, warm=bool(t & 1), sweeps=1 + t % 3, real=False)
        b = ob.friction_solve(n, f, d, warm=bool(t & 1), sweeps=1 + t % 3, real=True)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


# ------------------------------------------------------------------------------------------- whole steps
def _canon(m):
    """Manifold records keyed by the unordered pair, with the point fields that do not depend on which body is 'A'."""
    b = m["body"].astype(np.uint64)
    key = (np.maximum(b[:, 0], b[:, 1]) << np.uint64(32)) | np.minimum(b[:, 0], b[:, 1])
    return m[np.argsort(key, kind="stable")]


def _lockstep(scene, steps, iters=10, sleeping=False, check_manifolds_every=20):
    ref = ob.RefWorld(vel_iters=iters); ref.add_bodies(scene, sleeping_disabled=not sleeping)
    orc = ob.World(vel_iters=iters, order=ob.ORDER_EXTERNAL); orc.add_bodies(scene)
    if sleeping:
        orc.set_sleeping(True)
    dyn = scene["kind"] == scenes.KIND_DYNAMIC
    first_sleep = None
    for s in range(1, steps + 1):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order())
        orc.set_ext_restitution_walk(*ref.get_restitution_walk())
        orc.step(1)
        assert not orc.ext_order_mismatch(), f"step {s}: the reference's constraint list differs from the oracle's"
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all(), f"step {s}: {name} is not finite"
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {s}: {name} differs, max |d| = {np.abs(a - b).max()}"
        if sleeping:
            ra, oa = ref.get_asleep()[dyn], orc.get_asleep()[dyn]
            assert np.array_equal(ra, oa), f"step {s}: sleeping flags differ"
            if first_sleep is None and ra.any():
                first_sleep = s
        if s % check_manifolds_every == 0 or s == steps:
            rm, om = _canon(ref.get_manifolds()), _canon(orc.get_manifolds())
            assert len(rm) == len(om) and np.array_equal(np.sort(rm["body"], axis=1), np.sort(om["body"], axis=1)), f"step {s}: pair sets differ"
            assert np.array_equal(rm["body"], om["body"]), f"step {s}: pair orientation (body[0] = querying body) differs"
            assert np.array_equal(rm["num_points"], om["num_points"])
            # A sleeping manifold's stored `distance` is not compared: one step after its island fell asleep the engine
            # rewrites it with the value its contact_constraint last held (stale by one step, ~1e-7 m); nothing reads it
            # before update_contact_distances recomputes it on wake-up (collision_util.cpp:28-45).
            asleep = ref.get_asleep()
            awake = ~(asleep[rm["body"][:, 0]] | asleep[rm["body"][:, 1]]) if sleeping else np.ones(len(rm), bool)
            for fld in ("pivotA", "pivotB", "normal", "local_normal", "distance", "friction", "attachment", "lifetime",
                        "normal_impulse", "friction_impulse"):
                sel = awake if fld == "distance" else np.ones(len(rm), bool)
                assert np.array_equal(rm["pt"][fld][sel], om["pt"][fld][sel]), f"step {s}: contact field {fld} differs"
            ad, od = ref.get_derived(), orc.get_derived()
            assert np.array_equal(ad[0][dyn].view(np.uint32), od[0][dyn].view(np.uint32)), f"step {s}: AABBs differ"
            assert np.array_equal(ad[1][dyn].view(np.uint32), od[1][dyn].view(np.uint32)), f"step {s}: world inertias differ"
            assert np.array_equal(ad[2][dyn], od[2][dyn]), f"step {s}: island partition differs"
    return ref, orc, first_sleep


def _tumbling_box():
    sc = scenes.box_pile(1, 1, 1)
    sc["pos"][1] = (0.1, 2.0, 0.2); sc["orn"][1] = (0.1, 0.2, 0.3, 0.9273618495495704); sc["angvel"][1] = (1, 2, 3)
    return sc


def _rolling_sphere():
    sc = scenes.box_pile(1, 1, 1)
    sc["shape_type"][1] = scenes.SHAPE_SPHERE; sc["shape_param"][1] = (0.5, 0, 0, 0); sc["linvel"][1] = (1, 0, 0.5)
    return sc


def _leaning_tower():
    sc = scenes.box_pile(2, 8, 2)
    sc["pos"][1:, 0] += (sc["pos"][1:, 1] * 0.08).astype(np.float32)
    return sc


def _spinning_drop():
    sc = scenes.box_pile(3, 3, 3)
    sc["angvel"][1:] = (np.random.default_rng(1).normal(size=(27, 3)) * 5).astype(np.float32)
    sc["pos"][1:, 1] += 3
    return sc


@pytest.mark.parametrize("name,make,steps,iters", [
    ("tumbling_box", _tumbling_box, 300, 10),
    ("rolling_sphere", _rolling_sphere, 300, 10),
    ("pile_3x3x3", lambda: scenes.box_pile(3, 3, 3), 120, 10),
    ("pile_4x4x4", lambda: scenes.box_pile(4, 4, 4), 60, 10),
    ("mixed_4x4x4_20it", lambda: scenes.box_pile(4, 4, 4, mixed=True), 150, 20),
    ("pyramid_5", lambda: scenes.pyramid(5), 100, 10),
    ("mini_piles_3x3", lambda: scenes.mini_piles(3, 3), 40, 10),
    ("leaning_tower", _leaning_tower, 250, 10),
    ("spinning_drop", _spinning_drop, 250, 10),
    ("chains_8x8", lambda: scenes.c5_chains(8, 8), 400, 10),
])
def test_whole_steps_bit_exact_against_the_real_engine(name, make, steps, iters):
    """stepper_sequential.cpp:121-147 end to end: the restatement and the real engine stay bit-identical."""
    ref, orc, _ = _lockstep(make(), steps, iters)
    if name.startswith("mini_piles"):
        assert ref.num_islands == 9 == orc.get_stats()["num_islands"]


@pytest.mark.parametrize("name,make,steps", [
    ("pile_2x2x2", lambda: scenes.box_pile(2, 2, 2), 400),
    ("mini_piles_2x2", lambda: scenes.mini_piles(2, 2), 330),
])
def test_island_sleeping_matches_the_real_engine(name, make, steps):
    """island_manager.cpp:524-623: same islands fall asleep at the same step (velocities zeroed), state stays bit-identical."""
    ref, orc, first_sleep = _lockstep(make(), steps, sleeping=True)
    assert first_sleep is not None and first_sleep > 120   # island_time_to_sleep = 2 s
    assert ref.get_asleep()[1:].mean() > 0.5


def _drift_apart_scene():
    """Weightless boxes: two whose boxes overlap (one manifold without points: one island) while one drifts away at 0.004 m/s - below the
    sleep thresholds -, and a third far away. After 1.75 s the manifold is destroyed and the island splits, its sleep timer running."""
    s = scenes._empty(3)
    s["kind"][:] = scenes.KIND_DYNAMIC
    s["shape_type"][:] = scenes.SHAPE_BOX; s["shape_param"][:, :3] = 0.5
    s["pos"][0] = (0, 0, 0); s["pos"][1] = (1.019, 0, 0); s["pos"][2] = (10, 0, 0)
    s["linvel"][1] = (0.004, 0, 0)
    s["gravity"] = np.zeros((3, 3), np.float32)
    return s


def test_island_split_restarts_the_sleep_timers_like_the_real_engine():
    """split_islands (island_manager.cpp:411-447) move-assigns the largest component into the island entity - a freshly built `island`
    whose sleep_timestamp is empty - and creates new islands for the rest: every part of a split island starts its 2 s timer again.
    The far box falls asleep after 2 s, the two parts only 2 s after the split (the split half)."""
    ref, orc, first_sleep = _lockstep(_drift_apart_scene(), 300, sleeping=True)
    assert 120 < first_sleep < 125
    ref2 = ob.RefWorld(vel_iters=10); ref2.add_bodies(_drift_apart_scene(), sleeping_disabled=False)
    ref2.step(200)
    assert ref2.get_asleep().tolist() == [False, False, True] and len(ref2.get_manifolds()) == 0
    ref2.step(60)
    assert ref2.get_asleep().all()


def _drift_together_scene(mirror=False):
    """Weightless boxes. Bodies 1 and 2 overlap (one island), body 3 overlaps body 2 and drifts away at 0.004 m/s - below the sleep
    thresholds - so that after 1.75 s that island splits and its parts' timers start again; body 0, alone and timed from the first step,
    drifts towards body 1 and touches its box after 1.9 s: a small island with the LOWER label and the older timer merges into a bigger
    one with the younger timer. `mirror` reverses the body order: then the bigger island has the lower label."""
    s = scenes._empty(4)
    s["kind"][:] = scenes.KIND_DYNAMIC
    s["shape_type"][:] = scenes.SHAPE_BOX; s["shape_param"][:, :3] = 0.5
    g0 = 0.02 + 0.004 * 1.9
    s["pos"][0] = (0, 0, 0); s["pos"][1] = (1 + g0, 0, 0); s["pos"][2] = (1 + g0, 1.01, 0); s["pos"][3] = (2.019 + g0, 1.01, 0)
    s["linvel"][0] = (0.004, 0, 0); s["linvel"][3] = (0.004, 0, 0)
    s["gravity"] = np.zeros((4, 3), np.float32)
    if mirror:
        for k in ("pos", "linvel"):
            s[k] = s[k][::-1].copy()
    return s


@pytest.mark.parametrize("mirror", [False, True])
def test_island_merge_keeps_the_bigger_islands_sleep_timer_like_the_real_engine(mirror):
    """merge_islands (island_manager.cpp:297-350): the biggest island survives a merge with its sleep_timestamp. The split at step 106
    restarts the two-box island's timer, the merge at step 116 joins it with a one-box island timed from step 1: everything falls
    asleep 2 s after the SPLIT (step 226), not 2 s after the start (step 121, what keeping the lower label's timer gave until round 4 -
    VERDICT r03 item 9). Bit for bit in lock-step, sleeping flags and island partition included."""
    ref, orc, first_sleep = _lockstep(_drift_together_scene(mirror), 260, sleeping=True)
    assert first_sleep == 226
    assert ref.get_asleep().all()


def test_reference_order_is_a_permutation_of_the_canonical_order():
    """ORDER_EXTERNAL only permutes: same multiset of (pair, slot) as the canonical sequence of ORDER_SEQUENTIAL."""
    sc = scenes.box_pile(3, 3, 3)
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    ref.step(30)
    contacts, joints = ref.get_solve_order()
    m = ref.get_manifolds()
    assert len(contacts) == int(m["num_points"].sum()) and len(joints) == 0
    keys = {(int(a), int(b), int(s)) for a, b, s in contacts}
    expect = {(int(r["body"][0]), int(r["body"][1]), k) for r in m for k in range(int(r["num_points"]))}
    assert keys == expect


def test_canonical_and_coloured_orders_agree_with_the_reference_order_within_tolerance():
    """The GPU's specification is the restatement in coloured order (DESIGN.md §3 "Solve order"): same row arithmetic,
    another Gauss-Seidel visiting order, so an unconverged 10-iteration solve differs in the low digits and a collapsing
    pile diverges chaotically later on. Against the real engine running its own order: identical pair sets for the first
    30 steps of a 4x4x4 brick pile, positions within 1e-2 m and velocities within 0.1 m/s after 10 steps; and on the
    stable straight-column scene (C1) the same resting heights within 1e-3 m after 200 steps."""
    ob.set_libm_trig(False)
    sc = scenes.box_pile(4, 4, 4)
    for order in (ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED):
        ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
        orc = ob.World(vel_iters=10, order=order); orc.add_bodies(sc)
        for s in range(1, 31):
            ref.step(1); orc.step(1)
            assert np.array_equal(ref.get_pairs(), orc.get_pairs())
            if s == 10:
                (rp, rq, rv, rw), (op, oq, ov, ow) = ref.get_state(), orc.get_state()
                assert np.abs(rp - op).max() < 1e-2 and np.abs(rv - ov).max() < 0.1
    cols = scenes.subset(scenes.c1_columns(), np.r_[0, 1:1001:1][:1 + 10 * 10])   # plane + the first 10 columns
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(cols)
    orc = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); orc.add_bodies(cols)
    ref.step(200); orc.step(200)
    assert np.abs(ref.get_state()[0][:, 1] - orc.get_state()[0][:, 1]).max() < 1e-3


def test_solver_residual_and_penetration_of_the_coloured_order_against_the_real_engine():
    """SURVEY section 7 hard-part 1 (VERDICT r04 missing #4): the solver-residual invariant. After the last velocity iteration of a step,
    |J v - rhs| over the active normal rows (tests/invariants.py: the relative normal velocity the unconverged Gauss-Seidel sweep leaves at
    loaded, touching contacts) in the coloured order - the device's specification, with the reference's row arithmetic - against the real
    engine running its own order, free-running from the same start on a box pile (10 iterations) and a box / sphere mix (20 iterations), while
    the piles settle (steps 120, 240). The coloured order must not leave more than TWICE the engine's residual (measured: it leaves LESS -
    0.3-0.5x the maximum, 0.05-0.3x the mean: a body meets its contacts in colour order, bottom-up through the lattice, instead of in
    EnTT creation order), and the pile sits no deeper than the engine's (99th percentile and mean penetration within 2x). The device is bit-exact against this coloured
    order; the same invariant is taken on the device at BASELINE size in tests/test_gpu_parity.py."""
    from invariants import normal_row_residual, penetration_stats
    for name, sc, vel, height in (("pile_8x8x8", scenes.box_pile(8, 8, 8), 10, 8.0), ("mixed_7x7x7", scenes.box_pile(7, 7, 7, mixed=True), 20, 7.0)):
        ref = ob.RefWorld(vel_iters=vel); ref.add_bodies(sc)
        orc = ob.World(vel_iters=vel, order=ob.ORDER_COLOURED); orc.add_bodies(sc)
        for upto in (120, 240):
            ref.step(120); orc.step(120)
            (em, ea, en, e99, ex), (cm, ca, cn, c99, cx) = normal_row_residual(ref.get_state(), ref.get_manifolds()), normal_row_residual(orc.get_state(), orc.get_manifolds())
            (pe, pe99, pea), (pc, pc99, pca) = penetration_stats(ref.get_manifolds()), penetration_stats(orc.get_manifolds())
            print(f"\n[residual] {name} step {upto} [coloured / engine]: max among resting bodies {cm:.3e} / {em:.3e} m/s, 99th percentile {c99:.3e} / {e99:.3e}, mean {ca:.3e} / {ea:.3e} m/s, max over all {cx:.3e} / {ex:.3e} ({cn} / {en} active rows); "
                  f"penetration deepest {pc:.4f} / {pe:.4f} m, 99th percentile {pc99:.5f} / {pe99:.5f} m, mean {pca:.2e} / {pea:.2e} m")
            assert cn > 100 and en > 100
            assert cm <= 2.0 * em + 1e-3 and c99 <= 2.0 * e99 + 1e-4 and ca <= 2.0 * ea, (name, upto, cm, em, c99, e99, ca, ea)
            # the pile as a whole sits no deeper than the engine's (percentile and mean within 2x + 0.2 mm); the single deepest point is a
            # faller's transient on either side (see penetration_stats), bounded by one step of free fall from the pile's height
            assert pc99 <= 2.0 * pe99 + 2e-4 and pca <= 2.0 * pea + 2e-4, (name, upto, pc99, pe99, pca, pea)
            assert max(pc, pe) <= float(np.sqrt(2 * 9.8 * height)) / 60.0, (name, upto, pc, pe)


def test_jointed_scenes_in_lock_step_with_the_real_engine():
    """VERDICT r04 missing #3: how far the COLOURED joint order lands from the engine per step on the jointed configurations - C5-style
    chains (hinge_constraint.cpp:26-213, point_constraint.cpp:9-58) and a heap of the reference's rag dolls (cone / cvjoint / hinge, capsule
    contacts) - with every step restarted from the engine's own state: bodies, manifolds and the joints' applied impulses and tracked
    angles (set_joint_warm_start). Two statements:
      * the engine's own row order replayed (ORDER_EXTERNAL): ZERO difference for every step - the hand-over of an engine state (incl. the
        joint warm start) into another stepper is exact, so whatever the coloured order differs by below is its visiting order alone;
      * the coloured order (the device's; the device is bit-exact against it): pair sets and narrowphase output identical, the solved state
        within the bounds asserted here per step. Measured: chains 3.0e-3 m / 0.24 m/s; the collapsing rag-doll heap 2.9e-2 m / 1.7 m/s
        (limbs hitting the floor at 5-10 m/s, 36 stiff constraints per figure with bump stops, 10 iterations: the joints are far from
        converged in either order) - mean over the bodies 2.9e-3 m / 0.18 m/s in the worst step, median step 5e-3 m / 0.32 m/s."""
    from invariants import resync_lockstep_jointed
    chains = scenes.c5_chains(64, 16)
    dolls = scenes.figures(scenes.load_figure(os.path.join(GOLDEN, "ragdoll_capsule.npz")), 3, 2, pitch=1.0, ny=3, pitch_v=1.9)
    for name, sc, steps, contacts, tol_p, tol_v in (("chains_64x16", chains, 90, False, 1e-2, 0.6), ("ragdoll_heap_18", dolls, 90, True, 8e-2, 4.0)):
        # (a) the engine's order replayed: exact
        ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc); scenes.apply_figure_settings(ref, sc)
        orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc); scenes.apply_figure_settings(orc, sc)
        from invariants import canonical_records
        for step in range(1, 41):
            orc.set_state(*ref.get_state()); orc.refresh_derived()
            if contacts:
                orc.set_manifolds(canonical_records(ref.get_manifolds(), sc["kind"]))
            orc.set_joint_warm_start(ref.get_joint_impulses24(), ref.get_joint_impulses()[:, 9])
            ref.step(1)
            orc.set_ext_order(*ref.get_solve_order())
            orc.step(1)
            assert not orc.ext_order_mismatch(), (name, step)
            for a, b, f in zip(orc.get_state(), ref.get_state(), ("pos", "orn", "linvel", "angvel")):
                assert np.array_equal(a, b), (name, step, f)
        # (b) the coloured order: bounded per step
        ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc); scenes.apply_figure_settings(ref, sc)
        orc = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); orc.add_bodies(sc); scenes.apply_figure_settings(orc, sc)
        wp, wv, ww = resync_lockstep_jointed(orc, ref, sc["kind"], steps, tol_p, tol_v, contacts)
        vmax = float(np.abs(ref.get_state()[2]).max())
        print(f"\n[lock-step] {name}, coloured order vs engine, {steps} steps: worst per step |dpos| {wp:.3e} m, |dvel| {wv:.3e} m/s, |dangvel| {ww:.3e} rad/s "
              f"(bounds {tol_p} m / {tol_v} m/s; fastest body at the end {vmax:.2f} m/s)")


def test_two_phase_coloured_order_distance_to_the_engine():
    """VERDICT r05 missing #5 / next #7(c): the reference sweeps ALL normal rows of an island, then ALL friction rows, per iteration
    (island_solver.cpp:94-111); the coloured order - the device's specification - finishes each manifold (its normals, then its friction
    rows) before the next one. How much of the coloured order's per-step distance to the engine is that choice, how much the colour order
    itself? Inside one colour the two forms coincide (a colour's manifolds share no dynamic body), so the comparable variant is the
    island-wide two-phase sweep in colour order (oracle ARITH_TWO_PHASE, checker-only: a device would visit every body's chain twice per
    iteration). Lock-step against the engine (every step restarted from the engine's state and manifolds) on the 8^3 pile at 10 iterations
    and the 7^3 mixed pile at 20, first 40 steps (the collapse onto the 5 mm gaps, where the orders differ most). Result (printed; round 6):
    the island-wide two-phase variant lands 8-17 % closer to the engine (8^3 pile: worst step 1.31e-3 -> 1.12e-3 m, mean over bodies and
    steps 6.5e-5 -> 5.4e-5 m; 7^3 mixed pile: 6.9e-4 -> 6.4e-4 m, 4.3e-5 -> 3.9e-5 m): about a sixth of the coloured order's distance to the
    engine is the interleaving of row kinds, the rest is the order in which the bodies are visited - at the price of two passes over every
    body's chain per iteration on a device (twice the hand-offs of the dataflow solve), which is why the device keeps the per-manifold form."""
    from invariants import canonical_records
    fig = {}
    try:
        for name, gen, vel in (("pile_8x8x8", lambda: scenes.box_pile(8, 8, 8), 10), ("mixed_7x7x7", lambda: scenes.box_pile(7, 7, 7, mixed=True), 20)):
            for label, mode in (("per manifold (the device's order)", ob.ARITH_REFERENCE), ("two phases over the island", ob.ARITH_TWO_PHASE)):
                ob.set_arithmetic(mode)
                sc = gen()
                ref = ob.RefWorld(vel_iters=vel); ref.add_bodies(sc)
                orc = ob.World(vel_iters=vel, order=ob.ORDER_COLOURED); orc.add_bodies(sc)
                worst_p = worst_v = 0.0; mean_p = []
                for step in range(1, 41):
                    orc.set_state(*ref.get_state()); orc.refresh_derived()
                    orc.set_manifolds(canonical_records(ref.get_manifolds(), sc["kind"]))
                    orc.step(1); ref.step(1)
                    assert np.array_equal(orc.get_pairs(), ref.get_pairs()), (name, label, step)
                    (wp, _, wv, _), (rp, _, rv, _) = orc.get_state(), ref.get_state()
                    worst_p = max(worst_p, float(np.abs(wp - rp).max())); worst_v = max(worst_v, float(np.abs(wv - rv).max()))
                    mean_p.append(float(np.abs(wp - rp).max(axis=1).mean()))
                fig[(name, label)] = (worst_p, worst_v, float(np.mean(mean_p)))
                print(f"\n[figures] lock-step vs engine, {name}, {vel} iterations, coloured order {label}: worst per step |dpos| {worst_p:.3e} m, |dvel| {worst_v:.3e} m/s, "
                      f"mean over bodies and steps {np.mean(mean_p):.3e} m")
    finally:
        ob.set_arithmetic(ob.ARITH_REFERENCE)
    for name in ("pile_8x8x8", "mixed_7x7x7"):
        a, b = fig[(name, "per manifold (the device's order)")], fig[(name, "two phases over the island")]
        assert a[0] < 2e-3 and a[1] < 0.1, (name, a)           # the device's order: the lock-step contract of tests/test_gpu_parity.py
        assert b[0] < 2e-3 and b[1] < 0.1, (name, b)           # the variant: the same contract
        assert 0.6 * a[2] < b[2] < 1.05 * a[2], (name, a, b)    # somewhat closer on average, not a different regime


def _pair_bodies(keys):
    keys = np.asarray(keys, np.uint64)
    return (keys >> np.uint64(32)).astype(np.int64), (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)


def _checkerboard_filter(default):
    """A user predicate on top of the default: boxes whose indices have the same parity pass through each other."""
    return lambda a, b: default(a, b) and (a == 0 or b == 0 or (a + b) % 2 == 1)


def test_user_should_collide_predicate_matches_the_real_engine():
    """settings.should_collide_func (settings.hpp:43, edyn::set_should_collide, called at broadphase.cpp:145): a user predicate that
    replaces should_collide_default - here the default AND "index parities differ" - in the real engine and in the restatement:
    same pair sets, bit-identical state over 150 steps of a collapsing pile in which half of the box pairs never get a manifold;
    switched off again after 80 steps (existing manifolds live on, new ones follow the default)."""
    sc = scenes.box_pile(3, 3, 3)
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    ref.set_should_collide(_checkerboard_filter(ref.default_should_collide))
    orc.set_should_collide(_checkerboard_filter(orc.default_should_collide))
    for s in range(1, 151):
        if s == 81:
            ref.set_should_collide(None); orc.set_should_collide(None)
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order())
        orc.step(1)
        assert np.array_equal(ref.get_pairs(), orc.get_pairs()), s
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), s
        if s == 80:
            hi, lo = _pair_bodies(ref.get_pairs())
            boxes = (hi != 0) & (lo != 0)
            assert boxes.sum() > 10 and ((hi[boxes] + lo[boxes]) % 2 == 1).all()
    hi, lo = _pair_bodies(ref.get_pairs())
    boxes = (hi != 0) & (lo != 0)
    assert ((hi[boxes] + lo[boxes]) % 2 == 0).any()   # after the switch-off same-parity pairs appear


def test_collision_filter_and_exclusion_lists_match_the_real_engine():
    """should_collide_default (should_collide.cpp:11-57) inside whole steps: group/mask bits and exclude_collision lists decide
    which manifolds get created. A 3x3x3 pile where one column of boxes ignores its neighbours (filter) and six pairs are
    excluded explicitly: pair sets and state stay bit-identical with the real engine while boxes fall through each other."""
    sc = scenes.box_pile(3, 3, 3)
    sc["group"][5] = 0x2; sc["mask"][5] = np.uint64(0xFFFFFFFFFFFFFFFF) & ~np.uint64(0x4)
    sc["group"][14] = 0x4; sc["mask"][14] = np.uint64(0xFFFFFFFFFFFFFFFF) & ~np.uint64(0x2)
    excl = [(1, 10), (2, 11), (3, 12), (13, 22), (14, 23), (10, 19)]
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    for a, b in excl:
        ref.exclude_collision(a, b); orc.exclude_collision(a, b)
    for s in range(1, 121):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch()
        assert np.array_equal(ref.get_pairs(), orc.get_pairs()), s
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), s
    pairs = {(int(k >> 32), int(k & 0xFFFFFFFF)) for k in ref.get_pairs()}
    for a, b in excl:
        assert (max(a, b), min(a, b)) not in pairs


# ------------------------------------------------------------------------- optional joint rows, removal, settings
def _joint_lockstep(scene, steps, setup, iters=10):
    ref = ob.RefWorld(vel_iters=iters); ref.add_bodies(scene)
    orc = ob.World(vel_iters=iters, order=ob.ORDER_EXTERNAL); orc.add_bodies(scene)
    setup(ref); setup(orc)
    for s in range(1, steps + 1):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.set_ext_restitution_walk(*ref.get_restitution_walk()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all(), (s, name)   # (equal NaN bit patterns would pass the comparison below)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, name)
        assert np.array_equal(ref.get_joint_impulses().view(np.uint32), orc.get_joint_impulses().view(np.uint32)), s
    return ref, orc


def _chain_joints(scene):
    hinges = [i for i, j in enumerate(scene["joints"]) if j[0] == scenes.JOINT_HINGE]
    points = [i for i, j in enumerate(scene["joints"]) if j[0] == scenes.JOINT_POINT]
    return hinges, points


@pytest.mark.parametrize("name,params", [
    ("limits_bump_stop_restitution", [-0.3, 0.4, 0.2, 0.1, 5.0, 0, 0, 0, 0, 0]),
    ("spring_damping", [0, 0, 0, 0, 0, 0, 0, 0.2, 3.0, 0.05]),
    ("torque_speed", [0, 0, 0, 0, 0, 0.05, 0.5, 0, 0, 0]),
    ("all_rows", [-0.5, 0.5, 0.3, 0.2, 2.0, 0.01, 0.0, 0.1, 1.0, 0.02]),
])
def test_hinge_optional_rows_match_the_real_engine(name, params):
    """hinge_constraint.cpp:69-178: angle tracking across wraps, limit row with restitution, bump stop, spring, torque /
    damping rows, and their applied impulses (store_applied_impulses :215-257) - bit-identical with the real engine on
    swinging chains, joint impulses and tracked angle included."""
    sc = scenes.c5_chains(4, 6)
    hinges, _ = _chain_joints(sc)

    def setup(w):
        for i in hinges:
            w.set_joint_params(i, params)
    ref, _ = _joint_lockstep(sc, 300, setup)
    ji = ref.get_joint_impulses()
    assert np.abs(ji[hinges][:, 5:9]).max() > 0   # the optional rows did act


def test_point_friction_torque_matches_the_real_engine():
    """point_constraint.cpp:33-46 (row along the normalised relative spin, limits +-friction_torque*dt). The links start
    spinning so that the row exists from the first step: while the relative spin is zero the reference's
    store_applied_impulses reads one element past its impulse array (point_constraint.cpp:55-57), which cannot be mirrored."""
    sc = scenes.c5_chains(4, 6)
    sc["angvel"][:] = (np.random.default_rng(3).normal(size=sc["angvel"].shape) * 2).astype(np.float32)
    _, points = _chain_joints(sc)

    def setup(w):
        for i in points:
            w.set_joint_params(i, [0.03])
    ref, _ = _joint_lockstep(sc, 300, setup)
    assert np.abs(ref.get_joint_impulses()[points][:, 3]).max() > 0


def test_body_and_joint_removal_match_the_real_engine():
    """registry.destroy on bodies / constraint entities of a running world (island_manager.cpp:47-115: the node's edges -
    manifolds, contact points, joints - go with it, its island wakes and is split): the survivors stay bit-identical."""
    sc = scenes.box_pile(3, 3, 3)
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    keep = np.ones(len(sc["kind"]), bool)
    for s in range(1, 121):
        if s in (30, 60):
            for b in ((14, 5) if s == 30 else (1,)):
                ref.remove_body(b); orc.remove_body(b); keep[b] = False
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        assert np.array_equal(ref.get_pairs(), orc.get_pairs()), s
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a[keep].view(np.uint32), b[keep].view(np.uint32)), s
    ch = scenes.c5_chains(3, 6)
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(ch)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(ch)
    for s in range(1, 201):
        if s == 50:
            ref.remove_joint(8); orc.remove_joint(8)       # a chain falls apart in the middle
        if s == 100:
            body = 1 + 6 + 3                                # a link of the second chain disappears with both its joints
            ref.remove_body(body); orc.remove_body(body)
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        sel = np.ones(len(ch["kind"]), bool)
        if s >= 100:
            sel[1 + 6 + 3] = False
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a[sel].view(np.uint32), b[sel].view(np.uint32)), s


def test_runtime_settings_match_the_real_engine():
    """set_solver_*_iterations / set_gravity / set_fixed_dt on a running world (solver_iteration_config.cpp:9-75,
    gravity_util.cpp:12-20, edyn.cpp:203-207) keep every manifold and warm-start impulse: the trajectory stays bit-identical."""
    sc = scenes.box_pile(3, 3, 3)
    ref = ob.RefWorld(vel_iters=8); ref.add_bodies(sc)
    orc = ob.World(vel_iters=8, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    for s in range(1, 91):
        if s == 30:
            ref.set_params(1 / 60, 14, 2, (0.5, -6.0, 0.0)); orc.set_params(1 / 60, 14, 2, (0.5, -6.0, 0.0))
        if s == 60:
            ref.set_params(1 / 90, 5, 3, (0.0, -9.8, 0.0)); orc.set_params(1 / 90, 5, 3, (0.0, -9.8, 0.0))
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), s
    m = ref.get_manifolds()
    assert m["pt"]["lifetime"].max() > 60   # contacts (and their impulses) survived both changes


def test_update_accumulator_and_stretched_time_stamps_match_the_real_engine():
    """edyn::update(registry, time) (stepper_sequential.cpp:28-119): floor(acc / dt) steps per call, clamped to
    max_steps_per_update; when clamped, the steps that do run carry stretched time stamps (:60-66) - which only the island
    sleep timers see. Spheres at rest on a plane (one contact point per island: no visiting order to inject) driven by the
    same irregular clock through the real engine and through edyn_amd.world.fixed_step_plan + the oracle's step_timed:
    identical state and identical sleeping flags after every update, and sleep arrives after far fewer STEPS than 2 s / dt."""
    from edyn_amd.world import fixed_step_plan
    n = 6
    sc = scenes._empty(n + 1); scenes._add_plane(sc, 0)
    for i in range(n):
        sc["kind"][i + 1] = scenes.KIND_DYNAMIC; sc["pos"][i + 1] = (2.0 * i, 0.5 + 0.01 * i, 0.3 * i)
        sc["shape_type"][i + 1] = scenes.SHAPE_SPHERE; sc["shape_param"][i + 1] = (0.5, 0, 0, 0)
    ref = ob.RefWorld(vel_iters=8); ref.add_bodies(sc, sleeping_disabled=False)
    orc = ob.World(vel_iters=8, order=ob.ORDER_SEQUENTIAL); orc.add_bodies(sc); orc.set_sleeping(True)
    max_steps = 5
    ref.set_max_steps_per_update(max_steps)
    dt = float(np.float32(1 / 60))
    last, acc, t, total_steps, first_sleep_steps = 0.0, 0.0, 0.0, 0, None
    rng = np.random.default_rng(5)
    for k in range(40):
        t += float(rng.choice([0.004, 0.016, 0.021, 0.3, 0.45]))     # frames shorter than dt, normal frames, long stalls
        ref.update(t)
        sim_time = last - acc
        steps, acc, step_dt = fixed_step_plan(acc, t - last, dt, max_steps)
        if steps:
            orc.step_timed(steps, sim_time, step_dt)
        last = t
        total_steps += steps
        for a, b in zip(ref.get_state(), orc.get_state()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        ra, oa = ref.get_asleep()[1:], orc.get_asleep()[1:]
        assert np.array_equal(ra, oa), k
        if first_sleep_steps is None and ra.any():
            first_sleep_steps = total_steps
    assert first_sleep_steps is not None and first_sleep_steps < 90   # 2 s of stamps, far fewer than 120 steps


# ---------------------------------------------------------------------------------------------- restitution solver
def _bouncy(n, rest, shape="sphere", stacked=False, seed=1):
    rng = np.random.default_rng(seed)
    s = scenes._empty(n + 1); scenes._add_plane(s, 0); s["restitution"][0] = 1.0
    for i in range(n):
        s["kind"][i + 1] = scenes.KIND_DYNAMIC
        s["pos"][i + 1] = (0.02 * i, 1.0 + 1.3 * i, 0.01 * i) if stacked else (3.0 * i, 1.0 + 0.7 * i, 0.5 * i)
        if shape == "sphere":
            s["shape_type"][i + 1] = scenes.SHAPE_SPHERE; s["shape_param"][i + 1] = (0.5, 0, 0, 0)
        else:
            s["shape_type"][i + 1] = scenes.SHAPE_BOX; s["shape_param"][i + 1] = (0.5, 0.4, 0.3, 0)
            q = rng.normal(size=4); s["orn"][i + 1] = q / np.linalg.norm(q); s["angvel"][i + 1] = rng.normal(size=3)
        s["restitution"][i + 1] = rest[i % len(rest)]
        s["linvel"][i + 1] = (0.0, 0.0, 0.0) if stacked else (0.3 * i, 0, 0.1)
    return s


@pytest.mark.parametrize("name,make", [
    ("spheres", lambda: _bouncy(6, [0.9, 0.5, 0.2, 0.0, 0.7])),
    ("boxes", lambda: _bouncy(5, [0.8, 0.4, 0.6, 0.3], "box")),
])
def test_restitution_solver_matches_the_real_engine(name, make):
    """restitution_solver.cpp:86-408 (fastest closing tagged manifold, threshold -0.005 m/s, rows with the contact's
    restitution, 3 sweeps from zero impulses, velocities applied at once; the constraint solver then runs with zero
    restitution, solver.cpp:282-283). Bodies bouncing in islands of their own - one manifold each, so the graph walk has
    no order to choose - stay bit-identical with the real engine through hundreds of bounces, tumbling boxes included."""
    sc = make()
    ref, orc, _ = _lockstep(sc, 400)
    assert ref.get_state()[0][1:, 1].max() < 6 and np.isfinite(ref.get_state()[0]).all()


def test_restitution_shock_propagation_close_to_the_real_engine():
    """A column of bouncy spheres is ONE island: the engine walks it in its entity graph's adjacency order, the restatement
    in canonical (pair-key) order. Same physics: over 120 steps of impacts and rebounds the positions stay within 2 mm of
    the real engine's (bit-identical while the column is still falling), and spheres do rebound."""
    ob.set_libm_trig(False)
    sc = _bouncy(4, [0.8, 0.6], stacked=True)
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); orc.add_bodies(sc)
    rebound = 0.0
    for s in range(120):
        ref.step(1); orc.step(1)
        rebound = max(rebound, float(ref.get_state()[2][1:, 1].max()))
        assert np.abs(ref.get_state()[0] - orc.get_state()[0]).max() < 2e-3, s
    assert rebound > 2.0


def _bouncy_pile(nx, ny, nz, mixed, spin):
    sc = scenes.box_pile(nx, ny, nz, mixed=mixed)
    sc["restitution"][:] = 0.6 if not mixed else 0.5
    sc["pos"][1:, 1] += 0.4
    if spin:
        sc["angvel"][1:] = np.random.default_rng(3).normal(size=(len(sc["kind"]) - 1, 3)).astype(np.float32)
    return sc


@pytest.mark.parametrize("name,make,steps", [
    ("pile_3x2_dropped", lambda: _bouncy_pile(3, 2, 1, False, False), 400),
    ("pile_3x2_tumbling", lambda: _bouncy_pile(3, 2, 1, False, True), 400),
    ("mixed_4x4x4", lambda: _bouncy_pile(4, 4, 4, True, False), 300),
    ("sphere_column", lambda: _bouncy(4, [0.8, 0.6], stacked=True), 300),
])
def test_restitution_walk_in_large_islands_matches_the_real_engine(name, make, steps):
    """Islands with many manifolds: the restitution solver searches the island's edge list for the fastest closing manifold and then
    walks the entity graph breadth first from it, solving each body's star of closing manifolds (restitution_solver.cpp:86-385). Both
    orders are data-structure orders of the real engine (island.edges, the graph's adjacency lists); replayed from the engine step by
    step (RefWorld.get_restitution_walk -> World.set_ext_restitution_walk, like the constraint order), the restatement's restitution
    solve - star assembly, 3 sweeps, velocities applied per star - is bit-identical with the engine through the impacts and rebounds."""
    ref, orc, _ = _lockstep(make(), steps)
    assert np.isfinite(ref.get_state()[0]).all()


def _event_multiset(ev3):
    """(type, min body, max body) -> count"""
    from collections import Counter
    return Counter((int(t), int(min(a, b)), int(max(a, b))) for t, a, b in ev3)


def test_contact_events_match_what_the_engine_signals():
    """Contact events: the oracle's event list (manifold / contact point created / destroyed, the
    device's edynhip_get_contact_events spec) against what an application observes on the real engine's registry through
    on_construct / on_destroy<contact_manifold> and <contact_point>, step by step over a collapsing mixed pile and a body
    removal. Point ids are created once, destroyed once, and the live set equals the manifolds' points."""
    scene = scenes.box_pile(4, 4, 4, mixed=True)
    o = ob.World(order=ob.ORDER_EXTERNAL); o.add_bodies(scene); ob.set_libm_trig(True)
    r = ob.RefWorld(); r.add_bodies(scene)
    try:
        o.record_events(True); r.record_events(True)
        live = set()
        total = 0
        for step in range(120):
            if step == 60:   # registry.destroy(body): its manifolds and points go with it
                o.remove_body(10); r.remove_body(10)
            r.step(1)
            o.set_ext_order(*r.get_solve_order()); o.step(1)
            eo, er = o.get_events(), r.get_events()
            assert _event_multiset(zip(eo["type"], eo["body"][:, 0], eo["body"][:, 1])) == _event_multiset(er), step
            for e in eo:
                if e["type"] == 3:
                    assert int(e["point_id"]) not in live; live.add(int(e["point_id"]))
                elif e["type"] == 4:
                    live.remove(int(e["point_id"]))
            ids = o.get_point_ids()
            assert set(int(x) for x in ids.ravel() if x) == live, step
            total += len(eo)
            o.clear_events(); r.clear_events()
        assert total > 500
    finally:
        ob.set_libm_trig(False)


def _extras_scene(kind):
    """Spheres and boxes on the plane with contact_extras materials: rolling + spinning friction, or soft contacts."""
    sc = scenes.box_pile(3, 2, 3, mixed=True)
    n = len(sc["kind"])
    rng = np.random.default_rng(3)
    sc["linvel"][1:] = (rng.normal(size=(n - 1, 3)) * (1.5, 0.2, 1.5)).astype(np.float32)
    sc["angvel"][1:] = (rng.normal(size=(n - 1, 3)) * 3).astype(np.float32)
    ex = {}
    for i in range(n):
        if kind == "roll_spin":
            ex[i] = dict(spin=0.02 + 0.01 * (i % 3), roll=0.05 if i % 2 else 0.0)
        elif kind == "soft":
            ex[i] = dict(stiffness=4000.0 + 500.0 * (i % 4), damping=60.0) if (i % 3 != 1) else {}
        else:   # everything at once; some bodies keep the plain contact_constraint
            ex[i] = dict(spin=0.03, roll=0.04, stiffness=6000.0, damping=80.0) if i % 2 == 0 else (dict(roll=0.02) if i % 4 == 1 else {})
    return sc, ex


@pytest.mark.parametrize("kind", ["roll_spin", "soft", "both"])
def test_contact_extras_match_the_real_engine(kind):
    """contact_extras_constraint (contact_extras_constraint.cpp:12-110, constraint_row_spin_friction.cpp:5-36,
    island_solver.cpp:76-111 row-kind order, material_mixing.hpp:20-34): rolling and spinning friction rows and soft
    (stiffness / damping limited) normal rows - state, the rolling / spinning impulses carried by the contact points and
    the mixed material values, bit for bit against the real engine over a rolling, spinning, settling scene."""
    sc, ex = _extras_scene(kind)
    ref = ob.RefWorld(); ref.add_bodies(sc)
    orc = ob.World(order=ob.ORDER_EXTERNAL); orc.add_bodies(sc); ob.set_libm_trig(True)
    try:
        for i, kw in ex.items():
            if kw:
                ref.set_material_extras(i, **kw); orc.set_material_extras(i, **kw)
        seen_roll = seen_spin = seen_soft = 0
        for s in range(1, 241):
            ref.step(1)
            orc.set_ext_order(*ref.get_solve_order())
            orc.step(1)
            assert not orc.ext_order_mismatch(), s
            for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {s}: {name} differs, max |d| = {np.abs(a - b).max()}"
            if s % 10 == 0:
                rm, om = _canon(ref.get_manifolds()), _canon(orc.get_manifolds())
                assert np.array_equal(rm["body"], om["body"]) and np.array_equal(rm["num_points"], om["num_points"]), s
                for fld in ("normal_impulse", "friction_impulse", "distance"):
                    assert np.array_equal(rm["pt"][fld], om["pt"][fld]), (s, fld)
                rx, ox = ref.get_point_extras(), orc.get_point_extras()   # both in ascending (max, min) pair order
                ob_ = orc.get_manifolds()["body"].astype(np.uint64)
                okey = (np.maximum(ob_[:, 0], ob_[:, 1]) << np.uint64(32)) | np.minimum(ob_[:, 0], ob_[:, 1])
                ox = ox[np.argsort(okey, kind="stable")]
                assert np.array_equal(rx.view(np.uint32), ox.view(np.uint32)), f"step {s}: rolling / spinning impulses or mixed materials differ"
                seen_roll += int((rx[..., 0] != 0).sum()); seen_spin += int((rx[..., 2] != 0).sum()); seen_soft += int((rx[..., 5] < 1e17).sum() - (rx[..., 5] == 0).sum())
        if kind in ("roll_spin", "both"):
            assert seen_roll > 0 and seen_spin > 0
        if kind in ("soft", "both"):
            assert seen_soft > 0
    finally:
        ob.set_libm_trig(False)


def _capsule_scene():
    sc = scenes.box_pile(3, 3, 3, mixed=True)
    n = len(sc["kind"])
    for i in range(1, n, 2):
        sc["shape_type"][i] = scenes.SHAPE_CAPSULE
        sc["shape_param"][i] = (0.3, 0.2 + 0.05 * (i % 4), float(i % 3), 0)
    rng = np.random.default_rng(9)
    sc["angvel"][1:] = (rng.normal(size=(n - 1, 3)) * 2).astype(np.float32)
    sc["linvel"][1:] = (rng.normal(size=(n - 1, 3)) * (0.8, 0.1, 0.8)).astype(np.float32)
    return sc


def test_capsules_whole_steps_bit_exact_against_the_real_engine():
    """capsule_shape: AABB (aabb_util.cpp:81-88), solid-capsule inertia (moment_of_inertia.cpp:65-90), the
    four capsule pair routines and rolling_tag matching in the narrowphase - a tumbling heap of capsules, boxes and spheres
    stays bit-identical to the real engine for 250 steps."""
    _lockstep(_capsule_scene(), 250, 10)


def _cylinder_scene():
    """A tumbling heap of cylinders (flat discs and rods, all three axes), capsules, boxes and spheres."""
    sc = scenes.box_pile(3, 3, 4, mixed=True)
    n = len(sc["kind"])
    for i in range(1, n):
        if i % 3 == 0:
            sc["shape_type"][i] = scenes.SHAPE_CYLINDER
            sc["shape_param"][i] = (0.25 + 0.05 * (i % 5), 0.1 + 0.08 * (i % 4), float(i % 3), 0)
        elif i % 7 == 1:
            sc["shape_type"][i] = scenes.SHAPE_CAPSULE
            sc["shape_param"][i] = (0.3, 0.2 + 0.05 * (i % 4), float(i % 3), 0)
    rng = np.random.default_rng(11)
    sc["angvel"][1:] = (rng.normal(size=(n - 1, 3)) * 2).astype(np.float32)
    sc["linvel"][1:] = (rng.normal(size=(n - 1, 3)) * (0.8, 0.1, 0.8)).astype(np.float32)
    return sc


def test_cylinders_whole_steps_bit_exact_against_the_real_engine():
    """cylinder_shape: AABB (aabb_util.cpp:72-79), solid-cylinder inertia (moment_of_inertia.cpp:27-44), its five
    pair routines (cylinder-plane / sphere / cylinder / box, capsule-cylinder) with the circle-line and circle-circle Newton
    iterations, and rolling_tag matching in the narrowphase: a tumbling heap of cylinders, capsules, boxes and spheres stays
    bit-identical to the real engine for 250 steps."""
    _lockstep(_cylinder_scene(), 250, 10)


def _polyhedron_scene():
    """A tumbling heap of convex polyhedra (the ten meshes of tests/meshes.py) among cylinders, capsules, boxes and spheres."""
    lib, _ = meshes.registered()
    sc = scenes.box_pile(3, 3, 5, mixed=True)
    n = len(sc["kind"])
    rng = np.random.default_rng(12)
    for i in range(1, n):
        if i % 2 == 0:
            sc["shape_type"][i] = scenes.SHAPE_POLYHEDRON
            sc["shape_param"][i] = (float(i // 2 % len(lib)), 0, 0, 0)
        elif i % 7 == 1:
            sc["shape_type"][i] = scenes.SHAPE_CYLINDER
            sc["shape_param"][i] = (0.25 + 0.05 * (i % 5), 0.1 + 0.08 * (i % 4), float(i % 3), 0)
        elif i % 7 == 3:
            sc["shape_type"][i] = scenes.SHAPE_CAPSULE
            sc["shape_param"][i] = (0.3, 0.2 + 0.05 * (i % 4), float(i % 3), 0)
    q = rng.normal(size=(n - 1, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    sc["orn"][1:] = q.astype(np.float32)
    sc["angvel"][1:] = (rng.normal(size=(n - 1, 3)) * 2).astype(np.float32)
    sc["linvel"][1:] = (rng.normal(size=(n - 1, 3)) * (0.8, 0.1, 0.8)).astype(np.float32)
    sc["meshes"] = lib
    return sc


def test_polyhedra_whole_steps_bit_exact_against_the_real_engine():
    """polyhedron_shape: convex_mesh::initialize, mesh inertia, the rotated mesh refreshed after every integration
    (update_rotated_meshes.cpp, solver.cpp:458), its AABB (update_aabbs.cpp:22-32), and the six pair routines: a tumbling heap of
    polyhedra, cylinders, capsules, boxes and spheres stays bit-identical to the real engine for 300 steps."""
    _lockstep(_polyhedron_scene(), 300, 10)


def test_polyhedron_heap_whole_steps_bit_exact_against_the_real_engine():
    """The product's own polyhedron scene (edyn_amd.scenes.polyhedron_heap, 256 bodies: the six meshes of convex_library in random
    orientations falling into a heap) stays bit-identical to the real engine for 250 steps."""
    _lockstep(scenes.polyhedron_heap(8, 4, 8), 250, 10)


def test_capsule_rolling_friction_uses_the_roll_direction():
    """contact_extras rolling rows of bodies with a roll_direction (dynamic capsules: their axis, shapes.hpp:136-139): the
    tangent axes are scaled by the projection of the rolling direction (contact_extras_constraint.cpp:44-55)."""
    sc = _capsule_scene()
    ref = ob.RefWorld(); ref.add_bodies(sc)
    orc = ob.World(order=ob.ORDER_EXTERNAL); orc.add_bodies(sc); ob.set_libm_trig(True)
    try:
        for i in range(len(sc["kind"])):
            ref.set_material_extras(i, roll=0.05, spin=0.01); orc.set_material_extras(i, roll=0.05, spin=0.01)
        for s in range(1, 201):
            ref.step(1)
            orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
            assert not orc.ext_order_mismatch(), s
            for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {s}: {name} differs, max |d| = {np.abs(a - b).max()}"
    finally:
        ob.set_libm_trig(False)


def _distance_scene():
    """Chains whose links hang on distance / soft-distance constraints (mixed with the hinges and points of the C5 chains)."""
    sc = scenes.c5_chains(4, 6)
    joints = []
    for i, j in enumerate(sc["joints"]):
        jt = (scenes.JOINT_DISTANCE, scenes.JOINT_SOFT_DISTANCE, j[0])[i % 3]
        joints.append((jt,) + tuple(j[1:]))
    sc["joints"] = joints
    return sc


def _distance_setup(sc):
    def setup(w):
        for i, j in enumerate(sc["joints"]):
            if j[0] == scenes.JOINT_DISTANCE:
                w.set_joint_params(i, [0.1])
            elif j[0] == scenes.JOINT_SOFT_DISTANCE:
                w.set_joint_params(i, [0.15, 400.0, 3.0])
    return setup


def test_distance_and_soft_distance_constraints_match_the_real_engine():
    """distance_constraint.cpp:7-35 (one row along the unnormalised separation, error 0.5 (d^2 - L^2) / dt) and
    soft_distance_constraint.cpp:8-67 (spring row with impulse limits and +-large error, damping row), their applied
    impulses, and their place in the constraint order (constraint.hpp:23-34: before hinges and points) - bit-identical."""
    sc = _distance_scene()
    ref, _ = _joint_lockstep(sc, 300, _distance_setup(sc))
    ji = ref.get_joint_impulses()
    kinds = np.array([j[0] for j in sc["joints"]])
    assert np.abs(ji[kinds == scenes.JOINT_DISTANCE][:, 0]).max() > 0
    assert np.abs(ji[kinds == scenes.JOINT_SOFT_DISTANCE][:, :2]).max() > 0


def _frame(axis_x):
    """Orthonormal basis (row-major 3x3) whose first COLUMN is axis_x."""
    x = np.asarray(axis_x, np.float64); x /= np.linalg.norm(x)
    y = np.cross(x, (0.0, 0.0, 1.0) if abs(x[2]) < 0.9 else (1.0, 0.0, 0.0)); y /= np.linalg.norm(y)
    z = np.cross(x, y)
    return np.stack([x, y, z], axis=1).astype(np.float32)


def _ragdoll_like_scene():
    """Chains whose links hang on cvjoints (twist limits, springs, friction, bending) with a cone limiting every other link -
    the constraint pair make_ragdoll builds its limbs from (ragdoll.cpp:471-914)."""
    sc = scenes.c5_chains(4, 6)
    joints, defs = [], []
    down = _frame((0.0, -1.0, 0.0))
    for i, j in enumerate(sc["joints"]):
        joints.append((scenes.JOINT_CVJOINT,) + tuple(j[1:]))
        cv = [-0.6, 0.8, 0.2, 0.15, 3.0, 0.01, 0.1, 2.0, 0.05, 0.0, -1.0, 0.0, 1.5, 0.02, 0.03] if i % 2 == 0 else \
             [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.8, 0.0, 0.0]
        defs.append((len(joints) - 1, down, down, cv))
    for i, j in enumerate(sc["joints"]):
        if i % 2 == 0 and j[1] != 0:
            a, b = j[1], j[2]
            joints.append((scenes.JOINT_CONE, a, b, (0.0, 0.0, 0.0), (0.0, -0.5, 0.0)) + tuple(j[5:]))
            defs.append((len(joints) - 1, down, np.eye(3, dtype=np.float32), [0.6, 0.9, 0.3, 40.0, 0.1]))
    sc["joints"] = joints
    rng = np.random.default_rng(17)
    # gentle: a bend of 90 degrees makes |rest x twist| round to just above 1 and std::asin return NaN - in the reference too
    sc["angvel"][1:] = (rng.normal(size=(len(sc["kind"]) - 1, 3)) * 0.6).astype(np.float32)
    return sc, defs


def test_cone_and_cvjoint_constraints_match_the_real_engine():
    """cone_constraint.cpp:12-104 (point-in-elliptic-cone limit row with restitution + bump stop) and
    cvjoint_constraint.cpp:12-302 (point rows, twist limit with angle tracking, bump stop, spring, friction / damping, bending
    friction and spring, position correction of twist and pivots) - state, applied impulses and tracked twist angle
    bit-identical with the real engine on swinging, twisting chains."""
    sc, defs = _ragdoll_like_scene()

    def setup(w):
        for j, fa, fb, p in defs:
            w.set_joint_definition(j, fa, fb, p)
    ref, _ = _joint_lockstep(sc, 300, setup)
    ji = ref.get_joint_impulses()
    kinds = np.array([j[0] for j in sc["joints"]])
    assert np.abs(ji[kinds == scenes.JOINT_CONE][:, :2]).max() > 0
    assert (np.abs(ji[kinds == scenes.JOINT_CVJOINT][:, 3:9]).max(axis=0) > 0).all()   # every optional cvjoint row acted


def _gravity_scene():
    """A few heavy bodies in free space attracting each other (gravity_constraint), no uniform gravity."""
    sc = scenes.c5_chains(1, 4)
    n = len(sc["kind"])
    sc["joints"] = [(scenes.JOINT_GRAVITY, a, b, (0, 0, 0), (0, 0, 0), (1, 0, 0), (1, 0, 0)) for a in range(1, n) for b in range(a + 1, n)]
    sc["mass"][1:] = (2e9, 5e9, 1e9, 3e9)[: n - 1]
    sc["gravity"] = np.zeros((n, 3), np.float32)
    sc["pos"][1:] = ((0, 10, 0), (3, 10, 0.5), (0.5, 12, 2), (-2, 9, -1))[: n - 1]
    sc["linvel"][1:] = ((0, 0, 0.1), (0, 0.15, 0), (-0.1, 0, 0), (0, 0, -0.1))[: n - 1]
    return sc


def test_gravity_constraint_matches_the_real_engine():
    """gravity_constraint.cpp:6-34: Newtonian attraction F = G / (l^2 inv_mA inv_mB) as a row with impulse limits +-F dt and
    a large error - orbiting heavy bodies, bit-identical incl. the applied impulse; first in the constraint order."""
    sc = _gravity_scene()
    ref, _ = _joint_lockstep(sc, 300, lambda w: None)
    assert np.abs(ref.get_joint_impulses()[:, 0]).min() > 0
    assert np.abs(ref.get_state()[2][1:]).max() > 0.15   # they did accelerate towards each other


def _generic_scene():
    """Chains on generic_constraints: limited / sprung / damped translation along the link and about all three axes."""
    sc = scenes.c5_chains(4, 5)
    fr = _frame((0.0, -1.0, 0.0))
    defs = []
    joints = []
    for i, j in enumerate(sc["joints"]):
        joints.append((scenes.JOINT_GENERIC,) + tuple(j[1:]))
        lin = [[1, -0.05, 0.08, 0.2, 0.02, 300.0, 0.01, 0.0, 50.0, 0.5],    # along the link: limits, bump stop, spring, friction / damping
               [1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],            # locked
               [1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.02, 0.0, 0.0, 0.3]]           # locked + friction row
        ang = [[1, -0.05, 0.06, 0.3, 0.02, 2.0, 0.01, 0.01, 0.5, 0.02],     # twist: everything
               [1, -0.4, 0.4, 0.0, 0.0, 0.0, 0.0, 0.0, 0.3, 0.0] if i % 2 == 0 else [1, 0, 0, 0, 0, 0, 0, 0, 0, 0],
               [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.005, 0.0, 0.2, 0.01]]         # free, sprung and damped
        defs.append((i, fr, fr, lin + ang))
    sc["joints"] = joints
    rng = np.random.default_rng(23)
    sc["angvel"][1:] = (rng.normal(size=(len(sc["kind"]) - 1, 3)) * 0.8).astype(np.float32)
    sc["angvel"][1::2, 1] += 4.0   # spin every other link about the chain axis: the twist limits and bump stops get hit
    return sc, defs


def test_generic_constraint_matches_the_real_engine():
    """generic_constraint.cpp:10-330: six degrees of freedom with limits (erp 0.9 on the linear ones), bump stops, springs and
    friction / damping rows, the angle conventions of the three angular degrees of freedom, the linear position correction and
    all 24 applied impulses - bit-identical with the real engine."""
    sc, defs = _generic_scene()

    def setup(w):
        for j, fa, fb, d in defs:
            w.set_generic_definition(j, fa, fb, d)
    ref = ob.RefWorld(); ref.add_bodies(sc)
    orc = ob.World(order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    setup(ref); setup(orc)
    for s in range(1, 301):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all(), (s, name)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, name, float(np.abs(a - b).max()))
        assert np.array_equal(ref.get_joint_impulses24().view(np.uint32), orc.get_joint_impulses24().view(np.uint32)), s
    ji = ref.get_joint_impulses24()
    acted = np.abs(ji).max(axis=0) > 0
    assert acted[[0, 1, 2, 3, 4, 8, 11, 12, 13, 14, 15, 16, 18, 22, 23]].all(), acted


def _mix_table_setup(w, n):
    """Three material ids over the bodies (the plane is id 0) and a mix table covering some of the pairs."""
    for i in range(n):
        w.set_material_id(i, 0 if i == 0 else 1 + i % 3)
    w.insert_material_mixing(0, 1, restitution=0.0, friction=0.9)
    w.insert_material_mixing(1, 2, restitution=0.4, friction=0.1)
    w.insert_material_mixing(2, 2, restitution=0.0, friction=0.3, roll=0.05, spin=0.02)
    w.insert_material_mixing(0, 3, restitution=0.0, friction=0.6, stiffness=5000.0, damping=70.0)


def test_material_mix_table_matches_the_real_engine():
    """material_mix_table (material_mixing.hpp:36-82, assign_material_properties collision_util.cpp:291-299, the restitution tag
    of a manifold constraint_util.cpp:90-100): an entry for a pair of material ids replaces every mixing rule - friction,
    restitution (incl. whether the restitution solver sees the manifold), rolling / spinning friction, soft contacts."""
    sc = scenes.box_pile(3, 3, 3, mixed=True)
    n = len(sc["kind"])
    sc["linvel"][1:] = (np.random.default_rng(4).normal(size=(n - 1, 3)) * (1.0, 0.5, 1.0)).astype(np.float32)
    ref = ob.RefWorld(); ref.add_bodies(sc)
    orc = ob.World(order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    _mix_table_setup(ref, n); _mix_table_setup(orc, n)
    frictions = set()
    for s in range(1, 201):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all(), (s, name)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, name, float(np.abs(a - b).max()))
        if s % 20 == 0:
            rm, om = _canon(ref.get_manifolds()), _canon(orc.get_manifolds())
            assert np.array_equal(rm["num_points"], om["num_points"]), s
            for fld in ("friction", "restitution", "normal_impulse"):
                assert np.array_equal(rm["pt"][fld], om["pt"][fld]), (s, fld)
            frictions |= set(int(round(float(x) * 100)) for x in rm["pt"]["friction"][rm["pt"]["friction"] > 0])
    assert {90, 30} <= frictions and 50 in frictions   # table entries and the sqrt rule both occurred


# ------------------------------------------------------------------------------------------------ the reference's rag doll
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("shape", ["capsule", "box", "cylinder"])
def test_ragdoll_template_is_what_the_engine_builds(shape):
    """tests/golden/ragdoll_*.npz (what the -m gpu tests and the bench build their figures from) against edyn::make_ragdoll
    (util/ragdoll.cpp:65-914) run by the real engine now: every body and constraint field bit for bit."""
    r = ob.RefWorld()
    fig = r.export_figure(*r.make_ragdoll(shape, pos=(0, 0, 0), height=1.7, weight=72.0))
    tpl = scenes.load_figure(os.path.join(GOLDEN, f"ragdoll_{shape}.npz"))
    assert sorted(fig) == sorted(tpl)
    for k in fig:
        assert fig[k].dtype == tpl[k].dtype and np.array_equal(fig[k].view(np.uint8), tpl[k].view(np.uint8)), k
    assert len(fig["kind"]) == 22 and len(fig["joint_type"]) == 36 and len(fig["exclusions"]) == 21


@pytest.mark.parametrize("shape", ["capsule", "box", "cylinder"])
def test_ragdolls_match_the_real_engine(shape):
    """Four of the reference's rag dolls (22 bodies on 36 cone / cvjoint / hinge constraints with bump stops, twist limits,
    friction and damping; shapeless twist bodies with explicit inertia; 21 collision exclusions) collapse onto the floor:
    positions, velocities, applied impulses and tracked angles bit-identical with the real engine for 300 steps."""
    sc = scenes.figures(scenes.load_figure(os.path.join(GOLDEN, f"ragdoll_{shape}.npz")), 2, 2)
    ref, _ = _joint_lockstep(sc, 300, lambda w: scenes.apply_figure_settings(w, sc))
    ji = ref.get_joint_impulses()
    kinds = np.array([j[0] for j in sc["joints"]])
    assert np.abs(ji[kinds == scenes.JOINT_CONE][:, 1]).max() > 0            # cone bump stops
    assert (np.abs(ji[kinds == scenes.JOINT_CVJOINT][:, [3, 4, 6, 7]]).max(axis=0) > 0).all()   # twist limit, bump stop, friction rows
    assert np.abs(ji[kinds == scenes.JOINT_HINGE][:, [6, 8]]).max() > 0      # knee / elbow bump stops and torque
    assert len(ref.get_manifolds()) > 40 and ref.get_state()[0][1:, 1].max() < 0.6   # everybody is lying on the floor


def _null_constraint_scene():
    """Two boxes far apart on the floor, one resting, one dropped from 6 m, tied by a null_constraint; a third box rests alone."""
    sc = scenes.box_pile(3, 1, 1)
    sc["pos"][1] = (0.0, 0.5, 0.0); sc["pos"][2] = (20.0, 6.0, 0.0); sc["pos"][3] = (40.0, 0.5, 0.0)
    sc["joints"] = [(scenes.JOINT_NULL, 1, 2, (0, 0, 0), (0, 0, 0), (1, 0, 0), (1, 0, 0))]
    return sc


def test_null_constraint_keeps_its_bodies_in_one_island_like_the_real_engine():
    """null_constraint (null_constraint.hpp:12-17): no rows, only an edge of the island graph - the resting box cannot fall asleep
    before the one it is tied to has landed and settled, while the untied box sleeps on time. Sleep flags, islands and state
    bit-identical with the real engine."""
    ref, orc, first_sleep = _lockstep(_null_constraint_scene(), 420, sleeping=True)
    assert first_sleep is not None
    a = ref.get_asleep()
    assert a[1] and a[2] and a[3] and ref.num_islands == 2 == orc.get_stats()["num_islands"]


def _com_scene():
    """Bodies whose centre of mass is not their shape's origin (rigidbody_def::center_of_mass): loaded boxes and a weighted sphere
    tumbling onto the floor, two of them hinged together and one hanging from a static anchor by a point constraint."""
    sc = scenes.box_pile(3, 2, 1)
    n = len(sc["kind"])
    sc["com"] = np.zeros((n, 3), np.float32)
    sc["com"][1] = (0.2, -0.1, 0.05); sc["com"][3] = (-0.15, 0.2, 0.1); sc["com"][5] = (0.0, -0.3, 0.0)
    sc["shape_type"][6] = scenes.SHAPE_SPHERE; sc["shape_param"][6] = (0.5, 0, 0, 0); sc["com"][6] = (0.0, -0.25, 0.0)
    sc["angvel"][1:] = np.random.default_rng(3).normal(size=(n - 1, 3)).astype(np.float32)
    sc["pos"][1:, 1] += 0.4
    sc["joints"] = [(scenes.JOINT_HINGE, 1, 2, (0.51, 0.0, 0.0), (-0.51, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0)),
                    (scenes.JOINT_POINT, 4, 5, (0.0, 0.6, 0.0), (0.0, -0.6, 0.0), (1.0, 0.0, 0.0), (1.0, 0.0, 0.0))]
    return sc


@pytest.mark.parametrize("bouncy", [False, True])
def test_center_of_mass_matches_the_real_engine(bouncy):
    """rigidbody_def::center_of_mass (rigidbody.cpp:56-87,517-548): the parallel-axis shift of the shape's inertia, position and
    velocity moved to the centre of mass, shapes / contact pivots / joint pivots in the frame of the origin (update_origins.cpp,
    position_solver.hpp:34-41: origins follow position corrections, and are otherwise refreshed once per step) - initial state,
    AABBs and world inertias, then 300 steps bit-identical with the real engine."""
    sc = _com_scene()
    if bouncy:   # the restitution solver anchors its pivots at the origins too (restitution_solver.cpp:166-220)
        sc["restitution"][:] = 0.6
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    ad, od = ref.get_derived(), orc.get_derived()
    assert np.array_equal(ad[0][1:].view(np.uint32), od[0][1:].view(np.uint32)) and np.array_equal(ad[1][1:].view(np.uint32), od[1][1:].view(np.uint32))
    for s in range(1, 301):
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.set_ext_restitution_walk(*ref.get_restitution_walk()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, name)
    assert np.array_equal(ref.get_joint_impulses().view(np.uint32), orc.get_joint_impulses().view(np.uint32))
    ad, od = ref.get_derived(), orc.get_derived()
    assert np.array_equal(ad[0][1:].view(np.uint32), od[0][1:].view(np.uint32))
    # the loaded bodies came to rest heavy side down: their centres of mass sit below their origins
    assert bouncy or ref.get_state()[0][6, 1] < 0.35


def test_center_of_mass_moved_on_a_running_world_matches_the_real_engine():
    """edyn::set_center_of_mass between steps (rigidbody.cpp:364-370 -> apply_center_of_mass :517-548): position and linear velocity
    follow the centre of mass, the origin - with every contact and joint pivot - stays, the inertia is not touched; giving a body its
    first offset, changing one and removing one (offset zero drops the origin), each bit-identical with the real engine afterwards."""
    sc = _com_scene()
    ref = ob.RefWorld(vel_iters=10); ref.add_bodies(sc)
    orc = ob.World(vel_iters=10, order=ob.ORDER_EXTERNAL); orc.add_bodies(sc)
    edits = {40: (2, (0.1, 0.2, 0.0)), 80: (1, (0.0, 0.0, 0.0)), 120: (6, (0.1, -0.1, 0.2))}   # first offset, removal, change
    for s in range(1, 241):
        if s in edits:
            body, com = edits[s]
            ref.move_center_of_mass(body, com); orc.move_center_of_mass(body, com)
            for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, "edit", name)
        ref.step(1)
        orc.set_ext_order(*ref.get_solve_order()); orc.set_ext_restitution_walk(*ref.get_restitution_walk()); orc.step(1)
        assert not orc.ext_order_mismatch(), s
        for name, a, b in zip(("pos", "orn", "linvel", "angvel"), ref.get_state(), orc.get_state()):
            assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, name)
    ad, od = ref.get_derived(), orc.get_derived()
    assert np.array_equal(ad[0][1:].view(np.uint32), od[0][1:].view(np.uint32))
