"""Analytic / physical invariants of the oracle (the reference has no multi-step known answers,
so the restatement is additionally pinned by physics) and agreement between its two solver orders."""
import math
import numpy as np
import pytest
from oracle import binding as ob
from edyn_amd import scenes

HALF = [0.5, 0.5, 0.5, 0]


def test_free_fall_matches_semi_implicit_euler():
    w = ob.World()
    w.add_body(ob.KIND_DYNAMIC, pos=(0, 100, 0), inertia=np.eye(3).reshape(9))
    n = 60
    w.step(n)
    pos, _, v, _ = w.get_state()
    dt = np.float32(1 / 60); g = np.float32(-9.8)
    vy = np.float32(0); y = np.float32(100)
    for _ in range(n):
        vy = vy + g * dt
        y = y + vy * dt
    assert pos[0, 1] == y and v[0, 1] == vy


@pytest.mark.parametrize("order", [ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED])
def test_box_rests_on_plane(order):
    w = ob.World(order=order)
    w.add_body(ob.KIND_STATIC, shape_type=ob.SHAPE_PLANE, shape_param=[0, 1, 0, 0])
    w.add_body(ob.KIND_DYNAMIC, pos=(0, 0.6, 0), shape_type=ob.SHAPE_BOX, shape_param=HALF)
    w.step(180)
    pos, orn, v, av = w.get_state()
    assert abs(pos[1, 1] - 0.5) < 2e-3
    assert np.abs(v[1]).max() < 1e-3 and np.abs(av[1]).max() < 1e-3
    m = w.get_manifolds()
    assert len(m) == 1 and m["num_points"][0] == 4
    # the four normal impulses carry the weight: sum = m*g*dt
    assert abs(m["pt"]["normal_impulse"][0].sum() - 9.8 / 60) < 1e-3


@pytest.mark.parametrize("order", [ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED])
def test_sphere_rests_and_box_stack(order):
    w = ob.World(order=order, vel_iters=10)
    w.add_body(ob.KIND_STATIC, shape_type=ob.SHAPE_PLANE, shape_param=[0, 1, 0, 0])
    w.add_body(ob.KIND_DYNAMIC, pos=(3, 0.55, 0), shape_type=ob.SHAPE_SPHERE, shape_param=[0.5, 0, 0, 0])
    for k in range(5):
        w.add_body(ob.KIND_DYNAMIC, pos=(0, 0.505 + 1.005 * k, 0), shape_type=ob.SHAPE_BOX, shape_param=HALF)
    w.step(240)
    pos = w.get_state()[0]
    assert abs(pos[1, 1] - 0.5) < 3e-3
    for k in range(5):
        assert abs(pos[2 + k, 1] - (0.5 + k)) < 0.02, (k, pos[2 + k])


def test_c1_columns_settle():
    """C1 reduced to 3x3 columns of 5 boxes (same pitch): separate islands, tops at 4.5, bottoms at 0.5."""
    s = scenes._empty(46)
    scenes._add_plane(s)
    pos, _ = scenes._lattice(3, 5, 3, pitch_h=1.05, pitch_v=1.05, y0=0.55, brick=False)
    s["pos"][1:] = pos; s["shape_type"][1:] = scenes.SHAPE_BOX; s["shape_param"][1:, :3] = 0.5
    scenes._jitter(s, 1, 45)
    w = ob.World(order=ob.ORDER_SEQUENTIAL)
    w.add_bodies(s)
    w.step(300)
    p = w.get_state()[0]
    assert np.isfinite(p).all()
    assert w.get_stats()["num_islands"] == 9
    ys = np.sort(p[1:, 1])
    assert np.abs(ys[:9] - 0.5).max() < 0.02 and np.abs(ys[-9:] - 4.5).max() < 0.02


def test_orders_agree_on_invariants():
    """Reference (sequential) order vs the GPU's coloured order: different Gauss-Seidel sweeps, same physics."""
    scene = scenes.pyramid(5)
    res = []
    for order in (ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED):
        w = ob.World(order=order, vel_iters=10)
        w.add_bodies(scene)
        w.step(300)
        pos, _, v, av = w.get_state()
        m = w.get_manifolds()
        pen = min(float(m["pt"]["distance"][i, :k].min()) for i, k in enumerate(m["num_points"]) if k)
        res.append((pos, v, pen))
    (p0, v0, pen0), (p1, v1, pen1) = res
    assert pen0 > -0.02 and pen1 > -0.02                      # max penetration <= 0.02 m
    assert np.abs(v0[1:]).max() < 0.05 and np.abs(v1[1:]).max() < 0.05   # both settled
    assert abs(p0[1:, 1].mean() - p1[1:, 1].mean()) < 1e-3     # mean resting height
    assert np.abs(p0 - p1).max() < 0.02


def test_hinge_pendulum_period():
    """Point mass on a hinge: small-angle period of a physical pendulum, T = 2*pi*sqrt(I_p/(m g L))."""
    w = ob.World(vel_iters=20, pos_iters=3)
    L = 1.0
    I = np.diag([0.01, 0.01, 0.01]).astype(np.float32)
    a = w.add_body(ob.KIND_STATIC, pos=(0, 5, 0))
    th0 = 0.1
    b = w.add_body(ob.KIND_DYNAMIC, pos=(L * math.sin(th0), 5 - L * math.cos(th0), 0),
                   orn=(0, 0, math.sin(th0 / 2), math.cos(th0 / 2)), inertia=I.reshape(9))
    w.add_joint(ob.JOINT_HINGE, a, b, (0, 0, 0), (0, L, 0), (0, 0, 1), (0, 0, 1))
    xs = []
    for _ in range(240):
        w.step(1)
        xs.append(w.get_state()[0][b, 0])
    xs = np.array(xs)
    crossings = np.nonzero((xs[:-1] > 0) & (xs[1:] <= 0))[0]
    assert len(crossings) >= 2
    period = (crossings[1] - crossings[0]) / 60.0
    expect = 2 * math.pi * math.sqrt((0.01 + L * L) / (9.8 * L))
    assert abs(period - expect) < 0.06
    pos = w.get_state()[0][b]
    assert abs(np.linalg.norm(pos - np.array([0, 5, 0])) - L) < 2e-3   # joint holds


def _rot(q, v):
    q = np.asarray(q, np.float64); v = np.asarray(v, np.float64)
    r = q[:3]
    return v + np.cross(2 * r, np.cross(r, v) + q[3] * v)


def test_chain_joint_anchors_stay_together():
    sc = scenes.c5_chains(2, 6)
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
    w.add_bodies(sc)
    w.step(120)
    p, q, _, _ = w.get_state()
    assert np.isfinite(p).all()
    assert p[:, 1].min() < 9.5          # the chains did swing down
    for (jt, a, b, pa, pb, _, _) in sc["joints"]:
        wa = p[a] + _rot(q[a], pa); wb = p[b] + _rot(q[b], pb)
        assert np.linalg.norm(wa - wb) < 0.02, (jt, a, b)


# ------------------------------------------------------------------ island sleeping (island_manager.cpp:524-623)
def _stack2():
    """A floor and two boxes stacked exactly on top of each other (box_pile offsets alternate layers like bricks)."""
    from edyn_amd import scenes
    s = scenes.box_pile(1, 2, 1)
    s["pos"][2] = s["pos"][1] + np.float32([0.0, 1.005, 0.0])
    return s


def test_island_falls_asleep_after_two_seconds_and_freezes():
    from edyn_amd import scenes
    for order in (ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED):
        o = ob.World(vel_iters=10, pos_iters=3, order=order)
        o.add_bodies(scenes.box_pile(1, 1, 1))
        o.set_sleeping(True)
        first = None
        for k in range(400):
            o.step(1)
            if first is None and o.get_asleep()[1]:
                first = k
        assert first is not None and 120 <= first <= 200, first     # settle, then island_time_to_sleep = 2 s = 120 steps
        p0, q0, v0, w0 = o.get_state()
        assert not v0[1].any() and not w0[1].any()                    # put_to_sleep zeroes the velocities
        o.step(50)
        p1, q1, v1, w1 = o.get_state()
        assert np.array_equal(p0, p1) and np.array_equal(q0, q1) and not v1.any()
        assert not o.get_asleep()[0]                                  # the static floor never carries the tag


def test_sleeping_disabled_body_keeps_its_island_awake_and_default_is_off():
    from edyn_amd import scenes
    o = ob.World(vel_iters=10, pos_iters=3, order=ob.ORDER_COLOURED)
    o.add_bodies(_stack2())
    o.set_sleeping(True)
    o.set_sleeping_disabled(2, True)       # the upper box; the lower one shares its island
    o.step(400)
    assert not o.get_asleep().any()
    o1 = ob.World(vel_iters=10, pos_iters=3, order=ob.ORDER_COLOURED)
    o1.add_bodies(_stack2())
    o1.set_sleeping(True)
    o1.step(400)
    assert o1.get_asleep()[1] and o1.get_asleep()[2]
    o2 = ob.World(vel_iters=10, pos_iters=3, order=ob.ORDER_COLOURED)
    o2.add_bodies(_stack2())
    o2.step(400)
    assert not o2.get_asleep().any()       # sleeping is opt-in


def test_new_contact_wakes_a_sleeping_island():
    from edyn_amd import scenes
    o = ob.World(vel_iters=10, pos_iters=3, order=ob.ORDER_COLOURED)
    base = scenes.box_pile(1, 1, 1)
    o.add_bodies(base)
    o.set_sleeping(True)
    o.step(300)
    assert o.get_asleep()[1]
    falling = {k: (v[1:2].copy() if isinstance(v, np.ndarray) and len(v) == 2 else v) for k, v in base.items()}
    falling["pos"] = falling["pos"] + np.float32([0.05, 3.0, 0.0])
    falling.pop("joints", None)
    o.add_bodies(falling)
    woke = None
    for k in range(120):
        o.step(1)
        if woke is None and not o.get_asleep()[1]:
            woke = k
    assert woke is not None and 20 < woke < 80, woke                  # ~0.75 s of free fall
    assert len(o.get_manifolds()) == 2
    o.step(400)
    assert o.get_asleep()[1] and o.get_asleep()[2]                     # the stack goes back to sleep as one island
    assert abs(o.get_state()[0][2, 1] - 1.5) < 0.02


def test_serial_colour_bucket_holds_what_62_colours_cannot():
    """A dynamic plate on 81 bricks: its contacts beyond the 62 conflict-free colours share the serial bucket (colour 62,
    solved one after the other after the parallel colours). Colours below 62 stay conflict free, and the plate rests as it
    does in the reference's sequential order."""
    s = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scenes.box_pile(9, 1, 9).items()}
    n = len(s["kind"])
    top = float(s["pos"][:, 1].max()) + 0.5
    plate = dict(kind=scenes.KIND_DYNAMIC, pos=(float(s["pos"][1:, 0].mean()), top + 0.26, float(s["pos"][1:, 2].mean())),
                 orn=(0, 0, 0, 1), linvel=(0, 0, 0), angvel=(0, 0, 0), mass=40.0, shape_type=scenes.SHAPE_BOX,
                 shape_param=(5.1, 0.25, 5.1, 0), friction=0.5, restitution=0.0,
                 group=np.uint64(0xFFFFFFFFFFFFFFFF), mask=np.uint64(0xFFFFFFFFFFFFFFFF))
    for k, v in plate.items():
        s[k] = np.concatenate([s[k], np.asarray(v, dtype=s[k].dtype).reshape((1,) + s[k].shape[1:])], axis=0)
    for k in ("inertia", "has_inertia", "gravity"):
        s.pop(k, None)
    res = []
    for order in (ob.ORDER_SEQUENTIAL, ob.ORDER_COLOURED):
        w = ob.World(order=order, vel_iters=10)
        w.add_bodies(s)
        w.step(120)
        res.append((w.get_state(), w.get_manifolds()))
    (st0, _), (st1, m) = res
    live = m["num_points"] > 0
    on = live & (m["body"] == n).any(axis=1)
    assert on.sum() == 81 and (m["colour"][on] == 62).sum() >= 81 - 62
    for c in range(62):                                   # parallel colours: no dynamic body twice
        bodies = m["body"][live & (m["colour"] == c)].ravel()
        bodies = bodies[s["kind"][bodies] == ob.KIND_DYNAMIC]
        assert len(bodies) == len(set(bodies.tolist())), c
    assert np.isfinite(st1[0]).all() and abs(st1[0][n, 1] - st0[0][n, 1]) < 2e-3 and np.abs(st1[2][n]).max() < 0.02
