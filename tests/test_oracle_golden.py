"""Pin the CPU oracle against every golden vector the reference's own tests hold for the hot path
(SURVEY.md §8(i)). These are the reference's known-answer tests, re-expressed over the oracle's C ABI."""
import math
import numpy as np
import pytest
from oracle import binding as ob

FLT_EPS = np.finfo(np.float32).eps
HALF = [0.5, 0.5, 0.5, 0]
IDENT = [0, 0, 0, 1]


def _set_equal(points, expected):
    points = [tuple(p) for p in points]
    for e in expected:
        assert any(max(abs(p[i] - e[i]) for i in range(3)) <= FLT_EPS for p in points), (e, points)


def test_collide_box_box_face_face():
    # test/edyn/collision/test_collision.cpp:13-43
    r = ob.collide(ob.SHAPE_BOX, HALF, [0, 0, 0], IDENT, ob.SHAPE_BOX, HALF, [0, 1.0, 0], IDENT, 0.02)
    assert len(r) == 4
    _set_equal(r[:, 0:3], [(0.5, 0.5, 0.5), (-0.5, 0.5, 0.5), (-0.5, 0.5, -0.5), (0.5, 0.5, -0.5)])


def test_collide_box_box_face_edge():
    # test/edyn/collision/test_collision.cpp:45-90: B rotated pi/4 about x, raised 0.2
    a = math.pi / 4
    q = [math.sin(a / 2), 0, 0, math.cos(a / 2)]
    r = ob.collide(ob.SHAPE_BOX, HALF, [0, 0, 0], IDENT, ob.SHAPE_BOX, HALF, [0, 1.2, 0], q, 0.02)
    assert len(r) == 2
    _set_equal(r[:, 0:3], [(0.5, 0.5, 0), (-0.5, 0.5, 0)])
    _set_equal(r[:, 3:6], [(0.5, -0.5, 0.5), (-0.5, -0.5, 0.5)])


def test_collide_polyhedron_sphere():
    # test/edyn/collision/test_collision.cpp:92-147: the unit box as a convex mesh against a sphere of radius 0.5, both argument orders
    import meshes
    meshes.registered()   # id 0 = make_box_mesh({0.5, 0.5, 0.5})
    P, S = ob.SHAPE_POLYHEDRON, ob.SHAPE_SPHERE
    r = ob.collide(P, [0, 0, 0, 0], [0.5, 0.5, 0.5], IDENT, S, [0.5, 0, 0, 0], [0.5, 1.4, 0.5], IDENT, 1e18)
    assert len(r) == 1
    np.testing.assert_allclose(r[0, 6:9], (0, -1, 0), atol=1e-6)
    np.testing.assert_allclose(r[0, 0:3], (0, 0.5, 0), atol=1e-6)
    np.testing.assert_allclose(r[0, 3:6], (0, -0.5, 0), atol=1e-6)
    assert abs(r[0, 9] + 0.1) < 1e-6
    r = ob.collide(S, [0.5, 0, 0, 0], [1.5, 1.5, 0.5], IDENT, P, [0, 0, 0, 0], [0.5, 0.5, 0.5], IDENT, 1e18)
    assert len(r) == 1
    h = 0.707107
    np.testing.assert_allclose(r[0, 6:9], (h, h, 0), atol=1e-5)
    np.testing.assert_allclose(r[0, 0:3], (-h / 2, -h / 2, 0), atol=1e-5)
    np.testing.assert_allclose(r[0, 3:6], (0.5, 0.5, 0), atol=1e-5)
    assert abs(r[0, 9] - 0.2071067812) < 1e-5


def test_mesh_centroid_and_volume():
    # test/edyn/shapes/test_centroid.cpp:4-51, test_shape_volume.cpp:5-13: initialize() leaves the centroid of the unit box mesh at the
    # origin; an off-centre prism is moved there
    import meshes
    lib, _ = meshes.registered()
    v = ob.mesh_get(0, "vertices")
    assert np.array_equal(np.abs(v), np.full((8, 3), 0.5, np.float32))
    v4 = ob.mesh_get(4, "vertices")   # prism(6) built at offset (0.2, -0.1, 0.05)
    np.testing.assert_allclose(v4.mean(axis=0), 0, atol=1e-6)
    np.testing.assert_allclose(lib[4]["vertices"].mean(axis=0), (0.2, -0.1, 0.05), atol=1e-6)


@pytest.mark.parametrize("p0,p1,bmin,bmax,n,s", [
    ((0, 0.5), (1, 1.5), (-1, -0.5), (2, 1), 2, (-1, 0.5)),      # test_geom.cpp:3-13
    ((2, 1), (1, 1.5), (-1, -0.5), (2, 1), 1, (0,)),             # :15-24
    ((0, 0), (1, 1.5), (-1, 0.5), (0, 1), 0, ()),                # :26-34
    ((1, 1), (1, -1), (-2, -0.5), (1, 0.5), 2, (0.75, 0.25)),    # :36-46 vertical line
    ((0, -0.25), (1, -0.25), (-2, -0.5), (1, 0.5), 2, (-2, 1)),  # :48-58 horizontal line
])
def test_intersect_line_aabb(p0, p1, bmin, bmax, n, s):
    got_n, got_s = ob.leaf().intersect_line_aabb(p0, p1, bmin, bmax)
    assert got_n == n
    for i, e in enumerate(s):
        assert abs(float(got_s[i]) - e) <= 4 * FLT_EPS * max(1.0, abs(e))


def test_apply_gravity_10_calls():
    # test/edyn/sys/test_apply_gravity.cpp:4-23: v == g*dt*10 after 10 steps (no contacts, amorphous body)
    dt = np.float32(0.1666)
    w = ob.World(dt=float(dt), vel_iters=8, pos_iters=3, gravity=(0, -9.8, 0))
    w.add_body(ob.KIND_DYNAMIC, inertia=np.eye(3).reshape(9))
    w.step(10)
    v = w.get_state()[2][0]
    expect = np.float32(-9.8) * dt * np.float32(10)
    assert abs(v[1] - expect) <= 4 * np.spacing(np.float32(abs(expect)))
    assert v[0] == 0 and v[2] == 0


def test_should_collide_truth_table():
    # test/edyn/collision/test_broadphase.cpp:4-34 (filters; the exclusion list is outside the hot-path scope)
    ALL = 2**64 - 1
    first = (0x1, ALL & ~0x2)
    second = (0x2, ALL & ~0x1)
    third = (ALL, ALL)
    assert ob.should_collide(ALL, ALL, ALL, ALL)
    assert not ob.should_collide(*first, *second)
    assert ob.should_collide(*first, *third)
    assert ob.should_collide(*second, *third)


def test_connected_components():
    # test/edyn/core/test_entity_graph.cpp:54-97: 2 nodes joined by parallel edges + 1 isolated node -> 2 components
    w = ob.World()
    I = np.eye(3).reshape(9)
    a = w.add_body(ob.KIND_DYNAMIC, pos=(0, 0, 0), inertia=I)
    b = w.add_body(ob.KIND_DYNAMIC, pos=(1, 0, 0), inertia=I)
    c = w.add_body(ob.KIND_DYNAMIC, pos=(5, 0, 0), inertia=I)
    w.add_joint(ob.JOINT_POINT, a, b, (0.5, 0, 0), (-0.5, 0, 0))
    w.add_joint(ob.JOINT_POINT, a, b, (0.5, 0.1, 0), (-0.5, 0.1, 0))
    w.run_stage(2)
    isl = w.get_derived()[2]
    assert isl[a] == isl[b] and isl[c] != isl[a]
    assert w.get_stats()["num_islands"] == 2


def test_static_nodes_do_not_connect():
    # comp/island.hpp:34-41: non-procedural nodes are shared, never merge islands
    w = ob.World()
    w.add_body(ob.KIND_STATIC, shape_type=ob.SHAPE_PLANE, shape_param=[0, 1, 0, 0])
    w.add_body(ob.KIND_DYNAMIC, pos=(0, 0.5, 0), shape_type=ob.SHAPE_BOX, shape_param=HALF)
    w.add_body(ob.KIND_DYNAMIC, pos=(5, 0.5, 0), shape_type=ob.SHAPE_BOX, shape_param=HALF)
    w.step(2)
    assert w.get_stats()["num_islands"] == 2
    assert len(w.get_manifolds()) == 2
