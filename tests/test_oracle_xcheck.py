"""Cross-check the oracle's restated leaf functions against the REAL reference functions, compiled from the
EnTT-free translation units where they lie under /root/reference (oracle/_ref/libedynref.so, built by
`make -C oracle ref`). Bit-exact on random inputs. Skipped with a message where _ref was never built."""
import numpy as np
import pytest
from oracle import binding as ob

ref = ob.ref_leaf()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libedynref.so not built (needs /root/reference at build time)")
orc = ob.leaf()
rng = np.random.default_rng(12345)


def rq():
    q = rng.normal(size=4).astype(np.float32)
    return q / np.float32(np.linalg.norm(q))


def test_plane_space_and_rotate():
    for _ in range(2000):
        n = rng.normal(size=3).astype(np.float32); n /= np.float32(np.linalg.norm(n))
        for a, b in zip(orc.plane_space(n), ref.plane_space(n)):
            assert np.array_equal(a, b)
        q = rq(); v = rng.normal(size=3).astype(np.float32)
        assert np.array_equal(orc.rotate(q, v), ref.rotate(q, v))


def test_integrate():
    """quaternion.cpp:7-22. The Taylor branch (|w| < 0.001) is bit-exact. In the sin/cos branch the oracle (and the GPU)
    use correctly rounded fp32 sin/cos (double evaluation, one rounding) instead of the C library's sinf/cosf, whose
    last bit is library dependent: vs this container's glibc that differs in <2 % of calls, by at most 2 ulp."""
    for _ in range(1000):
        q = rq(); w = (rng.normal(size=3) * 3e-4).astype(np.float32)
        assert np.array_equal(orc.integrate(q, w, 1 / 60), ref.integrate(q, w, 1 / 60))
    total = mismatch = 0
    for scale in (1e-2, 1.0, 20.0):
        for _ in range(1000):
            q = rq(); w = (rng.normal(size=3) * scale).astype(np.float32)
            for dt in (1 / 60, -1 / 60):
                a, b = orc.integrate(q, w, dt), ref.integrate(q, w, dt)
                total += 1
                if not np.array_equal(a, b):
                    mismatch += 1
                    assert np.abs(a - b).max() <= 2 * np.finfo(np.float32).eps
    assert mismatch / total < 0.02


def test_intersect_line_aabb_random():
    for _ in range(5000):
        p0 = rng.uniform(-2, 2, 2).astype(np.float32); p1 = rng.uniform(-2, 2, 2).astype(np.float32)
        if rng.random() < 0.2: p1[0] = p0[0]
        if rng.random() < 0.2: p1[1] = p0[1]
        he = rng.uniform(0.1, 1.5, 2).astype(np.float32)
        a = orc.intersect_line_aabb(p0, p1, -he, he); b = ref.intersect_line_aabb(p0, p1, -he, he)
        assert a[0] == b[0]
        assert np.array_equal(a[1][:a[0]], b[1][:b[0]])


def test_insertion_point_index_random():
    for _ in range(5000):
        n = int(rng.integers(0, 5))
        pts = rng.uniform(-0.5, 0.5, (4, 3)).astype(np.float32)
        mode = rng.random()
        if mode < 0.3: pts[:, 1] = 0.5                 # coplanar (face contacts)
        if mode < 0.1 and n >= 3: pts[2] = (pts[0] + pts[1]) * np.float32(0.5)   # collinear
        newp = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
        if mode < 0.3: newp[1] = 0.5
        if rng.random() < 0.2 and n > 0: newp = pts[int(rng.integers(0, n))] + np.float32(0.001)
        assert orc.insertion_point_index(pts, n, newp) == ref.insertion_point_index(pts, n, newp)


def test_closest_segment_segment_random():
    for _ in range(3000):
        p1, q1, p2, q2 = (rng.uniform(-1, 1, 3).astype(np.float32) for _ in range(4))
        if rng.random() < 0.3: q2 = p2 + (q1 - p1) * np.float32(rng.uniform(0.2, 2))   # parallel segments
        a = orc.closest_segment_segment(p1, q1, p2, q2); b = ref.closest_segment_segment(p1, q1, p2, q2)
        assert a[0] == b[0] and a[3] == b[3]
        assert np.array_equal(a[1][:2], b[1][:2]) and np.array_equal(a[2][:6], b[2][:6])
        if a[3] == 2:
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_box_support_feature_and_projection():
    for _ in range(5000):
        h = rng.uniform(0.1, 2, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32); d /= np.float32(np.linalg.norm(d))
        if rng.random() < 0.4:   # near-axis directions exercise the edge/face tolerance branches
            d = np.zeros(3, np.float32); d[int(rng.integers(0, 3))] = rng.choice([-1, 1]); d += (rng.normal(size=3) * 0.003).astype(np.float32)
        assert orc.box_support_feature(h, d, 0.005) == ref.box_support_feature(h, d, 0.005)
        pos = rng.uniform(-3, 3, 3).astype(np.float32); q = rq()
        assert orc.box_support_projection(h, pos, q, d) == ref.box_support_projection(h, pos, q, d)


def test_prepare_row_solve_apply():
    for _ in range(3000):
        rd = np.zeros(38, np.float32)
        rd[0:12] = rng.normal(size=12)
        rd[12:14] = rng.uniform(0, 2, 2)
        for k in (14, 23):
            m = rng.normal(size=(3, 3)); m = (m @ m.T + np.eye(3)).astype(np.float32)
            rd[k:k + 9] = m.reshape(9)
        rd[32] = rng.normal(); rd[33] = 0.2; rd[34] = rng.uniform(0, 0.5)
        lim = sorted(rng.normal(size=2) * 3)
        rd[35], rd[36] = (0, 1e18) if rng.random() < 0.5 else lim
        rd[37] = rng.uniform(0, 1)
        vel = rng.normal(size=12).astype(np.float32); delta = rng.normal(size=12).astype(np.float32) * np.float32(0.1)
        a = orc.row_prepare_solve(rd, vel, delta); b = ref.row_prepare_solve(rd, vel, delta)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
