"""Host-side logic that needs no GPU: scene generators, the fixed-dt accumulator, island sharding."""
import numpy as np
from edyn_amd import scenes
from edyn_amd.world import fixed_step_plan
from edyn_amd.parallel import shard_range


def test_splitmix_is_deterministic_and_uniform():
    a = scenes.splitmix64_uniform(10000); b = scenes.splitmix64_uniform(10000)
    assert np.array_equal(a, b)
    assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.02
    # first SplitMix64 output for seed 0x9E3779B97F4A7C15 (state = 2*gamma): top 24 bits as a float
    z = (0x9E3779B97F4A7C15 * 2) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    z ^= z >> 31
    assert a[0] == np.float32((z >> 40) / float(1 << 24))


def test_headline_scene_shape():
    s = scenes.box_pile(32, 32, 32)
    assert len(s["kind"]) == 32769 and s["kind"][0] == scenes.KIND_STATIC and (s["kind"][1:] == 0).all()
    y = s["pos"][1:, 1]
    assert np.isclose(y.min(), 0.505) and np.isclose(y.max(), 0.505 + 31 * 1.005)
    assert np.abs(np.linalg.norm(s["orn"], axis=1) - 1).max() < 1e-6
    m = scenes.box_pile(4, 4, 4, mixed=True)
    assert set(np.unique(m["shape_type"][1:])) == {scenes.SHAPE_BOX, scenes.SHAPE_SPHERE}


def test_c4_shards_tile_the_full_scene():
    full = scenes.mini_piles(4, 4)
    parts = [scenes.mini_piles(4, 4, first_site=f, num_sites=c) for f, c in (shard_range(16, r, 3) for r in range(3))]
    pos = np.concatenate([p["pos"][1:] for p in parts]); orn = np.concatenate([p["orn"][1:] for p in parts])
    assert np.array_equal(pos, full["pos"][1:]) and np.array_equal(orn, full["orn"][1:])


def test_shard_range_balanced_and_covering():
    for total in (1, 7, 8, 4096):
        for ws in (1, 2, 3, 8):
            rs = [shard_range(total, r, ws) for r in range(ws)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == total
            assert all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(ws - 1))
            assert max(c for _, c in rs) - min(c for _, c in rs) <= 1


def test_fixed_step_accumulator():
    dt = 1 / 60
    steps, acc, sdt = fixed_step_plan(0.0, 0.05, dt, 10)          # 3 steps, remainder kept
    assert steps == 3 and abs(acc - (0.05 - 3 * dt)) < 1e-12
    steps, acc, sdt = fixed_step_plan(acc, dt - acc + 1e-9, dt, 10)
    assert steps == 1
    steps, acc, sdt = fixed_step_plan(0.0, 1.0, dt, 10)           # clamp to max_steps_per_update
    assert steps == 10 and acc < dt
    steps, acc, sdt = fixed_step_plan(0.0, -5.0, dt, 10)          # negative elapsed clamps to 0
    assert steps == 0 and acc == 0.0


def test_convex_library_meshes_are_closed_and_wound_outwards():
    """edyn_amd.scenes.convex_library: every mesh is a closed 2-manifold (each edge shared by exactly two faces, once in each
    direction), its faces wind counter-clockwise seen from outside (positive volume), no face has more vertices than a support
    polygon holds (32) - what edynhip_create_convex_mesh insists on."""
    from edyn_amd import scenes
    for m in scenes.convex_library():
        v, idx, faces = m["vertices"].astype(np.float64), m["indices"], m["faces"]
        directed = {}
        vol = 0.0
        for first, count in faces:
            f = idx[first:first + count]
            assert 3 <= count <= 32
            for k in range(count):
                e = (int(f[k]), int(f[(k + 1) % count]))
                assert e not in directed
                directed[e] = 1
            for k in range(1, count - 1):
                vol += np.dot(v[f[0]], np.cross(v[f[k]], v[f[k + 1]])) / 6
        assert all((b, a) in directed for a, b in directed)   # every edge has its opposite: closed, consistently oriented
        assert vol > 0.01
        assert len(np.unique(idx)) == len(v)                  # no unused vertex


def test_polyhedron_heap_scene_and_its_subsets_carry_the_mesh_list():
    from edyn_amd import scenes
    a, b = scenes.polyhedron_heap(4, 3, 4), scenes.polyhedron_heap(4, 3, 4)
    assert all(np.array_equal(a[k], b[k]) for k in a if k != "meshes")                    # deterministic
    assert (a["shape_type"][1:] == scenes.SHAPE_POLYHEDRON).all() and a["shape_type"][0] == scenes.SHAPE_PLANE
    assert set(a["shape_param"][1:, 0].astype(int)) == set(range(len(a["meshes"])))      # every mesh is used
    assert np.allclose(np.linalg.norm(a["orn"], axis=1), 1.0, atol=1e-6)
    sub = scenes.subset(a, np.array([0, 3, 7, 11]))
    assert sub["meshes"] is a["meshes"] and len(sub["kind"]) == 4                         # ids stay positions in the same list
