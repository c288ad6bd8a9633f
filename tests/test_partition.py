"""The island partitioner and the island-box sweep of the library (edynhip_partition_islands, edynhip_island_boxes_overlap:
host code of edyn_amd/csrc/multi.hip, shared by the single-process multi-GPU world and by ShardedWorld) - no GPU needed.
Checked against plain numpy restatements of the same rules, at world sizes up to 8."""
import numpy as np
import pytest
from edyn_amd import scenes
from edyn_amd.parallel import partition_islands, island_boxes_overlap, island_boxes


def lpt_numpy(labels, kind, weights, world_size):
    """longest-processing-time-first: islands by descending weight, ties by ascending label, each to the lightest rank (lowest on ties)"""
    labels = np.asarray(labels); kind = np.asarray(kind); weights = np.asarray(weights, np.float64)
    dyn = kind == scenes.KIND_DYNAMIC
    rank_of = np.full(len(kind), -1, np.int32)
    if not dyn.any():
        return rank_of
    isl, inv = np.unique(labels[dyn], return_inverse=True)
    w = np.bincount(inv, weights=weights[dyn], minlength=len(isl))
    load = np.zeros(world_size); owner = np.zeros(len(isl), np.int32)
    for k in np.lexsort((isl, -w)):
        r = int(np.argmin(load)); owner[k] = r; load[r] += w[k]
    rank_of[dyn] = owner[inv]
    return rank_of


@pytest.mark.parametrize("world_size", [1, 2, 3, 8])
def test_partitioner_matches_the_rule_and_balances(world_size):
    rng = np.random.default_rng(7 + world_size)
    n = 5000
    kind = np.where(rng.random(n) < 0.05, scenes.KIND_STATIC, scenes.KIND_DYNAMIC).astype(np.int32)
    # islands of very different sizes: label = lowest body index of the island
    island_of = np.maximum.accumulate(np.where(rng.random(n) < 0.04, np.arange(n), 0))
    weights = 1.0 + rng.integers(0, 12, n)
    got = partition_islands(island_of, kind, weights, world_size)
    want = lpt_numpy(island_of, kind, weights, world_size)
    assert np.array_equal(got, want)
    assert (got[kind != scenes.KIND_DYNAMIC] == -1).all() and (got[kind == scenes.KIND_DYNAMIC] >= 0).all()
    # every island on one rank
    for lab in np.unique(island_of[kind == scenes.KIND_DYNAMIC]):
        assert len(np.unique(got[(island_of == lab) & (kind == scenes.KIND_DYNAMIC)])) == 1
    # LPT bound: no rank carries more than the mean plus the heaviest island
    dyn = kind == scenes.KIND_DYNAMIC
    load = np.bincount(got[dyn], weights=weights[dyn], minlength=world_size)
    heaviest = max(weights[dyn & (island_of == lab)].sum() for lab in np.unique(island_of[dyn]))
    assert load.max() <= load.sum() / world_size + heaviest + 1e-9


def test_partitioner_on_the_c4_scene_at_world_size_8():
    """C4's 4096 mini-piles over 8 ranks: 512 islands and 32 768 boxes each, the plane replicated."""
    sc = scenes.mini_piles(64, 64)
    n = len(sc["kind"])
    labels = np.arange(n, dtype=np.uint32)
    labels[1:] = 1 + 64 * ((np.arange(n - 1)) // 64)     # 64 boxes per site, label = the site's first box
    rank_of = partition_islands(labels, sc["kind"], np.ones(n), 8)
    assert rank_of[0] == -1
    counts = np.bincount(rank_of[1:], minlength=8)
    assert (counts == 32768).all()
    for site in range(0, 4096, 257):
        assert len(np.unique(rank_of[1 + 64 * site: 1 + 64 * (site + 1)])) == 1


def overlap_numpy(aabb, labels, kind, rank_of, margin, any_owner):
    isl, boxes, inv = island_boxes(np.asarray(aabb, np.float64), labels, kind)
    dyn = np.asarray(kind) == scenes.KIND_DYNAMIC
    owner = np.zeros(len(isl), np.int32); owner[inv] = np.asarray(rank_of)[dyn]
    lo, hi = boxes[:, :3] - np.float32(margin), boxes[:, 3:] + np.float32(margin)
    out = set()
    for a in range(len(isl)):
        for b in range(a + 1, len(isl)):
            if (any_owner or owner[a] != owner[b]) and np.all(lo[a] < hi[b]) and np.all(lo[b] < hi[a]):
                out.add((int(min(isl[a], isl[b])), int(max(isl[a], isl[b]))))
    return sorted(out)


@pytest.mark.parametrize("any_owner", [False, True])
def test_island_box_sweep_matches_the_all_pairs_test(any_owner):
    rng = np.random.default_rng(11)
    n = 600
    kind = np.full(n, scenes.KIND_DYNAMIC, np.int32); kind[0] = scenes.KIND_STATIC
    centre = rng.uniform(-20, 20, (n, 3))
    half = rng.uniform(0.2, 1.5, (n, 3))
    aabb = np.concatenate([centre - half, centre + half], axis=1).astype(np.float32)
    labels = np.arange(n, dtype=np.uint32)
    labels[1:] = 1 + 3 * ((np.arange(n - 1)) // 3)      # islands of three bodies
    rank_of = np.where(kind == scenes.KIND_DYNAMIC, (labels // 3) % 4, -1).astype(np.int32)
    got = island_boxes_overlap(aabb, labels, kind, rank_of, margin=0.026, any_owner=any_owner)
    want = overlap_numpy(aabb, labels, kind, rank_of, 0.026, any_owner)
    assert got == want and len(got) > 5
