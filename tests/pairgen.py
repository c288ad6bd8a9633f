"""Random shape-pair batches for the closest-feature (collide) parity tests: a good share of the pairs are within the
contact threshold of each other, and degenerate SAT configurations (parallel faces / edges, equal boxes) are over-represented."""
import numpy as np
from edyn_amd import scenes


def random_quats(rng, n, snap_fraction=0.3):
    q = rng.normal(size=(n, 4)).astype(np.float32)
    # a share of exactly axis-aligned / 45-degree / 90-degree orientations: the degenerate SAT cases (parallel faces and
    # edges) are where the feature-selection branches of box_box differ most
    special = np.array([[0, 0, 0, 1], [0, 0.70710678, 0, 0.70710678], [0.38268343, 0, 0, 0.92387953],
                        [0, 0, 0.70710678, 0.70710678], [0.5, 0.5, 0.5, 0.5]], np.float32)
    pick = rng.random(n) < snap_fraction
    q[pick] = special[rng.integers(0, len(special), pick.sum())]
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    return q.astype(np.float32)


def pair_batch(rng, n, tA, tB, mesh_radii=None):
    """n shape pairs placed so that a good share are within the contact threshold of each other."""
    st = np.empty((n, 2), np.int32); st[:, 0] = tA; st[:, 1] = tB
    sp = np.zeros((n, 2, 4), np.float32)
    pos = np.zeros((n, 2, 3), np.float32)
    orn = np.stack([random_quats(rng, n), random_quats(rng, n)], axis=1)
    reach = np.zeros((n, 2), np.float32)
    for side, t in enumerate((tA, tB)):
        if t == scenes.SHAPE_BOX:
            h = rng.uniform(0.1, 0.6, size=(n, 3)).astype(np.float32)
            if side == 1:   # equal boxes in a share of the pairs (stacked-brick degeneracy)
                same = rng.random(n) < 0.3
                h[same] = sp[same, 0, :3] if tA == scenes.SHAPE_BOX else h[same]
            sp[:, side, :3] = h
            reach[:, side] = h.min(axis=1) + rng.random(n).astype(np.float32) * (np.linalg.norm(h, axis=1) - h.min(axis=1))
        elif t == scenes.SHAPE_SPHERE:
            sp[:, side, 0] = rng.uniform(0.1, 0.5, size=n)
            reach[:, side] = sp[:, side, 0]
        elif t == getattr(scenes, "SHAPE_CAPSULE", 4):   # (radius, half_length, axis)
            sp[:, side, 0] = rng.uniform(0.1, 0.4, size=n)
            sp[:, side, 1] = rng.uniform(0.05, 0.6, size=n)
            sp[:, side, 2] = rng.integers(0, 3, n)
            reach[:, side] = sp[:, side, 0] + rng.random(n).astype(np.float32) * sp[:, side, 1]
        elif t == getattr(scenes, "SHAPE_CYLINDER", 5):   # (radius, half_length, axis): flat discs and long rods alike
            sp[:, side, 0] = rng.uniform(0.1, 0.5, size=n)
            sp[:, side, 1] = rng.uniform(0.05, 0.6, size=n)
            sp[:, side, 2] = rng.integers(0, 3, n)
            r, hl = sp[:, side, 0], sp[:, side, 1]
            reach[:, side] = np.minimum(r, hl) + rng.random(n).astype(np.float32) * (np.sqrt(r * r + hl * hl) - np.minimum(r, hl))
        elif t == getattr(scenes, "SHAPE_POLYHEDRON", 6):   # shape_param[0] = mesh id; mesh_radii[id] = (inner, outer) about the centroid
            ids = rng.integers(0, len(mesh_radii), n)
            sp[:, side, 0] = ids
            rr = np.asarray(mesh_radii, np.float32)[ids]
            reach[:, side] = rr[:, 0] + rng.random(n).astype(np.float32) ** 2 * (rr[:, 1] - rr[:, 0])
        else:   # plane through a random offset with a random (or +Y) normal
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
            nrm[rng.random(n) < 0.5] = (0, 1, 0)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
            sp[:, side, :3] = nrm
            sp[:, side, 3] = rng.uniform(-0.5, 0.5, size=n)
            orn[:, side] = (0, 0, 0, 1)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    snap = rng.random(n) < 0.4
    axes = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n)] * rng.choice(np.float32([-1, 1]), size=(n, 1))
    d[snap] = axes[snap]
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    gap = rng.uniform(-0.08, 0.04, size=n).astype(np.float32)
    pos[:, 0] = rng.uniform(-2, 2, size=(n, 3))
    if tB == scenes.SHAPE_PLANE:
        pos[:, 1] = 0
        nrm = sp[:, 1, :3]
        pos[:, 0] = nrm * (sp[:, 1, 3:4] + reach[:, 0:1] + gap[:, None]) + np.cross(nrm, d) * 2
    elif tA == scenes.SHAPE_PLANE:
        pos[:, 0] = 0
        nrm = sp[:, 0, :3]
        pos[:, 1] = nrm * (sp[:, 0, 3:4] + reach[:, 1:2] + gap[:, None]) + np.cross(nrm, d) * 2
    else:
        pos[:, 1] = pos[:, 0] + d * (reach[:, 0] + reach[:, 1] + gap)[:, None]
    return st, sp, pos.astype(np.float32), orn.astype(np.float32)
