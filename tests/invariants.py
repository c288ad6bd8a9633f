"""Physical invariants shared by the parity tests (CPU and GPU): computed from the state and manifold records every stepper exposes
(device, checker, reference engine) - so the same number is taken on every side."""
import numpy as np


def _rotate(q, v):
    u, w = q[:, :3].astype(np.float64), q[:, 3:4].astype(np.float64)
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def normal_row_residual(state, manifolds, rest_speed=0.05):
    """The solver residual of SURVEY section 7 hard-part 1: |J v - rhs| of the ACTIVE normal rows after the last velocity iteration.

    A contact's normal row asks for J (v + dv) = -(error erp) with J = {n, rA x n, -n, -(rB x n)} (contact_constraint.cpp:15-56,
    constraint_row.cpp:6-57); for a touching or penetrating point (distance <= 0) the error term is zero (no Baumgarte in the velocity rows,
    SURVEY 8(g)(3)), so what the unconverged Gauss-Seidel sweep leaves behind is the relative normal velocity of the two contact points at
    the end of the step, n . ((vA + wA x rA) - (vB + wB x rB)) - evaluated for every point whose row is active (applied normal impulse > 0) and
    whose distance is <= 0, from the post-step state (the lever arms move by O(dt |v|) during the step: the same on every side).
    Returns (max over the rows whose two bodies are at rest, mean over all active rows, count, 99th percentile, max over all active rows)
    in m/s. The unrestricted maximum is an extreme statistic of a collapsing scene - the contact a faller has just landed on - like the
    deepest penetration (penetration_stats); comparisons between steppers use the resting-set maximum, the percentile and the mean."""
    pos, orn, v, w = (np.asarray(a) for a in state)
    v, w = v.astype(np.float64), w.astype(np.float64)
    slow = (np.linalg.norm(v, axis=1) <= rest_speed) & (np.linalg.norm(w, axis=1) <= 2.0 * rest_speed)   # the same surface speed at half a box
    m = manifolds
    out, rest = [], []
    for k in range(4):
        sel = m["num_points"] > k
        if not sel.any():
            continue
        a, b = m["body"][sel, 0], m["body"][sel, 1]
        pt = m["pt"]
        lam, dist = pt["normal_impulse"][sel, k], pt["distance"][sel, k]
        n = pt["normal"][sel, k].astype(np.float64)
        rA, rB = _rotate(orn[a], pt["pivotA"][sel, k].astype(np.float64)), _rotate(orn[b], pt["pivotB"][sel, k].astype(np.float64))
        vn = ((v[a] + np.cross(w[a], rA) - v[b] - np.cross(w[b], rB)) * n).sum(axis=1)
        act = (lam > 0) & (dist <= 0)
        out.append(np.abs(vn[act])); rest.append(np.abs(vn[act & slow[a] & slow[b]]))
    r = np.concatenate(out) if out else np.zeros(0)
    q = np.concatenate(rest) if rest else np.zeros(0)
    if not len(r):
        return 0.0, 0.0, 0, 0.0, 0.0
    return (float(q.max()) if len(q) else 0.0), float(r.mean()), int(len(r)), float(np.percentile(r, 99)), float(r.max())


def max_penetration(manifolds):
    d = manifolds["pt"]["distance"].astype(np.float64).copy()
    for k in range(4):
        d[manifolds["num_points"] <= k, k] = 1.0
    return -float(d.min()) if len(d) else 0.0


def penetration_stats(manifolds):
    """(deepest, 99th percentile, mean) penetration depth in metres over the live contact points that penetrate at all (distance < 0).
    The deepest value is an extreme statistic of a collapsing pile - one box that has just landed on a vertex under another faller sits
    centimetres deep for tens of steps in every stepper, the reference engine included - so comparisons between steppers use the
    percentile and the mean, and bound the deepest by what a faller can do in one step."""
    d = manifolds["pt"]["distance"].astype(np.float64)
    live = np.zeros(d.shape, bool)
    for k in range(4):
        live[:, k] = manifolds["num_points"] > k
    depth = -d[live & (d < 0)]
    if len(depth) == 0:
        return 0.0, 0.0, 0.0
    return float(depth.max()), float(np.percentile(depth, 99)), float(depth.mean())


def canonical_records(m, kind, dynamic=0):
    """Reference-engine manifold records in the order set_manifolds expects: ascending (owner << 32 | other), the owner being the
    dynamic body (the higher index when both are)."""
    a, b = m["body"][:, 0].astype(np.uint64), m["body"][:, 1].astype(np.uint64)
    da, db = kind[m["body"][:, 0]] == dynamic, kind[m["body"][:, 1]] == dynamic
    owner = np.where(da & db, np.maximum(a, b), np.where(da, a, b))
    other = np.where(owner == a, b, a)
    out = m[np.argsort((owner << np.uint64(32)) | other, kind="stable")].copy()
    out["colour"] = 0xFF
    return out


def resync_lockstep_jointed(world, ref, kind, steps, tol_pos, tol_vel, contacts=True):
    """Lock-step against the reference engine for scenes WITH joints: every step starts from the engine's own state - bodies, contact
    manifolds with their warm-start impulses, and the joints' applied impulses and tracked angles (set_joint_warm_start) - and both sides
    step once. The pair set and the narrowphase output (what the solver's visiting order cannot touch) must be identical; positions /
    velocities - joints by colour here, in the engine's edge order there - within tol_pos / tol_vel per step. `world` is the device
    (edyn_amd.World) or the checker (oracle World). Returns the worst (|dpos|, |dvel|, |dangvel|) of a single step."""
    worst = [0.0, 0.0, 0.0]
    for step in range(1, steps + 1):
        world.set_state(*ref.get_state())
        world.refresh_derived()
        if contacts:
            world.set_manifolds(canonical_records(ref.get_manifolds(), kind))
        world.set_joint_warm_start(ref.get_joint_impulses24(), ref.get_joint_impulses()[:, 9])
        world.step_simulation(1) if hasattr(world, "step_simulation") else world.step(1)
        ref.step(1)
        if contacts:
            assert np.array_equal(world.get_pairs(), ref.get_pairs()), step
            wm, rm = world.get_manifolds(), canonical_records(ref.get_manifolds(), kind)
            assert np.array_equal(wm["body"], rm["body"]) and np.array_equal(wm["num_points"], rm["num_points"]), step
            for fld in ("pivotA", "pivotB", "local_normal", "attachment", "lifetime", "friction"):
                assert np.array_equal(wm["pt"][fld], rm["pt"][fld]), (step, fld)
        (wp, wq, wv, ww), (rp, rq, rv, rw) = world.get_state(), ref.get_state()
        worst = [max(worst[0], float(np.abs(wp - rp).max())), max(worst[1], float(np.abs(wv - rv).max())), max(worst[2], float(np.abs(ww - rw).max()))]
        assert worst[0] < tol_pos and worst[1] < tol_vel, (step, worst)
    return tuple(worst)
