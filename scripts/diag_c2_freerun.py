"""Diagnostic for tests/test_gpu_parity.py::test_c2_300_steps_survey_invariants_*: where do the deepest penetrations and the
kinetic energy of the free-running C2 pile sit, on the device and in the reference engine? (developer aid; oracle/ use = checker)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
from oracle import binding as ob

scene = scenes.c2_pile()
g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3)); g.set_scene(scene)
r = ob.RefWorld(vel_iters=10); r.add_bodies(scene)

def report(tag, w, step):
    p, q, v, av = w.get_state()
    m = w.get_manifolds()
    sp = np.linalg.norm(v, axis=1)
    d = m["pt"]["distance"].copy()
    for k in range(4):
        d[m["num_points"] <= k, k] = 1.0
    dm = d.min(axis=1)
    order = np.argsort(dm)[:4]
    fast = sp > 0.05
    both_slow = ~fast[m["body"][:, 0]] & ~fast[m["body"][:, 1]]
    pen_all, pen_rest = -dm.min(), -dm[both_slow].min()
    ke = 0.5 * (sp[1:] ** 2)
    print(f"{tag} step {step}: pen all {pen_all:.4f} rest-only {pen_rest:.4f}; fast bodies {int(fast.sum())}; KE/body all {ke.mean():.3e} "
          f"rest-only {ke[~fast[1:]].mean():.3e}; max speed {sp.max():.3f}; mean h {p[1:,1].mean():.5f}")
    for i in order[:3]:
        a, b = m["body"][i]
        print(f"    deepest: dist {dm[i]:.4f} bodies {a},{b} y {p[a,1]:.2f},{p[b,1]:.2f} speed {sp[a]:.3f},{sp[b]:.3f} npts {m['num_points'][i]}")

done = 0
for upto in (60, 120, 180, 240, 270, 285, 295, 300, 330, 360):
    g.step_simulation(upto - done); r.step(upto - done); done = upto
    report("gpu   ", g, upto); report("engine", r, upto)
