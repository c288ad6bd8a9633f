"""Which contact is the deepest of the free-running C2 pile on the DEVICE, step by step (tests/test_gpu_parity.py::
test_c2_free_running_penetration_time_series_*: the device's whole-scene deepest contact stays near 0.08-0.09 m for over a hundred
steps while the engine's median is 0.004 m)? Prints, for steps 100..420, the deepest manifold: bodies, their heights and speeds, point
count, every point's distance and normal impulse, and how many manifolds load the deeper body from above."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes

scene = scenes.c2_pile()
g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3)); g.set_scene(scene)
g.step_simulation(100)
last = None
for step in range(101, 421):
    g.step_simulation(1)
    m = g.get_manifolds()
    d = m["pt"]["distance"].astype(np.float64).copy()
    for k in range(4):
        d[m["num_points"] <= k, k] = 1.0
    dm = d.min(axis=1)
    i = int(np.argmin(dm))
    a, b = (int(x) for x in m["body"][i])
    key = (a, b)
    if key != last or step % 20 == 0:
        p, q, v, w = g.get_state()
        npts = int(m["num_points"][i])
        touching = lambda x: int(((m["body"][:, 0] == x) | (m["body"][:, 1] == x)).sum())
        print(f"step {step}: deepest {-dm[i]:.4f} m, bodies {a} (y {p[a,1]:.3f}, |v| {np.linalg.norm(v[a]):.3f}, |w| {np.linalg.norm(w[a]):.3f}, {touching(a)} manifolds) / "
              f"{b} (y {p[b,1]:.3f}, |v| {np.linalg.norm(v[b]):.3f}, {touching(b)} manifolds), {npts} points: distances "
              f"{[round(float(x), 4) for x in m['pt']['distance'][i][:npts]]} normal impulses {[round(float(x), 4) for x in m['pt']['normal_impulse'][i][:npts]]} "
              f"normal {[round(float(x), 3) for x in m['pt']['normal'][i][0]]} lifetimes {[int(x) for x in m['pt']['lifetime'][i][:npts]]}", flush=True)
        last = key
