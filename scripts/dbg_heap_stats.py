"""Debug aid: colouring statistics of the polyhedron heap in its timed regime."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import edyn_amd
from edyn_amd import scenes
sc = scenes.polyhedron_heap(32, 32, 32)
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, exclusive_device=True)); w.set_scene(sc)
w.step_simulation(240)
prev = None
for s in range(8):
    w.step_simulation(1)
    st = w.get_stats()
    m = w.get_manifolds()
    act = m["num_points"] > 0
    col = m["colour"].astype(np.int64)
    key = (m["body"][:, 0].astype(np.int64) << 32) | m["body"][:, 1]
    cur = dict(zip(key[act].tolist(), col[act].tolist()))
    changed = -1 if prev is None else sum(1 for k, c in cur.items() if prev.get(k) != c)
    prev = cur
    print("step", s, "manifolds", st["num_manifolds"], "active", st["num_active_manifolds"], "colours", st["num_colours"], "rounds", st["colour_rounds"], "recoloured-or-new", changed)
