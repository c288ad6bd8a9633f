"""Where in the colour order do the points sit whose penetration exceeds the position solver's island threshold (0.005 m)? Decides how early
an island's "continue" verdict would be known if the three position iterations ran as one launch.
usage (GPU box): python scripts/pos_error_by_colour.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "pile32k"]
g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"]))
g.set_scene(wl["gen"]())
g.step_simulation(wl["settle"] + 20)
m = g.get_manifolds()
act = m["num_points"] > 0
d = m["pt"]["distance"].copy()
for k in range(4):
    d[m["num_points"] <= k, k] = 1.0
deep = d.min(axis=1)
print("active manifolds", int(act.sum()), "colours", int(m["colour"][act].max()) + 1, "deepest", float(deep.min()))
for thr in (0.005, 0.004, 0.003):
    sel = act & (deep <= -thr)
    cols = np.bincount(m["colour"][sel], minlength=int(m["colour"][act].max()) + 1)
    print(f"penetration >= {thr}: {int(sel.sum())} manifolds; per colour {cols.tolist()}")
