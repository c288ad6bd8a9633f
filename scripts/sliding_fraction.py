"""How many contact points of a settled workload sit ON the friction circle (sliding: |friction impulse| = mu * normal impulse),
how many carry no normal impulse, and how the sliders are spread over the solver's wave-tasks (32 manifolds in colour-sorted order)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "pile32k"
wl = bench.WORKLOADS[name]
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"]))
w.set_scene(wl["gen"]())
w.step_simulation(wl["settle"] + 30)
m = w.get_manifolds()
act = m["num_points"] > 0
mm = m[act]
order = np.lexsort((-(mm["num_points"].astype(np.int64)), mm["colour"]))   # colour, then 4-point manifolds first
mm = mm[order]
np_ = mm["num_points"]
valid = np.arange(4)[None, :] < np_[:, None]
ni = mm["pt"]["normal_impulse"]; fi = np.linalg.norm(mm["pt"]["friction_impulse"], axis=2); mu = mm["pt"]["friction"]
lim = mu * ni
sliding = valid & (ni > 0) & (np.abs(fi - lim) <= 1e-6 * np.maximum(lim, 1e-30))
unloaded = valid & (ni == 0)
print(f"{name}: {int(valid.sum())} points in {len(mm)} active manifolds; unloaded (normal impulse 0) {unloaded.sum() / valid.sum():.3f}; sliding {sliding.sum() / valid.sum():.4f} "
      f"({sliding.sum() / max((valid & (ni > 0)).sum(), 1):.4f} of the loaded ones)")
nt = len(mm) // 32
for k in range(4):
    per_task = sliding[: nt * 32, k].reshape(nt, 32).any(axis=1)
    print(f"  point slot {k}: wave-tasks (32 manifolds) with at least one sliding point: {per_task.mean():.3f}")
any_task = sliding[: nt * 32].reshape(nt, 32 * 4).any(axis=1)
print(f"  wave-tasks with a sliding point in any slot: {any_task.mean():.3f}")
