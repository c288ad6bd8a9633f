"""Rounds of the contact colouring per step (edynhip_stats.colour_rounds: k_col_rounds' rounds in one workgroup + the multi-block rounds) and
the uncoloured edges each step lists - how deep the priority-ordered first-fit rule is on a scene.   usage: colour_rounds.py [workload] [steps]"""
import sys; sys.path.insert(0, '.')
import collections
import numpy as np
import edyn_amd, bench
name = sys.argv[1] if len(sys.argv) > 1 else "pile32k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 420
wl = bench.WORKLOADS[name]
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"]))
w.set_scene(wl["gen"]())
rounds = []
for s in range(steps):
    w.step_simulation(1)
    rounds.append(w.get_stats()["colour_rounds"])
r = np.array(rounds)
print(f"{name}: {steps} steps; colouring rounds per step: median {np.median(r):.0f}, mean {r.mean():.1f}, max {r.max()}; last 100 steps: median {np.median(r[-100:]):.0f}, max {r[-100:].max()}")
print("histogram (rounds: steps):", sorted(collections.Counter((r // 8 * 8).tolist()).items()))
