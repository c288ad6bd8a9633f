"""Developer aid: settle a workload on the device and save its contact graph (bodies, point counts, colours, body kinds) for
offline critical-path modelling (scripts/chain_model.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
name = sys.argv[1]; settle = int(sys.argv[2]); out = sys.argv[3]
gen = {"pile32k": (lambda: scenes.box_pile(32, 32, 32), 10), "mixed32k": (scenes.c3_mixed, 20), "pile8k": (scenes.c2_pile, 10)}[name]
scene = gen[0]()
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=gen[1], num_solver_position_iterations=3)); w.set_scene(scene)
w.step_simulation(settle)
m = w.get_manifolds()
np.savez_compressed(out, body=m["body"], num_points=m["num_points"], colour=m["colour"], kind=scene["kind"], pos=w.get_state()[0])
print(name, len(m), w.get_stats()["num_colours"])
