import sys; sys.path.insert(0, '.')
import numpy as np, collections
import edyn_amd
from edyn_amd import scenes
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3))
w.set_scene(scenes.box_pile(32, 32, 32))
hist = collections.Counter(); newm = []
prev = 0
for s in range(420):
    w.step_simulation(1)
    st = w.get_stats()
    hist[st["colour_rounds"]] += 1
print(sorted(hist.items()))
