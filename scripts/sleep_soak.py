import sys; sys.path.insert(0, '.')
import numpy as np
import edyn_amd
from edyn_amd import scenes
from oracle import binding as ob
for name, scene, steps in (("pile7", scenes.box_pile(7, 7, 7), 700), ("mixed6", scenes.box_pile(6, 6, 6, mixed=True), 700)):
    g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3, sleeping=True))
    g.set_scene(scene)
    o = ob.World(vel_iters=10, pos_iters=3, order=ob.ORDER_COLOURED); o.add_bodies(scene); o.set_sleeping(True)
    ok = True; wakes = 0; prev = None
    for k in range(steps):
        g.step_simulation(1); o.step(1)
        a = g.get_asleep()
        if not np.array_equal(a, o.get_asleep()): print(name, "asleep differs at", k); ok = False; break
        if prev is not None: wakes += int((prev & ~a).sum())
        prev = a
        if k % 50 == 49:
            for x, y in zip(g.get_state(), o.get_state()):
                if not np.array_equal(x, y): print(name, "state differs at", k); ok = False; break
        if not ok: break
    print(name, "ok" if ok else "FAIL", "asleep at end", int(a.sum()), "of", len(a) - 1, "wake events (bodies)", wakes)
