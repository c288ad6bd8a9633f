"""Developer aid: settle C3 on the device, hand the state to the oracle, step once, print the pairs that differ with their AABBs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
from oracle import binding as ob

settle = int(sys.argv[1]) if len(sys.argv) > 1 else 120
scene = scenes.c3_mixed()
g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=20, num_solver_position_iterations=3)); g.set_scene(scene)
g.step_simulation(settle)
o = ob.World(vel_iters=20, pos_iters=3, order=ob.ORDER_COLOURED); o.add_bodies(scene)
o.set_state(*g.get_state()); o.refresh_derived(); o.set_manifolds(g.get_manifolds())
ga, oa = g.get_derived()[0], o.get_derived()[0]
print("aabb equal before:", np.array_equal(ga[1:], oa[1:]), np.abs(ga[1:] - oa[1:]).max())
gk0 = g.get_pairs()
g.step_simulation(1); o.step(1)
gk, ok = g.get_pairs(), o.get_pairs()
print(len(gk0), len(gk), len(ok))
only_o = np.setdiff1d(ok, gk); only_g = np.setdiff1d(gk, ok)
print("only oracle:", len(only_o), "only gpu:", len(only_g))
v = g.get_state()
for k in list(only_o[:6]) + list(only_g[:6]):
    hi, lo = int(k >> np.uint64(32)), int(k & np.uint64(0xFFFFFFFF))
    print("pair", hi, lo, "in prev:", k in gk0, "aabb hi", ga[hi], "aabb lo", ga[lo], "types", scene["shape_type"][hi], scene["shape_type"][lo])
    a, b = ga[hi], ga[lo]
    gap = np.maximum(a[:3] - b[3:], b[:3] - a[3:])
    print("   gap per axis (positive = separated):", gap, "|v|", np.linalg.norm(v[2][hi]), np.linalg.norm(v[2][lo]), "|w|", np.linalg.norm(v[3][hi]), np.linalg.norm(v[3][lo]))
