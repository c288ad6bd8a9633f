"""Long-horizon parity soak: GPU vs oracle (coloured order), full steps, reporting the first divergence if any."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
from oracle import binding as ob
name, steps = sys.argv[1], int(sys.argv[2])
def _figures(shape, *a, **k):
    return scenes.figures(scenes.load_figure(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"ragdoll_{shape}.npz")), *a, **k)


def _pile_and_figures():
    figs = _figures("capsule", 4, 3, pitch=1.4, floor=False)
    figs["pos"][:, 0] += np.float32(16.0)
    return scenes.merge(scenes.box_pile(12, 8, 12), figs)


# ragdoll_heap: the reference's rag dolls merging into one heap (island-fused schedule: register, LDS and global-list paths);
# pile_ragdolls: figures beside - and, once they topple, against - a pile (mixed schedule, per-colour launches when a figure joins the pile)
gen = {"pile10": lambda: scenes.box_pile(10, 10, 10), "mixed8": lambda: scenes.box_pile(8, 8, 8, mixed=True),
       "pile6": lambda: scenes.box_pile(6, 6, 6), "pyr8": lambda: scenes.pyramid(8),
       "ragdoll_heap": lambda: _figures("box", 3, 3, pitch=1.0, ny=3, pitch_v=1.9), "pile_ragdolls": _pile_and_figures}[name]
vel = 20 if name.startswith("mixed") else 10
scene = gen()
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=vel)); w.set_scene(scene); scenes.apply_figure_settings(w, scene)
o = ob.World(vel_iters=vel, order=ob.ORDER_COLOURED); o.add_bodies(scene); scenes.apply_figure_settings(o, scene)
worst = [0.0] * 4; first_inexact = None
t0 = time.time()
for s in range(steps):
    w.step_simulation(1); o.step(1)
    if not np.array_equal(w.get_pairs(), o.get_pairs()):
        print("PAIRS differ at step", s); sys.exit(1)
    d = [float(np.abs(a - b).max()) for a, b in zip(w.get_state(), o.get_state())]
    if scene.get("joints") and not np.array_equal(w.get_joint_impulses().view(np.uint32), o.get_joint_impulses().view(np.uint32)):
        print("JOINT IMPULSES differ at step", s); sys.exit(1)
    worst = [max(x, y) for x, y in zip(worst, d)]
    if first_inexact is None and max(d) > 0: first_inexact = s
    if s % 25 == 0 or s == steps - 1:
        st = w.get_stats()
        print(f"step {s}: max diff so far pos {worst[0]:.2e} orn {worst[1]:.2e} v {worst[2]:.2e} w {worst[3]:.2e}; pts {st['num_points']} colours {st['num_colours']} islands {st['num_islands']} first_inexact {first_inexact}", flush=True)
gm, om = w.get_manifolds(), o.get_manifolds()
print("manifolds equal:", len(gm) == len(om) and np.array_equal(gm["body"], om["body"]) and np.array_equal(gm["num_points"], om["num_points"]) and np.array_equal(gm["colour"], om["colour"]), "elapsed", round(time.time() - t0, 1))
