"""Developer aid: step a small scene on the GPU and on the oracle (coloured order) side by side and
report the first step / stage at which they diverge."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes
from oracle import binding as ob

def cmp_manifolds(g, o, tag):
    if len(g) != len(o):
        print(tag, "manifold count", len(g), len(o)); return False
    ok = True
    for f in ("body", "num_points", "colour"):
        if not np.array_equal(g[f], o[f]):
            bad = np.nonzero((g[f] != o[f]).reshape(len(g), -1).any(1))[0]
            print(tag, f, "differs at", bad[:5], g[f][bad[:3]], o[f][bad[:3]]); ok = False
    if ok:
        for k in range(4):
            sel = g["num_points"] > k
            for f in ("pivotA", "pivotB", "normal", "local_normal", "distance", "friction", "attachment", "lifetime", "normal_impulse", "friction_impulse"):
                a = g["pt"][f][sel, k]; b = o["pt"][f][sel, k]
                if not np.array_equal(a, b):
                    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
                    print(tag, f"pt[{k}].{f} max abs diff {d:.3e} (n={sel.sum()})"); ok = ok and d < 1e-5
    return ok

name = sys.argv[1] if len(sys.argv) > 1 else "pile4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene = {"pile4": lambda: scenes.box_pile(4, 4, 4), "pile8": lambda: scenes.box_pile(8, 8, 8), "mixed6": lambda: scenes.box_pile(6, 6, 6, mixed=True),
         "c1": scenes.c1_columns, "chains": lambda: scenes.c5_chains(16, 8)}[name]()
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10))
w.set_scene(scene)
o = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
o.add_bodies(scene)
for s in range(steps):
    for stage, mask in enumerate((1, 2, 4, 8)):
        w.run_stages(mask); o.run_stage(stage)
        if stage == 0:
            if not np.array_equal(w.get_pairs(), o.get_pairs()):
                print("step", s, "PAIRS differ", len(w.get_pairs()), len(o.get_pairs())); sys.exit(1)
        if stage in (1, 3):
            if not cmp_manifolds(w.get_manifolds(), o.get_manifolds(), f"step {s} stage {stage}"):
                sys.exit(1)
        if stage == 2:
            gi = w.get_derived()[2]; oi = o.get_derived()[2]
            if not np.array_equal(gi, oi): print("step", s, "islands differ"); sys.exit(1)
    gs = w.get_state(); os_ = o.get_state()
    d = [float(np.abs(a - b).max()) for a, b in zip(gs, os_)]
    ga = w.get_derived(); oa = o.get_derived()
    da = float(np.abs(ga[0][1:] - oa[0][1:]).max()); di = float(np.abs(ga[1] - oa[1]).max())
    print(f"step {s}: dpos {d[0]:.2e} dorn {d[1]:.2e} dv {d[2]:.2e} dw {d[3]:.2e} daabb {da:.2e} dIw {di:.2e}  gpu {w.get_stats()['num_points']} pts {w.get_stats()['num_colours']} colours / oracle {o.get_stats()['num_points']} pts {o.get_stats()['num_colours']} colours")
    if not all(np.isfinite(x).all() for x in gs): print("NaN"); sys.exit(1)
print("joint impulses diff", float(np.abs(w.get_joint_impulses() - o.get_joint_impulses()).max()) if w.nj else 0.0)
