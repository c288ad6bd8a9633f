"""Generates tests/golden/*.npz: inputs + expected outputs for the parity tests, produced by the CPU oracle
(oracle/, pinned against the reference's own golden vectors and against the reference translation units that
compile stand-alone - see tests/test_oracle_golden.py, tests/test_oracle_xcheck.py). The -m gpu tests compare
the HIP path against these files without needing the oracle or /root/reference at run time.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from edyn_amd import scenes               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def columns_small():
    s = scenes._empty(46)
    scenes._add_plane(s)
    pos, _ = scenes._lattice(3, 5, 3, pitch_h=1.05, pitch_v=1.05, y0=0.55, brick=False)
    s["pos"][1:] = pos; s["shape_type"][1:] = scenes.SHAPE_BOX; s["shape_param"][1:, :3] = 0.5
    scenes._jitter(s, 1, 45)
    return s


CASES = {
    "pile4": (lambda: scenes.box_pile(4, 4, 4), 30, 10),
    "mixed5": (lambda: scenes.box_pile(5, 5, 5, mixed=True), 24, 20),
    "pyramid5": (lambda: scenes.pyramid(5), 40, 10),
    "columns": (columns_small, 30, 8),
    "chains": (lambda: scenes.c5_chains(4, 6), 40, 10),
}
# island sleeping enabled: a 3x3x3 brick pile collapses into several islands that settle and fall asleep one by one
SLEEP_CASES = {
    "sleep3": (lambda: scenes.box_pile(3, 3, 3), 420, 10),
}


def main():
    for name, (gen, steps, vel) in CASES.items():
        scene = gen()
        w = ob.World(vel_iters=vel, pos_iters=3, order=ob.ORDER_COLOURED)
        w.add_bodies(scene)
        pairs_per_step = []
        for _ in range(steps):
            w.step(1)
            pairs_per_step.append(w.get_pairs())
        pos, orn, lv, av = w.get_state()
        aabb, iw, isl = w.get_derived()
        m = w.get_manifolds()
        out = {"steps": steps, "vel_iters": vel, "pos": pos, "orn": orn, "linvel": lv, "angvel": av, "aabb": aabb,
               "inertia_world": iw, "island": isl, "manifolds": m, "joint_impulses": w.get_joint_impulses(),
               "pairs_last": pairs_per_step[-1], "pair_counts": np.array([len(p) for p in pairs_per_step]),
               "pairs_xor": np.array([np.bitwise_xor.reduce(p) if len(p) else 0 for p in pairs_per_step], np.uint64)}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "bodies", len(scene["kind"]), "manifolds", len(m), "points", int(m["num_points"].sum()))


def main_sleep():
    for name, (gen, steps, vel) in SLEEP_CASES.items():
        scene = gen()
        w = ob.World(vel_iters=vel, pos_iters=3, order=ob.ORDER_COLOURED)
        w.add_bodies(scene)
        w.set_sleeping(True)
        asleep_count, first_sleep = [], np.full(len(scene["kind"]), -1, np.int32)
        for k in range(steps):
            w.step(1)
            a = w.get_asleep()
            asleep_count.append(int(a.sum()))
            first_sleep[(first_sleep < 0) & a] = k
        pos, orn, lv, av = w.get_state()
        out = {"steps": steps, "vel_iters": vel, "pos": pos, "orn": orn, "linvel": lv, "angvel": av, "asleep": w.get_asleep(),
               "asleep_count": np.array(asleep_count, np.int32), "first_sleep": first_sleep, "manifolds": w.get_manifolds()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "bodies", len(scene["kind"]), "asleep at the end", asleep_count[-1], "first sleep at steps", sorted(set(first_sleep[first_sleep >= 0].tolist()))[:6])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sleep":
        main_sleep()       # only the sleeping fixtures (leaves the other files untouched)
    else:
        main()
        main_sleep()
