"""Developer aid: which variant of the jointed bridge scene makes the sphere cross a shard boundary, and when does the world re-partition?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import edyn_amd
from test_multirank_gloo import _jointed_bridge_scene
sphere, per_site = 1 + 6 * 64, 64
target_site = {"z": 3, "x": 1}
for shards in (4, 3, 5, 2):
    for along in ("z", "x"):
        scene = _jointed_bridge_scene(along=along)
        mw = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0] * shards)
        mw.set_scene(scene)
        part = mw.get_partition()
        t = 1 + target_site[along] * per_site
        sites = [int(part[1 + k * per_site]) for k in range(6)]
        line = f"shards {shards} along {along}: sphere on {part[sphere]}, target site on {part[t]}, sites {sites}, chains {sorted(set(int(x) for x in part[sphere + 1:] if x >= 0))}"
        events = []
        for k in range(70):
            before = mw.get_stats()["repartitions"]
            mw.step_simulation(1)
            st = mw.get_stats()
            if st["repartitions"] > before:
                p1 = mw.get_partition()
                events.append((k, int(p1[sphere]), int(p1[t])))
        print(line, "| re-partitions (step, sphere shard, target shard):", events, "| approach checks", mw.get_stats()["approach_checks"], flush=True)
        del mw
