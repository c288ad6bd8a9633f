"""What a step costs when almost everything sleeps: the C4 scene (262 144 boxes in 4 096 islands) with island sleeping on, stepped until
every island is asleep; then one more box is dropped onto ONE island and the step time is measured while only that island is awake."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes

sites = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = scenes.mini_piles(sites, sites)
w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3, sleeping=True, exclusive_device=True, timing=True,
                                        max_bodies=len(scene["kind"]) + 16))
w.set_scene(scene)


def timed(steps):
    w.synchronize() if hasattr(w, "synchronize") else None
    t = time.perf_counter(); w.step_simulation(steps); w.get_asleep(); return (time.perf_counter() - t) / steps * 1e3


print("awake, ms/step:", round(timed(60), 3))
for k in range(40):
    ms = timed(30)
    a = w.get_asleep()
    print(f"t={(90 + 30 * k) / 60:.1f}s asleep {int(a.sum())}/{len(a)} ms/step {ms:.3f}", flush=True)
    if a[1:].all():
        break
print("all asleep, ms/step:", round(timed(60), 4))
one = scenes.subset(scene, np.array([1]))
one["pos"][0] = scene["pos"][1] + np.array([0.3, 6.0, 0.2], np.float32)
w.add_scene(one)
for k in range(6):
    ms = timed(20)
    a = w.get_asleep()
    print(f"one island awake: awake bodies {int((~a).sum()) - 1} ms/step {ms:.3f} islands {w.get_stats()['num_islands']}", flush=True)
tm = w.get_timings()
print({k: round(v / max(tm.get("steps", 1), 1), 4) if isinstance(v, float) else v for k, v in tm.items()})
