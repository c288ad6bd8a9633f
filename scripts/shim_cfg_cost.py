"""Why is the raw rate of the context the C++ shim creates lower than bench.py's? The shim's defaults differ from the bench's in four
ways - island sleeping on (the reference always sleeps islands), contact events on (contact entities), head-room in max_bodies, and
cooperative launches unless init_config::exclusive_device. This steps the headline pile under each and prints steps/s + stage times."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import edyn_amd
from edyn_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
scene = scenes.box_pile(n, n, n)
scene["sleeping_disabled"] = np.ones(len(scene["kind"]), np.uint8)
nb = len(scene["kind"])
variants = {
    "bench (exclusive, no sleeping, no events, max_bodies = n)": dict(exclusive_device=True),
    "+ sleeping": dict(exclusive_device=True, sleeping=True),
    "+ contact events": dict(exclusive_device=True, contact_events=True),
    "+ head-room (max_bodies = 1.5 n + 16)": dict(exclusive_device=True, max_bodies=nb + nb // 2 + 16),
    "shim default (all three, cooperative)": dict(sleeping=True, contact_events=True, max_bodies=nb + nb // 2 + 16),
    "shim exclusive (all three)": dict(exclusive_device=True, sleeping=True, contact_events=True, max_bodies=nb + nb // 2 + 16),
}
for name, kw in variants.items():
    for timing in (False, True):
        w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3, timing=timing, **kw))
        w.set_scene(scene)
        w.step_simulation(120)
        w.synchronize()
        t = time.perf_counter(); w.step_simulation(steps); w.synchronize(); el = time.perf_counter() - t
        if not timing:
            print(f"{name:60s} {steps / el:7.1f} steps/s  {1e3 * el / steps:.3f} ms/step", flush=True)
        else:
            tm = w.get_timings()
            print("    stages ms/step:", {k: round(v / max(tm.get("steps", 1), 1), 4) for k, v in tm.items() if isinstance(v, float)}, flush=True)
        del w
