"""Developer aid and test helper (tests/test_multirank_gpu.py): build a world, step it, end the process - the exit status tells whether the
process got through the runtime's exit handlers.   usage: exit_probe.py single|single_keep|single_thread|multi<N>[_gloo][_torch][_keep]   (STEPS=<n>)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
mode = sys.argv[1]
if "torch" in mode:
    import torch
import numpy as np
import edyn_amd
from edyn_amd import scenes
if "gloo" in mode:
    from test_multirank_gloo import _jointed_bridge_scene
    scene = _jointed_bridge_scene(along="z")
else:
    scene = scenes.c4_islands(4, 4) if hasattr(scenes, "c4_islands") else scenes.box_pile(4, 4, 4)
if "multi" in mode:
    shards = int(mode.split("multi")[1][:1])
    mw = edyn_amd.MultiWorld(edyn_amd.init_config(num_solver_velocity_iterations=10), devices=[0] * shards)
    mw.set_scene(scene)
    for k in range(int(os.environ.get("STEPS", "40"))): mw.step_simulation(1)
    print(mode, "repartitions", mw.get_stats()["repartitions"], flush=True)
    if "keep" not in mode:
        mw.close(); print("closed", flush=True)
        del mw
elif "thread" in mode:
    import threading
    def work():
        w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10))
        w.set_scene(scene); w.step_simulation(10); print(mode, "single in a thread ok", flush=True); w.detach()
    t = threading.Thread(target=work); t.start(); t.join()
else:
    w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10))
    w.set_scene(scene)
    w.step_simulation(10)
    print(mode, "single ok", flush=True)
    if "keep" not in mode: del w
print("end of script", flush=True)
