"""Host-side cost of the multi-GPU world's step measured on ONE GPU: the same scene stepped by one context and by
edynhip_world_* with S shards that all live on device 0 (one host thread per shard, a blocking state gather per shard and step, the
approach check). The shards' kernels share the one GPU, so (world ms/step - single ms/step) bounds what the pool barrier + S gathers +
the merge cost per step on top of the GPU work - on S real GPUs the GPU work divides by S and this host cost stays.
usage: python scripts/multi_overhead.py [workload=islands256k] [shards=8] [steps=60]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edyn_amd
from edyn_amd import scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "islands256k"
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
gen = {"islands256k": lambda: scenes.c4_islands(), "islands64k": lambda: scenes.mini_piles(32, 32), "islands4k": lambda: scenes.mini_piles(8, 8)}[wl]
scene = gen()
cfg = edyn_amd.init_config(num_solver_velocity_iterations=10, num_solver_position_iterations=3)
settle = 120


def timed(step, sync, stats=None):
    """per-step wall times of `steps` steps after the settle; with `stats`: the steps in which the world re-partitioned are told apart"""
    step(settle); sync()
    plain, repart = [], []
    for _ in range(steps):
        before = stats()["repartitions"] if stats else 0
        t0 = time.perf_counter(); step(1); sync()
        dt = 1e3 * (time.perf_counter() - t0)
        (repart if stats and stats()["repartitions"] != before else plain).append(dt)
    return plain, repart


one = edyn_amd.World(cfg); one.set_scene(scene)
t_one, _ = timed(one.step_simulation, lambda: one.get_state())
p_one = one.get_state()[0]
del one
mw = edyn_amd.MultiWorld(cfg, devices=[0] * shards); mw.set_scene(scene)
t_multi, t_repart = timed(mw.step_simulation, lambda: mw.get_state(), mw.get_stats)   # edynhip_world_step returns with the state gathered
same = bool(np.array_equal(mw.get_state()[0], p_one))
st = mw.get_stats()
med = lambda v: float(np.median(v)) if len(v) else None
print(json.dumps({"workload": wl, "bodies": len(scene["kind"]), "shards_on_one_gpu": shards, "settle_steps": settle, "timed_steps": steps,
                  "single_context_ms_per_step_median_incl_state_readback": med(t_one), "world_ms_per_step_median_without_repartition": med(t_multi),
                  "difference_us_per_step": 1e3 * (med(t_multi) - med(t_one)) if t_multi else None,
                  "steps_with_a_repartition": len(t_repart), "ms_per_repartitioning_step_median": med(t_repart),
                  "bit_identical_to_single_context": same, "approach_checks_total": st["approach_checks"], "repartitions_total": st["repartitions"],
                  "bodies_per_shard": st["bodies_per_shard"]}))
