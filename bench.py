#!/usr/bin/env python
"""bench.py - steps/sec of the MI355X-native Edyn stepper on the headline scene.

Contract (one JSON line on rank 0):
  metric   steps/sec (whole node), 32k-box pile, 10 SI iters           (BASELINE.json)
  value    pile-steps per second over all ranks, scene resident in HBM before the timed region
  N = 1    the 32 768-box brick-offset pile on a static plane, 10 velocity / 3 position iterations
  N > 1    the pile is ONE island and cannot be split: each rank steps its own replica of
           the pile (N islands sharded one per GPU, no data-path collective) and the integrated state
           (13 floats/body) is all-gathered over RCCL every step, as the registry write-back would need;
           value counts pile-steps, scaling = "weak".
  roofline the SI velocity-solve kernel (k_contact_solve_df2 / _df: ONE dataflow launch per step runs the warm start and
           every iteration over every colour; scenes with joints use one k_contact_solve launch per colour):
           algorithmic bytes (380 B per contact point per iteration) / time measured with HIP events
           recorded on the stepper's stream around that launch inside the timed region.
  cpu_baseline  the CPU oracle (reference-order restatement, 1 thread) timed on a bounded sample of the
           same scene on rank 0 at N=1. A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import edyn_amd
from edyn_amd import scenes

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_POINT_ITER = 380.0   # algorithmic bytes per contact point per velocity iteration
WORKLOADS = {
    "pile32k": dict(gen=lambda: scenes.box_pile(32, 32, 32), vel=10, pos=3, desc="32x32x32 = 32768-box brick-offset pile on a static plane"),
    "pile8k": dict(gen=lambda: scenes.box_pile(20, 20, 20), vel=10, pos=3, desc="20x20x20 = 8000-box pile (config C2)"),
    "mixed32k": dict(gen=lambda: scenes.box_pile(32, 32, 32, mixed=True), vel=20, pos=3, desc="32768 mixed box/sphere stack, 20 it (config C3)"),
    "pile512": dict(gen=lambda: scenes.box_pile(8, 8, 8), vel=10, pos=3, desc="8x8x8 pile (smoke)"),
    # C4 shards by islands: with N ranks each rank steps its contiguous block of the 4096 sites, no data-path
    # collective - strong scaling of ONE scene (value = scene-steps/s), unlike the single-island pile's replicas
    "islands256k": dict(gen=lambda: scenes.c4_islands(), vel=10, pos=3, desc="262144 boxes in 4096 independent 4x4x4 mini-piles (config C4)",
                        shard=lambda first, count: scenes.c4_islands(first_site=first, num_sites=count), shard_units=4096),
    "chains16k": dict(gen=lambda: scenes.c5_chains(1024, 16), vel=10, pos=3, desc="1024 chains x 16 links, hinge + point joints, no contacts (config C5)"),
}


def cpu_baseline(workload, sample_steps, warm_steps):
    """Time the CPU oracle (reference order, single thread) on a bounded sample. Checker code, timed - never shipped."""
    from oracle import binding as ob
    wl = WORKLOADS[workload]
    scene = wl["gen"]()
    o = ob.World(vel_iters=wl["vel"], pos_iters=wl["pos"], order=ob.ORDER_SEQUENTIAL)
    o.add_bodies(scene)
    o.step(warm_steps)
    t = o.time_steps(sample_steps)
    return {"value": sample_steps / t, "unit": "steps/sec", "cores": 1, "kind": "port",
            "sample": f"{sample_steps} steps after {warm_steps} warm-up steps of the same scene from its initial state "
                      f"({o.get_stats()['num_points']} contact points at the end), oracle in reference (sequential) row order"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=120)
    ap.add_argument("--workload", default="pile32k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-timing", action="store_true",
                    help="HIP events around every stage (adds stages_ms_per_step; each event idles the GPU ~6 us, so the headline\n                    value is measured without it: only the two events around the velocity solve are recorded)")
    ap.add_argument("--cpu-sample-steps", type=int, default=16)
    args = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU path")
    ngpu = torch.cuda.device_count()
    # One process per GPU over RCCL is the real configuration. If there are fewer GPUs than ranks (functional test of
    # this code path on a 1-GPU box: EDYN_BENCH_SHARE_GPU=1) ranks share devices and the gather is staged through gloo.
    share = world_size > ngpu
    if share and os.environ.get("EDYN_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py: {world_size} ranks but {ngpu} GPU(s); set EDYN_BENCH_SHARE_GPU=1 only for functional tests")
    device_index = local_rank % ngpu
    torch.cuda.set_device(device_index)
    distributed = world_size > 1
    backend = "gloo" if share else "nccl"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    wl = WORKLOADS[args.workload]
    sharded = distributed and "shard" in wl
    if sharded:
        from edyn_amd.parallel import shard_range
        first, count = shard_range(wl["shard_units"], rank, world_size)
        scene = wl["shard"](first, count)
        max_count = shard_range(wl["shard_units"], 0, world_size)[1]          # rank 0 holds a largest block
        n_gather = len(wl["shard"](0, max_count)["kind"]) if count != max_count else len(scene["kind"])
    else:
        scene = wl["gen"]()
        n_gather = len(scene["kind"])
    n_bodies = len(scene["kind"])
    cfg = edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"],
                               device=device_index, timing=args.stage_timing, timing_solve=not args.stage_timing,
                               # one stepper per GPU, its kernels and the RCCL gather serialised on one stream: the stepper owns
                               # the device (not so when several ranks share a GPU in the functional gloo mode)
                               exclusive_device=(backend != "gloo"))
    w = edyn_amd.World(cfg)
    w.set_scene(scene)
    stream = torch.cuda.current_stream()
    w.set_stream(stream.cuda_stream)   # stepper kernels and the RCCL gather share torch's stream => ordered

    state = torch.zeros((n_gather, 13), dtype=torch.float32, device="cuda")   # ragged shards are padded to the largest block
    gathered = torch.empty((world_size * n_gather, 13), dtype=torch.float32, device="cuda" if backend == "nccl" else "cpu") if distributed else None

    def one_step():
        w.step_simulation(1)
        if distributed:
            w.pack_state_device(state.data_ptr())
            if backend == "nccl":
                dist.all_gather_into_tensor(gathered.view(-1), state.view(-1))
            else:
                dist.all_gather_into_tensor(gathered.view(-1), state.cpu().view(-1))

    for _ in range(args.warmup):
        one_step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if distributed:
        for _ in range(args.steps):
            one_step()
    else:
        w.step_simulation(args.steps)   # K steps, stage events recorded per step on the stepper's stream
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    stats = w.get_stats()
    tm = w.get_timings()
    pos = w.get_state()[0]
    finite = bool(np.isfinite(pos).all())

    if rank == 0:
        dist_note = ""
        if distributed:
            how = (f"sharded by islands over {world_size} ranks ({wl['shard_units']} sites, contiguous blocks), no data-path collective"
                   if sharded else f"{world_size} replicas (one island per GPU)")
            dist_note = f"; {how}, per-step {'RCCL' if backend == 'nccl' else 'gloo (shared-GPU functional test)'} all-gather of state"
        value = (1 if sharded else world_size) * args.steps / elapsed   # sharded: one scene stepped once per step by all ranks together
        steps_timed = max(tm["steps"], 1)
        solve_ms = tm["solve_velocity_ms"] / steps_timed
        launches = tm["solve_velocity_launches"] / max(args.steps if not distributed else 1, 1)
        alg_bytes_step = BYTES_PER_POINT_ITER * stats["num_points"] * (wl["vel"] + 1)   # +1: warm start sweep
        achieved = (alg_bytes_step / 1e9) / (solve_ms / 1e3) if solve_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "steps/sec (whole node), 32k-box pile, 10 SI iters; HBM GB/s in solve",
            "value": value, "unit": "steps/sec", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}; {wl['vel']} velocity / {wl['pos']} position iterations, dt 1/60, "
                                   f"friction 0.5, restitution 0" + dist_note,
                       "bodies": n_bodies, "contact_points": stats["num_points"], "manifolds": stats["num_manifolds"],
                       "colours": stats["num_colours"], "colour_sizes": stats["colour_size"], "islands": stats["num_islands"], "finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": ("k_contact_solve_df2 (one dataflow launch per step: warm start + every iteration over every colour; two lanes per manifold - k_contact_solve_df, one lane, on bandwidth-bound scenes)" if launches < 1.5
                                    else "k_contact_solve<WARM,PUSH> (+ _tail): one launch per colour, every iteration + warm start"),
                         "algorithmic_bytes_per_launch": alg_bytes_step / max(launches, 1), "launches_per_step": launches,
                         "avg_launch_us": 1e3 * solve_ms / max(launches, 1), "solve_ms_per_step": solve_ms},
        }
        if args.stage_timing:
            out["stages_ms_per_step"] = {k: tm[k] / steps_timed for k in ("broadphase_ms", "narrowphase_ms", "islands_ms", "colouring_ms",
                                                                         "prepare_ms", "solve_velocity_ms", "integrate_ms", "solve_position_ms",
                                                                         "finish_ms", "step_ms")}
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_steps, 4)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
