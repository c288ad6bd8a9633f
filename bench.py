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
  cpu_baseline  Edyn's own multithreaded CPU path: the REAL reference engine (oracle/_ref/libedynref.so = the reference's
           translation units compiled where they lie, driven through edyn::attach / step_simulation in
           execution_mode::sequential_multithreaded on all host cores) timed on a bounded sample of the same scene on rank 0
           at N=1, in a subprocess with a wall-clock budget; next to it, in `sample`, the 1-thread restatement (oracle).
           Falls back to the restatement (kind "port", cores 1) where oracle/_ref is not built or the budget is exceeded.
           A reported baseline, not the target.
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
    # 1024 of the reference's own rag dolls (edyn::make_ragdoll, exported from the real engine: tests/golden/make_ragdoll.py)
    # collapsing on a plane: 22 529 bodies, 36 864 cone / cvjoint / hinge constraints, capsule contacts
    "ragdolls1k": dict(gen=lambda: scenes.figures(scenes.load_figure(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "ragdoll_capsule.npz")), 32, 32),
                       vel=10, pos=3, desc="1024 rag dolls (22 bodies, 36 cone/cvjoint/hinge constraints each) falling on a plane"),
    # north_star's body count on ONE GPU: 1 048 576 boxes in 16 384 islands (the C4 scene at 4x the sites)
    "islands1m": dict(gen=lambda: scenes.mini_piles(128, 128), vel=10, pos=3, desc="1048576 boxes in 16384 independent 4x4x4 mini-piles",
                      shard=lambda first, count: scenes.mini_piles(128, 128, first_site=first, num_sites=count), shard_units=16384),
    # the headline pile with 64 rag dolls standing beside it: islands with joints next to a large island without (mixed schedule)
    "pile32k_ragdolls": dict(gen=lambda: _pile_and_ragdolls(), vel=10, pos=3, desc="the 32768-box pile beside 64 rag dolls (36 constraints each)"),
    "chains16k": dict(gen=lambda: scenes.c5_chains(1024, 16), vel=10, pos=3, desc="1024 chains x 16 links, hinge + point joints, no contacts (config C5)"),
}


def _pile_and_ragdolls():
    figs = scenes.figures(scenes.load_figure(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "ragdoll_capsule.npz")), 8, 8, floor=False)
    figs["pos"][:, 0] += np.float32(60.0)
    return scenes.merge(scenes.box_pile(32, 32, 32), figs)


def _cpu_scene(workload):
    """The scene the CPU legs time: the workload itself, or - for the many-island scene, whose reference run would need
    tens of GB - a block of its independent sites (steps/s then scale with the site count, stated in `sample`)."""
    wl = WORKLOADS[workload]
    if "shard" in wl:
        sites = 256
        return wl["shard"](0, sites), wl["shard_units"] / sites, f"sites 0..{sites - 1} of {wl['shard_units']} (independent islands; steps/s divided by {wl['shard_units'] // sites})"
    return wl["gen"](), 1.0, "the whole scene"


def cpu_reference_leg(workload, sample_steps, warm_steps):
    """Child process (bench.py --cpu-reference-leg): time the real reference engine, multithreaded. Prints one JSON line."""
    from oracle import binding as ob
    wl = WORKLOADS[workload]
    scene, scale, what = _cpu_scene(workload)
    cores = os.cpu_count() or 1
    r = ob.RefWorld(vel_iters=wl["vel"], pos_iters=wl["pos"], mode=1, workers=0)   # sequential_multithreaded, hardware_concurrency - 1 workers + the caller
    r.add_bodies(scene); scenes.apply_figure_settings(r, scene)
    t0 = time.perf_counter()
    r.step(warm_steps)
    warm_s = time.perf_counter() - t0
    t = r.time_steps(sample_steps)
    print(json.dumps({"value": sample_steps / t / scale, "cores": cores, "warm_s": warm_s, "what": what,
                      "points": int(r.get_manifolds()["num_points"].sum())}))


def cpu_baseline(workload, sample_steps, warm_steps, budget_s):
    """CPU legs, timed on the host cores of the bench box. Checker code, timed - never shipped."""
    import subprocess
    from oracle import binding as ob
    wl = WORKLOADS[workload]
    scene, scale, what = _cpu_scene(workload)
    o = ob.World(vel_iters=wl["vel"], pos_iters=wl["pos"], order=ob.ORDER_SEQUENTIAL)
    o.add_bodies(scene); scenes.apply_figure_settings(o, scene)
    o.step(warm_steps)
    port = sample_steps / o.time_steps(sample_steps) / scale
    port_note = (f"1-thread restatement (oracle, reference row order): {port:.3f} steps/s over {sample_steps} steps after {warm_steps} "
                 f"warm-up steps of {what} from its initial state ({o.get_stats()['num_points']} contact points at the end)")
    ref = None
    if ob.ref() is not None and budget_s > 0:
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-reference-leg", "--workload", workload,
                                  "--cpu-sample-steps", str(sample_steps), "--cpu-warm-steps", str(warm_steps)],
                                 capture_output=True, text=True, timeout=budget_s)
            ref = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else None
        except (subprocess.TimeoutExpired, ValueError):
            ref = None
    if ref is not None:
        return {"value": ref["value"], "unit": "steps/sec", "cores": ref["cores"], "kind": "reference",
                "sample": f"the reference engine itself (edyn::attach, execution_mode::sequential_multithreaded, {ref['cores']} host threads): "
                          f"{sample_steps} steps after {warm_steps} warm-up steps ({ref['warm_s']:.1f} s, they build ~{ref['points']} contact points "
                          f"and their islands) of {ref['what']}; beside it the {port_note}"}
    return {"value": port, "unit": "steps/sec", "cores": 1, "kind": "port",
            "sample": port_note + "; the multithreaded reference-engine leg was unavailable (oracle/_ref not built or over its time budget)"}


def copy_ceiling_gbs():
    """Measured device-copy bandwidth (read + write bytes of a 1 GiB device-to-device copy, best of 5): SURVEY 8(d)'s
    practical ceiling, reported beside the 8 TB/s spec peak."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2.0 * n / 1e9 / (e0.elapsed_time(e1) / 1e3))
    del a, b
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=120)
    ap.add_argument("--workload", default="pile32k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-timing", action="store_true",
                    help="HIP events around every stage (adds stages_ms_per_step; each event idles the GPU ~6 us, so the headline\n                    value is measured without it: only the two events around the velocity solve are recorded)")
    ap.add_argument("--cpu-sample-steps", type=int, default=6)
    ap.add_argument("--cpu-warm-steps", type=int, default=2)
    ap.add_argument("--cpu-budget-s", type=float, default=240.0, help="wall-clock budget of the reference-engine CPU leg")
    ap.add_argument("--cpu-reference-leg", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_reference_leg:
        cpu_reference_leg(args.workload, args.cpu_sample_steps, args.cpu_warm_steps)
        return

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
    from edyn_amd.parallel import shard_range, StateGather
    if sharded:
        first, count = shard_range(wl["shard_units"], rank, world_size)
        scene = wl["shard"](first, count)
        per_site = (len(scene["kind"]) - 1) // count                          # bodies per site + the replicated static plane
        counts = [1 + per_site * shard_range(wl["shard_units"], r, world_size)[1] for r in range(world_size)]
    else:
        scene = wl["gen"]()
        counts = [len(scene["kind"])] * world_size
    n_bodies = len(scene["kind"])
    assert counts[rank] == n_bodies
    cfg = edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"],
                               device=device_index, timing=args.stage_timing, timing_solve=not args.stage_timing,
                               # a single-rank run owns its GPU: plain launches for the resident-grid kernels
                               exclusive_device=not distributed)   # RCCL kernels share the GPU in multi-rank runs: cooperative launches there
    w = edyn_amd.World(cfg)
    w.set_scene(scene)
    scenes.apply_figure_settings(w, scene)
    # The stepper, the pack kernel and the RCCL gather all run on ONE explicitly created torch stream (a non-zero
    # handle: edynhip_set_stream(NULL) would mean "a private stream"), entered for the whole run => ordered.
    stream = torch.cuda.Stream(device=device_index)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    w.set_stream(stream.cuda_stream)

    gath = StateGather(counts, "cuda", backend) if distributed else None   # edyn_amd.parallel: the registry write-back gather

    def one_step():
        w.step_simulation(1)
        if distributed:
            w.pack_state_device(gath.local.data_ptr())   # same stream as the stepper and (through torch) the collective
            gath.gather()

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
        # measured fabric traffic per launch: a KEPT rocprofv3 PMC profile of this command (profiles/traffic.json, keyed
        # by workload; FETCH_SIZE/WRITE_SIZE cannot be read from inside the run) - null when no profile was kept
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(args.workload) if isinstance(tj.get(args.workload), dict) else (tj if tj.get("workload") == args.workload else None)
                traffic = ent.get("hbm_bytes_per_launch") if ent else None
            except Exception:
                traffic = None
        per_launch_alg = alg_bytes_step / max(launches, 1)
        # achieved = min(algorithmic, measured) bytes / kernel time
        eff_bytes_step = min(alg_bytes_step, traffic * max(launches, 1)) if traffic else alg_bytes_step
        achieved = (eff_bytes_step / 1e9) / (solve_ms / 1e3) if solve_ms > 0 else 0.0
        # The denominator stays the 8 TB/s spec peak: the measured device-to-device copy rate (reported beside it) is NOT a
        # ceiling for this read-dominated kernel - on the many-island scene the solve streams rows faster than a copy moves
        # bytes (a copy alternates reads and writes on every channel), which would put the fraction above 1.
        ceiling = copy_ceiling_gbs()
        peak = HBM_PEAK_GBS
        out = {
            "metric": "steps/sec (whole node), 32k-box pile, 10 SI iters; HBM GB/s in solve",
            "value": value, "unit": "steps/sec", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}; {wl['vel']} velocity / {wl['pos']} position iterations, dt 1/60, "
                                   f"friction 0.5, restitution 0" + dist_note,
                       "bodies": n_bodies, "contact_points": stats["num_points"], "manifolds": stats["num_manifolds"],
                       "colours": stats["num_colours"], "colour_sizes": stats["colour_size"], "islands": stats["num_islands"], "finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": "kept rocprofv3 PMC profile (profiles/traffic.json), not measured in this run",
                         "measured_copy_ceiling": ceiling, "frac_of_measured_copy_ceiling": achieved / ceiling if ceiling > 0 else None,
                         "achieved_rule": "min(algorithmic bytes, measured traffic) / kernel time; peak = HBM3E spec; the measured device-to-device copy rate (read + write bytes) is reported beside it",
                         "kernel": ("k_contact_solve_df2 (one dataflow launch per step: warm start + every iteration over every colour; two lanes per manifold - k_contact_solve_df, one lane, on bandwidth-bound scenes)" if launches < 1.5
                                    else "k_contact_solve<WARM,PUSH> (+ _tail): one launch per colour, every iteration + warm start"),
                         "algorithmic_bytes_per_launch": per_launch_alg, "launches_per_step": launches,
                         "avg_launch_us": 1e3 * solve_ms / max(launches, 1), "solve_ms_per_step": solve_ms},
        }
        if args.stage_timing:
            out["stages_ms_per_step"] = {k: tm[k] / steps_timed for k in ("broadphase_ms", "narrowphase_ms", "islands_ms", "colouring_ms",
                                                                         "prepare_ms", "solve_velocity_ms", "integrate_ms", "solve_position_ms",
                                                                         "finish_ms", "step_ms")}
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_steps, args.cpu_warm_steps, args.cpu_budget_s)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
