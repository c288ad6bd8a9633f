#!/usr/bin/env python
"""bench.py - steps/sec of the MI355X-native Edyn stepper on the headline scene.

Contract (one JSON line on rank 0):
  metric   steps/sec (whole node), 32k-box pile, 10 SI iters           (BASELINE.json)
  value    pile-steps per second over all ranks, scene resident in HBM before the timed region
  state    the scene is SETTLED before anything is timed: `--settle` steps (default 120, SURVEY 8d "warm-up 120 steps") are part
           of scene construction - untimed, not counted in `warmup` - so the number does not depend on the driver's --warmup;
           `--warmup` steps then warm caches / clocks, `--steps` steps are timed.
  N = 1    the 32 768-box brick-offset pile on a static plane, 10 velocity / 3 position iterations
  N > 1    `python bench.py --gpus N` starts its N ranks itself (torch.distributed.run on 127.0.0.1) when it was not launched by
           torchrun. The pile is ONE island and cannot be split: each rank steps its own replica of the pile (N islands
           sharded one per GPU, no data-path collective) and the integrated state (13 floats/body) is all-gathered over RCCL every
           step, as the registry write-back would need; value counts pile-steps, scaling = "weak".
  north_star  (second result in the same line) north_star's own target: 1 048 576 boxes in 16 384 islands, the islands sharded
           over the N ranks (strong scaling of ONE scene, per-step RCCL gather of the state): bodies, steps/s, >= 60 Hz yes/no.
  roofline the SI velocity-solve kernel actually launched (edynhip_stats::solve_schedule): algorithmic bytes
           (380 B per contact point + 256 B per joint row, per iteration) / time measured with HIP events recorded
           on the stepper's stream around that launch inside the timed region; beside the 8 TB/s spec peak the measured read
           and copy ceilings of this GPU (edynhip_measure_bandwidth).
  cpu_baseline  Edyn's own multithreaded CPU path: the REAL reference engine (oracle/_ref/libedynref.so = the reference's
           translation units compiled where they lie, driven through edyn::attach / step_simulation in
           execution_mode::sequential_multithreaded on all host cores) timed on a bounded sample of the same scene IN THE SAME
           (settled) STATE the GPU leg times, on rank 0 at N=1, in a subprocess with a wall-clock budget; next to it, in
           `sample`, the 1-thread restatement (oracle). Falls back to the restatement (kind "port", cores 1) where oracle/_ref is
           not built or the budget is exceeded. A reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import edyn_amd
from edyn_amd import scenes
from edyn_amd import _capi

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_POINT_ITER = 380.0   # algorithmic bytes per contact point per velocity iteration
BYTES_PER_JOINT_ROW_ITER = 256.0   # per joint row per iteration
GOLDEN = os.path.join(ROOT, "tests", "golden")
# parity statement that accompanies the number (tests/test_gpu_parity.py; the figures are asserted there)
# Contact arithmetic (edynhip.h EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / EDYNHIP_FLAG_BLOCK_POSITION): the headline is quoted on "reference" - every
# contact row and position correction with the reference's operations in the reference's order; the other two are opt-in and reported beside it.
ARITHMETIC = {
    "reference": dict(),
    "fused_velocity_rows": dict(fused_velocity_rows=True),
    "fused_rows_block_position": dict(fused_velocity_rows=True, block_position=True),
}
# what each mode costs in parity, measured on the CPU checker and asserted by tests/test_arithmetic_fork.py (coloured order, against the coloured
# order with the reference's arithmetic): worst of box_pile(6,6,6) 10 it / mixed pile 20 it / c1_columns
ARITHMETIC_FORK = {
    "reference": "the reference's row arithmetic (constraint_row.cpp:24-57, constraint_row_friction.cpp:11-54, contact_constraint.cpp:58-90, position_solver.hpp:16-51) in the coloured "
                 "visiting order: SURVEY 8(d)(3) holds with zero error (device bit-exact vs the checker's coloured order with ARITH_REFERENCE)",
    "fused_velocity_rows": "fp-level deviation: lock-step <= 9.5e-7 m (1 ulp at 9.5 m) / 7.2e-7 m/s per step; 60 free-running steps <= 9.0e-5 m / 6.7e-4 m/s - inside SURVEY 8(d)(3) (1e-4 m / 1e-3)",
    "fused_rows_block_position": "algorithmic deviation: lock-step up to 5.1e-4 m per step; 60 free-running steps 4.7e-4 m (box pile) / 2.1e-4 m (mixed pile) / 4.8e-2 m (straight "
                                 "columns) - SURVEY 8(d)(3) is NOT met in this mode",
}
PARITY = ("bit-exact vs the oracle's coloured order in this settled state at full size (pairs, state, manifolds, colours: "
          "test_timed_regime_at_full_size_bit_exact; islands1m: test_islands1m_at_full_size_bit_exact) - the coloured order evaluates every row with the "
          "REFERENCE's arithmetic (round 5 default; only the Gauss-Seidel visiting order differs from the reference), "
          "the oracle's reference order is pinned to the reference engine bit for bit; "
          "free-running vs the reference engine itself, C2 8000 boxes: 60 steps max |dpos| 0.23 m, mean 0.059 m; 300 steps (SURVEY 8(d)(4)): "
          "mean resting height within 3.4e-4 m, penetration of the resting pile 0.0036 / 0.0018 m, kinetic energy of the resting set 1.1e-5 / 1.2e-5 J per "
          "body - a Gauss-Seidel visiting-order effect of the unconverged 10-iteration solve on a collapsing lattice, not fp "
          "rounding (lock-step: 2e-3 m per step, pair sets and narrowphase bit-exact, also on C3 at full size: 1.6e-3 m; "
          "test_c2_300_steps_survey_invariants_*, test_free_running_c2_*, test_c3_full_size_lock_step_*, test_gpu_against_the_real_reference_engine); "
          "solver residual |Jv - rhs| over active normal rows, coloured order vs engine: it leaves less than the engine does (0.3-0.7x the mean, "
          "test_solver_residual_*; the device on C2 after 300 steps: profiles/r05_parity_figures.txt); jointed configurations in lock-step with the engine: "
          "exact with the engine's order replayed, 3.0e-3 m / 0.24 m/s per step on chains and 2.9e-2 m / 1.7 m/s on a collapsing rag-doll heap in the coloured order "
          "(test_jointed_*)")
WORKLOADS = {
    "pile32k": dict(gen=lambda: scenes.box_pile(32, 32, 32), vel=10, pos=3, settle=120, desc="32x32x32 = 32768-box brick-offset pile on a static plane"),
    "pile8k": dict(gen=lambda: scenes.box_pile(20, 20, 20), vel=10, pos=3, settle=120, desc="20x20x20 = 8000-box pile (config C2)"),
    "mixed32k": dict(gen=lambda: scenes.box_pile(32, 32, 32, mixed=True), vel=20, pos=3, settle=120, desc="32768 mixed box/sphere stack, 20 it (config C3)"),
    "pile512": dict(gen=lambda: scenes.box_pile(8, 8, 8), vel=10, pos=3, settle=20, desc="8x8x8 pile (smoke)"),
    # C4 shards by islands: with N ranks each rank steps its contiguous block of the 4096 sites, no data-path
    # collective - strong scaling of ONE scene (value = scene-steps/s), unlike the single-island pile's replicas
    "islands256k": dict(gen=lambda: scenes.c4_islands(), vel=10, pos=3, settle=120, desc="262144 boxes in 4096 independent 4x4x4 mini-piles (config C4)",
                        shard=lambda first, count: scenes.c4_islands(first_site=first, num_sites=count), shard_units=4096),
    # 1024 of the reference's own rag dolls (edyn::make_ragdoll, exported from the real engine: tests/golden/make_ragdoll.py)
    # collapsing on a plane: 22 529 bodies, 36 864 cone / cvjoint / hinge constraints, capsule contacts
    "ragdolls1k": dict(gen=lambda: scenes.figures(scenes.load_figure(os.path.join(GOLDEN, "ragdoll_capsule.npz")), 32, 32),
                       vel=10, pos=3, settle=120, desc="1024 rag dolls (22 bodies, 36 cone/cvjoint/hinge constraints each) falling on a plane"),
    # north_star's body count: 1 048 576 boxes in 16 384 islands (the C4 scene at 4x the sites)
    "islands1m": dict(gen=lambda: scenes.mini_piles(128, 128), vel=10, pos=3, settle=120, desc="1048576 boxes in 16384 independent 4x4x4 mini-piles",
                      shard=lambda first, count: scenes.mini_piles(128, 128, first_site=first, num_sites=count), shard_units=16384),
    # a small scene of the same kind (functional tests of the north_star leg)
    "islands4k": dict(gen=lambda: scenes.mini_piles(8, 8), vel=10, pos=3, settle=20, desc="4096 boxes in 64 independent 4x4x4 mini-piles (smoke)",
                      shard=lambda first, count: scenes.mini_piles(8, 8, first_site=first, num_sites=count), shard_units=64),
    # the headline pile with 64 rag dolls standing beside it: islands with joints next to a large island without (mixed schedule)
    "pile32k_ragdolls": dict(gen=lambda: _pile_and_ragdolls(), vel=10, pos=3, settle=120, desc="the 32768-box pile beside 64 rag dolls (36 constraints each)"),
    # every polyhedron pair routine at work: 32768 convex polyhedra (six meshes, random orientations) collapsing into a heap
    "polyheap32k": dict(gen=lambda: scenes.polyhedron_heap(32, 32, 32), vel=10, pos=3, settle=240, desc="32768 convex polyhedra (cubes, tetrahedra, octahedra, prisms, wedges) in a heap"),
    "chains16k": dict(gen=lambda: scenes.c5_chains(1024, 16), vel=10, pos=3, settle=120, desc="1024 chains x 16 links, hinge + point joints, no contacts (config C5)"),
}


def _pile_and_ragdolls():
    figs = scenes.figures(scenes.load_figure(os.path.join(GOLDEN, "ragdoll_capsule.npz")), 8, 8, floor=False)
    figs["pos"][:, 0] += np.float32(60.0)
    return scenes.merge(scenes.box_pile(32, 32, 32), figs)


# ------------------------------------------------------------------------------------------------ CPU legs (checker code, timed)
def _cpu_scene(workload):
    """The scene the CPU legs time: the workload itself, or - for the many-island scenes, whose reference run would need
    tens of GB - a block of its independent sites (steps/s then scale with the site count, stated in `sample`)."""
    wl = WORKLOADS[workload]
    if "shard" in wl:
        sites = min(256, wl["shard_units"])
        return wl["shard"](0, sites), wl["shard_units"] / sites, f"sites 0..{sites - 1} of {wl['shard_units']} (independent islands; steps/s divided by {wl['shard_units'] // sites})", sites
    return wl["gen"](), 1.0, "the whole scene", None


def _load_state(path, n):
    if not path:
        return None
    z = np.load(path)
    return tuple(np.ascontiguousarray(z[k][:n]) for k in ("pos", "orn", "linvel", "angvel"))


def cpu_reference_leg(workload, sample_steps, warm_steps, state_path):
    """Child process (bench.py --cpu-reference-leg): time the real reference engine, multithreaded. Prints one JSON line."""
    from oracle import binding as ob
    wl = WORKLOADS[workload]
    scene, scale, what, _ = _cpu_scene(workload)
    cores = os.cpu_count() or 1
    r = ob.RefWorld(vel_iters=wl["vel"], pos_iters=wl["pos"], mode=1, workers=0)   # sequential_multithreaded, hardware_concurrency - 1 workers + the caller
    r.add_bodies(scene); scenes.apply_figure_settings(r, scene)
    st = _load_state(state_path, len(scene["kind"]))
    if st is not None:
        r.set_state(*st)
    t0 = time.perf_counter()
    r.step(warm_steps)
    warm_s = time.perf_counter() - t0
    per_step = [r.time_steps(1) for _ in range(sample_steps)]   # one by one: the spread of the sample is reported beside its mean
    t = sum(per_step)
    print(json.dumps({"value": sample_steps / t / scale, "cores": cores, "warm_s": warm_s, "what": what,
                      "slowest": 1.0 / max(per_step) / scale, "fastest": 1.0 / min(per_step) / scale,
                      "points": int(r.get_manifolds()["num_points"].sum())}))


def cpu_baseline(workload, sample_steps, warm_steps, budget_s, state_path, settle):
    """CPU legs, timed on the host cores of the bench box, started from the state the GPU leg times (the device's settled
    transforms and velocities; `warm_steps` steps rebuild the contact manifolds there). Checker code, timed - never shipped."""
    from oracle import binding as ob
    wl = WORKLOADS[workload]
    scene, scale, what, _ = _cpu_scene(workload)
    o = ob.World(vel_iters=wl["vel"], pos_iters=wl["pos"], order=ob.ORDER_SEQUENTIAL)
    o.add_bodies(scene); scenes.apply_figure_settings(o, scene)
    st = _load_state(state_path, len(scene["kind"]))
    if st is not None:
        o.set_state(*st); o.refresh_derived()
    o.step(warm_steps)
    port = sample_steps / o.time_steps(sample_steps) / scale
    state_note = (f"the device's state after its {settle} settle steps" if st is not None else "its initial state")
    port_note = (f"1-thread restatement (oracle, reference row order): {port:.3f} steps/s over {sample_steps} steps after {warm_steps} "
                 f"contact-building steps of {what} from {state_note} ({o.get_stats()['num_points']} contact points at the end)")
    ref = None
    if ob.ref() is not None and budget_s > 0:
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-reference-leg", "--workload", workload,
                   "--cpu-sample-steps", str(sample_steps), "--cpu-warm-steps", str(warm_steps)]
            if state_path:
                cmd += ["--cpu-state", state_path]
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s)
            ref = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else None
        except (subprocess.TimeoutExpired, ValueError):
            ref = None
    if ref is not None:
        return {"value": ref["value"], "unit": "steps/sec", "cores": ref["cores"], "kind": "reference",
                # the spread: over the steps of this sample, and what kept driver runs of earlier rounds recorded on the headline scene
                # (box and state dependent: BENCH_r01..r03 1.29 / 1.88 / 1.13, profiles/r04 1.17) - a baseline, not a precise figure
                "spread": {"slowest_step": ref.get("slowest"), "fastest_step": ref.get("fastest"), "earlier_rounds_pile32k": [1.29, 1.88, 1.13, 1.17]},
                "sample": f"the reference engine itself (edyn::attach, execution_mode::sequential_multithreaded, {ref['cores']} host threads): "
                          f"{sample_steps} steps after {warm_steps} contact-building steps ({ref['warm_s']:.1f} s, ~{ref['points']} contact points) "
                          f"of {ref['what']} from {state_note}; beside it the {port_note}"}
    return {"value": port, "unit": "steps/sec", "cores": 1, "kind": "port",
            "sample": port_note + "; the multithreaded reference-engine leg was unavailable (oracle/_ref not built or over its time budget)"}


# ------------------------------------------------------------------------------------------------ launching N ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
    and pass their output through. Fails loudly when the box has fewer GPUs (unless the functional-test switch is set)."""
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ngpu < n and os.environ.get("EDYN_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py: --gpus {n} but this box has {ngpu} GPU(s) (EDYN_BENCH_SHARE_GPU=1 shares devices, functional tests only)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------ one measured leg
class Leg:
    """One workload, built, settled, warmed and timed on this rank's GPU (all ranks call the same sequence)."""

    def __init__(self, workload, args, rank, world_size, device_index, backend, stream, arithmetic=None):
        from edyn_amd.parallel import shard_range, StateGather
        self.name, self.wl = workload, WORKLOADS[workload]
        wl = self.wl
        self.rank, self.world_size, self.backend = rank, world_size, backend
        self.distributed = world_size > 1
        self.sharded = self.distributed and "shard" in wl
        if self.sharded:
            first, count = shard_range(wl["shard_units"], rank, world_size)
            scene = wl["shard"](first, count)
            per_site = (len(scene["kind"]) - 1) // count                          # bodies per site + the replicated static plane
            counts = [1 + per_site * shard_range(wl["shard_units"], r, world_size)[1] for r in range(world_size)]
            self.total_bodies = 1 + per_site * wl["shard_units"]
        else:
            scene = wl["gen"]()
            counts = [len(scene["kind"])] * world_size
            self.total_bodies = len(scene["kind"])
        self.n_bodies = len(scene["kind"])
        assert counts[rank] == self.n_bodies
        cfg = edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"],
                                   device=device_index, timing=args.stage_timing, timing_solve=not args.stage_timing,
                                   # a single-rank run owns its GPU: plain launches for the resident-grid kernels;
                                   # RCCL kernels share the GPU in multi-rank runs: cooperative launches there
                                   exclusive_device=not self.distributed, **ARITHMETIC[arithmetic or args.arithmetic])
        self.w = edyn_amd.World(cfg)
        self.w.set_scene(scene)
        scenes.apply_figure_settings(self.w, scene)
        self.w.set_stream(stream.cuda_stream)
        self.gath = StateGather(counts, "cuda", backend) if self.distributed else None   # edyn_amd.parallel: the registry write-back gather
        self.settle = wl["settle"] if args.settle is None else args.settle

    def one_step(self):
        self.w.step_simulation(1)
        if self.distributed:
            self.w.pack_state_device(self.gath.local.data_ptr())   # same stream as the stepper and (through torch) the collective
            self.gath.gather()

    def run(self, steps, warmup):
        """settle (scene construction) -> warm-up -> barrier + sync -> `steps` timed steps -> sync + barrier; max over ranks."""
        self.w.step_simulation(self.settle)
        for _ in range(warmup):
            self.one_step()
        if self.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if self.distributed:
            for _ in range(steps):
                self.one_step()
        else:
            self.w.step_simulation(steps)   # K steps, stage events recorded per step on the stepper's stream
        torch.cuda.synchronize()
        if self.distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if self.distributed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        self.elapsed, self.steps = elapsed, steps
        # scene-steps per second: sharded = ONE scene stepped once per step by all ranks together; replicas = one scene per rank
        self.value = (1 if self.sharded else self.world_size) * steps / elapsed
        return self.value

    def close(self):
        self.w = None
        self.gath = None
        torch.cuda.empty_cache()


def per_rank_proxy(workload, args, device_index, stream, ranks):
    from edyn_amd.parallel import shard_range
    wl = WORKLOADS[workload]
    first, count = shard_range(wl["shard_units"], 0, ranks)
    scene = wl["shard"](first, count)
    cfg = edyn_amd.init_config(num_solver_velocity_iterations=wl["vel"], num_solver_position_iterations=wl["pos"], device=device_index,
                               exclusive_device=True, **ARITHMETIC[args.arithmetic])
    w = edyn_amd.World(cfg)
    w.set_scene(scene)
    w.set_stream(stream.cuda_stream)
    n = len(scene["kind"])
    buf = torch.zeros((n, 13), dtype=torch.float32, device="cuda")
    settle = wl["settle"] if args.settle is None else args.settle
    w.step_simulation(settle)
    def one():
        w.step_simulation(1)
        w.pack_state_device(buf.data_ptr())
    for _ in range(min(args.warmup, 10)):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.north_star_steps):
        one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = w.get_stats()
    finite = bool(torch.isfinite(buf).all().item())
    rate = args.north_star_steps / el
    return {"what": f"rank 0's block of a {ranks}-rank run of {workload} (sites {first}..{first + count - 1}) stepped on ONE GPU with the per-step state pack "
                    f"(the send buffer of the all-gather) in the loop; settled {settle} steps",
            "bodies": n, "contact_points": st["num_points"], "islands": st["num_islands"], "steps": args.north_star_steps,
            "steps_per_sec": rate, "ms_per_step": 1e3 * el / args.north_star_steps, "target_hz": 60.0, "meets_60hz": bool(rate >= 60.0), "finite": finite,
            "reading": f"measured upper bound of the {ranks}-GPU rate of {workload}: no data-path collective exists between islands; the state gather over xGMI "
                       f"(13 floats per body) and the slowest of the {ranks} ranks come on top - not a {ranks}-GPU measurement"}


def shim_leg(steps, settle):
    """The drop-in itself: tests/cpp/bench_update.cpp - the headline pile built with edyn::make_rigidbody,
    edyn::update(registry, t) once per step through include/edyn/edyn.hpp - in execution_mode::sequential and ::asynchronous, default
    and exclusive_device launch modes, against edynhip_step on the very same context. A separate process (plain C++ over the C-ABI)."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    exe = os.path.join(cpp, "bench_update")
    try:
        subprocess.check_call(["make", "-s", "-C", cpp, "bench_update"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res = subprocess.run([exe, "32", str(settle), str(steps)], capture_output=True, text=True, timeout=600)
    except Exception as e:   # no compiler on the box, a time-out: the bench line says so instead of failing
        return {"error": f"{type(e).__name__}: {e}"}
    runs = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    if not runs or "BENCH_UPDATE_OK" not in res.stdout:
        return {"error": "bench_update failed", "tail": (res.stdout + res.stderr)[-400:]}
    return {"what": "edyn::update(registry, t) over the registry, one fixed step per call, on the headline pile built with edyn::make_rigidbody (tests/cpp/bench_update.cpp); "
                    "bundled mini registry; the shim's defaults: island sleeping on, contact_manifold / contact_point entities kept in step with the device "
                    f"(~{runs[0]['contact_events_per_update']:.0f} creations + destructions per update); settled {settle} updates, {steps} timed",
            "steps_per_sec": {r["run"]: r["update_steps_per_sec"] for r in runs},
            "raw_steps_per_sec_same_context": {r["run"]: r["raw_steps_per_sec"] for r in runs},
            "ratio_to_raw_same_context": {r["run"]: r["ratio"] for r in runs},
            "host_ms_per_update": {r["run"]: r["host_ms_per_update"] for r in runs},
            "ms_per_update": {r["run"]: r["ms_per_update"] for r in runs},
            "contact_point_entities": runs[0]["contact_point_entities"],
            "reading": "host_ms_per_update = registry write-back + contact entities + removal hooks + presentation (the latter now computed on the device and "
                       "delivered with the state); in sequential mode the contact entities are built while the step's solve still runs on the device, in "
                       "asynchronous mode the whole import overlaps the next step; step_call = host time inside edynhip_step (the step's counter fetches), "
                       "state_wait = waiting for the rest of the step and the 96-byte-per-body record copy"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=None, help="settle steps before anything is timed (scene construction; default per workload: 120)")
    ap.add_argument("--workload", default="pile32k", choices=sorted(WORKLOADS))
    ap.add_argument("--north-star", default="auto", help="workload of the north_star leg (sharded islands): a workload name, 'none', or 'auto' = islands1m on the default workload")
    ap.add_argument("--north-star-steps", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shim", action="store_true", help="skip the edyn::update leg (tests/cpp/bench_update: the C++ drop-in over the registry)")
    ap.add_argument("--shim-steps", type=int, default=300)
    ap.add_argument("--arithmetic", default="reference", choices=sorted(ARITHMETIC),
                    help="contact arithmetic of the measured run (default: the reference's operations; the others are the opt-in faster forms)")
    ap.add_argument("--other-arithmetic-steps", type=int, default=None,
                    help="timed steps of the two short legs in the OTHER contact arithmetics (config.arithmetic.steps_per_sec; default: --steps on the default workload at N=1, else 0)")
    ap.add_argument("--stage-timing", action="store_true",
                    help="HIP events around every stage (adds stages_ms_per_step; each event idles the GPU ~6 us, so the headline\n                    value is measured without it: only the two events around the velocity solve are recorded)")
    ap.add_argument("--cpu-sample-steps", type=int, default=6)
    ap.add_argument("--cpu-warm-steps", type=int, default=3)
    ap.add_argument("--cpu-budget-s", type=float, default=240.0, help="wall-clock budget of the reference-engine CPU leg")
    ap.add_argument("--cpu-reference-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-state", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_reference_leg:
        cpu_reference_leg(args.workload, args.cpu_sample_steps, args.cpu_warm_steps, args.cpu_state)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)   # does not return

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and rank == 0:
        print(f"bench.py: launched with WORLD_SIZE={world_size} but --gpus {args.gpus}; measuring {world_size} ranks", file=sys.stderr)
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

    # The stepper, the pack kernel and the RCCL gather all run on ONE explicitly created torch stream (a non-zero
    # handle: edynhip_set_stream(NULL) would mean "a private stream"), entered for the whole run => ordered.
    stream = torch.cuda.Stream(device=device_index)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0

    wl = WORKLOADS[args.workload]
    leg = Leg(args.workload, args, rank, world_size, device_index, backend, stream)
    leg.run(args.steps, args.warmup)
    w = leg.w
    stats = w.get_stats()
    tm = w.get_timings()
    state = w.get_state()
    finite = bool(np.isfinite(state[0]).all())
    read_gbs, copy_gbs = w.measure_bandwidth(1 << 30) if rank == 0 else (0.0, 0.0)

    out = None
    state_path = None
    if rank == 0:
        dist_note = ""
        if distributed:
            how = (f"sharded by islands over {world_size} ranks ({wl['shard_units']} sites, contiguous blocks), no data-path collective"
                   if leg.sharded else f"{world_size} replicas (one island per GPU)")
            dist_note = f"; {how}, per-step {'RCCL' if backend == 'nccl' else 'gloo (shared-GPU functional test)'} all-gather of state"
        steps_timed = max(tm["steps"], 1)
        solve_ms = tm["solve_velocity_ms"] / steps_timed
        launches = tm["solve_velocity_launches"] / max(args.steps if not distributed else 1, 1)
        sweeps = wl["vel"] + 1   # +1: the warm-start sweep
        alg_bytes_step = (BYTES_PER_POINT_ITER * stats["num_points"] + BYTES_PER_JOINT_ROW_ITER * stats["num_joint_rows"]) * sweeps
        # Measured fabric traffic: a KEPT rocprofv3 PMC profile of this workload (profiles/traffic.json, keyed by workload, made by
        # scripts/profile_round.sh; FETCH_SIZE / WRITE_SIZE cannot be read from inside the run). The profile stores the traffic of its
        # timed steps PER ALGORITHMIC BYTE, so it scales to this run's own contact-point / joint-row counts; it only counts for the
        # schedule (the kernels) it was measured on. No profile for the schedule that ran => traffic, achieved and frac are null.
        schedule = _capi.SCHEDULE_NAMES.get(stats["solve_schedule"], str(stats["solve_schedule"]))
        traffic = None
        traffic_note = "no kept rocprofv3 PMC profile of this workload on this solve schedule (profiles/traffic.json): traffic, achieved and frac are null"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(args.workload)
                if isinstance(ent, dict) and ent.get("schedule") == schedule and ent.get("traffic_per_algorithmic_byte"):
                    traffic = float(ent["traffic_per_algorithmic_byte"]) * alg_bytes_step / max(launches, 1)
                    traffic_note = (f"kept rocprofv3 PMC profile (profiles/traffic.json: {ent['traffic_per_algorithmic_byte']:.3f} fabric bytes per algorithmic byte over "
                                    f"the timed steps of a run with {ent['contact_points']} contact points / {ent['joint_rows']} joint rows on the same schedule), scaled to this "
                                    f"run's counts; not measured in this run")
            except Exception:
                traffic = None
        per_launch_alg = alg_bytes_step / max(launches, 1)
        # achieved = min(algorithmic, measured) bytes / kernel time
        achieved = None
        if traffic is not None and solve_ms > 0:
            achieved = (min(alg_bytes_step, traffic * max(launches, 1)) / 1e9) / (solve_ms / 1e3)
        achieved_alg = (alg_bytes_step / 1e9) / (solve_ms / 1e3) if solve_ms > 0 else 0.0
        # The denominator stays the 8 TB/s spec peak. Beside it: what this GPU streams when every CU reads 16 B per lane
        # (the practical ceiling of a read-dominated kernel like the solve) and what a device-to-device copy moves (read +
        # write bytes; a copy alternates reads and writes on every channel and is slower than a pure read stream).
        peak = HBM_PEAK_GBS
        out = {
            "metric": "steps/sec (whole node), 32k-box pile, 10 SI iters; HBM GB/s in solve",
            "value": leg.value, "unit": "steps/sec", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * leg.elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if leg.sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}; {wl['vel']} velocity / {wl['pos']} position iterations, dt 1/60, "
                                   f"friction 0.5, restitution 0; settled for {leg.settle} steps before warm-up and timing" + dist_note,
                       "settle_steps": leg.settle, "bodies": leg.n_bodies, "contact_points": stats["num_points"], "manifolds": stats["num_manifolds"],
                       "joint_rows": stats["num_joint_rows"],
                       "colours": stats["num_colours"], "colour_sizes": stats["colour_size"], "islands": stats["num_islands"], "finite": finite,
                       "arithmetic": {"mode": args.arithmetic, "parity": ARITHMETIC_FORK[args.arithmetic], "steps_per_sec": {args.arithmetic: leg.value}},
                       "parity": PARITY},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": (achieved / peak) if achieved is not None else None,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "achieved_from_algorithmic_bytes_only": achieved_alg, "frac_from_algorithmic_bytes_only": achieved_alg / peak,
                         "measured_read_ceiling": read_gbs, "frac_of_measured_read_ceiling": (achieved / read_gbs) if achieved is not None and read_gbs > 0 else None,
                         "measured_copy_ceiling": copy_gbs, "frac_of_measured_copy_ceiling": (achieved / copy_gbs) if achieved is not None and copy_gbs > 0 else None,
                         "achieved_rule": "min(algorithmic bytes, measured traffic) / kernel time; peak = HBM3E spec; the measured read-stream and device-to-device copy rates of this GPU are reported beside it",
                         "kernel": schedule,
                         "algorithmic_bytes_per_launch": per_launch_alg,
                         "algorithmic_bytes_rule": "(380 B x contact points + 256 B x joint rows) x (velocity iterations + 1 warm-start sweep)",
                         "launches_per_step": launches,
                         "avg_launch_us": 1e3 * solve_ms / max(launches, 1), "solve_ms_per_step": solve_ms},
        }
        # the kept 300-step figure of this workload beside the one of this run (a short driver run samples 20 steps: VERDICT r05 next #8)
        kpath = os.path.join(ROOT, "profiles", "kept_runs.json")
        if os.path.exists(kpath):
            try:
                kept = json.load(open(kpath)).get(args.workload)
                if isinstance(kept, dict):
                    out["roofline"]["frac_profile_300"] = kept.get("frac")
                    out["roofline"]["frac_profile_300_from_algorithmic_bytes_only"] = kept.get("frac_from_algorithmic_bytes_only")
                    out["roofline"]["frac_profile_300_source"] = kept.get("source")
            except Exception:
                pass
        if args.stage_timing:
            out["stages_ms_per_step"] = {k: tm[k] / steps_timed for k in ("broadphase_ms", "narrowphase_ms", "islands_ms", "colouring_ms",
                                                                         "prepare_ms", "solve_velocity_ms", "integrate_ms", "solve_position_ms",
                                                                         "finish_ms", "step_ms")}
        if world_size == 1 and not args.no_cpu_baseline:
            # the CPU legs start from the state the GPU leg was timed in (the first bodies of a sharded scene: its first sites)
            fd, state_path = tempfile.mkstemp(suffix=".npz", prefix="edyn_bench_state_")
            os.close(fd)
            np.savez(state_path, pos=state[0], orn=state[1], linvel=state[2], angvel=state[3])
    leg.close()
    del w

    # ---- the same workload in the other contact arithmetics (single rank): both numbers beside the headline
    other_steps = args.other_arithmetic_steps
    if other_steps is None:
        other_steps = args.steps if (args.workload == "pile32k" and world_size == 1) else 0
    if other_steps > 0 and world_size == 1:
        for mode in ARITHMETIC:
            if mode == args.arithmetic:
                continue
            other = Leg(args.workload, args, rank, world_size, device_index, backend, stream, arithmetic=mode)
            other.run(other_steps, args.warmup)
            otm = other.w.get_timings()
            out["config"]["arithmetic"]["steps_per_sec"][mode] = other.value
            out["config"]["arithmetic"].setdefault("solve_ms_per_step", {args.arithmetic: out["roofline"]["solve_ms_per_step"]})[mode] = otm["solve_velocity_ms"] / max(otm["steps"], 1)
            out["config"]["arithmetic"].setdefault("other_modes", {})[mode] = ARITHMETIC_FORK[mode]
            other.close()

    # ---- north_star: 1M bodies in 16k islands, sharded over the ranks (strong scaling), >= 60 Hz?
    ns_name = args.north_star
    if ns_name == "auto":
        ns_name = "islands1m" if args.workload == "pile32k" else "none"
    if ns_name != "none":
        if ns_name not in WORKLOADS or "shard" not in WORKLOADS[ns_name]:
            raise SystemExit(f"bench.py: --north-star {ns_name}: not a shardable workload")
        ns = Leg(ns_name, args, rank, world_size, device_index, backend, stream)
        ns.run(args.north_star_steps, min(args.warmup, 10))
        ns_stats = ns.w.get_stats()
        ns_finite = bool(np.isfinite(ns.w.get_state()[0]).all())
        if rank == 0:
            out["north_star"] = {
                "workload": f"{ns_name}: {WORKLOADS[ns_name]['desc']}; islands sharded over {world_size} rank(s) in contiguous site blocks, no data-path "
                            f"collective" + (f", per-step {'RCCL' if backend == 'nccl' else 'gloo'} all-gather of the state" if distributed else "")
                            + f"; settled {ns.settle} steps",
                "bodies": ns.total_bodies, "n_gpus": world_size, "scaling": "strong", "steps": args.north_star_steps,
                "steps_per_sec": ns.value, "ms_per_step": 1e3 * ns.elapsed / args.north_star_steps,
                "target_hz": 60.0, "meets_60hz": bool(ns.value >= 60.0),
                "bodies_this_rank": ns.n_bodies, "contact_points_this_rank": ns_stats["num_points"], "finite": ns_finite}
        ns.close()
        # north_star.per_rank_proxy: what ONE rank of an 8-GPU run has to do - its 1/8 block of the sites
        # (131 072 boxes, 2 048 islands) stepped on this GPU with the per-step state pack (edynhip_pack_state_device into the buffer
        # an RCCL all-gather would send) in the loop. Islands need no data-path collective, so this is a MEASURED UPPER BOUND of the
        # 8-GPU rate (the gather over xGMI and the slowest of eight ranks come on top), stated as such.
        if rank == 0 and world_size == 1 and "shard" in WORKLOADS[ns_name] and WORKLOADS[ns_name]["shard_units"] >= 8:
            out["north_star"]["per_rank_proxy"] = per_rank_proxy(ns_name, args, device_index, stream, 8)

    if rank == 0 and world_size == 1 and args.workload == "pile32k" and not args.no_shim:
        out["shim"] = shim_leg(args.shim_steps, leg.settle)
        if "steps_per_sec" in out["shim"]:
            out["shim"]["ratio_to_value"] = {k: v / out["value"] for k, v in out["shim"]["steps_per_sec"].items()}
    if rank == 0:
        if state_path is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_steps, args.cpu_warm_steps, args.cpu_budget_s, state_path, leg.settle)
            finally:
                os.unlink(state_path)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
