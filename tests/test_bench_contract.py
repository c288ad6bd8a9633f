"""bench.py prints ONE JSON line with the driver's contract keys plus roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "pile512", "--steps", "10", "--warmup", "5",
                          "--cpu-sample-steps", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "steps/sec" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["finite"] is True
    assert d["config"]["settle_steps"] == 20 and d["config"]["colours"] >= 1 and "parity" in d["config"]
    # the headline is quoted on the reference's contact arithmetic; the opt-in forms are separate legs (default workload only)
    a = d["config"]["arithmetic"]
    assert a["mode"] == "reference" and list(a["steps_per_sec"]) == ["reference"] and abs(a["steps_per_sec"]["reference"] - d["value"]) < 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if r["traffic"] is None:   # no kept counter profile for this workload / schedule (pile512 has none): nothing is claimed
        assert r["achieved"] is None and r["frac"] is None and "no kept" in r["traffic_source"]
    else:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["frac_from_algorithmic_bytes_only"] > 0 and r["algorithmic_bytes_per_launch"] > 0
    assert r["kernel"].startswith("k_contact_solve_df") and r["measured_read_ceiling"] > 1000 and r["measured_copy_ceiling"] > 1000
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    if c["kind"] == "reference":   # the real engine, multithreaded: both thread counts are in the sample text
        assert c["cores"] > 1 and "1-thread" in c["sample"] and "sequential_multithreaded" in c["sample"]


def _two_rank_checks(out):
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    ns = d["north_star"]   # the sharded-islands leg: ONE scene over both ranks
    assert ns["n_gpus"] == 2 and ns["scaling"] == "strong" and ns["bodies"] == 64 * 64 + 1 and ns["bodies_this_rank"] == 32 * 64 + 1
    assert ns["steps_per_sec"] > 0 and ns["finite"] is True and isinstance(ns["meets_60hz"], bool)


def test_bench_two_ranks_on_one_gpu_functional():
    """The N>1 code path (rendezvous, per-step pack + all-gather, max-over-ranks timing, the sharded north_star leg) on a 1-GPU
    box: two ranks share cuda:0 and gather through gloo. The 8-GPU RCCL run itself is the driver's; this guards everything
    around it. Launched the way the driver launches it (torch.distributed.run) ..."""
    env = dict(os.environ, EDYN_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "pile512",
           "--steps", "8", "--warmup", "4", "--north-star", "islands4k", "--north-star-steps", "6"]
    _two_rank_checks(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env))


def test_bench_gpus_flag_starts_its_own_ranks():
    """... and as plain `python bench.py --gpus 2`: without torchrun's environment bench.py starts the two ranks itself
    (--gpus used to be parsed and ignored)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EDYN_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "pile512", "--steps", "8", "--warmup", "4",
           "--north-star", "islands4k", "--north-star-steps", "6"]
    _two_rank_checks(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env))


def test_bench_reports_every_contact_arithmetic_beside_the_headline():
    """the line carries steps/s of the headline workload in all three contact arithmetics - the default (the
    reference's operations) is `value`, the two opt-in forms are measured beside it with what each costs in parity."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "pile512", "--steps", "10", "--warmup", "5", "--north-star", "none",
                          "--other-arithmetic-steps", "10", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    a = d["config"]["arithmetic"]
    assert a["mode"] == "reference" and set(a["steps_per_sec"]) == {"reference", "fused_velocity_rows", "fused_rows_block_position"}
    assert all(v > 0 for v in a["steps_per_sec"].values()) and a["steps_per_sec"]["reference"] == d["value"]
    assert "NOT met" in a["other_modes"]["fused_rows_block_position"] and "zero error" in a["parity"]


def test_bench_eight_ranks_on_one_gpu_functional():
    """the north_star leg at the world size it is meant for - 8 ranks, the islands sharded in contiguous site blocks, the
    per-step pack + all-gather of every rank's state - as a functional run on the one GPU of the test box (ranks share cuda:0, the gather goes
    through gloo). The 8-GPU RCCL measurement itself is the driver's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EDYN_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "islands4k", "--steps", "6", "--warmup", "2",
           "--north-star", "islands4k", "--north-star-steps", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["finite"] is True
    ns = d["north_star"]
    assert ns["n_gpus"] == 8 and ns["scaling"] == "strong" and ns["bodies"] == 64 * 64 + 1 and ns["bodies_this_rank"] == 8 * 64 + 1 and ns["finite"] is True
