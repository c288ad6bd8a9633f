"""HBM-side traffic of the velocity-solve kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases),
over the TIMED steps of the profiled bench run only (its last `steps` steps: the scene is settled there, the contact count is the
one the bench line reports), stored so that bench.py can scale it to ITS run's own counts:

    traffic_per_algorithmic_byte = measured bytes per step / ((380 B x contact points + 256 B x joint rows) x sweeps)

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B -> x2 for these
kernels' 16 B/lane row streams (the dataflow kernels' hand-off polls are device-coherent 16-B reads and are counted too: they are
real fabric traffic); WRITE_SIZE is uncalibrated and taken as is. Units: FETCH/WRITE_SIZE in KiB. Infinity-Cache hits are included
in these fabric-side counters.
usage: pmc_traffic.py <FETCH_SIZE dir> <WRITE_SIZE dir> <workload> <bench json of the FETCH_SIZE pass>"""
import json, sqlite3, sys, glob, os

SOLVE_KERNELS = ("k_contact_solve", "k_joint_solve", "k_island_velocity")   # every schedule's velocity-solve kernels


def launches(path, counter):
    """[(kernel name, value)] of the solve kernels in dispatch order"""
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    order = next((c for c in ("dispatch_id", "start", "id", "event_id") if c in cols), None)
    q = "select name, counter_value from pmc_events where counter_name = ?" + (f" order by {order}" if order else "")
    return [(n, v) for n, v in cur.execute(q, (counter,)) if any(k in n for k in SOLVE_KERNELS)]


fetch_dir, write_dir, workload, bench_json = sys.argv[1:5]
bench = json.loads([l for l in open(bench_json).read().splitlines() if l.strip().startswith("{")][-1])
steps = int(bench["steps"])
roof, cfg = bench["roofline"], bench["config"]
per_step = float(roof["launches_per_step"])
timed = int(round(per_step * steps))
f, w = launches(fetch_dir, "FETCH_SIZE"), launches(write_dir, "WRITE_SIZE")
if timed <= 0 or len(f) < timed or len(w) < timed:
    raise SystemExit(f"pmc_traffic: {len(f)} / {len(w)} solve launches profiled, {timed} expected in the timed region")
ft, wt = f[-timed:], w[-timed:]
fetch_bytes = 2.0 * 1024 * sum(v for _, v in ft)      # the gfx950 wide-read correction
write_bytes = 1024.0 * sum(v for _, v in wt)
alg_step = float(roof["algorithmic_bytes_per_launch"]) * per_step
kernels = {}
for n, v in ft:
    k = n.replace("void ", "").split("(")[0][:60]
    kernels.setdefault(k, [0, 0.0])
    kernels[k][0] += 1; kernels[k][1] += 2.0 * 1024 * v
out = {"workload": workload, "schedule": roof["kernel"], "contact_points": cfg["contact_points"], "joint_rows": cfg["joint_rows"],
       "timed_steps": steps, "launches_per_step": per_step,
       "fetch_bytes_per_step_corrected": fetch_bytes / steps, "write_bytes_per_step": write_bytes / steps,
       "hbm_bytes_per_step": (fetch_bytes + write_bytes) / steps, "hbm_bytes_per_launch": (fetch_bytes + write_bytes) / timed,
       "algorithmic_bytes_per_step": alg_step, "traffic_per_algorithmic_byte": (fetch_bytes + write_bytes) / steps / alg_step,
       "kernels": {k: {"launches_per_step": c / steps, "fetch_bytes_per_launch_corrected": b / c} for k, (c, b) in kernels.items()},
       "note": "timed region of the profiled bench run only; FETCH_SIZE doubled per the gfx950 wide-read correction; WRITE_SIZE uncalibrated; "
               "Infinity-Cache hits are included in these fabric-side counters"}
print(json.dumps(out, indent=1))
