"""profiles/kept_runs.json: the roofline figures of the round's kept 300-step bench runs, keyed by workload - what bench.py prints as
roofline.frac_profile_300 beside the figure of its own (possibly 20-step) run.  usage: keep_runs.py profiles/r06_bench_default.json [profiles/r06_bench_pile8k.json ...]"""
import json, os, sys
out = {}
for path in sys.argv[1:]:
    try:
        j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as e:
        print("skipped", path, e, file=sys.stderr); continue
    wl = j["config"]["workload"].split(":")[0]
    r = j["roofline"]
    out[wl] = {"frac": r.get("frac"), "frac_from_algorithmic_bytes_only": r.get("frac_from_algorithmic_bytes_only"), "solve_ms_per_step": r.get("solve_ms_per_step"),
               "steps": j["steps"], "value": j["value"], "source": path}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "kept_runs.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
