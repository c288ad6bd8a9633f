"""Per-kernel averages of PMC counters from a rocprofv3 --pmc rocpd database."""
import sqlite3, sys, glob, os, collections
path = sys.argv[1]
if os.path.isdir(path): path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(path).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
    k = name.replace("void ", "").split("(")[0][:44]
    a = agg[k][cname]; a[0] += 1; a[1] += val
names = sorted({c for k in agg for c in agg[k]})
print(f"{'kernel':44s} {'n':>7s} " + " ".join(f"{c[-14:]:>14s}" for c in names))
for k in sorted(agg, key=lambda k: -sum(v[1] for v in agg[k].values())):
    n = max(v[0] for v in agg[k].values())
    print(f"{k:44s} {n:7d} " + " ".join(f"{agg[k][c][1] / max(agg[k][c][0], 1):14.1f}" for c in names))
