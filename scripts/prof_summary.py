"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a per-kernel table."""
import sqlite3, sys, glob, os
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
db = sqlite3.connect(path)
cur = db.cursor()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
# Scene SET-UP (uploads, list builds, per-call copies of edynhip_set_* / edynhip_exclude_collision) is told apart from the STEPS: the
# stepping window opens 20 ms before the first k_integrate launch (a kernel only steps run; a step's earlier kernels precede it by ~1 ms).
# Per-step columns count the window only (43 042 set-up copies used to show as 224 us "per step").
launches = list(cur.execute("select name, start, end from kernels order by start"))
first_step = min((s for n, s, e in launches if "k_integrate" in n), default=None)
t_open = first_step - 20_000_000 if first_step is not None else (launches[0][1] if launches else 0)
agg = {}
for n, s, e in launches:
    a = agg.setdefault(n, [0, 0.0, 0, 0.0])   # window calls, window ns, set-up calls, set-up ns
    if s >= t_open: a[0] += 1; a[1] += e - s
    else: a[2] += 1; a[3] += e - s
total_window = sum(a[1] for a in agg.values()) or 1.0
rows = sorted(((n, a[0], a[1] / 1e3, (a[1] / a[0] / 1e3) if a[0] else 0.0, 100.0 * a[1] / total_window, a[2], a[3] / 1e3) for n, a in agg.items()), key=lambda r: -(r[2] + r[6]))
print(f"# {os.path.basename(path)}  (durations in us; per_step = total of the stepping window / {steps:g} steps; set-up = launches before the first step, not in the per-step columns)")
print(f"{'kernel':60s} {'calls':>8s} {'total_us':>12s} {'avg_us':>9s} {'pct':>6s} {'us/step':>10s} {'setup calls':>12s} {'setup us':>10s}")
for name, calls, total, avg, pct, su_calls, su_us in rows:
    short = name.replace("void ", "").split("(")[0]
    if "rocprim" in short:   # keep what tells the library kernels apart
        import re
        m = re.search(r"(radix_sort_\w+|onesweep\w*|histogram\w*|scan\w*|merge\w*|block_sort\w*|lookback\w*)", name)
        short = "rocprim::" + (m.group(1) if m else "?") + " " + short[-24:]
    short = short[:60]
    print(f"{short:60s} {calls:8d} {total:12.1f} {avg:9.3f} {pct:6.2f} {total / steps:10.1f} {su_calls:12d} {su_us:10.1f}")

# The bench's roofline figure covers the TIMED region only (the last `timed` steps of the run): report the roofline
# kernel's average over exactly those launches as well, so that it can be compared with bench.py's avg_launch_us.
if len(sys.argv) > 4:
    pat, timed = sys.argv[3], int(sys.argv[4])
    d = [(e - s) / 1e3 for name, s, e in cur.execute("select name, start, end from kernels order by start") if pat in name]
    per_step = len(d) / steps
    tail = d[-int(round(timed * per_step)):]
    print(f"# {pat}: {len(d)} launches; timed region = last {timed} steps = {len(tail)} launches: avg {sum(tail) / len(tail):.3f} us, "
          f"{sum(tail) / timed:.1f} us/step")
