"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a per-kernel table."""
import sqlite3, sys, glob, os
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
db = sqlite3.connect(path)
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"# {os.path.basename(path)}  (durations in us; per_step = total / {steps:g} steps)")
print(f"{'kernel':60s} {'calls':>8s} {'total_us':>12s} {'avg_us':>9s} {'pct':>6s} {'us/step':>10s}")
for name, calls, total, avg, pct in rows:
    short = name.replace("void ", "").split("(")[0]
    if "rocprim" in short:   # keep what tells the library kernels apart
        import re
        m = re.search(r"(radix_sort_\w+|onesweep\w*|histogram\w*|scan\w*|merge\w*|block_sort\w*|lookback\w*)", name)
        short = "rocprim::" + (m.group(1) if m else "?") + " " + short[-24:]
    short = short[:60]
    print(f"{short:60s} {calls:8d} {total:12.1f} {avg:9.3f} {pct:6.2f} {total / steps:10.1f}")

# The bench's roofline figure covers the TIMED region only (the last `timed` steps of the run): report the roofline
# kernel's average over exactly those launches as well, so that it can be compared with bench.py's avg_launch_us.
if len(sys.argv) > 4:
    pat, timed = sys.argv[3], int(sys.argv[4])
    d = [(e - s) / 1e3 for name, s, e in cur.execute("select name, start, end from kernels order by start") if pat in name]
    per_step = len(d) / steps
    tail = d[-int(round(timed * per_step)):]
    print(f"# {pat}: {len(d)} launches; timed region = last {timed} steps = {len(tail)} launches: avg {sum(tail) / len(tail):.3f} us, "
          f"{sum(tail) / timed:.1f} us/step")
