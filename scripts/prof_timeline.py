"""Timeline of ONE step from a rocprofv3 rocpd database: every kernel with its start offset, duration and the gap to the
previous kernel's end - shows launch bubbles (e.g. around cooperative launches and host syncs)."""
import sqlite3, sys, glob, os
path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if os.path.isdir(path): path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(path).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_bp_refit" in r[0]]   # first kernel of every step
a, b = starts[which], starts[which + 1]
t0 = rows[a][1]; prev_end = t0
tot_gap = 0
for name, s, e in rows[a:b]:
    short = name.replace("void ", "").split("(")[0][-44:]
    gap = (s - prev_end) / 1e3
    tot_gap += max(gap, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {short}")
    prev_end = max(prev_end, e)
print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us, sum of gaps {tot_gap:.1f} us, kernels {b - a}")
