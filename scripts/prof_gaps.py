"""Inter-kernel gap analysis from a rocprofv3 rocpd database: duration and gap-to-next for one kernel family."""
import sqlite3, sys, glob, os
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_contact_solve"
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
db = sqlite3.connect(path); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select name, start, end from kernels order by start"))
import statistics
durs, gaps = [], []
for i, (name, s, e) in enumerate(rows[:-1]):
    if pat in name:
        durs.append((e - s) / 1e3)
        gaps.append((rows[i + 1][1] - e) / 1e3)
print(f"{pat}: n={len(durs)} dur median {statistics.median(durs):.2f} us mean {statistics.mean(durs):.2f}; gap-to-next median {statistics.median(gaps):.2f} us mean {statistics.mean(gaps):.2f}")
