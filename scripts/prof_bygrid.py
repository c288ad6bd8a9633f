import sqlite3, sys, glob, os, statistics, collections
path = sys.argv[1]; pat = sys.argv[2]
if os.path.isdir(path): path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
db = sqlite3.connect(path); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
by = collections.defaultdict(list)
for name, s, e, g, wg in rows:
    if pat in name: by[(name.split('(')[0][-30:], g // max(wg,1))].append((e - s) / 1e3)
for k in sorted(by, key=lambda k: k[1]):
    v = by[k]
    print(f"{k[0]:32s} blocks {k[1]:6d}  n {len(v):5d}  median {statistics.median(v):7.2f} us  min {min(v):7.2f}")
