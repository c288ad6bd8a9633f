"""Fabric-side bytes per launch of ANY kernel from the two rocprofv3 --pmc passes profile_round.sh takes (FETCH_SIZE, WRITE_SIZE; rocpd
databases) - the last `launches` launches of every kernel whose name contains one of the given substrings. Same corrections as
pmc_traffic.py (FETCH_SIZE x2 for wide coalesced reads on gfx950, WRITE_SIZE as is, KiB units, Infinity-Cache hits included).
usage: pmc_kernel_traffic.py <FETCH_SIZE dir> <WRITE_SIZE dir> <launches> <kernel substring> [...]"""
import json, sqlite3, sys, glob, os


def rows(path, counter, names):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    order = next((c for c in ("dispatch_id", "start", "id", "event_id") if c in cols), None)
    q = "select name, counter_value from pmc_events where counter_name = ?" + (f" order by {order}" if order else "")
    out = {}
    for n, v in cur.execute(q, (counter,)):
        for k in names:
            if k in n:
                out.setdefault(n.replace("void ", "").split("(")[0][:70], []).append(v)
    return out


fetch_dir, write_dir, last = sys.argv[1], sys.argv[2], int(sys.argv[3])
names = sys.argv[4:]
f, w = rows(fetch_dir, "FETCH_SIZE", names), rows(write_dir, "WRITE_SIZE", names)
res = {}
for k in sorted(f):
    fv, wv = f[k][-last:], w.get(k, [0])[-last:]
    res[k] = {"launches": len(fv), "fetch_bytes_per_launch_corrected": 2.0 * 1024 * sum(fv) / max(len(fv), 1), "write_bytes_per_launch": 1024.0 * sum(wv) / max(len(wv), 1)}
    res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch_corrected"] + res[k]["write_bytes_per_launch"]
print(json.dumps(res, indent=1))
