"""What rocprofv3's FETCH_SIZE / WRITE_SIZE charge per 16-byte access, by access pattern (scripts/ubench/gather.hip).
usage: pmc_gather_calibration.py <FETCH_SIZE dir> <WRITE_SIZE dir>   (two rocprofv3 --pmc passes of scripts/ubench/gather, rocpd output)
Prints, per kernel, the RAW counter bytes per launch (counter value x 1024: the counters are in KiB) divided by the lanes of a launch."""
import glob, json, os, sqlite3, sys

LANES = 8 << 20


def per_kernel(path, counter):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
    cur = sqlite3.connect(path).cursor()
    out = {}
    for name, v in cur.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        out.setdefault(name.replace("void ", "").split("(")[0], []).append(v)
    return out


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
expect = {"k_gather_stream": (16, 16), "k_gather_stride4": (16, 64), "k_gather_stride8": (16, 128), "k_gather_permuted": (20, 68),
          "k_scatter_store16": (4, 4), "k_stream_store16": (0, 0)}
res = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fb = 1024.0 * sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1) / LANES
    wb = 1024.0 * sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1) / LANES
    res[k] = {"launches": len(f.get(k, [])), "FETCH_SIZE_bytes_per_lane_raw": round(fb, 2), "WRITE_SIZE_bytes_per_lane_raw": round(wb, 2),
              "read_bytes_per_lane_useful": expect.get(k, (None, None))[0], "read_bytes_per_lane_in_whole_sectors_or_lines": expect.get(k, (None, None))[1]}
print(json.dumps(res, indent=1))
