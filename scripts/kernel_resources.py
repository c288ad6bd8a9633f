"""Kernel resource table of the current sources: every .hip of edyn_amd/csrc compiled for gfx950 with
-Rpass-analysis=kernel-resource-usage (device code only; no GPU needed), reduced to one line per kernel.
usage: python scripts/kernel_resources.py rNN > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edyn_amd", "csrc")
tag = sys.argv[1] if len(sys.argv) > 1 else "rNN"
rows = []
for f in sorted(os.listdir(CSRC)):
    if not f.endswith(".hip"):
        continue
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-c", "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, f)]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*$", "", name), "file": f}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("VGPR", r"remark:\s+VGPRs: (\d+)"), ("AGPR", r"remark:\s+AGPRs: (\d+)"), ("SGPR", r"remark:\s+SGPRs: (\d+)"),
                         ("scratch", r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)"), ("LDS", r"remark:\s+LDS Size \[bytes/block\]: (\d+)"),
                         ("occ", r"remark:\s+Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
print(f"# Kernel resource usage of the {tag} build (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage)")
print("# VGPRs/AGPRs per lane, scratch bytes per lane, LDS bytes per workgroup, occupancy in waves per SIMD\n")
print(f"{'kernel':<86} {'file':<16} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}")
for r in sorted(rows, key=lambda r: (r["file"], r["name"])):
    if "VGPR" not in r:
        continue
    print(f"{r['name'][:86]:<86} {r['file']:<16} {r.get('VGPR', 0):>5} {r.get('AGPR', 0):>5} {r.get('SGPR', 0):>5} {r.get('scratch', 0):>8} {r.get('LDS', 0):>7} {r.get('occ', 0):>4}")
