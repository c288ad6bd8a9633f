# round 6, iteration g: how should the row preparation read its contact points? (scripts/ubench/points.hip)
( cd scripts/ubench && timeout 300 ./points > ../../gpurun_out/$TAG/points_times.txt 2>&1; cat ../../gpurun_out/$TAG/points_times.txt )
( cd /tmp && rm -rf /tmp/pt_F && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pt_F -o r -- $OLDPWD/scripts/ubench/points > /dev/null 2> /tmp/pt_F.log )
python - <<'PY' > gpurun_out/$TAG/points_fetch.json
import glob, json, sqlite3
path = sorted(glob.glob("/tmp/pt_F/**/*.db", recursive=True))[0]
cur = sqlite3.connect(path).cursor()
out = {}
for name, v in cur.execute("select name, counter_value from pmc_events where counter_name = 'FETCH_SIZE'"):
    out.setdefault(name.replace("void ", "").split("(")[0], []).append(v)
print(json.dumps({k: {"launches": len(v), "fetch_bytes_per_manifold_x2": round(2 * 1024.0 * sum(v) / len(v) / (2 << 20), 1)} for k, v in out.items() if k.startswith("k_")}, indent=1))
PY
cat gpurun_out/$TAG/points_fetch.json
