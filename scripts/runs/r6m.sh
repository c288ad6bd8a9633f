# round 6, iteration m: EDYNHIP_SNAPSHOT_DIRECT in the shim's synchronous write-back - the final bench_update figures
( cd tests/cpp && timeout 600 ./bench_update 32 120 300 > ../../gpurun_out/$TAG/update_mini.txt 2>&1; timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 )
python - <<'PY'
import json
for f in ("update_mini", "update_entt"):
    for l in open(f"gpurun_out/r6m/{f}.txt"):
        if l.startswith("{"):
            r = json.loads(l); print(f, r["run"], r["update_steps_per_sec"], r["raw_steps_per_sec"], r["ratio"], r["host_ms_per_update"])
PY
timeout 600 python bench.py > gpurun_out/$TAG/bench_default.json 2> /dev/null; python -c "
import json; j=json.loads([l for l in open('gpurun_out/r6m/bench_default.json') if l.startswith('{')][-1]); print(j['value'], j['shim']['steps_per_sec'], j['shim']['ratio_to_value'])"
