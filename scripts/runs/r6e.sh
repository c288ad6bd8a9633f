# round 6, iteration e: sleep kernels skipped when no body can sleep; contact tables flat; the deep contact of free-running C2; FETCH_SIZE
# calibration for gathers; where a re-partition's 1.2 s go
timeout 300 python scripts/diag_c2_deep_contact.py > gpurun_out/$TAG/c2_deep_contact.txt 2>&1; head -40 gpurun_out/$TAG/c2_deep_contact.txt | cut -c1-420
( cd tests/cpp && timeout 600 ./bench_update 32 120 300 > ../../gpurun_out/$TAG/update_mini.txt 2>&1; timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 )
cat gpurun_out/$TAG/update_mini.txt gpurun_out/$TAG/update_entt.txt | cut -c1-640
timeout 900 python scripts/shim_cfg_cost.py 32 300 2>&1 | grep -v stages > gpurun_out/$TAG/shim_cfg_cost.txt; cat gpurun_out/$TAG/shim_cfg_cost.txt
( cd scripts/ubench && timeout 300 ./gather > ../../gpurun_out/$TAG/gather_times.txt 2>&1; cat ../../gpurun_out/$TAG/gather_times.txt )
for C in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && rm -rf /tmp/gc_$C && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/gc_$C -o r -- $OLDPWD/scripts/ubench/gather > /dev/null 2> /tmp/gc_$C.log ); done
python scripts/pmc_gather_calibration.py /tmp/gc_FETCH_SIZE /tmp/gc_WRITE_SIZE > gpurun_out/$TAG/gather_calibration.json 2>&1; cat gpurun_out/$TAG/gather_calibration.json
EDYNHIP_WORLD_TRACE=1 timeout 900 python scripts/multi_overhead.py islands256k 8 40 > gpurun_out/$TAG/multi_overhead.json 2> gpurun_out/$TAG/multi_trace.txt; cat gpurun_out/$TAG/multi_overhead.json; grep -c "sticky re-partition" gpurun_out/$TAG/multi_trace.txt; tail -60 gpurun_out/$TAG/multi_trace.txt
