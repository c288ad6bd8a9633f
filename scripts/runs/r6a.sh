# round 6, iteration a: the drop-in's own cost - edyn::update with the in-place record write-back, both registry branches; what the
# shim's default context configuration costs against the bench's
( cd tests/cpp && timeout 600 ./bench_update 32 120 300 > ../../gpurun_out/$TAG/update_mini.txt 2>&1; timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 )
cat gpurun_out/$TAG/update_mini.txt gpurun_out/$TAG/update_entt.txt
timeout 900 python scripts/shim_cfg_cost.py 32 300 > gpurun_out/$TAG/shim_cfg_cost.txt 2>&1; cat gpurun_out/$TAG/shim_cfg_cost.txt
