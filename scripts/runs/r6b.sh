# round 6, iteration b: sleep-state atomics aggregated per wave (islands stage of a sleeping-enabled world), contact events counted; graphs for the chain model
( cd tests/cpp && timeout 600 ./bench_update 32 120 300 > ../../gpurun_out/$TAG/update_mini.txt 2>&1; timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 )
cat gpurun_out/$TAG/update_mini.txt gpurun_out/$TAG/update_entt.txt
timeout 900 python scripts/shim_cfg_cost.py 32 300 > gpurun_out/$TAG/shim_cfg_cost.txt 2>&1; cat gpurun_out/$TAG/shim_cfg_cost.txt
timeout 300 python scripts/dump_graph.py pile32k 420 gpurun_out/$TAG/graph_pile32k.npz
timeout 300 python scripts/dump_graph.py mixed32k 420 gpurun_out/$TAG/graph_mixed32k.npz
