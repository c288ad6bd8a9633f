# round 6, iteration c: contact entities built from prefetched events while the solve runs (sequential modes)
( cd tests/cpp && timeout 600 ./bench_update 32 120 300 > ../../gpurun_out/$TAG/update_mini.txt 2>&1; timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 )
cat gpurun_out/$TAG/update_mini.txt gpurun_out/$TAG/update_entt.txt
