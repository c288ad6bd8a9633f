# round 6, iteration d: full GPU suite on the shim rework + the default bench line with the shim leg
timeout 900 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -c 3000 gpurun_out/$TAG/bench_default.json
( cd tests/cpp && timeout 600 ./bench_update_entt 32 120 300 > ../../gpurun_out/$TAG/update_entt.txt 2>&1 ); cat gpurun_out/$TAG/update_entt.txt | cut -c1-700
