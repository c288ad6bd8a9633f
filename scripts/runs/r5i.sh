# round 5, iteration i: functional check of the multi-device additions (pair filter, sleep timers, com offsets, joints) + the host-side cost of the world's step
python scripts/multi_overhead.py islands256k 8 40 > gpurun_out/$TAG/multi_overhead_islands256k.json 2> gpurun_out/$TAG/multi_overhead.err; cat gpurun_out/$TAG/multi_overhead_islands256k.json; tail -3 gpurun_out/$TAG/multi_overhead.err
python scripts/multi_overhead.py islands64k 8 60 > gpurun_out/$TAG/multi_overhead_islands64k.json 2>> gpurun_out/$TAG/multi_overhead.err; cat gpurun_out/$TAG/multi_overhead_islands64k.json
python scripts/multi_overhead.py islands256k 2 40 > gpurun_out/$TAG/multi_overhead_islands256k_2.json 2>> gpurun_out/$TAG/multi_overhead.err; cat gpurun_out/$TAG/multi_overhead_islands256k_2.json
