#!/bin/bash
# quick GPU iteration: tests (-x), default bench, rocprofv3 kernel stats of the default bench (short), all into gpurun_out/
R=${1:-rq}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${R}_pytest_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/${R}_bench.json 2>gpurun_out/${R}_bench.err; python -c "
import json; j=json.load(open('gpurun_out/${R}_bench.json')); print('bench', round(j['value'],1), 'solve_ms', round(j['roofline']['solve_ms_per_step'],4), 'frac', round(j['roofline']['frac'],3))"
W=/tmp/prof_$R; rm -rf $W; mkdir -p $W
( cd /tmp && rocprofv3 --kernel-trace --stats -d $W/kt -o r -- python $OLDPWD/bench.py --steps 300 --warmup 20 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline > /dev/null 2> $W/kt.log )
python scripts/prof_summary.py $W/kt 440 k_contact_solve 300 > gpurun_out/${R}_kernel_stats_pile32k.txt; head -24 gpurun_out/${R}_kernel_stats_pile32k.txt | cut -c1-110
