#!/bin/bash
# One gpurun call: GPU tests, the bench line of every BASELINE workload, then the profiles. Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh r02 [skip-tests]
R=${1:-r02}
mkdir -p gpurun_out
if [ "${2:-}" != skip-tests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${R}_pytest_gpu.log
  tail -5 gpurun_out/${R}_pytest_gpu.log
fi
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo "bench rc=$?"; cat gpurun_out/${R}_bench_default.json | cut -c1-1500
for WL in pile8k mixed32k islands256k chains16k ragdolls1k polyheap32k; do
  case $WL in islands256k) A="--steps 60 --warmup 10";; polyheap32k) A="--steps 100 --warmup 10";; *) A="";; esac
  timeout 600 python bench.py --workload $WL $A --north-star none --other-arithmetic-steps 0 --no-cpu-baseline > gpurun_out/${R}_bench_$WL.json 2> gpurun_out/${R}_bench_$WL.err; echo "$WL rc=$?"; cut -c1-400 gpurun_out/${R}_bench_$WL.json
done
timeout 2400 bash scripts/profile_round.sh $R pile32k pile8k mixed32k islands256k chains16k ragdolls1k polyheap32k > gpurun_out/${R}_profile.log 2>&1; echo "profile rc=$?"
