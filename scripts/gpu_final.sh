#!/bin/bash
# Round-end gpurun call: the whole GPU test suite (with the figures its parity tests print), the bench line of every workload, the
# profile round of every workload (scripts/profile_round.sh) and the dataflow traces of the headline scene. Everything lands in
# gpurun_out/ with the round tag (copy the summaries into profiles/).   usage: scripts/gpu_final.sh r04
R=${1:-r04}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/${R}_pytest_gpu_full.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${R}_pytest_gpu_full.log
grep -a "passed\|failed\|error" gpurun_out/${R}_pytest_gpu_full.log | tail -3
grep -a "free-running C2\|C2 free-running\|C3 full size\|\[figures\]\|\[lock-step\]\|\[residual\]\|\[fork\]" gpurun_out/${R}_pytest_gpu_full.log | sed 's/^\.*//' > gpurun_out/${R}_parity_figures.txt; cat gpurun_out/${R}_parity_figures.txt
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/${R}_bench_default.json
for WL in pile8k mixed32k islands256k chains16k ragdolls1k polyheap32k; do
  # (the BASELINE configurations also in the two opt-in contact arithmetics: config.arithmetic.steps_per_sec)
  case $WL in islands256k) A="--steps 60 --warmup 10 --other-arithmetic-steps 60";; polyheap32k) A="--steps 100 --warmup 10 --other-arithmetic-steps 0";; pile8k|mixed32k) A="--other-arithmetic-steps 300";; *) A="--other-arithmetic-steps 0";; esac
  case $WL in pile8k|mixed32k) CPU="";; *) CPU="--no-cpu-baseline";; esac   # cpu_baseline of C2 and C3 once per round
  timeout 900 python bench.py --workload $WL $A --north-star none $CPU > gpurun_out/${R}_bench_$WL.json 2> gpurun_out/${R}_bench_$WL.err; echo "$WL rc=$?"; cut -c1-260 gpurun_out/${R}_bench_$WL.json
done
python scripts/multi_overhead.py islands256k 8 40 > gpurun_out/${R}_multi_overhead_islands256k.json 2> /dev/null; cat gpurun_out/${R}_multi_overhead_islands256k.json
timeout 2400 bash scripts/profile_round.sh $R pile32k pile8k mixed32k islands256k chains16k ragdolls1k polyheap32k > gpurun_out/${R}_profile.log 2>&1; echo "profile rc=$?"
EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DFP_TRACE=/tmp/dfp.bin EDYNHIP_DF_TRACE_STEP=200 timeout 200 python bench.py --steps 150 --warmup 100 --no-cpu-baseline --north-star none --other-arithmetic-steps 0 --no-shim > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin > gpurun_out/${R}_dftrace_velocity_pile32k.txt 2>&1
python scripts/df_trace.py /tmp/dfp.bin > gpurun_out/${R}_dftrace_position_pile32k.txt 2>&1
head -12 gpurun_out/${R}_kernel_stats_pile32k.txt | cut -c1-118; cat gpurun_out/${R}_timeline_pile32k.txt | tail -36
