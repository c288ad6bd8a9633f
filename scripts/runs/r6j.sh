# round 6, iteration j: the candidate lists' look-ahead adapts to how often the lists are rebuilt (A/B by knob, one box)
for WL in polyheap32k pile32k mixed32k islands256k ragdolls1k pile8k; do
  run fixed_$WL $WL EDYNHIP_BP_ADAPT=0
  run adapt_$WL $WL
done
EDYNHIP_BP_STATS=1 timeout 300 python bench.py --workload polyheap32k --steps 100 --warmup 10 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > /dev/null 2> gpurun_out/$TAG/bp_stats_polyheap32k.txt; grep "bp stats" gpurun_out/$TAG/bp_stats_polyheap32k.txt | cut -c1-400
EDYNHIP_BP_STATS=1 timeout 300 python bench.py --workload mixed32k --steps 100 --warmup 10 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > /dev/null 2> gpurun_out/$TAG/bp_stats_mixed32k.txt; grep "bp stats" gpurun_out/$TAG/bp_stats_mixed32k.txt | cut -c1-400
PROF_WL=polyheap32k PROF_LINES=16 prof polyheap_adapt
