for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
  run pile32k_w1024_$REP pile32k EDYNHIP_DFP_WAVES=1024
done
for WL in pile8k mixed32k islands256k polyheap32k ragdolls1k; do
  run ${WL}_base $WL EDYNHIP_LIB=$BASE
  run ${WL}_new $WL X=0
done
PROF_LINES=12
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_poly/kt -o r -- python $OLDPWD/bench.py --workload polyheap32k --steps 100 --warmup 10 --north-star none --no-cpu-baseline > /dev/null 2> /tmp/prof_poly.log )
python scripts/prof_summary.py /tmp/prof_poly/kt 230 k_contact_solve 100 > gpurun_out/$TAG/kernel_stats_polyheap.txt 2>&1; head -14 gpurun_out/$TAG/kernel_stats_polyheap.txt | cut -c1-118
