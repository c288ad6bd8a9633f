# round 6, iteration w: dropped certificate manifolds repaired locally (k_bp_compact's extra block) instead of a full island relabel
for wl in pile32k mixed32k pile8k islands256k polyheap32k; do
  case $wl in islands256k) A="--steps 60 --warmup 10";; polyheap32k) A="--steps 100 --warmup 10";; *) A="";; esac
  for rep in 1 0; do
    EDYNHIP_TREE_REPAIR=$rep EDYNHIP_TREE_STATS=1 timeout 600 python bench.py --workload $wl $A --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > gpurun_out/$TAG/${wl}_rep$rep.json 2> gpurun_out/$TAG/${wl}_rep$rep.err
    line gpurun_out/$TAG/${wl}_rep$rep.json "$wl repair=$rep"; grep "island labels" gpurun_out/$TAG/${wl}_rep$rep.err | tail -1
  done
done
PROF_WL=mixed32k PROF_LINES=24 prof mixed32k
