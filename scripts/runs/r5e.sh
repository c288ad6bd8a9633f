# round 5, iteration e: two-level direct colour sort + direct pair compaction, k_push_links back as its own launch - A/B on one box
for REP in 1 2 3; do
  run base_pile32k_$REP pile32k EDYNHIP_LIB=$BASE
  run new_pile32k_$REP pile32k
done
run base_mixed32k mixed32k EDYNHIP_LIB=$BASE
run new_mixed32k mixed32k
run base_pile8k pile8k EDYNHIP_LIB=$BASE
run new_pile8k pile8k
prof new
python scripts/prof_timeline.py /tmp/prof_new/kt 390 > gpurun_out/$TAG/timeline_new.txt 2>&1; tail -28 gpurun_out/$TAG/timeline_new.txt
