# round 5, iteration b: launch-floor fusions (direct counting sort, direct pair compaction, push_links in the row preparation) - A/B on one box
run base_pile32k pile32k EDYNHIP_LIB=$BASE
run new_pile32k pile32k
run base_pile32k_2 pile32k EDYNHIP_LIB=$BASE
run new_pile32k_2 pile32k
run base_mixed32k mixed32k EDYNHIP_LIB=$BASE
run new_mixed32k mixed32k
run base_pile8k pile8k EDYNHIP_LIB=$BASE
run new_pile8k pile8k
run new_ragdolls ragdolls1k
run base_ragdolls ragdolls1k EDYNHIP_LIB=$BASE
prof new
python scripts/prof_timeline.py /tmp/prof_new/kt 390 > gpurun_out/$TAG/timeline_new.txt 2>&1; tail -40 gpurun_out/$TAG/timeline_new.txt
