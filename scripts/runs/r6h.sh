# round 6, iteration h: contact points as one 320-byte record per manifold (A/B against the slot-major layout on one box)
for REP in 1 2; do
  run base_pile32k_$REP pile32k EDYNHIP_LIB=$BASE
  run recs_pile32k_$REP pile32k
done
run base_mixed32k mixed32k EDYNHIP_LIB=$BASE
run recs_mixed32k mixed32k
run base_islands256k islands256k EDYNHIP_LIB=$BASE
run recs_islands256k islands256k
run base_polyheap32k polyheap32k EDYNHIP_LIB=$BASE
run recs_polyheap32k polyheap32k
run base_ragdolls1k ragdolls1k EDYNHIP_LIB=$BASE
run recs_ragdolls1k ragdolls1k
PROF_LINES=14 prof recs
PROF_LINES=14 prof base EDYNHIP_LIB=$BASE
