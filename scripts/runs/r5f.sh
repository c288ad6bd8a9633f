# round 5, iteration f: which of the two direct paths moves the velocity solve? (developer knobs, same build, one box)
for REP in 1 2 3; do
  run base_$REP pile32k EDYNHIP_LIB=$BASE
  run both_$REP pile32k
  run sort_only_$REP pile32k EDYNHIP_DIRECT_COMPACT=0
  run compact_only_$REP pile32k EDYNHIP_DIRECT_SORT=0
  run neither_$REP pile32k EDYNHIP_DIRECT_SORT=0 EDYNHIP_DIRECT_COMPACT=0
done
