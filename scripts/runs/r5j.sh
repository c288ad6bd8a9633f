# round 5, iteration j: contact points as 80-byte records (the row preparation gathers whole records) - A/B on one box
for REP in 1 2; do
  run base_pile32k_$REP pile32k EDYNHIP_LIB=$BASE
  run new_pile32k_$REP pile32k
done
run base_mixed32k mixed32k EDYNHIP_LIB=$BASE
run new_mixed32k mixed32k
run base_islands256k islands256k EDYNHIP_LIB=$BASE
run new_islands256k islands256k
run base_ragdolls ragdolls1k EDYNHIP_LIB=$BASE
run new_ragdolls ragdolls1k
prof new
