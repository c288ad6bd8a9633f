for WL in polyheap32k ragdolls1k mixed32k pile32k; do
  run ${WL}_l4 $WL EDYNHIP_PAIR_LANES=4
  run ${WL}_l16 $WL EDYNHIP_PAIR_LANES=16
  run ${WL}_auto $WL X=0
done
