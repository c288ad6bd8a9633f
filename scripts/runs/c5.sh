for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
  run pile32k_nopf_$REP pile32k EDYNHIP_DFP_PREFETCH=0
  run pile32k_pf1024_$REP pile32k EDYNHIP_DFP_WAVES=1024
done
for WL in pile8k mixed32k islands256k polyheap32k; do
  run ${WL}_base $WL EDYNHIP_LIB=$BASE
  run ${WL}_new $WL X=0
  run ${WL}_nopf $WL EDYNHIP_DFP_PREFETCH=0
done
trace new X=0
