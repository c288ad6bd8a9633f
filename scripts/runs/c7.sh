for REP in 1 2 3; do
  run pile32k_new_$REP pile32k X=0
  run pile32k_nopf_$REP pile32k EDYNHIP_DFP_PREFETCH=0
done
run pile32k_w1024 pile32k EDYNHIP_DFP_WAVES=1024
run pile32k_w768 pile32k EDYNHIP_DFP_WAVES=768
for WL in pile8k mixed32k islands256k; do
  run ${WL}_new $WL X=0
  run ${WL}_nopf $WL EDYNHIP_DFP_PREFETCH=0
done
trace new X=0
