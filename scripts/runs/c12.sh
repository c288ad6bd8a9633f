for REP in 1 2 3; do
  run pile32k_fused_$REP pile32k X=0
  run pile32k_unfused_$REP pile32k EDYNHIP_DFP_FUSED=0
done
for WL in pile8k mixed32k islands256k polyheap32k; do
  run ${WL}_fused $WL X=0
  run ${WL}_unfused $WL EDYNHIP_DFP_FUSED=0
done
run pile32k_w1024 pile32k EDYNHIP_DFP_WAVES=1024
run pile32k_w768 pile32k EDYNHIP_DFP_WAVES=768
trace fused X=0
