for REP in 1 2; do
  run pile32k_new_$REP pile32k X=0
  run pile32k_skip_$REP pile32k EDYNHIP_DFP_SKIPSTORE=1
  run pile32k_skipnopf_$REP pile32k EDYNHIP_DFP_SKIPSTORE=1 EDYNHIP_DFP_PREFETCH=0
done
trace skip EDYNHIP_DFP_SKIPSTORE=1
trace skipnopf EDYNHIP_DFP_SKIPSTORE=1 EDYNHIP_DFP_PREFETCH=0
