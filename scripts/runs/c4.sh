PROF_LINES=16
prof new X=0
prof pw0 EDYNHIP_POS_PW=0
for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
  run pile32k_pw0_$REP pile32k EDYNHIP_POS_PW=0
  run pile32k_pil1_$REP pile32k EDYNHIP_POS_PIL=1
  run pile32k_pw0pil1_$REP pile32k EDYNHIP_POS_PW=0 EDYNHIP_POS_PIL=1
done
for WL in pile8k mixed32k islands256k; do
  run ${WL}_base $WL EDYNHIP_LIB=$BASE
  run ${WL}_new $WL X=0
  run ${WL}_pw0 $WL EDYNHIP_POS_PW=0
done
trace new X=0
