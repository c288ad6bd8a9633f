PROF_LINES=26
prof base EDYNHIP_LIB=$BASE
prof new X=0
PROF_LINES=14
prof pwprep EDYNHIP_PW_IN_PREP=1
prof prep1 EDYNHIP_PREP_PER_POINT=0
for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
  run pile32k_pwprep_$REP pile32k EDYNHIP_PW_IN_PREP=1
  run pile32k_prep1_$REP pile32k EDYNHIP_PREP_PER_POINT=0
done
