# round 5, iteration l: XCD-local task lists in the two-lane velocity kernel and the position kernel (EDYNHIP_DF_XCD=0: tasks by p alone)
for REP in 1 2 3; do
  run plain_$REP pile32k EDYNHIP_DF_XCD=0
  run xcd_$REP pile32k
done
run mixed_plain mixed32k EDYNHIP_DF_XCD=0
run mixed_xcd mixed32k
run pile8k_plain pile8k EDYNHIP_DF_XCD=0
run pile8k_xcd pile8k
run mixed_plain2 mixed32k EDYNHIP_DF_XCD=0
run mixed_xcd2 mixed32k
