# round 5, iteration k: resident waves of the dataflow position kernel in the reference arithmetic (the block form was tuned at 512)
for W in 512 768 1024 384; do
  run dfp_waves_$W pile32k EDYNHIP_DFP_WAVES=$W
done
run dfp_waves_512b pile32k EDYNHIP_DFP_WAVES=512
run dfp_waves_1024b pile32k EDYNHIP_DFP_WAVES=1024
run mixed_512 mixed32k EDYNHIP_DFP_WAVES=512
run mixed_1024 mixed32k EDYNHIP_DFP_WAVES=1024
