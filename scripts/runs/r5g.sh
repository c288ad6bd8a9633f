# round 5, iteration g: a pointer-jumping pass between the initial forest and the unions of a full island relabel (C3: 371 of 440 steps relabel)
run mixed_c0 mixed32k EDYNHIP_CC_COMPRESS=0
run mixed_c1 mixed32k EDYNHIP_CC_COMPRESS=1
run mixed_c2 mixed32k EDYNHIP_CC_COMPRESS=2
run mixed_c0b mixed32k EDYNHIP_CC_COMPRESS=0
run mixed_c1b mixed32k EDYNHIP_CC_COMPRESS=1
run pile_c0 pile32k EDYNHIP_CC_COMPRESS=0
run pile_c1 pile32k EDYNHIP_CC_COMPRESS=1
PROF_LINES=30
PROF_WL=mixed32k; prof mixed_c1 EDYNHIP_CC_COMPRESS=1
