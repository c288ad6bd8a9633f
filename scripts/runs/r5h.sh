# round 5, iteration h: edge-centric hooking on the compressed forest (full island relabel)
run mixed_bodies mixed32k
run mixed_edges mixed32k EDYNHIP_CC_EDGES=1
run mixed_bodies2 mixed32k
run mixed_edges2 mixed32k EDYNHIP_CC_EDGES=1
run pile_bodies pile32k
run pile_edges pile32k EDYNHIP_CC_EDGES=1
PROF_LINES=30
PROF_WL=mixed32k; prof mixed_edges EDYNHIP_CC_EDGES=1
