# round 5, iteration t: separating-axis hints (a pair its last deciding axis still separates never reaches the axis kernel)
run poly_hint polyheap32k
run poly_nohint polyheap32k EDYNHIP_POLY_HINT=0
run poly_g0 polyheap32k EDYNHIP_POLY_GROUP=0
run prof_hint polyheap32k EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_hint.err | tail -1
PROF_WL=polyheap32k PROF_LINES=14 prof poly
