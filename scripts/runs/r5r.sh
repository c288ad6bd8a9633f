# round 5, iteration r: separating axes by one lane per pair at five waves per SIMD + contacts of the survivors by groups
run poly_g1_8 polyheap32k
run poly_g1_4 polyheap32k EDYNHIP_POLY_GROUP2=4
run poly_g0 polyheap32k EDYNHIP_POLY_GROUP=0
run prof_g1_8 polyheap32k EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g1_8.err | tail -1
PROF_WL=polyheap32k PROF_LINES=12 prof poly
