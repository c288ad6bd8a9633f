# round 5, iteration p: edge pairs of the separating-axis kernel by DPP row rotation (G = 16)
run poly_g16_8 polyheap32k
run poly_g8_8 polyheap32k EDYNHIP_POLY_GROUP=8
run prof_g16_8 polyheap32k EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g16_8.err | tail -1
PROF_WL=polyheap32k PROF_LINES=12 prof poly
