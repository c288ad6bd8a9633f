# round 5, iteration q: rotated vertices computed on the fly in the lane-group kernels (no reads of the bodies' rotated meshes)
run poly_g16_8 polyheap32k
run poly_g8_8 polyheap32k EDYNHIP_POLY_GROUP=8
run poly_g0 polyheap32k EDYNHIP_POLY_GROUP=0
run prof_g16_8 polyheap32k EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g16_8.err | tail -1
PROF_WL=polyheap32k PROF_LINES=12 prof poly
