# round 5, iteration n: phase profile of k_np_detect_pp
run prof_g16 polyheap32k EDYNHIP_POLY_GROUP=16 EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g16.err | tail -2
run prof_g8 polyheap32k EDYNHIP_POLY_GROUP=8 EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g8.err | tail -2
