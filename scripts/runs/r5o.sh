# round 5, iteration o: polyhedron pairs in two kernels (axes by lane groups at high occupancy, contacts of the survivors)
EDYNHIP_POLY_GROUP=8 EDYNHIP_POLY_GROUP2=4 timeout 600 python -m pytest tests -m gpu -x -q -k "polyhedron_collide_routines or polyhedron_heap_at_size or polyhedra" > gpurun_out/$TAG/pytest_g8.log 2>&1; echo "pytest G=8/4 rc=$?"; tail -3 gpurun_out/$TAG/pytest_g8.log
EDYNHIP_POLY_GROUP=16 EDYNHIP_POLY_GROUP2=16 timeout 600 python -m pytest tests -m gpu -x -q -k "polyhedron_collide_routines or polyhedron_heap_at_size or polyhedra" > gpurun_out/$TAG/pytest_g16.log 2>&1; echo "pytest G=16/16 rc=$?"; tail -3 gpurun_out/$TAG/pytest_g16.log
run poly_g0 polyheap32k EDYNHIP_POLY_GROUP=0
run poly_g16_8 polyheap32k
run poly_g8_8 polyheap32k EDYNHIP_POLY_GROUP=8
run poly_g16_4 polyheap32k EDYNHIP_POLY_GROUP2=4
run poly_g16_16 polyheap32k EDYNHIP_POLY_GROUP2=16
run prof_g16_8 polyheap32k EDYNHIP_PP_PROF=1
grep "pp prof" gpurun_out/$TAG/prof_g16_8.err | tail -1
run prof_g8_4 polyheap32k EDYNHIP_PP_PROF=1 EDYNHIP_POLY_GROUP=8 EDYNHIP_POLY_GROUP2=4
grep "pp prof" gpurun_out/$TAG/prof_g8_4.err | tail -1
PROF_WL=polyheap32k PROF_LINES=12 prof poly
