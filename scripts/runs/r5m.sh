# round 5, iteration m: polyhedron-polyhedron pairs by lane groups (k_np_detect_pp; EDYNHIP_POLY_GROUP=0: one lane per pair, the round-4 form)
EDYNHIP_POLY_GROUP=8 timeout 600 python -m pytest tests -m gpu -x -q -k "polyhedron_collide_routines or polyhedron_heap_at_size" > gpurun_out/$TAG/pytest_g8.log 2>&1; echo "pytest G=8 rc=$?"; tail -3 gpurun_out/$TAG/pytest_g8.log
run poly_g0 polyheap32k EDYNHIP_POLY_GROUP=0
run poly_g16 polyheap32k EDYNHIP_POLY_GROUP=16
run poly_g8 polyheap32k EDYNHIP_POLY_GROUP=8
PROF_WL=polyheap32k PROF_LINES=8 prof poly_g16
PROF_WL=polyheap32k PROF_LINES=8 prof poly_g8 EDYNHIP_POLY_GROUP=8
