# round 5, iteration u: trace of k_col_rounds on the polyhedron heap; the contact kernel compiled for three waves per SIMD
run col_trace polyheap32k EDYNHIP_COL_TRACE=1
grep "col trace" gpurun_out/$TAG/col_trace.err | cut -c1-3000
run poly_occ3 polyheap32k EDYNHIP_POLY_OCC=3
run poly_occ2 polyheap32k
run poly_g2_4 polyheap32k EDYNHIP_POLY_GROUP2=4
