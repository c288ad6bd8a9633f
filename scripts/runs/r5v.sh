# round 5, iteration v: k_col_rounds with owned edges and batched fetches
run poly polyheap32k
run poly_trace polyheap32k EDYNHIP_COL_TRACE=1
grep "col trace" gpurun_out/$TAG/poly_trace.err | cut -c1-1200
run pile pile32k
run mixed mixed32k
PROF_WL=polyheap32k PROF_LINES=16 prof poly
