# round 6, iteration r: k_col_rounds - LDS loops four edges at a time, point count in the LDS record (no info load), the fence behind the next phase 1
run poly_lds polyheap32k
run isl_lds islands256k
PROF_WL=polyheap32k PROF_LINES=8 prof poly_lds
