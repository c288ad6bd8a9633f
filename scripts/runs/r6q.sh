# round 6, iteration q: k_col_rounds warms the L2 with what the winners read
run poly_lds polyheap32k
run isl_lds islands256k
PROF_WL=polyheap32k PROF_LINES=8 prof poly_lds
