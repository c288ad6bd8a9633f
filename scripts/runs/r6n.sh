# round 6, iteration n: k_col_rounds with its marks in a hashed LDS table (A/B against the global-memory rounds: EDYNHIP_COL_LDS=0)
run poly_lds polyheap32k
run poly_global polyheap32k EDYNHIP_COL_LDS=0
run isl_lds islands256k
run isl_global islands256k EDYNHIP_COL_LDS=0
run pile_lds pile32k
run pile_global pile32k EDYNHIP_COL_LDS=0
PROF_WL=polyheap32k PROF_LINES=14 prof poly_lds
PROF_WL=polyheap32k PROF_LINES=14 prof poly_global EDYNHIP_COL_LDS=0
