# round 6, iteration p: k_col_rounds (LDS marks) takes a thread's winners four at a time, loads batched
run poly_lds polyheap32k
run isl_lds islands256k
run pile_lds pile32k
PROF_WL=polyheap32k PROF_LINES=12 prof poly_lds
PROF_WL=islands256k PROF_LINES=22 prof isl_lds
