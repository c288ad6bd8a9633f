# round 6, iteration s: a long list of uncoloured edges gets its first rounds as multi-block launches, k_col_rounds (LDS marks) the rest
run poly_pre8 polyheap32k
run poly_pre0 polyheap32k EDYNHIP_COL_PRE_ROUNDS=0
run poly_pre4 polyheap32k EDYNHIP_COL_PRE_ROUNDS=4
run poly_pre12 polyheap32k EDYNHIP_COL_PRE_ROUNDS=12
run poly_pre16_min3k polyheap32k EDYNHIP_COL_PRE_ROUNDS=16 EDYNHIP_COL_PRE_MIN=3000
run poly_global polyheap32k EDYNHIP_COL_LDS=0
PROF_WL=polyheap32k PROF_LINES=12 prof poly_pre8
