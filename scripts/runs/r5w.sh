# round 5, iteration w: round-stamped endpoint marks in k_col_rounds (no zeroing stores)
run poly polyheap32k
run islands islands256k
run pile pile32k
run mixed mixed32k
PROF_WL=polyheap32k PROF_LINES=6 prof poly
