# round 6, iteration v: k_cc_hook_bodies skips an edge whose two bodies already hang under the same node
PROF_WL=mixed32k PROF_LINES=24 prof mixed32k
run polyheap polyheap32k
run pile32k pile32k
