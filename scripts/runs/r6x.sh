# round 6, iteration x: k_cc_hook_bodies remembers the manifolds without points in a bit mask instead of reading every point count twice
PROF_WL=mixed32k PROF_LINES=10 prof mixed32k
PROF_WL=polyheap32k PROF_LINES=12 prof polyheap32k
run pile32k pile32k
