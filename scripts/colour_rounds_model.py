"""A model of the colouring rounds on a heap-like contact graph (bodies in raster order, owner = the higher index, canonical pair order): the rule of
k_col_rounds (an edge wins when it is the best uncoloured edge at both bodies) against a variant in which a thread colours a whole run of one owner's
edges in a single round. DESIGN section 3, round 6 item 2c.   usage: colour_rounds_model.py [listed fraction]"""
import numpy as np, sys
rng = np.random.default_rng(1)
n = 24; N = n**3
def idx(i,j,k): return (k*n + j)*n + i
edges = set()
# heap-like contact graph: each body touches a random subset of its 26 neighbours (mean degree ~8), a fraction of the edges uncoloured
for k in range(n):
    for j in range(n):
        for i in range(n):
            a = idx(i,j,k)
            for d in [(1,0,0),(0,1,0),(0,0,1),(1,1,0),(1,0,1),(0,1,1),(1,-1,0),(1,0,-1),(0,1,-1)]:
                ii,jj,kk = i+d[0], j+d[1], k+d[2]
                if 0<=ii<n and 0<=jj<n and 0<=kk<n and rng.random() < 0.45:
                    b = idx(ii,jj,kk); edges.add((max(a,b), min(a,b)))
edges = sorted(edges)                       # canonical order: owner (higher index) ascending, other ascending
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
listed = [e for e in edges if rng.random() < frac]
print("bodies", N, "edges", len(edges), "listed", len(listed))
def simulate(chain, chunk):
    live = list(range(len(listed)))
    prio = {e: -e for e in live}            # lower list position = higher priority
    at = {}
    for e in live:
        for b in listed[e]: at.setdefault(b, []).append(e)
    done = set(); rounds = 0; visits = 0
    while len(done) < len(listed):
        rounds += 1
        best = {}
        for e in range(len(listed)):
            if e in done: continue
            visits += 1
            for b in listed[e]:
                if b not in best or e < best[b]: best[b] = e
        won = set()
        for e in range(len(listed)):
            if e in done: continue
            o, y = listed[e]
            ok_y = best[y] == e
            ok_o = best[o] == e
            if chain and not ok_o and e > 0 and (e - 1) in won and listed[e-1][0] == o and (e // chunk) == ((e - 1) // chunk): ok_o = True
            if ok_o and ok_y: won.add(e)
        done |= won
    return rounds, visits / len(listed)
print("edge rule:           rounds %d, visits per edge %.1f" % simulate(False, 1))
c = max(1, -(-len(listed)//1024)) | 1
print("owner-run chains (chunk %d): rounds %d, visits per edge %.1f" % ((c,) + simulate(True, c)))
print("owner-run chains (no chunk limit): rounds %d, visits per edge %.1f" % simulate(True, 10**9))
