"""Offline model of the dataflow solve's critical path on a dumped contact graph (scripts/dump_graph.py).
Task = (sweep, manifold); it starts when both bodies' previous tasks have published (+ hand-off latency) and, with wave
batching, when every manifold of its wave-task is ready and the wave has finished its previous task."""
import sys, numpy as np
f = np.load(sys.argv[1])
body, npnt, col, kind = f["body"], f["num_points"], f["colour"], f["kind"]
act = npnt > 0
body, npnt, col = body[act], npnt[act], col[act]
order = np.lexsort((np.arange(len(col)), -npnt.astype(int), col))
body, npnt, col = body[order], npnt[order], col[order]
na = len(col); nb = len(kind)
dyn = kind == 0 if (kind == 0).sum() > 1 else kind == kind.max()
print("active manifolds", na, "colours", col.max() + 1, "dynamic?", dyn.sum())
deg = np.bincount(body[:, 0], minlength=nb) + np.bincount(body[:, 1], minlength=nb)
deg_d = deg[dyn]
print("degree: mean %.1f max %d; bodies with deg>=13: %d, >=16: %d" % (deg_d.mean(), deg_d.max(), (deg_d >= 13).sum(), (deg_d >= 16).sum()))

def simulate(compute, handoff, W=1, G=None, sweeps=11, owner=None, verbose=False):
    """compute(np)->us, handoff us; W manifolds per wave-task (lockstep); G resident waves (None = unlimited);
    owner: optional array[na] of a body whose hand-off is free when consecutive tasks of that body share the owner."""
    bt = np.zeros(nb)            # time at which the body's latest deltas are visible to the next task
    last_owner = np.full(nb, -1)
    cstart = np.r_[0, np.flatnonzero(np.diff(col)) + 1, na]
    wave_of = np.arange(na) // W
    nw = wave_of.max() + 1
    wave_free = np.zeros(G) if G else None
    ends = []
    for s in range(sweeps):
        for ci in range(len(cstart) - 1):
            a, e = cstart[ci], cstart[ci + 1]
            A, B = body[a:e, 0], body[a:e, 1]
            ready = np.maximum(np.where(dyn[A], bt[A], 0), np.where(dyn[B], bt[B], 0))
            if W > 1:      # wave lockstep: all manifolds of a wave-task start together
                wv = wave_of[a:e]
                wr = np.zeros(nw); np.maximum.at(wr, wv, ready)
                if G:
                    uw = np.unique(wv)
                    slot = uw % G
                    wr[uw] = np.maximum(wr[uw], wave_free[slot])
                ready = wr[wv]
            c = compute(npnt[a:e])
            if W > 1:
                wc = np.zeros(nw); np.maximum.at(wc, wv, c); c = wc[wv]
            fin = ready + c
            if W > 1 and G:
                wave_free[slot] = np.maximum(wave_free[slot], (wr + wc)[uw])
            bt[A] = np.where(dyn[A], fin + handoff, bt[A]); bt[B] = np.where(dyn[B], fin + handoff, bt[B])
        ends.append(bt[dyn].max())
    per = np.diff(ends)
    return ends[-1], per[-3:].mean()

c_df2 = lambda n: 0.25 + 0.35 * n
for name, comp, h, W, G in [
    ("ideal per-manifold, c=1.65@4, h=1.1", c_df2, 1.1, 1, None),
    ("wave lockstep 32, unlimited waves", c_df2, 1.1, 32, None),
    ("wave lockstep 32, 1024 waves", c_df2, 1.1, 32, 1024),
    ("wave lockstep 16, 2048 waves", c_df2, 1.1, 16, 2048),
    ("ideal, compute halved", lambda n: 0.5 * c_df2(n), 1.1, 1, None),
    ("ideal, hand-off 0.3", c_df2, 0.3, 1, None),
    ("ideal, hand-off 0", c_df2, 0.0, 1, None),
]:
    total, period = simulate(comp, h, W, G)
    print(f"{name:45s} total {total:7.1f} us  sweep period {period:6.2f} us")

# ---- ownership model: the lane that owns body X keeps X's deltas in registers across consecutive manifolds it owns
def simulate_owner(compute, handoff, rule, sweeps=11, local=0.0):
    deg_a, deg_b = deg[body[:, 0]], deg[body[:, 1]]
    da, db = dyn[body[:, 0]], dyn[body[:, 1]]
    if rule == "degree":
        own_a = (da & ~db) | (da & db & ((deg_a > deg_b) | ((deg_a == deg_b) & (body[:, 0] > body[:, 1]))))
    elif rule == "none":
        own_a = np.zeros(na, bool); 
    if not isinstance(rule, tuple): owner = np.where(own_a, body[:, 0], body[:, 1]) if rule != "none" else np.full(na, -1)
    if isinstance(rule, tuple):   # ("hot", H): only bodies with at least H active manifolds own their chains
        H = rule[1]
        ha, hb = da & (deg_a >= H), db & (deg_b >= H)
        own_a = ha & (~hb | (deg_a > deg_b) | ((deg_a == deg_b) & (body[:, 0] > body[:, 1])))
        own_b = hb & ~own_a
        owner = np.where(own_a, body[:, 0].astype(np.int64), np.where(own_b, body[:, 1].astype(np.int64), np.int64(-1)))
    bt = np.zeros(nb); last_owned = np.zeros(nb, bool)   # was the body's previous task owned by the body itself?
    cstart = np.r_[0, np.flatnonzero(np.diff(col)) + 1, na]
    ends = []
    for s in range(sweeps):
        for ci in range(len(cstart) - 1):
            a, e = cstart[ci], cstart[ci + 1]
            A, B = body[a:e, 0], body[a:e, 1]
            oA, oB = owner[a:e] == A, owner[a:e] == B
            # arrival of each body's delta at this task: free if it stays in the owner's registers
            inA = bt[A] + np.where(oA & last_owned[A], local, handoff)
            inB = bt[B] + np.where(oB & last_owned[B], local, handoff)
            ready = np.maximum(np.where(dyn[A], inA, 0), np.where(dyn[B], inB, 0))
            fin = ready + compute(npnt[a:e])
            bt[A] = np.where(dyn[A], fin, bt[A]); bt[B] = np.where(dyn[B], fin, bt[B])
            last_owned[A] = oA; last_owned[B] = oB
        ends.append(bt[dyn].max())
    return ends[-1], np.diff(ends)[-3:].mean(), owner

# round 6: the judge's proposal - the hottest bodies' chains each inside one wave (intra-wave hand-off 0.1 us), with this round's measured
# figures: 1.55 us of arithmetic per four-point task, 1.4 us per fabric hand-off
c_r5 = lambda n: 0.23 + 0.33 * n
print("degree histogram of the dynamic bodies (active manifolds):", dict(zip(*np.unique(deg_d, return_counts=True))))
for H in (16, 15, 14, 13, 12, 11, 10, 8, 1):
    total, period, owner = simulate_owner(c_r5, 1.4, ("hot", H), local=0.1)
    nh = int((deg_d >= H).sum())
    print(f"hot chains, degree >= {H:2d} ({nh:5d} bodies, {int((owner >= 0).sum()):6d} manifolds in hot waves): total {total:7.1f} us  sweep period {period:6.2f} us")
total, period, _ = simulate_owner(c_r5, 1.4, "none")
print(f"{'r05 figures, no ownership':45s} total {total:7.1f} us  sweep period {period:6.2f} us")
for name, rule in [("no ownership", "none"), ("owner = higher degree", "degree")]:
    total, period, owner = simulate_owner(c_df2, 1.1, rule)
    print(f"{name:45s} total {total:7.1f} us  sweep period {period:6.2f} us")
    if rule != "none":
        cnt = np.bincount(owner[owner >= 0], minlength=nb)[dyn]
        print("   owned manifolds per body: mean %.2f max %d; bodies owning none: %d" % (cnt.mean(), cnt.max(), (cnt == 0).sum()))

if "--clusters" not in sys.argv: sys.exit(0)
# ---- cluster model: one wave owns a spatial cluster of manifolds and walks it colour by colour; hand-offs inside the cluster
# are free (LDS), hand-offs between clusters cost `handoff` (slot through the fabric)
pos = f["pos"]
def morton_rank(pos):
    p = pos - pos.min(0); p = (p / p.max() * 1023).astype(np.uint64)
    def ex(v):
        v = v & 0x3FF; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    code = (ex(p[:, 0]) << 2) | (ex(p[:, 1]) << 1) | ex(p[:, 2])
    order = np.argsort(code, kind="stable"); rank = np.empty(len(pos), np.int64); rank[order] = np.arange(len(pos)); return rank

def simulate_clusters(compute, handoff, bodies_per_cluster, lanes_per_task=32, local=0.1, sweeps=11, assign="A"):
    rank = morton_rank(pos)
    rank_dyn = np.where(dyn, rank, -1)
    # rank among dynamic bodies only
    rd = np.full(nb, -1); rd[dyn] = np.argsort(np.argsort(rank[dyn]))
    cl_body = rd // bodies_per_cluster
    A, B = body[:, 0], body[:, 1]
    if assign == "A": clm = np.where(dyn[A], cl_body[A], cl_body[B])
    else: clm = np.where(dyn[A] & dyn[B], np.minimum(cl_body[A], cl_body[B]), np.where(dyn[A], cl_body[A], cl_body[B]))
    ncl = clm.max() + 1
    sizes = np.bincount(clm, minlength=ncl)
    # per body: previous task's cluster -> decides local / remote hand-off
    bt = np.zeros(nb); bcl = np.full(nb, -1)
    wave_t = np.zeros(ncl)
    cstart = np.r_[0, np.flatnonzero(np.diff(col)) + 1, na]
    ends = []; cross = 0; tot = 0
    for s in range(sweeps):
        for ci in range(len(cstart) - 1):
            a, e = cstart[ci], cstart[ci + 1]
            Ai, Bi, cm = A[a:e], B[a:e], clm[a:e]
            inA = bt[Ai] + np.where(bcl[Ai] == cm, 0.0, handoff); inB = bt[Bi] + np.where(bcl[Bi] == cm, 0.0, handoff)
            if s == sweeps - 1: cross += (bcl[Ai][dyn[Ai]] != cm[dyn[Ai]]).sum() + (bcl[Bi][dyn[Bi]] != cm[dyn[Bi]]).sum(); tot += dyn[Ai].sum() + dyn[Bi].sum()
            ready = np.maximum(np.where(dyn[Ai], inA, 0), np.where(dyn[Bi], inB, 0))
            # the cluster's wave handles this colour's manifolds in passes of lanes_per_task, each pass waits for its inputs and for the wave
            cnt = np.bincount(cm, minlength=ncl)
            rdy = np.zeros(ncl); np.maximum.at(rdy, cm, ready)
            cmax = np.zeros(ncl); np.maximum.at(cmax, cm, compute(npnt[a:e]))
            passes = -(-cnt // lanes_per_task)
            start = np.maximum(rdy, wave_t)
            finw = start + passes * (cmax + local)
            act_cl = cnt > 0
            wave_t[act_cl] = finw[act_cl]
            fin = finw[cm]
            bt[Ai] = np.where(dyn[Ai], fin, bt[Ai]); bt[Bi] = np.where(dyn[Bi], fin, bt[Bi])
            bcl[Ai] = np.where(dyn[Ai], cm, bcl[Ai]); bcl[Bi] = np.where(dyn[Bi], cm, bcl[Bi])
        ends.append(bt[dyn].max())
    return ends[-1], np.diff(ends)[-3:].mean(), ncl, sizes, cross / max(tot, 1)

for bpc in (32, 64, 96, 128, 192, 256):
    for assign in ("A", "min"):
        total, period, ncl, sizes, cf = simulate_clusters(c_df2, 1.1, bpc, assign=assign)
        print(f"clusters of {bpc:4d} bodies ({assign:3s}): {ncl:5d} clusters, manifolds/cluster mean {sizes.mean():6.1f} max {sizes.max():5d}; cross hand-offs {100*cf:4.1f} %; total {total:7.1f} us, sweep period {period:6.2f} us")
