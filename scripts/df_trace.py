"""Analyse a dataflow-solve trace (EDYNHIP_DF_TRACE): where does a sweep's time go?
Timestamps are wall_clock64 ticks (100 MHz): w0 task start, w1 first poll returned (rows + first slot read arrived),
w2 all inputs of the wave's current colour arrived, w3 task finished (hand-offs issued)."""
import sys, numpy as np
raw = open(sys.argv[1], "rb").read()
na, stride, sweeps, wl = (int(x) for x in np.frombuffer(raw[:16], np.uint32))
wl = wl or 64
keys = np.frombuffer(raw[16:16 + 4 * na], np.uint32)
tr = np.frombuffer(raw[16 + 4 * na:], np.uint64)
rounds = (na + stride - 1) // stride; nw = stride // wl
tr = tr.reshape(sweeps, rounds, nw, 4).astype(np.int64)
valid = tr[..., 3] > 0
t_begin = tr[..., 0][valid].min()
us = lambda x: x / 100.0
print(f"na {na} stride {stride} sweeps {sweeps} rounds {rounds} waves {nw}; kernel span {us(tr[..., 3].max() - t_begin):.1f} us")
col = keys >> 2
for s in range(sweeps):
    v = valid[s]
    st, en = tr[s][..., 0][v].min(), tr[s][..., 3][v].max()
    rows = (tr[s][..., 1] - tr[s][..., 0])[v]; wait = (tr[s][..., 2] - tr[s][..., 1])[v]; comp = (tr[s][..., 3] - tr[s][..., 2])[v]
    print(f"sweep {s:2d}: start {us(st - t_begin):8.1f} end {us(en - t_begin):8.1f} span {us(en - st):6.1f} | per task: first-poll {us(rows.mean()):5.2f} wait {us(wait.mean()):6.2f} compute+publish {us(comp.mean()):5.2f} (max {us(comp.max()):5.2f})")
# per colour in the last sweep: when do its tasks become ready / finish
s = sweeps - 1
print("last sweep, per colour: tasks, mean ready time, mean finish, (relative to sweep start of colour 0)")
base = tr[s][..., 0][valid[s]].min()
for c in range(int(col.max()) + 1):
    ps = np.nonzero(col == c)[0]
    if len(ps) == 0: continue
    wv = np.unique(ps // wl)            # global wave-task index -> (round, wave)
    r, w = (wv * wl) // stride, ((wv * wl) % stride) // wl
    ready = tr[s, r, w, 2]; fin = tr[s, r, w, 3]; start = tr[s, r, w, 0]
    print(f"  colour {c:2d}: {len(ps):6d} manifolds {len(wv):4d} wave-tasks  start {us(start.mean() - base):7.1f}  ready {us(ready.mean() - base):7.1f} (min {us(ready.min() - base):7.1f} max {us(ready.max() - base):7.1f})  done {us(fin.mean() - base):7.1f}")
