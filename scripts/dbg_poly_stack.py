"""Debug aid: the tests/cpp/polyhedra.cpp scene on the device against the oracle, step by step."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import edyn_amd
from oracle import binding as ob
import meshes
lib = [meshes.box_mesh((0.5, 0.5, 0.5)), meshes.wedge()]
for m in lib: ob.create_mesh(m)
P, B, S = 6, 1, 2
first = [(3, (0, 1, 0, 0), (0, 0, 0))]
first += [(P, (0, 0, 0, 0), (0.0, 0.52 + 1.03 * i, 0.0)) for i in range(3)]
first += [(B, (0.5, 0.5, 0.5, 0), (3.0, 0.52 + 1.03 * i, 0.0)) for i in range(3)]
first += [(P, (1, 0, 0, 0), (-2.0, 0.6, 0.5)), (S, (0.3, 0, 0, 0), (-2.2, 1.4, 0.5)), (P, (1, 0, 0, 0), (0.1, 3.8, 0.05))]
n = len(first)
sc = dict(kind=np.full(n, 0, np.int32), pos=np.float32([it[2] for it in first]), orn=np.tile(np.float32([0, 0, 0, 1]), (n, 1)),
          linvel=np.zeros((n, 3), np.float32), angvel=np.zeros((n, 3), np.float32), mass=np.full(n, 2, np.float32),
          shape_type=np.int32([it[0] for it in first]), shape_param=np.float32([it[1] for it in first]),
          friction=np.full(n, 0.5, np.float32), restitution=np.zeros(n, np.float32), group=np.full(n, 2**64 - 1, np.uint64),
          mask=np.full(n, 2**64 - 1, np.uint64), meshes=lib)
sc["kind"][0] = 2

def run(tag, initialized=False, **kw):
    o = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); o.add_bodies(sc)
    if kw.get("sleeping"): o.set_sleeping(True)
    g = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, **kw))
    sc2 = dict(sc)
    if initialized:
        g.attach(64)
        for k, m in enumerate(lib):
            g.create_convex_mesh(ob.mesh_get(k, "vertices"), m["indices"], m["faces"], initialized=True)
        sc2.pop("meshes")
        g.cfg.max_bodies = 64
        import ctypes as C
        n_, keep, b = g._body_arrays(sc2)
        g._check(g._L.edynhip_set_bodies(g._h, n_, C.byref(b))); g.n = n_; g._upload_joints([]); g._uploaded = (n_, 0); g._dirty = False
    else:
        g.set_scene(sc2)
    for s_ in range(120):
        g.step_simulation(1); o.step(1)
        gs, os_ = np.concatenate(g.get_state(), 1), np.concatenate(o.get_state(), 1)
        if not np.array_equal(gs.view(np.uint32), os_.view(np.uint32)):
            print(tag, "differs at step", s_, "bodies", np.nonzero((gs != os_).any(axis=1))[0], "finite", np.isfinite(gs).all()); return
    print(tag, "120 steps equal")

def regrow():
    later = [(P, (0, 0, 0, 0), (6.0 + 1.2 * (i % 8), 0.6 + 1.1 * (i // 8), 2.0)) for i in range(40)]
    a = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10)); a.set_scene(sc)
    a.step_simulation(120)
    pos, orn, lv, av = a.get_state()
    items = first + later
    n2 = len(items)
    sc2 = dict(kind=np.full(n2, 0, np.int32), pos=np.float32([it[2] for it in items]), orn=np.tile(np.float32([0, 0, 0, 1]), (n2, 1)),
               linvel=np.zeros((n2, 3), np.float32), angvel=np.zeros((n2, 3), np.float32), mass=np.full(n2, 2, np.float32),
               shape_type=np.int32([it[0] for it in items]), shape_param=np.float32([it[1] for it in items]),
               friction=np.full(n2, 0.5, np.float32), restitution=np.zeros(n2, np.float32), group=np.full(n2, 2**64 - 1, np.uint64),
               mask=np.full(n2, 2**64 - 1, np.uint64), meshes=lib)
    sc2["kind"][0] = 2
    sc2["pos"][:n] = pos; sc2["orn"][:n] = orn; sc2["linvel"][:n] = lv; sc2["angvel"][:n] = av
    sc2["sleeping_disabled"] = np.ones(n2, np.uint8)
    for variant in ("poly", "boxes"):
        sc3 = dict(sc2)
        if variant == "boxes":
            sc3["shape_type"] = sc2["shape_type"].copy(); sc3["shape_param"] = sc2["shape_param"].copy()
            sel = (sc3["shape_type"] == 6) & (sc3["shape_param"][:, 0] == 0)
            sc3["shape_type"][sel] = 1; sc3["shape_param"][sel] = (0.5, 0.5, 0.5, 0)
        b = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, sleeping=True)); b.set_scene(sc3)
        b.set_manifolds(a.get_manifolds())
        b.set_asleep(np.zeros(n2, np.uint8))
        for stage in (1, 2, 4, 8):
            b.run_stages(stage)
            st = np.concatenate(b.get_state(), 1)
            m = b.get_manifolds()
            isl = b.get_derived()[2]
            fin = {f: bool(np.isfinite(m["pt"][f]).all()) for f in m["pt"].dtype.names if m["pt"][f].dtype.kind == "f"}
            d = b.get_derived()
            print("   pt finite", fin, "aabb finite", np.isfinite(d[0]).all(), "iw finite", np.isfinite(d[1]).all(), "bad bodies", np.nonzero(~np.isfinite(st).all(axis=1))[0].tolist())
            if stage == 2 and variant == "poly":
                k = 8
                print("   manifold 8 body", m["body"][k], "pivotA", m["pt"]["pivotA"][k][:2], "normal", m["pt"]["normal"][k][:2], "dist", m["pt"]["distance"][k][:2], "iw[1]", d[1][1], "iw[8]", d[1][8])
            print(variant, "after stage", stage, "state finite", np.isfinite(st).all(), "manifolds", len(m), "pts", m["num_points"].tolist(),
                  "imp finite", np.isfinite(m["pt"]["normal_impulse"]).all(), "islands", isl[:12].tolist(), "colours", m["colour"].tolist())
        print(variant, "stats", b.get_stats())
regrow()
