"""The C++ drop-in shim (include/edyn/edyn.hpp: edyn::attach / make_rigidbody / make_constraint / update over an
EnTT-compatible registry) builds with plain g++ against libedynhip.so, and - on a GPU - runs the reference's
README main loop."""
import os
import subprocess
import pytest
from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def test_shim_compiles_and_links():
    subprocess.check_call(["make", "-s", "-C", CPP, "all"])
    assert os.path.exists(os.path.join(CPP, "hello_world")) and os.path.exists(os.path.join(CPP, "lifecycle"))


def test_reference_include_paths_resolve_to_the_shim():
    """Code written against the reference includes <edyn/comp/...>, <edyn/constraints/...>, <edyn/util/...> piecemeal: the shim ships
    forwarding headers for those paths (tests/cpp/includes.cpp builds and creates bodies / a constraint without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", CPP, "includes"])
    out = subprocess.run([os.path.join(CPP, "includes")], capture_output=True, text=True, timeout=60)
    assert "INCLUDES_OK 1" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("prog", ["host_logic", "host_logic_entt"])
def test_shim_host_logic(prog):
    """No GPU: the write-back's fork-join pool visits every index exactly once (1-8 threads, sleeping and pre-woken workers), the
    on_destroy hooks that replace the per-update scan for destroyed bodies / constraints, the host form of update_presentation /
    snap_presentation - tests/cpp/host_logic.cpp on both registry branches."""
    subprocess.check_call(["make", "-s", "-C", CPP, prog])
    out = subprocess.run([os.path.join(CPP, prog)], capture_output=True, text=True, timeout=300)
    assert "HOST_LOGIC_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["bench_update", "bench_update_entt"])
def test_shim_update_bench_small(prog):
    """tests/cpp/bench_update.cpp on a 10^3 pile: edyn::update in sequential / asynchronous mode, default / exclusive launches, with and
    without contact entities - every run steps exactly once per update, stays finite, rests at the right height and keeps its
    contact entities in step with the device (the program's own checks); the timing lines are printed for the log."""
    import json
    subprocess.check_call(["make", "-s", "-C", CPP, prog])
    out = subprocess.run([os.path.join(CPP, prog), "10", "90", "60"], capture_output=True, text=True, timeout=600)
    assert "BENCH_UPDATE_OK" in out.stdout, out.stdout + out.stderr
    runs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(runs) == 5
    seq = [r for r in runs if r["run"] == "sequential"][0]
    assert seq["contact_point_entities"] == seq["contact_points"] > 1000   # the registry mirrors the device's contact points


@pytest.mark.gpu
def test_shim_hello_world_runs():
    subprocess.check_call(["make", "-s", "-C", CPP, "hello_world"])
    out = subprocess.run([os.path.join(CPP, "hello_world")], capture_output=True, text=True, timeout=300)
    assert "HELLO_WORLD_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_lifecycle_runs():
    """registry.destroy on bodies and constraints, clear_rigidbody, runtime settings, constraint overloads with optional rows,
    exclude_collision, capacity growth that carries the contacts, update(registry) - tests/cpp/lifecycle.cpp."""
    subprocess.check_call(["make", "-s", "-C", CPP, "lifecycle"])
    out = subprocess.run([os.path.join(CPP, "lifecycle")], capture_output=True, text=True, timeout=300)
    assert "LIFECYCLE_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_contact_entities_and_asynchronous_mode():
    """contact_manifold / contact_point entities mirrored from the device's event list; execution_mode::asynchronous delivers
    the synchronous trajectory one update late - tests/cpp/contacts.cpp."""
    subprocess.check_call(["make", "-s", "-C", CPP, "contacts"])
    out = subprocess.run([os.path.join(CPP, "contacts")], capture_output=True, text=True, timeout=300)
    assert "contacts OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_polyhedra():
    """convex_mesh / polyhedron_shape / make_box_mesh through the shim - tests/cpp/polyhedra.cpp: meshes shared between bodies, created
    once per context and again when the context is re-created; polyhedral cubes rest like boxes. The final transforms agree with
    the same scene driven through the C ABI from Python (where the mesh is initialised by the library and the late bodies are
    appended instead of the context being re-created: same physics, not the same rounding)."""
    import numpy as np
    import edyn_amd
    from edyn_amd import scenes
    subprocess.check_call(["make", "-s", "-C", CPP, "polyhedra"])
    out = subprocess.run([os.path.join(CPP, "polyhedra")], capture_output=True, text=True, timeout=300)
    assert "POLYHEDRA_OK" in out.stdout, out.stdout + out.stderr
    got = np.array([[float(x) for x in (ln.split()[3:6] + ln.split()[7:11])] for ln in out.stdout.splitlines() if ln.startswith("body ")], np.float32)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import meshes
    cube, wedge = meshes.box_mesh((0.5, 0.5, 0.5)), meshes.wedge()
    P, B, S = scenes.SHAPE_POLYHEDRON, scenes.SHAPE_BOX, scenes.SHAPE_SPHERE
    first = [(scenes.SHAPE_PLANE, (0, 1, 0, 0), (0, 0, 0))]
    first += [(P, (0, 0, 0, 0), (0.0, 0.52 + 1.03 * i, 0.0)) for i in range(3)]
    first += [(B, (0.5, 0.5, 0.5, 0), (3.0, 0.52 + 1.03 * i, 0.0)) for i in range(3)]
    first += [(P, (1, 0, 0, 0), (-2.0, 0.6, 0.5)), (S, (0.3, 0, 0, 0), (-2.2, 1.4, 0.5)), (P, (1, 0, 0, 0), (0.1, 3.8, 0.05))]
    later = [(P, (0, 0, 0, 0), (6.0 + 1.2 * (i % 8), 0.6 + 1.1 * (i // 8), 2.0)) for i in range(40)]

    def scene(items, meshes_=None):
        n = len(items)
        sc = dict(kind=np.full(n, scenes.KIND_DYNAMIC, np.int32), pos=np.float32([it[2] for it in items]), orn=np.tile(np.float32([0, 0, 0, 1]), (n, 1)),
                  linvel=np.zeros((n, 3), np.float32), angvel=np.zeros((n, 3), np.float32), mass=np.full(n, 2, np.float32),
                  shape_type=np.int32([it[0] for it in items]), shape_param=np.float32([it[1] for it in items]),
                  friction=np.full(n, 0.5, np.float32), restitution=np.zeros(n, np.float32), group=np.full(n, 2**64 - 1, np.uint64),
                  mask=np.full(n, 2**64 - 1, np.uint64), sleeping_disabled=np.ones(n, np.uint8))
        sc["kind"][np.int32([it[0] for it in items]) == scenes.SHAPE_PLANE] = scenes.KIND_STATIC
        if meshes_:
            sc["meshes"] = meshes_
        return sc

    w = edyn_amd.World(edyn_amd.init_config(num_solver_velocity_iterations=10, max_bodies=64))
    w.set_scene(scene(first, [cube, wedge]))
    w.step_simulation(120)
    w.add_scene(scene(later))
    w.step_simulation(120)
    pos, orn, _, _ = w.get_state()
    want = np.concatenate([pos, orn], axis=1)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 0.02, float(np.abs(got - want).max())


def _parse_dump(text):
    bodies, joints, excl = [], [], []
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] in ("body", "joint"):
            rec, key = {}, None
            for w in tok[2:]:
                try:
                    rec[key].append(float(w))
                except (ValueError, KeyError):
                    key = w; rec[key] = []
            (bodies if tok[0] == "body" else joints).append(rec)
        elif tok[0] == "exclude":
            excl.append((int(tok[1]), int(tok[2])))
    return bodies, joints, excl


@pytest.mark.parametrize("shape", ["capsule", "box", "cylinder"])
def test_shim_make_ragdoll_builds_the_reference_figure(shape):
    """edyn::make_ragdoll of the shim (include/edyn/util/ragdoll.hpp, tables of parts and joints) against the real engine's own
    rag doll (tests/golden/ragdoll_*.npz: edyn::make_ragdoll run by the reference and exported): every body - mass, transform,
    shape, the explicit inertia of the shapeless shoulders and twist bodies - every constraint - type, bodies, pivots, axes,
    frames, all parameters - and the exclusion list. No GPU needed: the shim uploads lazily."""
    import numpy as np
    subprocess.check_call(["make", "-s", "-C", CPP, "ragdoll"])
    out = subprocess.run([os.path.join(CPP, "ragdoll"), shape], capture_output=True, text=True, timeout=60)
    assert "RAGDOLL_DUMP_OK" in out.stdout, out.stdout + out.stderr
    bodies, joints, excl = _parse_dump(out.stdout)
    ref = np.load(os.path.join(ROOT, "tests", "golden", f"ragdoll_{shape}.npz"))
    close = lambda a, b: np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=2e-6, atol=2e-7)
    assert len(bodies) == len(ref["kind"]) == 22 and len(joints) == len(ref["joint_type"]) == 36
    for i, b in enumerate(bodies):
        assert close(b["mass"], [ref["mass"][i]]) and close(b["pos"], ref["pos"][i]), (i, b["pos"], ref["pos"][i])
        assert close(np.abs(b["orn"]), np.abs(ref["orn"][i])), (i, b["orn"], ref["orn"][i])     # (q and -q are one rotation)
        assert int(b["shape"][0]) == int(ref["shape_type"][i]) and close(b["param"], ref["shape_param"][i][:3]), (i, b["param"], ref["shape_param"][i])
        assert close(b["friction"], [ref["friction"][i]]) and close(b["restitution"], [ref["restitution"][i]])
        if int(b["shape"][0]) == 0:
            assert close(b["inertia"], ref["inertia"][i]), (i, b["inertia"], ref["inertia"][i])
    for j, c in enumerate(joints):
        assert int(c["kind"][0]) == int(ref["joint_type"][j]) and [int(x) for x in c["bodies"]] == list(ref["joint_body"][j]), j
        assert close(c["pivotA"], ref["pivotA"][j]) and close(c["pivotB"], ref["pivotB"][j]), (j, c["pivotA"], ref["pivotA"][j], c["pivotB"], ref["pivotB"][j])
        kind = int(c["kind"][0])
        if kind == 1:
            assert close(c["axisA"], ref["axisA"][j]) and close(c["axisB"], ref["axisB"][j]) and close(c["p"], ref["params10"][j]), (j, c["p"], ref["params10"][j])
        elif kind == 4:
            assert close(c["frameA"], ref["frameA"][j]) and close(c["p"], ref["params16"][j][:5]), (j, c["frameA"], ref["frameA"][j], c["p"], ref["params16"][j][:5])
        else:
            assert close(c["frameA"], ref["frameA"][j]) and close(c["frameB"], ref["frameB"][j]) and close(c["p"], ref["params16"][j][:15]), (j, c["p"], ref["params16"][j])
    assert sorted(tuple(sorted(x)) for x in excl) == sorted(tuple(sorted(int(v) for v in x)) for x in ref["exclusions"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["capsule", "box", "cylinder"])
def test_shim_ragdoll_falls_on_the_floor(shape):
    """The shim's rag doll dropped on a plane: cone + cvjoint on one constraint entity, shapeless parts, exclusions - one second of
    simulation leaves a finite, connected figure lying on the floor."""
    subprocess.check_call(["make", "-s", "-C", CPP, "ragdoll"])
    out = subprocess.run([os.path.join(CPP, "ragdoll"), shape, "run"], capture_output=True, text=True, timeout=300)
    assert "RAGDOLL_RUN_OK" in out.stdout, out.stdout + out.stderr


# ---- the shim's EnTT branch -------------------------------------------------------------------------------
# `make entt` builds every program a second time with -I oracle/entt_min, so that <entt/entt.hpp> resolves and the
# `__has_include(<entt/entt.hpp>)` branch of include/edyn/edyn.hpp:22-26 is what compiles and runs: an EnTT-API registry with
# sparse-set pools, signals and scoped connections instead of the bundled 144-line mini registry. oracle/entt_min is the checker's
# from-scratch EnTT subset (EnTT 3.15 is not in this image); this is test-only use of oracle/.
def test_shim_compiles_against_an_entt_api_registry():
    subprocess.check_call(["make", "-s", "-C", CPP, "entt"])
    out = subprocess.run([os.path.join(CPP, "includes_entt")], capture_output=True, text=True, timeout=60)
    assert "INCLUDES_OK 1" in out.stdout, out.stdout + out.stderr
    # the EnTT branch really is what was compiled: the mini registry has no signals, listeners.cpp needs them
    syms = subprocess.run(["nm", "-C", os.path.join(CPP, "listeners_entt")], capture_output=True, text=True).stdout
    assert "sigh" in syms or "sink" in syms, "listeners_entt does not contain EnTT signal code"


@pytest.mark.gpu
def test_shim_entt_listeners_see_contact_points_and_destroy_hooks():
    """registry.on_construct / on_destroy<contact_point | contact_manifold> listeners, registry.destroy(body) hooks, a scoped
    connection, detach destroying the engine's entities - tests/cpp/listeners.cpp, EnTT branch only."""
    subprocess.check_call(["make", "-s", "-C", CPP, "listeners_entt"])
    out = subprocess.run([os.path.join(CPP, "listeners_entt")], capture_output=True, text=True, timeout=300)
    assert "LISTENERS_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog,ok", [("hello_world", "HELLO_WORLD_OK"), ("lifecycle", "LIFECYCLE_OK"), ("contacts", "contacts OK"),
                                     ("polyhedra", "POLYHEDRA_OK")])
def test_shim_programs_run_over_the_entt_api_registry(prog, ok):
    """The same programs as above, through the shim's EnTT branch (EnTT's reverse iteration, swap-and-pop pools, entity recycling
    with versions - what the mini registry only approximates)."""
    subprocess.check_call(["make", "-s", "-C", CPP, prog + "_entt"])
    out = subprocess.run([os.path.join(CPP, prog + "_entt")], capture_output=True, text=True, timeout=300)
    assert ok in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_ragdoll_over_the_entt_api_registry():
    subprocess.check_call(["make", "-s", "-C", CPP, "ragdoll_entt"])
    out = subprocess.run([os.path.join(CPP, "ragdoll_entt"), "capsule", "run"], capture_output=True, text=True, timeout=300)
    assert "RAGDOLL_RUN_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_multi_gpu_world_through_the_c_abi():
    """tests/cpp/multi.cpp: edynhip_world_* with two shards (on the one GPU of the test box) against one context, bit-equal through
    an approach-triggered and a forced re-partition; manifolds identical at the end."""
    subprocess.check_call(["make", "-s", "-C", CPP, "multi"])
    out = subprocess.run([os.path.join(CPP, "multi")], capture_output=True, text=True, timeout=300)
    assert "MULTI_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_attach_with_several_devices():
    """edyn::attach with init_config::devices = {0, 0}: the registry program of the reference over the multi-GPU world - bit-equal to the
    single-device stepper until the running world is edited, rebuilt from the registry afterwards (tests/cpp/multi_shim.cpp)."""
    subprocess.check_call(["make", "-s", "-C", CPP, "multi_shim"])
    out = subprocess.run([os.path.join(CPP, "multi_shim")], capture_output=True, text=True, timeout=300)
    assert "MULTI_SHIM_OK" in out.stdout, out.stdout + out.stderr
