"""Pins the ARITHMETIC FORK of the coloured order (CPU).

The device's specification is the checker's coloured order (oracle/oworld.hpp solve_coloured). Its contact arithmetic exists in three
forms (orc_set_arithmetic; the device: EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / EDYNHIP_FLAG_BLOCK_POSITION):

  ARITH_REFERENCE (0, the DEFAULT of checker and device)  every contact row and contact position correction with the reference's
        operations in the reference's order (constraint_row.cpp:24-57, constraint_row_friction.cpp:11-54, contact_constraint.cpp:58-90,
        position_solver.hpp:16-51): the coloured order differs from the reference in the Gauss-Seidel VISITING order only. SURVEY 8(d)(3)
        (device vs the coloured order with the reference's arithmetic: <= 1e-4 m, <= 1e-3 m/s after 60 steps) holds with ZERO error,
        because the device is bit-exact against this mode (tests/test_gpu_parity.py).
  | ARITH_FUSED_VELOCITY (1)   fma velocity rows: an fp-level change. Measured below: <= 2e-7 m/s per step in lock-step, and still
        inside SURVEY 8(d)(3) after 60 free-running steps.
  | ARITH_BLOCK_POSITION (2)   the points of a manifold corrected as one block: an ALGORITHMIC change. Measured below: up to 3e-4 m per
        step in lock-step; 8(d)(3) is exceeded after 60 free-running steps (2-5x on piles, ~500x on straight columns).

The figures asserted here are the ones bench.py quotes in `config.parity.arithmetic_fork`.
"""
import numpy as np
import pytest

from edyn_amd import scenes
from oracle import binding as ob

SCENES = {
    "pile_6x6x6_10it": (lambda: scenes.box_pile(6, 6, 6), 10),
    "mixed_6x6x6_20it": (lambda: scenes.box_pile(6, 6, 6, mixed=True), 20),
    "c1_columns_10it": (lambda: scenes.c1_columns(), 10),
}
FUSED, BLOCK = ob.ARITH_FUSED_VELOCITY, ob.ARITH_FUSED_VELOCITY | ob.ARITH_BLOCK_POSITION
# positions <= 1e-4 m, velocities <= 1e-3 (m/s, rad/s) after N = 60 steps against the coloured order with the
# reference's arithmetic
SURVEY_POS, SURVEY_VEL = 1e-4, 1e-3


@pytest.fixture(autouse=True)
def _reference_arithmetic_afterwards():
    yield
    ob.set_arithmetic(ob.ARITH_REFERENCE)   # the switch is process-wide


def _world(scene, vel):
    w = ob.World(vel_iters=vel, order=ob.ORDER_COLOURED)
    w.add_bodies(scene)
    return w


def _diff(a, b):
    (ap, aq, av, aw), (bp, bq, bv, bw) = a, b
    return float(np.abs(ap - bp).max()), float(np.abs(aq - bq).max()), float(max(np.abs(av - bv).max(), np.abs(aw - bw).max()))


def free_running(scene, vel, mode, steps=60):
    """Both arithmetics run `steps` steps from the same start, nothing resynchronised."""
    ob.set_arithmetic(ob.ARITH_REFERENCE)
    r = _world(scene, vel); r.step(steps)
    ob.set_arithmetic(mode)
    t = _world(scene, vel); t.step(steps)
    return _diff(t.get_state(), r.get_state())


def lock_step(scene, vel, mode, steps=60):
    """Every step starts from the reference-arithmetic world's state and manifolds; worst difference of ONE step."""
    ob.set_arithmetic(ob.ARITH_REFERENCE)
    r, t = _world(scene, vel), _world(scene, vel)
    worst = (0.0, 0.0, 0.0)
    for _ in range(steps):
        t.set_state(*r.get_state()); t.refresh_derived(); t.set_manifolds(r.get_manifolds())
        ob.set_arithmetic(ob.ARITH_REFERENCE); r.step(1)
        ob.set_arithmetic(mode); t.step(1)
        assert np.array_equal(t.get_pairs(), r.get_pairs())   # the visiting order and the arithmetic never touch the broadphase
        worst = tuple(max(a, b) for a, b in zip(worst, _diff(t.get_state(), r.get_state())))
    return worst


@pytest.mark.parametrize("name", list(SCENES))
def test_fused_velocity_rows_are_an_fp_level_change(name):
    gen, vel = SCENES[name]
    scene = gen()
    lp, lq, lv = lock_step(scene, vel, FUSED)
    fp, fq, fv = free_running(scene, vel, FUSED)
    print(f"\n[fork] {name} fused velocity rows: lock-step per step dpos {lp:.2e} dorn {lq:.2e} dvel {lv:.2e}; 60 steps free-running dpos {fp:.2e} dorn {fq:.2e} dvel {fv:.2e}")
    # one step: rounding only (measured <= 6e-8 m, <= 2.5e-7 m/s on these scenes)
    assert lp < 1e-6 and lq < 1e-6 and lv < 2e-6, (lp, lq, lv)
    # 60 steps: inside SURVEY 8(d)(3) (measured 5.2e-5 / 9.0e-5 / 6.2e-5 m and 3.6e-4 / 6.7e-4 / 1.3e-4 m/s)
    assert fp <= SURVEY_POS and fq <= SURVEY_POS and fv <= SURVEY_VEL, (fp, fq, fv)


@pytest.mark.parametrize("name", list(SCENES))
def test_block_position_correction_is_an_algorithmic_change(name):
    gen, vel = SCENES[name]
    scene = gen()
    lp, lq, lv = lock_step(scene, vel, BLOCK)
    fp, fq, fv = free_running(scene, vel, BLOCK)
    print(f"\n[fork] {name} fused rows + block position: lock-step per step dpos {lp:.2e} dorn {lq:.2e} dvel {lv:.2e}; 60 steps free-running dpos {fp:.2e} dorn {fq:.2e} dvel {fv:.2e}")
    # one step: the second-order coupling term between the points of a manifold (measured: 1.5e-5 m on the box pile, 2.8e-4 m on the
    # mixed pile at 20 iterations) - far beyond rounding, far below the visiting-order difference to the engine (1e-3 m per step)
    assert 1e-6 < lp < 1e-3, lp
    assert lv < 1e-2, lv
    # 60 steps: SURVEY 8(d)(3) is NOT met in this mode - that is why it is opt-in and why the headline is not quoted on it. Bounds = what
    # was measured with ~2x head-room (4.7e-4 / 2.1e-4 / 4.8e-2 m; 3.4e-3 / 4.9e-3 / 0.10 m/s); if this ever falls inside the
    # SURVEY bound the statement in DESIGN.md section 4 / edynhip.h has to change with it
    bound_p, bound_v = (0.1, 0.25) if name.startswith("c1") else (1e-3, 1e-2)
    assert fp < bound_p and fv < bound_v, (fp, fv)
    assert fp > SURVEY_POS or fv > SURVEY_VEL, ("the block correction now meets SURVEY 8(d)(3): update the documents", fp, fv)


def test_default_arithmetic_is_the_references():
    """The checker starts in ARITH_REFERENCE, and in that mode the coloured order runs the SAME row functions as the sequential /
    external orders (which tests/test_reference_engine.py pins to the real engine bit for bit): on a scene whose colour order equals
    its sequential order - one box on a plane: one manifold - the two orders agree to the last bit for 120 steps."""
    assert ob.get_arithmetic() == ob.ARITH_REFERENCE
    scene = scenes.box_pile(1, 1, 1)
    a = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); a.add_bodies(scene)
    b = ob.World(vel_iters=10, order=ob.ORDER_SEQUENTIAL); b.add_bodies(scene)
    for _ in range(120):
        a.step(1); b.step(1)
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y)
    ob.set_arithmetic(BLOCK)
    c = ob.World(vel_iters=10, order=ob.ORDER_COLOURED); c.add_bodies(scene)
    c.step(120)
    assert not np.array_equal(c.get_state()[0], a.get_state()[0])   # the switch does switch
