"""Generates tests/golden/ragdoll_{capsule,box,cylinder}.npz: the reference's own rag doll (edyn::make_ragdoll,
util/ragdoll.cpp:65-914 - 22 bodies, 36 cone / cvjoint / hinge constraints, 21 collision exclusions) built by the REAL
engine (oracle/_ref/libedynref.so) with its hip at the origin, exported body by body and constraint by constraint
(RefWorld.export_figure). edyn_amd.scenes.figures() replicates the template into scenes for the parity tests and the
bench; tests/test_reference_engine.py checks that the committed files are what the engine builds.

Run from the repo root (needs `make -C oracle ref`):  python tests/golden/make_ragdoll.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build(shape):
    r = ob.RefWorld()
    first_body, first_joint = r.make_ragdoll(shape, pos=(0, 0, 0), height=1.7, weight=72.0)
    return r.export_figure(first_body, first_joint)


def main():
    for shape in ("capsule", "box", "cylinder"):
        fig = build(shape)
        np.savez_compressed(os.path.join(HERE, f"ragdoll_{shape}.npz"), **fig)
        print(shape, "bodies", len(fig["kind"]), "constraints", len(fig["joint_type"]), "exclusions", len(fig["exclusions"]))


if __name__ == "__main__":
    main()
