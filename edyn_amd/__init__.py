"""edyn_amd - MI355X-native stepper for Edyn's per-step simulation loop.

The product is edyn_amd/libedynhip.so (hand-written HIP kernels for gfx950 behind the C-ABI of
include/edynhip.h). This package is the thin Python host mirror of the reference's stepping API used
by the tests and the bench; it never falls back to a CPU path.
"""
from .world import (World, init_config, rigidbody_def, attach, detach, make_rigidbody, update, step_simulation,
                    KIND_DYNAMIC, KIND_KINEMATIC, KIND_STATIC, SHAPE_NONE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_PLANE,
                    JOINT_POINT, JOINT_HINGE, ALL_GROUPS)
from ._capi import EdynHipError, MANIFOLD_DTYPE, POINT_DTYPE
from . import scenes
from .multi import MultiWorld

__all__ = ["World", "init_config", "rigidbody_def", "attach", "detach", "make_rigidbody", "update",
           "step_simulation", "EdynHipError", "scenes", "MultiWorld"]
