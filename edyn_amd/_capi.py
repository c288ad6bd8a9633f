"""ctypes binding of the C-ABI in include/edynhip.h (libedynhip.so, built in-tree by csrc/Makefile).

The library is the product: there is no Python or CPU fallback. Importing this module without the
built library raises ImportError; creating a context without a usable GPU raises EdynHipError.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDYNHIP_LIB") or os.path.join(_HERE, "libedynhip.so")   # EDYNHIP_LIB: developer A/B runs of two builds on one box


class EdynHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"edynhip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_bodies", C.c_uint32), ("max_manifolds", C.c_uint32),
                ("max_joints", C.c_uint32), ("fixed_dt", C.c_float), ("num_velocity_iterations", C.c_uint32),
                ("num_position_iterations", C.c_uint32), ("gravity", C.c_float * 3), ("flags", C.c_uint32)]


class Bodies(C.Structure):
    _fields_ = [("kind", C.c_void_p), ("pos", C.c_void_p), ("orn", C.c_void_p), ("linvel", C.c_void_p),
                ("angvel", C.c_void_p), ("mass", C.c_void_p), ("inertia", C.c_void_p), ("has_inertia", C.c_void_p),
                ("shape_type", C.c_void_p), ("shape_param", C.c_void_p), ("friction", C.c_void_p),
                ("restitution", C.c_void_p), ("group", C.c_void_p), ("mask", C.c_void_p), ("gravity", C.c_void_p),
                ("sleeping_disabled", C.c_void_p), ("center_of_mass", C.c_void_p)]


class Joints(C.Structure):
    _fields_ = [("type", C.c_void_p), ("body", C.c_void_p), ("pivot", C.c_void_p), ("axis", C.c_void_p), ("params", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("fixed_dt", C.c_float), ("num_velocity_iterations", C.c_uint32), ("num_position_iterations", C.c_uint32),
                ("gravity", C.c_float * 3), ("num_restitution_iterations", C.c_uint32),
                ("num_individual_restitution_iterations", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("broadphase_ms", "narrowphase_ms", "islands_ms", "colouring_ms", "prepare_ms",
                                         "solve_velocity_ms", "integrate_ms", "solve_position_ms", "finish_ms", "step_ms")] + \
               [("solve_velocity_launches", C.c_uint32), ("steps", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_bodies", "num_manifolds", "num_points", "num_active_manifolds",
                                          "num_islands", "num_colours", "num_joint_colours", "colour_rounds",
                                          "num_joints", "num_joint_rows", "solve_schedule")] + [("colour_size", C.c_uint32 * 64)]


class WorldStats(C.Structure):   # edynhip_world_stats
    _fields_ = [(n, C.c_uint32) for n in ("num_shards", "num_bodies", "steps", "approach_checks", "repartitions")] + [("bodies_per_shard", C.c_uint32 * 16), ("rebalances", C.c_uint32)]


SCHEDULE_NAMES = {0: "none", 1: "k_contact_solve_df2 (dataflow, one launch per step, two lanes per manifold)",
                  2: "k_contact_solve_df (dataflow, one launch per step, one lane per manifold)",
                  3: "k_island_velocity (island-fused: one wave per island)",
                  4: "k_contact_solve_df2 + k_island_velocity (mixed: dataflow launch for islands without joints, one wave per jointed island)",
                  5: "k_contact_solve<WARM,PUSH> / k_joint_solve (one launch per colour and sweep)",
                  6: "k_contact_solve_df4 (dataflow, one launch per step, four lanes per manifold)"}


POINT_DTYPE = np.dtype([
    ("pivotA", np.float32, 3), ("pivotB", np.float32, 3), ("normal", np.float32, 3),
    ("local_normal", np.float32, 3), ("distance", np.float32), ("friction", np.float32),
    ("restitution", np.float32), ("attachment", np.int32), ("lifetime", np.uint32),
    ("normal_impulse", np.float32), ("friction_impulse", np.float32, 2)])
MANIFOLD_DTYPE = np.dtype([
    ("body", np.uint32, 2), ("num_points", np.uint32), ("colour", np.uint32), ("pt", POINT_DTYPE, 4)])

FLAG_TIMING, FLAG_SLEEPING, FLAG_EXCLUSIVE_DEVICE, FLAG_TIMING_SOLVE, FLAG_CONTACT_EVENTS = 1, 4, 8, 16, 32
FLAG_FUSED_VELOCITY_ROWS, FLAG_BLOCK_POSITION = 64, 128   # opt-in contact arithmetic (edynhip.h); default: the reference's operations
PAIR_FILTER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)   # edynhip_pair_filter: int filter(void *user, uint32_t body, uint32_t other)
EVENT_DTYPE = np.dtype([("type", np.uint32), ("step", np.uint32), ("body", np.uint32, 2), ("point_id", np.uint64)])
EVENT_MANIFOLD_CREATED, EVENT_MANIFOLD_DESTROYED, EVENT_POINT_CREATED, EVENT_POINT_DESTROYED = 1, 2, 3, 4
# edynhip_body_record (96 B) and edynhip_record_view: the registry write-back read in place (edynhip_snapshot_records / _map)
RECORD_DTYPE = np.dtype([("pos", np.float32, 3), ("orn", np.float32, 4), ("linvel", np.float32, 3), ("angvel", np.float32, 3),
                         ("present_pos", np.float32, 3), ("present_orn", np.float32, 4), ("origin", np.float32, 3), ("flags", np.uint32)])
RECORD_DYNAMIC, RECORD_ASLEEP, RECORD_HAS_ORIGIN, RECORD_REMOVED = 1, 2, 4, 8


class RecordView(C.Structure):
    _fields_ = [("records", C.c_void_p), ("num_bodies", C.c_uint32), ("step_index", C.c_uint32), ("events", C.c_void_p),
                ("num_events", C.c_uint32), ("total_events", C.c_uint32)]
STAGE_BROADPHASE, STAGE_NARROWPHASE, STAGE_ISLANDS, STAGE_SOLVE, STAGE_ALL = 1, 2, 4, 8, 15

# every symbol include/edynhip.h declares (checked by tests/test_abi.py)
SYMBOLS = ["edynhip_create", "edynhip_destroy", "edynhip_last_error", "edynhip_set_stream", "edynhip_set_bodies",
           "edynhip_set_joints", "edynhip_step", "edynhip_run_stages", "edynhip_synchronize", "edynhip_get_state",
           "edynhip_set_state", "edynhip_pack_state_device", "edynhip_get_derived", "edynhip_num_manifolds",
           "edynhip_get_manifolds", "edynhip_set_manifolds", "edynhip_get_pairs", "edynhip_get_joint_impulses",
           "edynhip_get_timings", "edynhip_get_stats", "edynhip_abi_version", "edynhip_set_pair_filter", "edynhip_default_should_collide", "edynhip_debug_collide", "edynhip_add_bodies", "edynhip_get_asleep", "edynhip_wake_all", "edynhip_wake_bodies", "edynhip_set_center_of_mass",
           "edynhip_refresh_derived", "edynhip_exclude_collision", "edynhip_remove_collision_exclusion", "edynhip_add_joints",
           "edynhip_remove_joints", "edynhip_set_joint_params", "edynhip_remove_bodies", "edynhip_get_params", "edynhip_set_params",
           "edynhip_step_timed", "edynhip_get_contact_events", "edynhip_get_point_ids", "edynhip_snapshot", "edynhip_snapshot_read", "edynhip_snapshot_records", "edynhip_snapshot_map", "edynhip_set_event_prefetch", "edynhip_prefetched_events",
           "edynhip_set_material_extras", "edynhip_get_point_extras", "edynhip_set_joint_definition",
           "edynhip_set_generic_definition", "edynhip_get_joint_slot_impulses", "edynhip_set_material_ids", "edynhip_insert_material_mixing",
           "edynhip_measure_bandwidth", "edynhip_set_joint_warm_start", "edynhip_set_asleep", "edynhip_create_convex_mesh",
           "edynhip_get_convex_mesh",
           # multi-GPU world (ABI 12)
           "edynhip_world_create", "edynhip_world_destroy", "edynhip_world_last_error", "edynhip_world_create_convex_mesh",
           "edynhip_world_set_bodies", "edynhip_world_set_joints", "edynhip_world_set_joint_definition", "edynhip_world_exclude_collision",
           "edynhip_world_step", "edynhip_world_get_state", "edynhip_world_get_partition", "edynhip_world_repartition",
           "edynhip_world_get_manifolds", "edynhip_world_get_stats", "edynhip_world_context", "edynhip_partition_islands",
           "edynhip_island_boxes_overlap", "edynhip_get_island_boxes",
           "edynhip_world_set_pair_filter", "edynhip_world_default_should_collide", "edynhip_get_sleep_timers", "edynhip_set_sleep_timers"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing - build it with __graft_entry__.build() (make -C edyn_amd/csrc); "
                              "there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.edynhip_create.restype = C.c_void_p
        L.edynhip_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int)]
        L.edynhip_destroy.argtypes = [C.c_void_p]
        L.edynhip_last_error.restype = C.c_char_p
        L.edynhip_last_error.argtypes = [C.c_void_p]
        L.edynhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_set_bodies.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Bodies)]
        L.edynhip_set_joints.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Joints)]
        L.edynhip_step.argtypes = [C.c_void_p, C.c_uint32]
        L.edynhip_run_stages.argtypes = [C.c_void_p, C.c_uint32]
        L.edynhip_synchronize.argtypes = [C.c_void_p]
        L.edynhip_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.edynhip_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.edynhip_pack_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_get_derived.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.edynhip_num_manifolds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.edynhip_get_manifolds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_set_manifolds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.edynhip_get_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_get_joint_impulses.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
        L.edynhip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.edynhip_debug_collide.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
        L.edynhip_get_asleep.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_wake_all.argtypes = [C.c_void_p]
        L.edynhip_wake_bodies.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_set_center_of_mass.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_refresh_derived.argtypes = [C.c_void_p]
        L.edynhip_get_contact_events.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_get_point_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_snapshot.argtypes = [C.c_void_p]
        L.edynhip_snapshot_records.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32]
        L.edynhip_snapshot_map.argtypes = [C.c_void_p, C.POINTER(RecordView)]
        L.edynhip_set_event_prefetch.argtypes = [C.c_void_p, C.c_uint32]
        L.edynhip_prefetched_events.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.edynhip_set_joint_definition.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_set_material_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.edynhip_insert_material_mixing.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.edynhip_set_generic_definition.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_get_joint_slot_impulses.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_set_material_extras.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_get_point_extras.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_snapshot_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.edynhip_add_joints.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Joints), C.POINTER(C.c_uint32)]
        L.edynhip_remove_joints.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_set_joint_params.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_remove_bodies.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.edynhip_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.edynhip_step_timed.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double]
        L.edynhip_exclude_collision.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_remove_collision_exclusion.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_set_pair_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_default_should_collide.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_set_joint_warm_start.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_set_asleep.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_get_sleep_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.edynhip_set_sleep_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.edynhip_measure_bandwidth.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.edynhip_create_convex_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_get_convex_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_abi_version.restype = C.c_uint32
        L.edynhip_world_create.restype = C.c_void_p
        L.edynhip_world_create.argtypes = [C.POINTER(Config), C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
        L.edynhip_world_destroy.argtypes = [C.c_void_p]
        L.edynhip_world_last_error.restype = C.c_char_p
        L.edynhip_world_last_error.argtypes = [C.c_void_p]
        L.edynhip_world_create_convex_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_world_set_bodies.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Bodies)]
        L.edynhip_world_set_joints.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Joints)]
        L.edynhip_world_set_joint_definition.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.edynhip_world_exclude_collision.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_world_step.argtypes = [C.c_void_p, C.c_uint32]
        L.edynhip_world_set_pair_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.edynhip_world_default_should_collide.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.edynhip_world_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.edynhip_world_get_partition.argtypes = [C.c_void_p, C.c_void_p]
        L.edynhip_world_repartition.argtypes = [C.c_void_p]
        L.edynhip_world_get_manifolds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_world_get_stats.argtypes = [C.c_void_p, C.POINTER(WorldStats)]
        L.edynhip_world_context.restype = C.c_void_p
        L.edynhip_world_context.argtypes = [C.c_void_p, C.c_uint32]
        L.edynhip_partition_islands.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.edynhip_island_boxes_overlap.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.edynhip_get_island_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        _lib = L
    return _lib
