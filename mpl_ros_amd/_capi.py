"""ctypes declarations for libmplx.so (include/mplx.h).  No compute happens in Python.

The library is built in-tree by `make -C mpl_ros_amd/csrc` (hipcc --offload-arch=gfx950) or by
__graft_entry__.build().  Loading fails loudly if it is missing: there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPLX_LIB") or os.path.join(_HERE, "csrc", "libmplx.so")

VEL, ACC, JRK, SNP = 1, 3, 7, 15
PLAN_OK, PLAN_NO_PATH, PLAN_START_OCCUPIED, PLAN_MAX_EXPAND, PLAN_POOL_FULL, PLAN_INTERNAL, PLAN_TRAJ_TOO_LONG = 0, 1, 2, 3, 4, 5, 6
OK, ERR_HIP, ERR_ARG, ERR_CAPACITY, ERR_TIMEOUT = 0, -1, -2, -3, -4


class Waypoint(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("vel", C.c_double * 3), ("acc", C.c_double * 3),
                ("jrk", C.c_double * 3), ("yaw", C.c_double), ("t", C.c_double),
                ("control", C.c_int32), ("enable_t", C.c_int32)]


class Primitive(C.Structure):
    _fields_ = [("c", (C.c_double * 6) * 3), ("t", C.c_double), ("control", C.c_int32), ("pad", C.c_int32),
                ("cyaw", C.c_double * 6)]


class Config(C.Structure):
    _fields_ = [("control", C.c_int32), ("n_u", C.c_int32), ("U", C.POINTER(C.c_double)),
                ("dt", C.c_double), ("v_max", C.c_double), ("a_max", C.c_double), ("j_max", C.c_double),
                ("w", C.c_double), ("eps", C.c_double),
                ("tol_pos", C.c_double), ("tol_vel", C.c_double), ("tol_acc", C.c_double),
                ("t_max", C.c_double), ("max_expand", C.c_int32), ("heur_ignore_dynamics", C.c_int32),
                ("U_yaw", C.POINTER(C.c_double)), ("yaw_max", C.c_double), ("tol_yaw", C.c_double)]


class Succ(C.Structure):
    _fields_ = [("wp", Waypoint), ("cost", C.c_double), ("action", C.c_int32), ("valid", C.c_int32),
                ("key", C.c_int32 * 12), ("nkey", C.c_int32), ("voxel_reads", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("traj_len", C.c_int32), ("cost", C.c_double)] + \
               [(n, C.c_uint64) for n in ("n_expanded", "n_closed", "n_nodes", "n_edges", "n_primitives", "n_succ",
                                          "n_succ_finite", "voxel_reads", "n_push", "n_reopen", "n_refill", "n_evict",
                                          "expand_hash")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = [
    "mplx_ctx_create", "mplx_ctx_destroy", "mplx_last_error", "mplx_set_stream",
    "mplx_map_set", "mplx_map_set_device", "mplx_map_free_unknown", "mplx_map_get", "mplx_map_info", "mplx_map_query",
    "mplx_map_dilate", "mplx_map_cells", "mplx_map_raytrace", "mplx_map_cloud",
    "mplx_planner_config", "mplx_set_capacity", "mplx_set_bucket_width", "mplx_set_speculation", "mplx_set_helpers", "mplx_helper_stats",
    "mplx_expand_batch", "mplx_heuristic_batch", "mplx_plan", "mplx_plan_batch",
    "mplx_result_traj", "mplx_set_record", "mplx_result_expanded", "mplx_result_nodes", "mplx_result_edges", "mplx_result_blocked", "mplx_result_timing", "mplx_result_cycles", "mplx_result_speculation",
    "mplx_last_kernel_ms", "mplx_version", "mplx_kernel_name", "mplx_plan_epoch",
    "mplx_grid_create", "mplx_grid_destroy", "mplx_grid_last_error", "mplx_grid_allocate", "mplx_grid_info", "mplx_grid_clear",
    "mplx_grid_add_cloud", "mplx_grid_add_cloud_inflate", "mplx_grid_decay", "mplx_grid_clear_column", "mplx_grid_fill_column",
    "mplx_grid_fill_cell", "mplx_grid_get_map", "mplx_grid_get_cloud", "mplx_grid_to_map",
    "mplx_potential_weights", "mplx_potential_update", "mplx_search_region_set", "mplx_potential_clear", "mplx_aux_get", "mplx_aux_cloud", "mplx_aux_token",
    "mplx_lpa_create", "mplx_lpa_destroy", "mplx_lpa_last_error", "mplx_lpa_set_capacity", "mplx_lpa_set_record", "mplx_lpa_plan", "mplx_lpa_initialized",
    "mplx_lpa_reset", "mplx_lpa_update_blocked", "mplx_lpa_update_cleared", "mplx_lpa_sub_state_space", "mplx_lpa_set_reroot", "mplx_lpa_traj_len", "mplx_lpa_result_traj",
    "mplx_lpa_counts", "mplx_lpa_result_nodes", "mplx_lpa_result_edges", "mplx_lpa_result_expanded", "mplx_lpa_last_kernel_ms",
    "mplx_poly_create", "mplx_poly_destroy", "mplx_poly_last_error", "mplx_poly_config", "mplx_poly_begin", "mplx_poly_set_world",
    "mplx_poly_add_static", "mplx_poly_add_linear", "mplx_poly_add_nonlinear", "mplx_poly_commit", "mplx_poly_get_succ_batch", "mplx_poly_set_capacity", "mplx_poly_plan_batch", "mplx_poly_result_traj",
    "mplx_poly_set_record", "mplx_poly_result_expanded", "mplx_poly_last_kernel_ms", "mplx_poly_result_cycles", "mplx_poly_set_helpers", "mplx_poly_last_helpers", "mplx_poly_set_deadline", "mplx_plpa_create", "mplx_plpa_destroy", "mplx_plpa_last_error", "mplx_plpa_set_capacity", "mplx_plpa_initialized", "mplx_plpa_reset",
    "mplx_plpa_plan", "mplx_plpa_update_nodes", "mplx_plpa_changed", "mplx_plpa_sub_state_space", "mplx_plpa_traj_len", "mplx_plpa_result_traj", "mplx_plpa_last_kernel_ms", "mplx_plpa_result_cycles",
    "mplx_plpa_counts", "mplx_plpa_result_expanded", "mplx_plpa_result_nodes", "mplx_plpa_result_entries",
    "mplx_traj_solve", "mplx_traj_sample", "mplx_traj_effort",
    "mplx_plan_batch_submit", "mplx_plan_batch_wait", "mplx_plan_batch_done", "mplx_set_helper_limit", "mplx_release_pools",
    "mplx_set_deadline", "mplx_set_pool_recycling", "mplx_debug_hang_next_launch", "mplx_debug_query_records",
    "mplx_stream_create", "mplx_stream_destroy", "mplx_stream_last_error", "mplx_stream_depth", "mplx_stream_configure",
    "mplx_stream_submit", "mplx_stream_done", "mplx_stream_wait",
]


class PolySucc(C.Structure):
    _fields_ = [("state", C.c_double * 9), ("cost", C.c_double), ("action", C.c_int32), ("valid", C.c_int32)]

_lib = None


class MplxError(RuntimeError):
    pass


def load():
    """Load libmplx.so; raises MplxError when the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MplxError(f"{LIB_PATH} is missing: build it with `make -C mpl_ros_amd/csrc` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    I3 = C.POINTER(C.c_int32)
    D3 = C.POINTER(C.c_double)
    L.mplx_ctx_create.argtypes = [C.c_int, C.POINTER(P)]
    L.mplx_ctx_destroy.argtypes = [P]
    L.mplx_ctx_destroy.restype = None
    L.mplx_last_error.argtypes = [P]
    L.mplx_last_error.restype = C.c_char_p
    L.mplx_set_stream.argtypes = [P, C.c_void_p]
    L.mplx_map_set.argtypes = [P, C.c_void_p, I3, D3, C.c_double]
    L.mplx_map_set_device.argtypes = [P, C.c_void_p, I3, D3, C.c_double]
    L.mplx_map_free_unknown.argtypes = [P]
    L.mplx_map_get.argtypes = [P, C.c_void_p]
    L.mplx_map_info.argtypes = [P, I3, D3, D3]
    L.mplx_map_query.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mplx_map_dilate.argtypes = [P, C.c_int, C.c_void_p]
    L.mplx_map_cells.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p]
    L.mplx_map_raytrace.argtypes = [P, D3, D3, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.mplx_map_cloud.argtypes = [P, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mplx_planner_config.argtypes = [P, C.POINTER(Config)]
    L.mplx_set_capacity.argtypes = [P, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mplx_set_bucket_width.argtypes = [P, C.c_double]
    L.mplx_set_speculation.argtypes = [P, C.c_int32]
    L.mplx_set_helpers.argtypes = [P, C.c_int32, C.c_int32, C.c_uint64]
    L.mplx_helper_stats.argtypes = [P, C.POINTER(C.c_uint32)]
    L.mplx_expand_batch.argtypes = [P, C.c_int, C.POINTER(Waypoint), C.POINTER(Succ)]
    L.mplx_heuristic_batch.argtypes = [P, C.c_int, C.POINTER(Waypoint), C.POINTER(Waypoint), C.c_void_p, C.c_void_p]
    L.mplx_plan.argtypes = [P, C.POINTER(Waypoint), C.POINTER(Waypoint), C.POINTER(Result)]
    L.mplx_plan_batch.argtypes = [P, C.c_int, C.POINTER(Waypoint), C.POINTER(Waypoint), C.POINTER(Result)]
    L.mplx_result_traj.argtypes = [P, C.c_int, C.POINTER(Primitive), C.POINTER(Waypoint), I3, I3]
    L.mplx_set_record.argtypes = [P, C.c_uint32]
    L.mplx_result_expanded.argtypes = [P, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mplx_result_nodes.argtypes = [P, C.c_uint64, C.POINTER(Waypoint), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mplx_result_edges.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mplx_result_blocked.argtypes = [P, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mplx_result_timing.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), I3]
    L.mplx_result_cycles.argtypes = [P, C.c_int, C.POINTER(C.c_uint64)]
    L.mplx_result_speculation.argtypes = [P, C.c_int, C.POINTER(C.c_uint64)]
    L.mplx_last_kernel_ms.argtypes = [P, C.POINTER(C.c_float)]
    L.mplx_version.restype = C.c_char_p
    L.mplx_kernel_name.argtypes = [P]
    L.mplx_kernel_name.restype = C.c_char_p
    L.mplx_plan_epoch.argtypes = [P]
    L.mplx_plan_epoch.restype = C.c_uint64
    L.mplx_traj_solve.argtypes = [C.c_int32, C.c_int32, C.POINTER(Waypoint), D3, C.POINTER(Primitive)]
    L.mplx_traj_sample.argtypes = [C.c_int32, C.POINTER(Primitive), C.c_int32, C.POINTER(Waypoint), C.c_void_p]
    L.mplx_traj_effort.argtypes = [C.c_int32, C.POINTER(Primitive), C.c_int32]
    L.mplx_traj_effort.restype = C.c_double
    U64 = C.POINTER(C.c_uint64)
    L.mplx_potential_weights.argtypes = [P, C.c_double, C.c_double]
    L.mplx_potential_update.argtypes = [P, D3, D3, D3, C.c_int32]
    L.mplx_search_region_set.argtypes = [P, C.c_int, C.c_void_p, D3, C.c_int]
    L.mplx_potential_clear.argtypes = [P]
    L.mplx_aux_get.argtypes = [P, C.c_void_p]
    L.mplx_aux_token.argtypes = [P, C.c_uint64, C.c_int32, U64]
    L.mplx_aux_cloud.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, U64]
    L.mplx_lpa_create.argtypes = [P, C.POINTER(P)]
    L.mplx_lpa_destroy.argtypes = [P]
    L.mplx_lpa_destroy.restype = None
    L.mplx_lpa_last_error.argtypes = [P]
    L.mplx_lpa_last_error.restype = C.c_char_p
    L.mplx_lpa_set_capacity.argtypes = [P, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mplx_lpa_set_record.argtypes = [P, C.c_uint32]
    L.mplx_lpa_plan.argtypes = [P, C.POINTER(Waypoint), C.POINTER(Waypoint), C.POINTER(Result)]
    L.mplx_lpa_initialized.argtypes = [P]
    L.mplx_lpa_reset.argtypes = [P]
    L.mplx_lpa_update_blocked.argtypes = [P, C.c_int, C.c_void_p, U64]
    L.mplx_lpa_update_cleared.argtypes = [P, C.c_int, C.c_void_p, U64]
    L.mplx_lpa_sub_state_space.argtypes = [P, C.c_int32]
    L.mplx_lpa_set_reroot.argtypes = [P, C.c_int32]
    L.mplx_lpa_traj_len.argtypes = [P]
    L.mplx_lpa_result_traj.argtypes = [P, C.POINTER(Primitive), C.POINTER(Waypoint), I3, I3]
    L.mplx_lpa_counts.argtypes = [P, U64, U64, U64]
    L.mplx_lpa_result_nodes.argtypes = [P, C.c_uint64, C.POINTER(Waypoint)] + [C.c_void_p] * 6
    L.mplx_lpa_result_edges.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, U64]
    L.mplx_lpa_result_expanded.argtypes = [P, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mplx_lpa_last_kernel_ms.argtypes = [P, C.POINTER(C.c_float)]
    G = C.c_void_p
    L.mplx_grid_create.argtypes = [C.c_int, D3, D3, C.c_float, C.POINTER(G)]
    L.mplx_grid_destroy.argtypes = [G]
    L.mplx_grid_destroy.restype = None
    L.mplx_grid_last_error.argtypes = [G]
    L.mplx_grid_last_error.restype = C.c_char_p
    L.mplx_grid_allocate.argtypes = [G, D3, D3, C.POINTER(C.c_int)]
    L.mplx_grid_info.argtypes = [G, I3, D3, C.POINTER(C.c_float)]
    L.mplx_grid_clear.argtypes = [G]
    L.mplx_grid_add_cloud.argtypes = [G, C.c_int, C.c_void_p]
    L.mplx_grid_add_cloud_inflate.argtypes = [G, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.mplx_grid_decay.argtypes = [G]
    L.mplx_grid_clear_column.argtypes = [G, C.c_int, C.c_int]
    L.mplx_grid_fill_column.argtypes = [G, C.c_int, C.c_int]
    L.mplx_grid_fill_cell.argtypes = [G, C.c_int, C.c_int, C.c_int]
    L.mplx_grid_get_map.argtypes = [G, C.c_int, C.c_void_p]
    L.mplx_grid_get_cloud.argtypes = [G, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mplx_grid_to_map.argtypes = [G, C.c_int, P]
    D2 = C.POINTER(C.c_double)
    L.mplx_poly_create.argtypes = [C.c_int, C.POINTER(P)]
    L.mplx_poly_destroy.argtypes = [P]
    L.mplx_poly_destroy.restype = None
    L.mplx_poly_last_error.argtypes = [P]
    L.mplx_poly_last_error.restype = C.c_char_p
    L.mplx_poly_config.argtypes = [P, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    L.mplx_poly_begin.argtypes = [P, C.c_int32]
    L.mplx_poly_set_world.argtypes = [P, C.c_int32, D2, D2, C.c_double]
    L.mplx_poly_add_static.argtypes = [P, C.c_int32, C.c_int32, C.c_void_p, D2]
    L.mplx_poly_add_linear.argtypes = [P, C.c_int32, C.c_int32, C.c_void_p, D2, D2, C.c_double]
    L.mplx_poly_add_nonlinear.argtypes = [P, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_int32, C.c_int32]
    L.mplx_poly_commit.argtypes = [P]
    L.mplx_poly_get_succ_batch.argtypes = [P, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(PolySucc)]
    L.mplx_poly_set_capacity.argtypes = [P, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mplx_poly_plan_batch.argtypes = [P, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(Result)]
    L.mplx_poly_result_traj.argtypes = [P, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mplx_poly_set_record.argtypes = [P, C.c_uint32]
    L.mplx_poly_result_expanded.argtypes = [P, C.c_int32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mplx_poly_last_kernel_ms.argtypes = [P, C.POINTER(C.c_float)]
    L.mplx_poly_result_cycles.argtypes = [P, C.c_int32, C.POINTER(C.c_uint64)]
    L.mplx_poly_set_helpers.argtypes = [P, C.c_int32]
    L.mplx_poly_last_helpers.argtypes = [P]
    L.mplx_poly_set_deadline.argtypes = [P, C.c_double]
    L.mplx_plpa_create.argtypes = [P, C.POINTER(P)]
    L.mplx_plpa_destroy.argtypes = [P]
    L.mplx_plpa_destroy.restype = None
    L.mplx_plpa_last_error.argtypes = [P]
    L.mplx_plpa_last_error.restype = C.c_char_p
    L.mplx_plpa_set_capacity.argtypes = [P, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mplx_plpa_initialized.argtypes = [P]
    L.mplx_plpa_reset.argtypes = [P]
    L.mplx_plpa_plan.argtypes = [P, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(Result)]
    L.mplx_plpa_update_nodes.argtypes = [P, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mplx_plpa_changed.argtypes = [P, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.mplx_plpa_sub_state_space.argtypes = [P, C.c_int32, C.c_int32]
    L.mplx_plpa_traj_len.argtypes = [P]
    L.mplx_plpa_result_traj.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mplx_plpa_last_kernel_ms.argtypes = [P, C.POINTER(C.c_float)]
    L.mplx_plpa_result_cycles.argtypes = [P, C.POINTER(C.c_uint64)]
    L.mplx_plpa_counts.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mplx_plpa_result_expanded.argtypes = [P, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mplx_plpa_result_nodes.argtypes = [P, C.c_uint64] + [C.c_void_p] * 7
    L.mplx_plpa_result_entries.argtypes = [P, C.c_uint64] + [C.c_void_p] * 4
    L.mplx_plan_batch_submit.argtypes = [P, C.c_int, C.POINTER(Waypoint), C.POINTER(Waypoint)]
    L.mplx_plan_batch_wait.argtypes = [P, C.POINTER(Result)]
    L.mplx_plan_batch_done.argtypes = [P]
    L.mplx_set_helper_limit.argtypes = [P, C.c_int32]
    L.mplx_release_pools.argtypes = [P]
    L.mplx_set_deadline.argtypes = [P, C.c_double]
    L.mplx_set_pool_recycling.argtypes = [P, C.c_int32]
    L.mplx_debug_hang_next_launch.argtypes = [P]
    L.mplx_debug_query_records.argtypes = [P, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L.mplx_stream_create.argtypes = [P, C.c_int, C.POINTER(P)]
    L.mplx_stream_destroy.argtypes = [P]
    L.mplx_stream_destroy.restype = None
    L.mplx_stream_last_error.argtypes = [P]
    L.mplx_stream_last_error.restype = C.c_char_p
    L.mplx_stream_depth.argtypes = [P]
    L.mplx_stream_configure.argtypes = [P, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
    L.mplx_stream_submit.argtypes = [P, C.c_int, C.POINTER(Waypoint), C.POINTER(Waypoint), C.POINTER(C.c_int64)]
    L.mplx_stream_done.argtypes = [P, C.c_int64]
    L.mplx_stream_wait.argtypes = [P, C.c_int64, C.POINTER(Result), C.POINTER(P)]
    _lib = L
    return L
