"""ctypes binding of libmp2p_hip.so (include/mp2p_hip.h).

The product path is the HIP library; there is no CPU fallback.  Loading fails loudly when the
library is missing (and cannot be built), and every compute call fails loudly without a GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _build

PAIR_PT2PT = np.dtype(
    [("globalIdx", "<u4"), ("localIdx", "<u4"), ("global", "<f4", (3,)), ("local", "<f4", (3,)),
     ("errorSquareAfterTransformation", "<f4")])
PAIR_PT2LN = np.dtype([("ln_base", "<f8", (3,)), ("ln_director", "<f8", (3,)), ("pt_local", "<f8", (3,))])
PAIR_PL2PL = np.dtype([("pl_global", "<f8", (4,)), ("c_global", "<f8", (3,)), ("pl_local", "<f8", (4,)),
                       ("c_local", "<f8", (3,))])
PAIR_PT2PL = np.dtype(
    [("plane", "<f8", (4,)), ("centroid", "<f8", (3,)), ("pt_local", "<f4", (3,)), ("_pad", "<f4")])
assert PAIR_PT2PT.itemsize == 36 and PAIR_PT2PL.itemsize == 72

ABI_VERSION = 4  # MP2P_HIP_ABI_VERSION of include/mp2p_hip.h
OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_CAPACITY, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5
KERNEL_NONE, KERNEL_GEMANMCCLURE, KERNEL_CAUCHY = 0, 1, 2
MAX_WEIGHT_BLOCKS = 32  # MP2P_HIP_MAX_WEIGHT_BLOCKS
GN_NSUMS = 48


class MapParams(C.Structure):
    _fields_ = [("cell_size", C.c_float), ("target_per_cell", C.c_float),
                ("max_levels", C.c_uint32), ("no_occupancy_bitmap", C.c_uint32)]


class InlierRatioParams(C.Structure):
    _fields_ = [("inliersRatio", C.c_double), ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("allowMatchAlreadyMatchedGlobalPoints", C.c_int32),
                ("bounding_box_intersection_check_epsilon", C.c_double)]


class AdaptiveParams(C.Structure):
    """mp2p_hip_adaptive_params (Matcher_Adaptive.h:67-77)"""
    _fields_ = [("confidenceInterval", C.c_double), ("firstToSecondDistanceMax", C.c_double),
                ("absoluteMaxSearchDistance", C.c_double), ("minimumCorrDist", C.c_double),
                ("enableDetectPlanes", C.c_int32), ("maxPt2PtCorrespondences", C.c_uint32),
                ("planeSearchPoints", C.c_uint32), ("planeMinimumFoundPoints", C.c_uint32),
                ("planeMinimumDistance", C.c_double), ("planeEigenThreshold", C.c_double),
                ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("allowMatchAlreadyMatchedGlobalPoints", C.c_int32),
                ("bounding_box_intersection_check_epsilon", C.c_double)]


ADAPTIVE_BINS = 50


class AdaptiveHist(C.Structure):
    _fields_ = [("valid", C.c_int32), ("minSqr", C.c_float), ("maxSqr", C.c_float),
                ("count", C.c_uint64), ("bins", C.c_uint64 * ADAPTIVE_BINS)]


class HornParams(C.Structure):
    """mp2p_hip_horn_params (WeightParameters.h:34-72 + Pairings::point_weights)"""
    _fields_ = [("use_scale_outlier_detector", C.c_int32), ("scale_outlier_threshold", C.c_double),
                ("w_pt2pt", C.c_double), ("w_ln2ln", C.c_double), ("w_pl2pl", C.c_double),
                ("robust_kernel", C.c_int32), ("robust_kernel_param", C.c_double),
                ("has_current_estimate", C.c_int32), ("current_estimate", C.c_double * 12),
                ("n_weight_blocks", C.c_uint32), ("weight_block_count", C.POINTER(C.c_size_t)),
                ("weight_block_w", C.POINTER(C.c_double))]


class HornResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("solved", C.c_int32), ("n_outliers", C.c_uint64)]


class DecimateParams(C.Structure):
    _fields_ = [("voxel_filter_resolution", C.c_float), ("decimate_method", C.c_int32),
                ("has_flatten_to", C.c_int32), ("flatten_to", C.c_float)]


DECIMATE_FIRST_POINT, DECIMATE_CLOSEST_TO_AVERAGE, DECIMATE_VOXEL_AVERAGE = 0, 1, 2


class MapInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("cell_size", C.c_float), ("n_levels", C.c_uint32), ("n_cells_total", C.c_uint64),
                ("n_cells_level0", C.c_uint64), ("hash_capacity", C.c_uint64),
                ("device_bytes", C.c_uint64), ("build_ms", C.c_double)]


class Pt2PtParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("thresholdAngularDeg", C.c_double),
                ("pairingsPerPoint", C.c_uint32),
                ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("allowMatchAlreadyMatchedGlobalPoints", C.c_int32),
                ("bounding_box_intersection_check_epsilon", C.c_double),
                ("local_index_offset", C.c_uint64), ("initial_radius_cells", C.c_float),
                ("queries_per_wave", C.c_uint32), ("group_radius_factor", C.c_float),
                ("cell_budget", C.c_uint32), ("defer_radius_cells", C.c_float),
                ("disable_warm_start", C.c_int32), ("brick_budget", C.c_uint32),
                ("tile_order", C.c_int32), ("multi_search_radius_mode", C.c_int32)]


class Pt2PlParams(C.Structure):
    _fields_ = [("distanceThreshold", C.c_double), ("searchRadius", C.c_double),
                ("knn", C.c_uint32), ("minimumPlanePoints", C.c_uint32),
                ("planeEigenThreshold", C.c_double),
                ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("bounding_box_intersection_check_epsilon", C.c_double),
                ("initial_radius_cells", C.c_float), ("queries_per_wave", C.c_uint32)]


class NearestPlane(C.Structure):
    _fields_ = [("found", C.c_int32), ("plane", C.c_double * 4), ("centroid", C.c_double * 3),
                ("distance", C.c_float)]


class GNParams(C.Structure):
    _fields_ = [("maxInnerLoopIterations", C.c_uint32), ("minDelta", C.c_double),
                ("maxCost", C.c_double), ("kernel", C.c_int32), ("kernelParam", C.c_double),
                ("w_pt2pt", C.c_double), ("w_pt2pl", C.c_double), ("has_prior", C.c_int32),
                ("prior_mean", C.c_double * 12), ("prior_cov_inv", C.c_double * 36),
                ("n_weight_blocks", C.c_uint32), ("weight_block_count", C.c_uint64 * MAX_WEIGHT_BLOCKS),
                ("weight_block_w", C.c_double * MAX_WEIGHT_BLOCKS), ("w_pt2ln", C.c_double), ("w_pl2pl", C.c_double)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if "w_pt2ln" not in k and len(a) < 14:
            self.w_pt2ln = 1.0   # PairWeights defaults (PairWeights.h)
        if "w_pl2pl" not in k and len(a) < 15:
            self.w_pl2pl = 1.0


class GNResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("H", C.c_double * 36), ("g", C.c_double * 6),
                ("cost", C.c_double), ("iterations", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("ms_nn", C.c_double), ("ms_nn_tile", C.c_double), ("ms_nn_single", C.c_double),
                ("ms_compact", C.c_double), ("ms_gn", C.c_double),
                ("nn_tiles", C.c_uint64), ("nn_passes", C.c_uint64),
                ("nn_cells_visited", C.c_uint64), ("nn_candidates_tested", C.c_uint64),
                ("nn_points_staged", C.c_uint64), ("nn_queries", C.c_uint64),
                ("nn_unresolved_after_first_pass", C.c_uint64),
                ("nn_max_candidates_one_tile", C.c_uint64), ("nn_max_passes_one_tile", C.c_uint64),
                ("nn_tile_ticks_sum", C.c_uint64), ("nn_tile_ticks_max", C.c_uint64),
                ("nn_coop_passes", C.c_uint64), ("nn_single_queries", C.c_uint64),
                ("nn_single_passes", C.c_uint64), ("nn_single_cells", C.c_uint64),
                ("nn_single_candidates", C.c_uint64), ("nn_single_max_candidates", C.c_uint64),
                ("nn_tile_ticks_hist", C.c_uint64 * 24),
                ("nn_single_ticks_sum", C.c_uint64), ("nn_single_ticks_max", C.c_uint64),
                ("nn_single_max_passes", C.c_uint64), ("nn_single_max_cells", C.c_uint64),
                ("ms_nn_lane", C.c_double), ("nn_lane_searched", C.c_uint64),
                ("nn_lane_candidates", C.c_uint64), ("nn_lane_voxels", C.c_uint64),
                ("nn_lane_pending", C.c_uint64), ("nn_lane_skipped", C.c_uint64),
                ("nn_wave_path", C.c_uint64), ("nn_sel_voxels_listed", C.c_uint64),
                ("nn_sel_voxels_needed", C.c_uint64), ("nn_wave_inserts", C.c_uint64),
                ("nn_wave_overflows", C.c_uint64), ("nn_wave_rounds", C.c_uint64),
                ("nn_wave_toobig", C.c_uint64), ("nn_wave_phase_ticks", C.c_uint64 * 6),
                ("pl_certified", C.c_uint64), ("pl_searched", C.c_uint64)]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COMM_ID_BYTES = 128

# name -> (restype, argtypes).  tests/test_abi.py checks this table against include/mp2p_hip.h.
SIGNATURES = {
    "mp2p_hip_abi_version": (C.c_int, []),
    "mp2p_hip_abi_check": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "mp2p_hip_device_count": (C.c_int, []),
    "mp2p_hip_ctx_create": (C.c_int, [C.c_int, _P, _PP]),
    "mp2p_hip_ctx_destroy": (None, [_P]),
    "mp2p_hip_last_error": (C.c_char_p, [_P]),
    "mp2p_hip_sync": (C.c_int, [_P]),
    "mp2p_hip_ctx_stream": (_P, [_P]),
    "mp2p_hip_debug_alloc_count": (C.c_ulonglong, []),
    "mp2p_hip_map_upload": (C.c_int, [_P, _fp, _fp, _fp, C.c_size_t, C.POINTER(MapParams), _PP]),
    "mp2p_hip_map_upload_device": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(MapParams), _PP]),
    "mp2p_hip_map_free": (None, [_P, _P]),
    "mp2p_hip_map_get_info": (C.c_int, [_P, _P, C.POINTER(MapInfo)]),
    "mp2p_hip_cloud_upload": (C.c_int, [_P, _fp, _fp, _fp, C.c_size_t, _PP]),
    "mp2p_hip_cloud_upload_device": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _PP]),
    "mp2p_hip_cloud_free": (None, [_P, _P]),
    "mp2p_hip_cloud_size": (C.c_size_t, [_P]),
    "mp2p_hip_mstate_create": (C.c_int, [_P, C.c_size_t, C.c_size_t, _PP]),
    "mp2p_hip_mstate_reset": (C.c_int, [_P, _P]),
    "mp2p_hip_mstate_free": (None, [_P, _P]),
    "mp2p_hip_mstate_download": (C.c_int, [_P, _P, _u8p, _u8p]),
    "mp2p_hip_mstate_upload": (C.c_int, [_P, _P, _u8p, _u8p]),
    "mp2p_hip_mstate_upload_bits": (C.c_int, [_P, _P, _u64p, _u64p]),
    "mp2p_hip_mstate_download_bits": (C.c_int, [_P, _P, _u64p, _u64p]),
    "mp2p_hip_host_alloc": (_P, [_P, C.c_size_t]),
    "mp2p_hip_host_free": (None, [_P, _P]),
    "mp2p_hip_pairs_download_pt2pt_from": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t),
                                                     _u64p]),
    "mp2p_hip_pairs_download_pt2pl_from": (C.c_int, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_uint32), C.c_size_t,
                                                     C.POINTER(C.c_size_t), _u64p]),
    "mp2p_hip_pairs_copy_pt2pt": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P]),
    "mp2p_hip_pairs_copy_pt2pt_begin": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.POINTER(C.c_uint32),
                                                  C.POINTER(C.c_uint32)]),
    "mp2p_hip_pairs_copy_pt2pt_begin_soa": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.POINTER(C.c_uint32),
                                                      C.POINTER(C.c_uint32), _fp, _fp, _fp, C.c_size_t, C.c_uint64]),
    "mp2p_hip_pairs_copy_wait_idx": (C.c_int, [_P]),
    "mp2p_hip_pairs_copy_end": (C.c_int, [_P]),
    "mp2p_hip_pairs_copy_pt2pl": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.POINTER(C.c_uint32)]),
    "mp2p_hip_pairs_create": (C.c_int, [_P, C.c_size_t, C.c_size_t, _PP]),
    "mp2p_hip_pairs_free": (None, [_P, _P]),
    "mp2p_hip_pairs_clear": (C.c_int, [_P, _P]),
    "mp2p_hip_pairs_reserve": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t]),
    "mp2p_hip_pairs_counts": (C.c_int, [_P, _P, _u64p, _u64p, _u64p]),
    "mp2p_hip_pairs_download_pt2pt": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mp2p_hip_pairs_download_pt2pl": (C.c_int, [_P, _P, _P, C.POINTER(C.c_uint32), C.c_size_t,
                                                C.POINTER(C.c_size_t)]),
    "mp2p_hip_covariance": (C.c_int, [_P, _P, _dp, C.c_double, C.c_double, _dp, _dp, C.POINTER(C.c_int32)]),
    "mp2p_hip_filter_decimate_voxels": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(DecimateParams),
                                                  _P, _P, _P, _P, C.POINTER(C.c_size_t)]),
    "mp2p_hip_filter_decimate_voxels_device": (C.c_int, [_P, _P, _P, _P, C.c_size_t,
                                                         C.POINTER(DecimateParams), _P, _P, _P, _P,
                                                         C.POINTER(C.c_size_t)]),
    "mp2p_hip_pairs_upload_lines_planes": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_size_t]),
    "mp2p_hip_pairs_counts_lines_planes": (C.c_int, [_P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mp2p_hip_pairs_upload": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_size_t]),
    "mp2p_hip_match_pt2pt": (C.c_int, [_P, _P, _P, _dp, C.POINTER(Pt2PtParams), _P, _P]),
    "mp2p_hip_match_pt2pt_phase1": (C.c_int, [_P, _P, _P, _dp, C.POINTER(Pt2PtParams), _P]),
    "mp2p_hip_match_pt2pt_phase2": (C.c_int, [_P, _P, _P, C.POINTER(Pt2PtParams), _P, _P]),
    "mp2p_hip_exchange_pack": (C.c_int, [_P, _P, _P, C.POINTER(Pt2PtParams), C.POINTER(_P), C.POINTER(_P),
                                         C.POINTER(C.c_size_t)]),
    "mp2p_hip_cloud_set_visit_order": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mp2p_hip_exchange_unpack": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mp2p_hip_map_claims_ptr": (_P, [_P]),
    "mp2p_hip_map_claims_count": (C.c_size_t, [_P]),
    "mp2p_hip_ctx_local_bbox_ptr": (_P, [_P]),
    "mp2p_hip_match_inlier_ratio": (C.c_int, [_P, _P, _P, _dp, C.POINTER(InlierRatioParams), _P, _P]),
    "mp2p_hip_adaptive_search": (C.c_int, [_P, _P, _P, _dp, C.POINTER(AdaptiveParams), _P,
                                           C.POINTER(AdaptiveHist)]),
    "mp2p_hip_adaptive_ci_high": (C.c_double, [C.POINTER(AdaptiveHist), C.c_double]),
    "mp2p_hip_adaptive_select": (C.c_int, [_P, _P, _P, C.POINTER(AdaptiveParams), C.c_double, _P, _P]),
    "mp2p_hip_match_adaptive": (C.c_int, [_P, _P, _P, _dp, C.POINTER(AdaptiveParams), _P, _P,
                                          C.POINTER(C.c_double), C.POINTER(AdaptiveHist)]),
    "mp2p_hip_match_pt2pl": (C.c_int, [_P, _P, _P, _dp, C.POINTER(Pt2PlParams), _P, _P]),
    "mp2p_hip_nn_search_pt2pl": (C.c_int, [_P, _P, _fp, C.c_float, C.POINTER(Pt2PlParams), C.POINTER(NearestPlane)]),
    "mp2p_hip_gn_solve": (C.c_int, [_P, _P, _dp, C.POINTER(GNParams), C.POINTER(GNResult)]),
    "mp2p_hip_gn_begin": (C.c_int, [_P, _P, _dp, C.POINTER(GNParams)]),
    "mp2p_hip_gn_accumulate": (C.c_int, [_P]),
    "mp2p_hip_gn_sums_ptr": (_P, [_P]),
    "mp2p_hip_gn_step": (C.c_int, [_P]),
    "mp2p_hip_gn_end": (C.c_int, [_P, C.POINTER(GNResult)]),
    "mp2p_hip_horn_solve": (C.c_int, [_P, _P, C.c_double, _dp, C.POINTER(C.c_int32)]),
    "mp2p_hip_horn_solve_wp": (C.c_int, [_P, _P, C.POINTER(HornParams), C.POINTER(HornResult)]),
    "mp2p_hip_horn_outlier_flags": (C.c_int, [_P, C.POINTER(C.c_uint8), C.c_size_t]),
    "mp2p_hip_pairs_pt2ln_pl_to_pt2pt": (C.c_int, [_P, _P, _dp, _P]),
    "mp2p_hip_comm_get_unique_id": (C.c_int, [_P]),
    "mp2p_hip_comm_available": (C.c_int, [_P]),
    "mp2p_hip_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "mp2p_hip_comm_init_hooks": (C.c_int, [_P, C.c_int, C.c_int, ALLREDUCE_FN, ALLGATHER_FN, _P]),
    "mp2p_hip_comm_destroy": (C.c_int, [_P]),
    "mp2p_hip_comm_rank": (C.c_int, [_P]),
    "mp2p_hip_comm_size": (C.c_int, [_P]),
    "mp2p_hip_comm_allreduce_f64": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "mp2p_hip_step_sharded": (C.c_int, [_P, _P, _P, _dp, C.POINTER(Pt2PtParams), C.POINTER(GNParams), _P,
                                        C.POINTER(GNResult), C.POINTER(C.c_int32)]),
    "mp2p_hip_step_sharded_pt2pl": (C.c_int, [_P, _P, _P, _dp, C.POINTER(Pt2PlParams), C.c_uint64, C.POINTER(GNParams), _P,
                                              C.POINTER(GNResult)]),
    "mp2p_hip_set_profiling": (C.c_int, [_P, C.c_int]),
    "mp2p_hip_set_tune": (C.c_int, [_P, C.c_char_p]),
    "mp2p_hip_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "mp2p_hip_get_timeline": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t)]),
}


class Mp2pHipError(RuntimeError):
    """Raised for every non-zero return code of libmp2p_hip (the adapter rethrows these as
    std::runtime_error, like the reference's THROW_EXCEPTION / ASSERT_)."""

    def __init__(self, code, msg):
        super().__init__(f"libmp2p_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib_path():
    return _build.LIB


def load():
    """dlopen the HIP library (building it first if the sources are newer and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    try:
        # torch wheels bundle their own HIP runtime; loading it FIRST makes our library bind to
        # the same libamdhip64 (two runtimes in one process do not both see the device)
        import torch  # noqa: F401
    except Exception:
        pass
    if _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # stale or missing and cannot rebuild
            if not os.path.exists(path):
                raise ImportError(
                    f"libmp2p_hip.so is missing and could not be built ({e}); "
                    "mp2p_icp_amd has no CPU fallback") from e
            # a stale library is only acceptable when it still speaks this ABI (checked below) -- say so, loudly
            import sys
            print(f"[mp2p_icp_amd] WARNING: libmp2p_hip.so is older than its sources and could not be rebuilt ({e}); "
                  "loading it as it is (the ABI check below decides)", file=sys.stderr)
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    # the binding's view of the header against the library's (MP2P_HIP_ABI_CHECK of include/mp2p_hip.h): a stale .so, or
    # ctypes structs that lag behind the header, fail HERE and not as a solver reading its weights from the wrong offsets
    if L.mp2p_hip_abi_version() != ABI_VERSION:
        raise ImportError(f"libmp2p_hip.so is ABI version {L.mp2p_hip_abi_version()}, this binding is version {ABI_VERSION}: rebuild the library")
    if L.mp2p_hip_abi_check(ABI_VERSION, C.sizeof(Pt2PtParams), C.sizeof(Pt2PlParams), C.sizeof(GNParams), C.sizeof(GNResult),
                            C.sizeof(Stats)) != 0:
        msg = L.mp2p_hip_last_error(None)
        raise ImportError("libmp2p_hip.so ABI mismatch: " + (msg.decode() if msg else "?"))
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != 0:
        msg = load().mp2p_hip_last_error(ctx)
        raise Mp2pHipError(rc, msg.decode() if msg else "?")
