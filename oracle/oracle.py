"""ctypes binding of oracle/_build/libmp2p_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the mp2p_icp_amd product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmp2p_oracle.so")

PAIR_PT2PT = np.dtype(
    [("globalIdx", "<u4"), ("localIdx", "<u4"), ("gx", "<f4"), ("gy", "<f4"), ("gz", "<f4"),
     ("lx", "<f4"), ("ly", "<f4"), ("lz", "<f4"), ("errSq", "<f4")])
PAIR_PT2PL = np.dtype(
    [("plane", "<f8", (4,)), ("centroid", "<f8", (3,)), ("lx", "<f4"), ("ly", "<f4"),
     ("lz", "<f4"), ("_pad", "<f4")])
PAIR_PT2LN = np.dtype(
    [("pbase", "<f8", (3,)), ("director", "<f8", (3,)), ("lx", "<f8"), ("ly", "<f8"), ("lz", "<f8")])
PAIR_PL2PL = np.dtype(
    [("pl_global", "<f8", (4,)), ("c_global", "<f8", (3,)), ("pl_local", "<f8", (4,)),
     ("c_local", "<f8", (3,))])
assert PAIR_PT2PT.itemsize == 36 and PAIR_PT2PL.itemsize == 72 and PAIR_PT2LN.itemsize == 72
assert PAIR_PL2PL.itemsize == 112

KERNEL_NONE, KERNEL_GEMANMCCLURE, KERNEL_CAUCHY = 0, 1, 2


class Pt2PtParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("thresholdAngularDeg", C.c_double),
                ("pairingsPerPoint", C.c_uint32),
                ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("allowMatchAlreadyMatchedGlobalPoints", C.c_int32),
                ("bbox_eps", C.c_double), ("multi_search_radius_mode", C.c_int32)]


class Pt2PlParams(C.Structure):
    _fields_ = [("distanceThreshold", C.c_double), ("searchRadius", C.c_double),
                ("knn", C.c_uint32), ("minimumPlanePoints", C.c_uint32),
                ("planeEigenThreshold", C.c_double),
                ("allowMatchAlreadyMatchedPoints", C.c_int32), ("bbox_eps", C.c_double)]


class AdaptiveParams(C.Structure):
    _fields_ = [("confidenceInterval", C.c_double), ("firstToSecondDistanceMax", C.c_double),
                ("absoluteMaxSearchDistance", C.c_double), ("minimumCorrDist", C.c_double),
                ("enableDetectPlanes", C.c_int32), ("maxPt2PtCorrespondences", C.c_uint32),
                ("planeSearchPoints", C.c_uint32), ("planeMinimumFoundPoints", C.c_uint32),
                ("planeMinimumDistance", C.c_double), ("planeEigenThreshold", C.c_double),
                ("allowMatchAlreadyMatchedPoints", C.c_int32),
                ("allowMatchAlreadyMatchedGlobalPoints", C.c_int32), ("bbox_eps", C.c_double)]


ADAPTIVE_BINS = 50


class AdaptiveHist(C.Structure):
    _fields_ = [("valid", C.c_int32), ("minSq", C.c_float), ("maxSq", C.c_float),
                ("count", C.c_uint64), ("bins", C.c_uint64 * ADAPTIVE_BINS)]


class HornParams(C.Structure):
    _fields_ = [("use_scale_outlier_detector", C.c_int32), ("scale_outlier_threshold", C.c_double),
                ("w_pt2pt", C.c_double), ("w_ln2ln", C.c_double), ("w_pl2pl", C.c_double),
                ("robust_kernel", C.c_int32), ("robust_kernel_param", C.c_double),
                ("has_current_estimate", C.c_int32), ("current_estimate", C.c_double * 12),
                ("n_weight_blocks", C.c_uint32), ("weight_block_count", C.POINTER(C.c_size_t)),
                ("weight_block_w", C.POINTER(C.c_double))]


class GNParams(C.Structure):
    _fields_ = [("maxInnerLoopIterations", C.c_uint32), ("minDelta", C.c_double),
                ("maxCost", C.c_double), ("kernel", C.c_int32), ("kernelParam", C.c_double),
                ("w_pt2pt", C.c_double), ("w_pt2pl", C.c_double), ("w_pt2ln", C.c_double),
                ("w_pl2pl", C.c_double), ("has_prior", C.c_int32), ("prior_mean", C.c_double * 12),
                ("prior_cov_inv", C.c_double * 36), ("n_weight_blocks", C.c_uint32),
                ("weight_block_count", C.POINTER(C.c_size_t)),
                ("weight_block_w", C.POINTER(C.c_double)),
                ("reset_weight_cursor_each_iter", C.c_int32)]


def build(force=False):
    """Compile the oracle (gcc).  Building the checker is not using it."""
    if force or not os.path.exists(_SO) or (
            os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f))
                                        for f in ("mp2p_oracle.c", "mp2p_oracle.h", "Makefile"))):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def _declare(L):
    L.orc_version.restype = C.c_char_p
    L.orc_pose_from_xyzypr.argtypes = [C.c_double] * 6 + [_dp]
    L.orc_pose_to_xyzypr.argtypes = [_dp, _dp]
    L.orc_pose_compose.argtypes = [_dp, _dp, _dp]
    L.orc_pose_inverse.argtypes = [_dp, _dp]
    L.orc_se3_exp.argtypes = [_dp, _dp]
    L.orc_se3_log.argtypes = [_dp, _dp]
    L.orc_jacob_dDexpe_de.argtypes = [_dp, _dp]
    for n in ("orc_error_point2point", "orc_error_point2plane", "orc_error_point2line",
              "orc_error_plane2plane"):
        getattr(L, n).argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.orc_robust_weight.argtypes = [C.c_int32, C.c_double, C.c_double]
    L.orc_robust_weight.restype = C.c_double
    L.orc_transform_local_to_global.argtypes = [_fp, _fp, _fp, C.c_size_t, _dp, _fp, _fp, _fp,
                                                _fp, _fp]
    L.orc_kdtree_build.argtypes = [_fp, _fp, _fp, C.c_size_t, C.c_int]
    L.orc_kdtree_build.restype = C.c_void_p
    L.orc_kdtree_free.argtypes = [C.c_void_p]
    L.orc_kdtree_knn.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                 C.c_float, _u32p, _fp]
    L.orc_kdtree_knn.restype = C.c_int
    L.orc_brute_knn.argtypes = [_fp, _fp, _fp, C.c_size_t, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_float, _u32p, _fp]
    L.orc_brute_knn.restype = C.c_int
    L.orc_match_pt2pt.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                  C.c_size_t, _dp, C.POINTER(Pt2PtParams), _u8p, _u8p,
                                  C.c_void_p, C.POINTER(C.c_uint64)]
    L.orc_match_pt2pt.restype = C.c_size_t
    L.orc_match_pt2pt_mt.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                     C.c_size_t, _dp, C.POINTER(Pt2PtParams), C.c_void_p,
                                     C.c_int]
    L.orc_match_pt2pt_mt.restype = C.c_size_t
    L.orc_match_pt2pt_mt_ms.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                        C.c_size_t, _dp, C.POINTER(Pt2PtParams), _u8p, _u8p, C.c_void_p,
                                        C.c_int]
    L.orc_match_pt2pt_mt_ms.restype = C.c_size_t
    L.orc_match_pt2pl_mt.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                     C.c_size_t, _dp, C.POINTER(Pt2PlParams), _u8p, C.c_void_p,
                                     _u32p, C.POINTER(C.c_uint64), C.c_int]
    L.orc_match_pt2pl_mt.restype = C.c_size_t
    L.orc_match_pt2pl.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                  C.c_size_t, _dp, C.POINTER(Pt2PlParams), _u8p, C.c_void_p,
                                  _u32p, C.POINTER(C.c_uint64)]
    L.orc_match_pt2pl.restype = C.c_size_t
    L.orc_match_pt2pt_subset.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                         C.c_size_t, _u32p, C.c_size_t, _dp, C.POINTER(Pt2PtParams),
                                         _u8p, _u8p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.orc_match_pt2pt_subset.restype = C.c_size_t
    L.orc_match_pt2pl_subset.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp,
                                         C.c_size_t, _u32p, C.c_size_t, _dp, C.POINTER(Pt2PlParams),
                                         _u8p, C.c_void_p, _u32p, C.POINTER(C.c_uint64)]
    L.orc_match_pt2pl_subset.restype = C.c_size_t
    L.orc_adaptive_ci_high.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_uint64), C.c_int, C.c_uint64,
                                       C.c_double]
    L.orc_adaptive_ci_high.restype = C.c_double
    L.orc_match_adaptive.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp, C.c_size_t, _dp,
                                     C.c_void_p, _u8p, _u8p, C.c_int, _dp, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_size_t), C.c_void_p, _u32p, C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_uint64)]
    L.orc_match_adaptive.restype = C.c_int
    L.orc_match_inlier_ratio.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp, C.c_size_t,
                                         _u32p, C.c_size_t, _dp, C.c_double, C.c_int, C.c_int, C.c_double,
                                         _u8p, _u8p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.orc_match_inlier_ratio.restype = C.c_size_t
    L.orc_covariance.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.c_void_p, C.c_size_t, _dp, C.c_double, C.c_double, _dp, _dp]
    L.orc_covariance.restype = C.c_int
    L.orc_filter_decimate_voxels.argtypes = [_fp, _fp, _fp, C.c_size_t, C.c_float, C.c_int, C.c_int,
                                             C.c_float, _fp, _fp, _fp, _u32p]
    L.orc_filter_decimate_voxels.restype = C.c_size_t
    L.orc_estimate_points_eigen.argtypes = [_fp, _fp, _fp, C.c_size_t, _fp, _dp, _dp, _dp]
    L.orc_optimal_tf_gauss_newton.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _dp,
                                              C.POINTER(GNParams), _dp, _dp, _dp]
    L.orc_optimal_tf_gauss_newton.restype = C.c_int
    L.orc_optimal_tf_gauss_newton_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                 _dp, C.POINTER(GNParams), _dp, C.c_int]
    L.orc_optimal_tf_gauss_newton_mt.restype = C.c_int
    L.orc_optimal_tf_horn.argtypes = [C.c_void_p, C.c_size_t, C.c_double, _dp]
    L.orc_optimal_tf_horn.restype = C.c_int
    L.orc_optimal_tf_horn_wp.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, _dp, _u8p]
    L.orc_optimal_tf_horn_wp.restype = C.c_int
    L.orc_pt2ln_pl_to_pt2pt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _dp, C.c_void_p]
    L.orc_pt2ln_pl_to_pt2pt.restype = C.c_size_t


# ------------------------------------------------------------------------------------------
def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pose_from_xyzypr(x, y, z, yaw=0.0, pitch=0.0, roll=0.0):
    T = np.zeros(12)
    lib().orc_pose_from_xyzypr(x, y, z, yaw, pitch, roll, _d(T))
    return T


def pose_to_xyzypr(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    o = np.zeros(6)
    lib().orc_pose_to_xyzypr(_d(T), _d(o))
    return o


def pose_identity():
    return pose_from_xyzypr(0, 0, 0)


def pose_compose(A, B):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    o = np.zeros(12)
    lib().orc_pose_compose(_d(A), _d(B), _d(o))
    return o


def pose_inverse(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    o = np.zeros(12)
    lib().orc_pose_inverse(_d(A), _d(o))
    return o


def pose_compose_point(T, p):
    R = np.asarray(T[:9]).reshape(3, 3)
    return R @ np.asarray(p, dtype=np.float64) + np.asarray(T[9:12])


def pose_inverse_compose_point(T, g):
    R = np.asarray(T[:9]).reshape(3, 3)
    return R.T @ (np.asarray(g, dtype=np.float64) - np.asarray(T[9:12]))


def se3_exp(xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    T = np.zeros(12)
    lib().orc_se3_exp(_d(xi), _d(T))
    return T


def se3_log(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    xi = np.zeros(6)
    lib().orc_se3_log(_d(T), _d(xi))
    return xi


def pose_err(A, B):
    """|| log(B^-1 A) || -- the check used all over the reference tests
    (e.g. test-mp2p_optimize_pt2pl.cpp:75)."""
    return float(np.linalg.norm(se3_log(pose_compose(pose_inverse(B), A))))


def pose_err_split(A, B):
    xi = se3_log(pose_compose(pose_inverse(B), A))
    return float(np.linalg.norm(xi[:3])), float(np.linalg.norm(xi[3:]))


def jacob_dDexpe_de(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    J = np.zeros(72)
    lib().orc_jacob_dDexpe_de(_d(T), _d(J))
    return J.reshape(12, 6)


def _err_term(fn, pair, T, want_jac=True):
    T = np.ascontiguousarray(T, dtype=np.float64)
    e = np.zeros(3)
    J = np.zeros(36)
    getattr(lib(), fn)(pair.ctypes.data, _d(T), _d(e), _d(J) if want_jac else None)
    return e, J.reshape(3, 12)


def error_point2point(pair, T):
    return _err_term("orc_error_point2point", pair, T)


def error_point2plane(pair, T):
    return _err_term("orc_error_point2plane", pair, T)


def error_point2line(pair, T):
    return _err_term("orc_error_point2line", pair, T)


def error_plane2plane(pair, T):
    return _err_term("orc_error_plane2plane", pair, T)


def robust_weight(kernel, c, esq):
    return lib().orc_robust_weight(kernel, c, esq)


def transform_local_to_global(lx, ly, lz, T):
    lx, ly, lz = _f32(lx), _f32(ly), _f32(lz)
    T = np.ascontiguousarray(T, dtype=np.float64)
    n = lx.size
    ox, oy, oz = (np.empty(n, np.float32) for _ in range(3))
    bmin, bmax = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_transform_local_to_global(_f(lx), _f(ly), _f(lz), n, _d(T), _f(ox), _f(oy), _f(oz),
                                        _f(bmin), _f(bmax))
    return ox, oy, oz, bmin, bmax


class KDTree:
    """Exact fp32 KD-tree over SoA points (kept alive with the arrays)."""

    def __init__(self, x, y, z, leaf_max=10):
        self.x, self.y, self.z = _f32(x), _f32(y), _f32(z)
        self.n = self.x.size
        self._h = lib().orc_kdtree_build(_f(self.x), _f(self.y), _f(self.z), self.n, leaf_max)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kdtree_free(self._h)
            self._h = None

    def knn(self, q, k=1, max_d2=-1.0):
        idx = np.zeros(k, np.uint32)
        d2 = np.zeros(k, np.float32)
        n = lib().orc_kdtree_knn(self._h, float(q[0]), float(q[1]), float(q[2]), k, max_d2,
                                 idx.ctypes.data_as(_u32p), _f(d2))
        return idx[:n], d2[:n]


def brute_knn(x, y, z, q, k=1, max_d2=-1.0):
    x, y, z = _f32(x), _f32(y), _f32(z)
    idx = np.zeros(k, np.uint32)
    d2 = np.zeros(k, np.float32)
    n = lib().orc_brute_knn(_f(x), _f(y), _f(z), x.size, float(q[0]), float(q[1]), float(q[2]),
                            k, max_d2, idx.ctypes.data_as(_u32p), _f(d2))
    return idx[:n], d2[:n]


def match_pt2pt(gx, gy, gz, lx, ly, lz, T, threshold, thresholdAngularDeg, pairingsPerPoint=1,
                allowMatchAlreadyMatchedPoints=False, allowMatchAlreadyMatchedGlobalPoints=False,
                bbox_eps=0.20, tree=None, local_taken=None, global_taken=None,
                multi_search_radius_mode=0, threads=0, idxs=None):
    """Matcher_Points_DistanceThreshold::implMatchOneLayer.  Returns (pairs, potential).
    idxs: the visit list of maxLocalPointsPerLayer (Matcher_Points_Base.cpp:222-246) or None."""
    gx, gy, gz, lx, ly, lz = map(_f32, (gx, gy, gz, lx, ly, lz))
    T = np.ascontiguousarray(T, dtype=np.float64)
    prm = Pt2PtParams(threshold, thresholdAngularDeg, pairingsPerPoint,
                      int(allowMatchAlreadyMatchedPoints),
                      int(allowMatchAlreadyMatchedGlobalPoints), bbox_eps,
                      multi_search_radius_mode)
    out = np.zeros(max(1, lx.size * pairingsPerPoint), PAIR_PT2PT)
    pot = C.c_uint64(0)
    th = tree._h if tree is not None else None
    lt = local_taken.ctypes.data_as(_u8p) if local_taken is not None else None
    gt = global_taken.ctypes.data_as(_u8p) if global_taken is not None else None
    if threads and threads > 0:
        # the per-query searches spread over threads, the unique-global filter resolved sequentially:
        # the same lists as the sequential loop (K == 1, no visit list)
        assert tree is not None and pairingsPerPoint == 1 and idxs is None
        n = lib().orc_match_pt2pt_mt_ms(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                                        lx.size, _d(T), C.byref(prm), lt, gt, out.ctypes.data, threads)
        return out[:n].copy(), lx.size * pairingsPerPoint
    if idxs is not None:
        ii = np.ascontiguousarray(idxs, dtype=np.uint32)
        n = lib().orc_match_pt2pt_subset(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                                         lx.size, ii.ctypes.data_as(_u32p), ii.size,
                                         _d(T), C.byref(prm), lt, gt, out.ctypes.data, C.byref(pot))
        return out[:n].copy(), pot.value
    n = lib().orc_match_pt2pt(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                              lx.size, _d(T), C.byref(prm), lt, gt, out.ctypes.data,
                              C.byref(pot))
    return out[:n].copy(), pot.value


def match_inlier_ratio(gx, gy, gz, lx, ly, lz, T, inliersRatio, allowMatchAlreadyMatchedPoints=False,
                       allowMatchAlreadyMatchedGlobalPoints=False, bbox_eps=0.20, tree=None,
                       local_taken=None, global_taken=None, idxs=None):
    """Matcher_Points_InlierRatio::implMatchOneLayer.  Returns (pairs, potential); raises when no
    local point has a candidate (the reference's ASSERT_(nTotal > 0))."""
    gx, gy, gz, lx, ly, lz = map(_f32, (gx, gy, gz, lx, ly, lz))
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros(max(1, lx.size), PAIR_PT2PT)
    pot = C.c_uint64(0)
    th = tree._h if tree is not None else None
    lt = local_taken.ctypes.data_as(_u8p) if local_taken is not None else None
    gt = global_taken.ctypes.data_as(_u8p) if global_taken is not None else None
    ii = np.ascontiguousarray(idxs, dtype=np.uint32) if idxs is not None else None
    n = lib().orc_match_inlier_ratio(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz), lx.size,
                                     ii.ctypes.data_as(_u32p) if ii is not None else None,
                                     ii.size if ii is not None else 0, _d(T), float(inliersRatio),
                                     int(allowMatchAlreadyMatchedPoints), int(allowMatchAlreadyMatchedGlobalPoints),
                                     float(bbox_eps), lt, gt, out.ctypes.data, C.byref(pot))
    if n == C.c_size_t(-1).value:
        raise RuntimeError("ASSERT_(nTotal > 0)")
    return out[:n].copy(), pot.value


def adaptive_ci_high(minSq, maxSq, bins, count, confidenceInterval):
    """The restated CHistogram::getHistogramNormalized + confidenceIntervalsFromHistogram upper
    limit (MRPT, un-vendored: parity unpinned)."""
    b = (C.c_uint64 * len(bins))(*[int(v) for v in bins])
    return lib().orc_adaptive_ci_high(float(minSq), float(maxSq), b, len(bins), int(count),
                                      float(confidenceInterval))


def match_adaptive(gx, gy, gz, lx, ly, lz, T, confidenceInterval=0.80, firstToSecondDistanceMax=1.2,
                   absoluteMaxSearchDistance=5.0, minimumCorrDist=0.1, enableDetectPlanes=False,
                   maxPt2PtCorrespondences=1, planeSearchPoints=8, planeMinimumFoundPoints=4,
                   planeMinimumDistance=0.10, planeEigenThreshold=0.01,
                   allowMatchAlreadyMatchedPoints=False, allowMatchAlreadyMatchedGlobalPoints=False,
                   bbox_eps=0.20, tree=None, local_taken=None, global_taken=None, ci_high=None):
    """Matcher_Adaptive::implMatchOneLayer.  ci_high=None: the restated MRPT histogram rule.
    Returns dict(pt2pt, pt2pl, pl_local_idx, potential, ci_high, hist, no_neighbours)."""
    gx, gy, gz, lx, ly, lz = map(_f32, (gx, gy, gz, lx, ly, lz))
    T = np.ascontiguousarray(T, dtype=np.float64)
    prm = AdaptiveParams(confidenceInterval, firstToSecondDistanceMax, absoluteMaxSearchDistance,
                         minimumCorrDist, int(enableDetectPlanes), maxPt2PtCorrespondences,
                         planeSearchPoints, planeMinimumFoundPoints, planeMinimumDistance,
                         planeEigenThreshold, int(allowMatchAlreadyMatchedPoints),
                         int(allowMatchAlreadyMatchedGlobalPoints), bbox_eps)
    o1 = np.zeros(max(1, lx.size * max(1, maxPt2PtCorrespondences)), PAIR_PT2PT)
    o2 = np.zeros(max(1, lx.size), PAIR_PT2PL)
    oi = np.zeros(max(1, lx.size), np.uint32)
    n1, n2, pot = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
    ci = C.c_double(0.0 if ci_high is None else float(ci_high))
    h = AdaptiveHist()
    th = tree._h if tree is not None else None
    lt = local_taken.ctypes.data_as(_u8p) if local_taken is not None else None
    gt = global_taken.ctypes.data_as(_u8p) if global_taken is not None else None
    rc = lib().orc_match_adaptive(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz), lx.size,
                                  _d(T), C.byref(prm), lt, gt, int(ci_high is not None), C.byref(ci),
                                  C.byref(h), o1.ctypes.data, C.byref(n1), o2.ctypes.data,
                                  oi.ctypes.data_as(_u32p), C.byref(n2), C.byref(pot))
    if rc < 0:
        raise ValueError("unsupported Matcher_Adaptive parameters")
    return dict(pt2pt=o1[:n1.value].copy(), pt2pl=o2[:n2.value].copy(), pl_local_idx=oi[:n2.value].copy(),
                potential=pot.value, ci_high=ci.value, no_neighbours=(rc == 1),
                hist=dict(valid=bool(h.valid), minSq=h.minSq, maxSq=h.maxSq, count=h.count,
                          bins=np.array(list(h.bins), np.uint64)))


def match_pt2pl(gx, gy, gz, lx, ly, lz, T, distanceThreshold, searchRadius, knn,
                minimumPlanePoints, planeEigenThreshold, allowMatchAlreadyMatchedPoints=False,
                bbox_eps=0.20, tree=None, local_taken=None, idxs=None, threads=0):
    """Matcher_Point2Plane::implMatchOneLayer with the declared nn_search_pt2pl.
    Returns (pairs, local_idx, potential).  threads > 0: the (independent) queries spread over
    threads, gathered in the sequential loop's order."""
    gx, gy, gz, lx, ly, lz = map(_f32, (gx, gy, gz, lx, ly, lz))
    T = np.ascontiguousarray(T, dtype=np.float64)
    prm = Pt2PlParams(distanceThreshold, searchRadius, knn, minimumPlanePoints,
                      planeEigenThreshold, int(allowMatchAlreadyMatchedPoints), bbox_eps)
    out = np.zeros(max(1, lx.size), PAIR_PT2PL)
    oidx = np.zeros(max(1, lx.size), np.uint32)
    pot = C.c_uint64(0)
    th = tree._h if tree is not None else None
    lt = local_taken.ctypes.data_as(_u8p) if local_taken is not None else None
    if idxs is not None:
        ii = np.ascontiguousarray(idxs, dtype=np.uint32)
        n = lib().orc_match_pt2pl_subset(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                                         lx.size, ii.ctypes.data_as(_u32p), ii.size, _d(T),
                                         C.byref(prm), lt, out.ctypes.data,
                                         oidx.ctypes.data_as(_u32p), C.byref(pot))
        return out[:n].copy(), oidx[:n].copy(), pot.value
    if threads and threads > 0:
        assert tree is not None
        n = lib().orc_match_pt2pl_mt(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                                     lx.size, _d(T), C.byref(prm), lt, out.ctypes.data,
                                     oidx.ctypes.data_as(_u32p), C.byref(pot), threads)
        return out[:n].copy(), oidx[:n].copy(), pot.value
    n = lib().orc_match_pt2pl(th, _f(gx), _f(gy), _f(gz), gx.size, _f(lx), _f(ly), _f(lz),
                              lx.size, _d(T), C.byref(prm), lt, out.ctypes.data,
                              oidx.ctypes.data_as(_u32p), C.byref(pot))
    return out[:n].copy(), oidx[:n].copy(), pot.value


DECIMATE_FIRST_POINT, DECIMATE_CLOSEST_TO_AVERAGE, DECIMATE_VOXEL_AVERAGE = 0, 1, 2


def covariance(pt2pt, pt2pl, pt2ln, pl2pl, T, finDif_xyz=1e-7, finDif_angles=1e-7):
    """mp2p_icp::covariance -> (cov 6x6, H 6x6, ok)"""
    a = np.ascontiguousarray(pt2pt if pt2pt is not None else np.zeros(0, PAIR_PT2PT))
    b = np.ascontiguousarray(pt2pl if pt2pl is not None else np.zeros(0, PAIR_PT2PL))
    c = np.ascontiguousarray(pt2ln if pt2ln is not None else np.zeros(0, PAIR_PT2LN))
    d = np.ascontiguousarray(pl2pl if pl2pl is not None else np.zeros(0, PAIR_PL2PL))
    T = np.ascontiguousarray(T, dtype=np.float64)
    H, cov = np.zeros(36), np.zeros(36)
    ok = lib().orc_covariance(a.ctypes.data, a.size, b.ctypes.data, b.size, c.ctypes.data, c.size,
                              d.ctypes.data, d.size, _d(T), finDif_xyz, finDif_angles, _d(H), _d(cov))
    return cov.reshape(6, 6), H.reshape(6, 6), bool(ok)


def filter_decimate_voxels(x, y, z, resolution, method, flatten_to=None):
    """FilterDecimateVoxels::filter on one layer -> (xyz [m,3], source index [m])"""
    x, y, z = map(_f32, (x, y, z))
    n = x.size
    ox, oy, oz = np.zeros(max(1, n), np.float32), np.zeros(max(1, n), np.float32), np.zeros(max(1, n), np.float32)
    src = np.zeros(max(1, n), np.uint32)
    m = lib().orc_filter_decimate_voxels(_f(x), _f(y), _f(z), n, float(resolution), int(method),
                                         int(flatten_to is not None), float(flatten_to or 0.0),
                                         _f(ox), _f(oy), _f(oz), src.ctypes.data_as(_u32p))
    return np.stack([ox[:m], oy[:m], oz[:m]], 1), src[:m].copy()


def estimate_points_eigen(xs, ys, zs):
    xs, ys, zs = map(_f32, (xs, ys, zs))
    mean = np.zeros(3, np.float32)
    cov, ev, evec = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().orc_estimate_points_eigen(_f(xs), _f(ys), _f(zs), xs.size, _f(mean), _d(cov), _d(ev),
                                    _d(evec))
    return mean, cov.reshape(3, 3), ev, evec.reshape(3, 3)


def make_gn_params(maxIterations, kernel=KERNEL_NONE, kernelParam=1.0, w_pt2pt=1.0, w_pt2pl=1.0,
                   w_pt2ln=1.0, prior_mean=None, prior_cov_inv=None, minDelta=1e-7, maxCost=0.0,
                   weight_blocks=None, reset_weight_cursor_each_iter=0, w_pl2pl=1.0):
    p = GNParams()
    p.maxInnerLoopIterations = maxIterations
    p.minDelta, p.maxCost = minDelta, maxCost
    p.kernel, p.kernelParam = kernel, kernelParam
    p.w_pt2pt, p.w_pt2pl, p.w_pt2ln, p.w_pl2pl = w_pt2pt, w_pt2pl, w_pt2ln, w_pl2pl
    p.has_prior = 0
    keep = []
    if prior_mean is not None:
        p.has_prior = 1
        p.prior_mean = (C.c_double * 12)(*np.asarray(prior_mean, dtype=np.float64))
        p.prior_cov_inv = (C.c_double * 36)(*np.asarray(prior_cov_inv, dtype=np.float64).ravel())
    p.n_weight_blocks = 0
    if weight_blocks:
        cnt = (C.c_size_t * len(weight_blocks))(*[int(c) for c, _ in weight_blocks])
        ww = (C.c_double * len(weight_blocks))(*[float(w) for _, w in weight_blocks])
        keep += [cnt, ww]
        p.n_weight_blocks = len(weight_blocks)
        p.weight_block_count = C.cast(cnt, C.POINTER(C.c_size_t))
        p.weight_block_w = C.cast(ww, C.POINTER(C.c_double))
    p.reset_weight_cursor_each_iter = reset_weight_cursor_each_iter
    p._keep = keep
    return p


def optimal_tf_gauss_newton(pt2pt, pt2pl, pt2ln, T0, prm, threads=0, pl2pl=None):
    """Returns (T, iters, H, g)."""
    pl2pl = np.ascontiguousarray(pl2pl if pl2pl is not None else np.zeros(0, PAIR_PL2PL))
    T0 = np.ascontiguousarray(T0, dtype=np.float64)
    pt2pt = np.ascontiguousarray(pt2pt if pt2pt is not None else np.zeros(0, PAIR_PT2PT))
    pt2pl = np.ascontiguousarray(pt2pl if pt2pl is not None else np.zeros(0, PAIR_PT2PL))
    pt2ln = np.ascontiguousarray(pt2ln if pt2ln is not None else np.zeros(0, PAIR_PT2LN))
    T = np.zeros(12)
    H, g = np.zeros(36), np.zeros(6)
    if threads and threads > 0:
        it = lib().orc_optimal_tf_gauss_newton_mt(pt2pt.ctypes.data, pt2pt.size,
                                                  pt2pl.ctypes.data, pt2pl.size, _d(T0),
                                                  C.byref(prm), _d(T), threads)
        return T, it, None, None
    it = lib().orc_optimal_tf_gauss_newton(pt2pt.ctypes.data, pt2pt.size, pt2pl.ctypes.data,
                                           pt2pl.size, pt2ln.ctypes.data, pt2ln.size,
                                           pl2pl.ctypes.data, pl2pl.size, _d(T0),
                                           C.byref(prm), _d(T), _d(H), _d(g))
    return T, it, H.reshape(6, 6), g


def optimal_tf_horn_wp(pt2pt, pl2pl=None, use_scale_outlier_detector=False, scale_outlier_threshold=1.2,
                       w_pt2pt=1.0, w_ln2ln=1.0, w_pl2pl=1.0, robust_kernel=KERNEL_NONE,
                       robust_kernel_param=1.0, current_estimate=None, point_weights=None):
    """optimal_tf_horn with WeightParameters.  Returns (T, rc, outlier_flags): rc 1 solved, 0 fewer
    than 3 pairings, -1 where the reference throws."""
    pt2pt = np.ascontiguousarray(pt2pt if pt2pt is not None else np.zeros(0, PAIR_PT2PT))
    pl2pl = np.ascontiguousarray(pl2pl if pl2pl is not None else np.zeros(0, PAIR_PL2PL))
    w = HornParams()
    w.use_scale_outlier_detector, w.scale_outlier_threshold = int(use_scale_outlier_detector), scale_outlier_threshold
    w.w_pt2pt, w.w_ln2ln, w.w_pl2pl = w_pt2pt, w_ln2ln, w_pl2pl
    w.robust_kernel, w.robust_kernel_param = int(robust_kernel), robust_kernel_param
    if current_estimate is not None:
        w.has_current_estimate = 1
        w.current_estimate[:] = [float(v) for v in current_estimate]
    keep = None
    if point_weights:
        cnt = (C.c_size_t * len(point_weights))(*[int(c) for c, _ in point_weights])
        ws = (C.c_double * len(point_weights))(*[float(v) for _, v in point_weights])
        w.n_weight_blocks, w.weight_block_count, w.weight_block_w = len(point_weights), cnt, ws
        keep = (cnt, ws)
    T = np.zeros(12)
    flags = np.zeros(max(1, pt2pt.size), np.uint8)
    rc = lib().orc_optimal_tf_horn_wp(pt2pt.ctypes.data, pt2pt.size, pl2pl.ctypes.data, pl2pl.size,
                                      C.byref(w), _d(T), flags.ctypes.data_as(_u8p))
    del keep
    return T, rc, flags[:pt2pt.size]


def pt2ln_pl_to_pt2pt(pt2pl, pt2ln, T):
    """pt2ln_pl_to_pt2pt.cpp:47-113 -> the paired_pt2pt list Solver_Horn then solves on"""
    pt2pl = np.ascontiguousarray(pt2pl if pt2pl is not None else np.zeros(0, PAIR_PT2PL))
    pt2ln = np.ascontiguousarray(pt2ln if pt2ln is not None else np.zeros(0, PAIR_PT2LN))
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros(max(1, pt2pl.size + pt2ln.size), PAIR_PT2PT)
    n = lib().orc_pt2ln_pl_to_pt2pt(pt2pl.ctypes.data, pt2pl.size, pt2ln.ctypes.data, pt2ln.size, _d(T),
                                    out.ctypes.data)
    return out[:n].copy()


def optimal_tf_horn(pt2pt, w_pt2pt=1.0):
    pt2pt = np.ascontiguousarray(pt2pt)
    T = np.zeros(12)
    ok = lib().orc_optimal_tf_horn(pt2pt.ctypes.data, pt2pt.size, w_pt2pt, _d(T))
    return T, bool(ok)
