"""ctypes binding of libmp2p_hip_hostpath.so: the reference-side adapter's per-call host logic
(adapter/mp2p_hip_host.hpp) on plain host containers, behind a C ABI (adapter/hostpath_capi.cpp).

A Session plays the caller's part of ICP::align: begin_iteration() = the fresh MatchState and empty
Pairings of run_matchers (Matcher.cpp:46-66), match_*() = Matcher::match of one plugin matcher against
HOST containers (packed bit-fields in, pair records out), solve_gn() = the plugin's solver handed the
host Pairings.  tests/test_gpu_boundary_hostpath.py checks it against the oracle; bench.py times it
("host_boundary")."""
import ctypes as C

import numpy as np

from . import _build, _lib

_L = None
_P = C.c_void_p
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)

SIGNATURES = {
    "mp2p_hostpath_last_error": (C.c_char_p, []),
    "mp2p_hostpath_open": (_P, [_fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp, C.c_size_t]),
    "mp2p_hostpath_close": (None, [_P]),
    "mp2p_hostpath_begin_iteration": (C.c_int, [_P]),
    "mp2p_hostpath_match_pt2pt": (C.c_int, [_P, _dp, C.POINTER(_lib.Pt2PtParams), C.c_uint32, _P, C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_match_pt2pl": (C.c_int, [_P, _dp, C.POINTER(_lib.Pt2PlParams), C.c_uint32,
                                            C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_quality_paired_ratio": (C.c_int, [_P, _dp, C.POINTER(_lib.Pt2PtParams), C.c_double, _dp, C.POINTER(C.c_int),
                                                     C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_match_inlier_ratio": (C.c_int, [_P, _dp, C.POINTER(_lib.InlierRatioParams), C.c_uint32, _P, C.c_size_t,
                                                   C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_match_adaptive": (C.c_int, [_P, _dp, C.POINTER(_lib.AdaptiveParams), C.c_uint32, C.POINTER(C.c_size_t),
                                               C.POINTER(C.c_size_t), _dp]),
    "mp2p_hostpath_filter_decimate_local": (C.c_int, [_P, C.POINTER(_lib.DecimateParams), _fp, _fp, _fp, _P,
                                                      C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_cache": (C.c_int, [C.c_size_t, C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_release_layers": (C.c_int, [_P]),
    "mp2p_hostpath_solve_gn": (C.c_int, [_P, _dp, C.POINTER(_lib.GNParams), C.POINTER(_lib.GNResult)]),
    "mp2p_hostpath_set_pairings": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "mp2p_hostpath_pairs_pt2pt": (_P, [_P, C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_pairs_pt2pl": (_P, [_P, C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_bits": (C.POINTER(C.c_uint64), [_P, C.c_int, C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_potential": (C.c_uint64, [_P]),
    "mp2p_hostpath_counters": (C.c_int, [C.POINTER(C.c_size_t)]),
    "mp2p_hostpath_last_ms": (None, [_P, _dp]),
    "mp2p_hostpath_invalidate_layers": (None, []),
    "mp2p_hostpath_stage_ms": (None, [_dp]),
    "mp2p_hostpath_set_strict": (None, [C.c_int]),
    "mp2p_hostpath_set_trust_reseen": (None, [C.c_int]),
}


def lib_path():
    return _build.HOSTPATH_LIB


def load():
    global _L
    if _L is not None:
        return _L
    _lib.load()  # the HIP library first (same libamdhip64 as torch)
    if _build.hostpath_needs_build():
        _build.build_hostpath()
    L = C.CDLL(_build.HOSTPATH_LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _L = L
    return L


class HostPathError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise HostPathError(load().mp2p_hostpath_last_error().decode())


def _f(a):
    return a.ctypes.data_as(_fp)


class Session:
    def __init__(self, glob, local):
        self._L = load()
        self._g = [np.ascontiguousarray(glob[:, k], dtype=np.float32) for k in range(3)]
        self._l = [np.ascontiguousarray(local[:, k], dtype=np.float32) for k in range(3)]
        self._n_local = self._l[0].size
        self._h = self._L.mp2p_hostpath_open(_f(self._g[0]), _f(self._g[1]), _f(self._g[2]), self._g[0].size,
                                             _f(self._l[0]), _f(self._l[1]), _f(self._l[2]), self._l[0].size)

    def close(self):
        if self._h:
            self._L.mp2p_hostpath_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin_iteration(self):
        _check(self._L.mp2p_hostpath_begin_iteration(self._h))

    def match_pt2pt(self, pose, prm, icp_iteration=0, visit=None):
        T = np.ascontiguousarray(pose, dtype=np.float64)
        n = C.c_size_t(0)
        v = None if visit is None else np.ascontiguousarray(visit, dtype=np.uint32)
        _check(self._L.mp2p_hostpath_match_pt2pt(self._h, T.ctypes.data_as(_dp), C.byref(prm), int(icp_iteration),
                                                 None if v is None else v.ctypes.data, 0 if v is None else v.size,
                                                 C.byref(n)))
        return n.value

    def match_pt2pl(self, pose, prm, icp_iteration=0):
        T = np.ascontiguousarray(pose, dtype=np.float64)
        n = C.c_size_t(0)
        _check(self._L.mp2p_hostpath_match_pt2pl(self._h, T.ctypes.data_as(_dp), C.byref(prm), int(icp_iteration),
                                                 C.byref(n)))
        return n.value

    def match_inlier_ratio(self, pose, prm, icp_iteration=0, visit=None):
        T = np.ascontiguousarray(pose, dtype=np.float64)
        n = C.c_size_t(0)
        v = None if visit is None else np.ascontiguousarray(visit, dtype=np.uint32)
        _check(self._L.mp2p_hostpath_match_inlier_ratio(self._h, T.ctypes.data_as(_dp), C.byref(prm), int(icp_iteration),
                                                        None if v is None else v.ctypes.data, 0 if v is None else v.size,
                                                        C.byref(n)))
        return n.value

    def match_adaptive(self, pose, prm, icp_iteration=0):
        """-> (point pairs added, plane pairs added, the threshold ci_high)"""
        T = np.ascontiguousarray(pose, dtype=np.float64)
        a, b, ci = C.c_size_t(0), C.c_size_t(0), C.c_double(float("nan"))
        _check(self._L.mp2p_hostpath_match_adaptive(self._h, T.ctypes.data_as(_dp), C.byref(prm), int(icp_iteration),
                                                    C.byref(a), C.byref(b), C.byref(ci)))
        return a.value, b.value, ci.value

    def filter_decimate_local(self, prm):
        """FilterDecimateVoxels over the session's local layer -> (points [m, 3] float32, source indices [m] uint32)"""
        n = self._n_local
        x, y, z = (np.zeros(max(1, n), np.float32) for _ in range(3))
        src = np.zeros(max(1, n), np.uint32)
        m = C.c_size_t(0)
        _check(self._L.mp2p_hostpath_filter_decimate_local(self._h, C.byref(prm), _f(x), _f(y), _f(z), src.ctypes.data,
                                                           C.byref(m)))
        k = m.value
        return np.stack([x[:k], y[:k], z[:k]], 1), src[:k]

    def quality_paired_ratio(self, pose, prm, absolute_minimum_pairing_ratio=0.20):
        """QualityEvaluator_PairedRatio::evaluate with reuse_icp_pairings = false (the plugin's
        mp2p_icp_hip::QualityEvaluator_PairedRatio): -> (quality, hard_discard, pairs)"""
        T = np.ascontiguousarray(pose, dtype=np.float64)
        q, hd, n = C.c_double(0), C.c_int(0), C.c_size_t(0)
        _check(self._L.mp2p_hostpath_quality_paired_ratio(self._h, T.ctypes.data_as(_dp), C.byref(prm),
                                                          float(absolute_minimum_pairing_ratio), C.byref(q), C.byref(hd),
                                                          C.byref(n)))
        return float(q.value), bool(hd.value), int(n.value)

    def release_layers(self):
        _check(self._L.mp2p_hostpath_release_layers(self._h))

    def solve_gn(self, pose0, gn_prm):
        T = np.ascontiguousarray(pose0, dtype=np.float64)
        res = _lib.GNResult()
        _check(self._L.mp2p_hostpath_solve_gn(self._h, T.ctypes.data_as(_dp), C.byref(gn_prm), C.byref(res)))
        return np.array(res.pose), int(res.iterations)

    def set_pairings(self, pt2pt=None, pt2pl=None):
        a = np.ascontiguousarray(pt2pt if pt2pt is not None else np.zeros(0, _lib.PAIR_PT2PT))
        b = np.ascontiguousarray(pt2pl if pt2pl is not None else np.zeros(0, _lib.PAIR_PT2PL))
        _check(self._L.mp2p_hostpath_set_pairings(self._h, a.ctypes.data, a.size, b.ctypes.data, b.size))

    def pairs_pt2pt(self):
        n = C.c_size_t(0)
        p = self._L.mp2p_hostpath_pairs_pt2pt(self._h, C.byref(n))
        if not n.value:
            return np.zeros(0, _lib.PAIR_PT2PT)
        buf = (C.c_char * (n.value * _lib.PAIR_PT2PT.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=_lib.PAIR_PT2PT, count=n.value).copy()

    def pairs_pt2pl(self):
        n = C.c_size_t(0)
        p = self._L.mp2p_hostpath_pairs_pt2pl(self._h, C.byref(n))
        if not n.value:
            return np.zeros(0, _lib.PAIR_PT2PL)
        buf = (C.c_char * (n.value * _lib.PAIR_PT2PL.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=_lib.PAIR_PT2PL, count=n.value).copy()

    def bits(self, which):
        """the packed MatchState field (0 = global, 1 = local) as a bool array"""
        nb = C.c_size_t(0)
        p = self._L.mp2p_hostpath_bits(self._h, int(which), C.byref(nb))
        nw = (nb.value + 63) // 64
        w = np.ctypeslib.as_array(p, shape=(max(nw, 1),))[:nw].copy()
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:nb.value].astype(bool)

    def set_bits(self, which, mask):
        nb = C.c_size_t(0)
        p = self._L.mp2p_hostpath_bits(self._h, int(which), C.byref(nb))
        nw = (nb.value + 63) // 64
        w = np.ctypeslib.as_array(p, shape=(max(nw, 1),))
        packed = np.packbits(np.asarray(mask, dtype=bool), bitorder="little")
        packed = np.concatenate([packed, np.zeros(nw * 8 - packed.size, np.uint8)])
        w[:nw] = packed.view(np.uint64)

    @property
    def potential_pairings(self):
        return int(self._L.mp2p_hostpath_potential(self._h))

    def last_ms(self):
        out = (C.c_double * 2)()
        self._L.mp2p_hostpath_last_ms(self._h, out)
        return float(out[0]), float(out[1])


def counters():
    out = (C.c_size_t * 4)()
    _check(load().mp2p_hostpath_counters(out))
    return dict(map_uploads=out[0], cloud_uploads=out[1], mstate_uploads=out[2], pairings_uploads=out[3])


def cache(max_layers=0):
    """the layer cache of this thread's runtime: dict(layers, bytes, evictions, full_checks, reseen_checks)"""
    out = (C.c_size_t * 5)()
    _check(load().mp2p_hostpath_cache(int(max_layers), out))
    return dict(zip(("layers", "bytes", "evictions", "full_checks", "reseen_checks"), [int(v) for v in out]))


def stage_ms():
    """wall time [ms] of the stages of the last matcher call: layers + MatchState in, device work until the
    list length is known, container resize, pair copy-out (the whole window), marks (run INSIDE that window,
    on the index arrays, while the records are on the link), list fingerprint"""
    out = (C.c_double * 6)()
    load().mp2p_hostpath_stage_ms(out)
    return dict(zip(("state_in", "device", "resize", "copy_out_window", "marks_inside_window", "fingerprint"),
                    [float(v) for v in out]))


def set_strict(on):
    """every solver call uploads the host Pairings (the plugin's MP2P_HIP_HOST_STRICT=1)"""
    load().mp2p_hostpath_set_strict(int(bool(on)))


def set_trust_reseen(on):
    """re-seen layers are re-verified on every 61st point instead of hashed in full at ICP iteration 0
    (the plugin's MP2P_HIP_HOST_TRUST_RESEEN=1; off by default: ADVICE r3)"""
    load().mp2p_hostpath_set_trust_reseen(int(bool(on)))


def invalidate_layers():
    load().mp2p_hostpath_invalidate_layers()
