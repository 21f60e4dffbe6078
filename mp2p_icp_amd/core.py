"""Thin object wrappers over the C ABI handles (context, map index, local cloud, MatchState,
device-resident Pairings).  No numerics here: everything is computed by libmp2p_hip.so."""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import check


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pose(T):
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1)
    if T.size != 12:
        raise ValueError("pose must be 12 doubles: R row-major (9) + t (3)")
    return T


def _free_child(ctx, free_fn, handle):
    """finalizer of a map / cloud / MatchState / Pairings handle.  It holds the Context OBJECT, not its raw handle: the
    garbage collector finalizes unreachable objects in no particular order, and a context destroyed before one of its
    children made the child's free call run on freed memory (one aborted test run in round 3)."""
    free_fn(ctx.handle, handle)


class Context:
    """One HIP stream on one device.  stream=None: the context creates its own (non-blocking)
    stream.  Otherwise `stream` is a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that torch ops and torch.distributed collectives
    are ordered with the library's kernels.  torch's DEFAULT stream has the raw value 0 (the
    null stream): it is passed on as hipStreamLegacy, never silently replaced by a private
    stream -- that would race with torch."""
    HIP_STREAM_LEGACY = 1  # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)

    def __init__(self, device=0, stream=None):
        self._L = _lib.load()
        h = C.c_void_p()
        if stream is not None and int(stream) == 0:
            stream = self.HIP_STREAM_LEGACY
        check(self._L.mp2p_hip_ctx_create(int(device), C.c_void_p(int(stream)) if stream is not None else None,
                                          C.byref(h)))
        self._h = h
        self.device = int(device)
        self._fin = weakref.finalize(self, self._L.mp2p_hip_ctx_destroy, h)

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(self._L.mp2p_hip_sync(self._h), self._h)

    def set_profiling(self, on):
        check(self._L.mp2p_hip_set_profiling(self._h, int(on)), self._h)

    def set_tune(self, settings):
        """measurement knobs at run time ("name=value,..." as in MP2P_HIP_TUNE; csrc/common.hpp)"""
        check(self._L.mp2p_hip_set_tune(self._h, str(settings).encode()), self._h)

    def stats(self):
        s = _lib.Stats()
        check(self._L.mp2p_hip_get_stats(self._h, C.byref(s)), self._h)
        d = {k: getattr(s, k) for k, _ in s._fields_}
        d["nn_tile_ticks_hist"] = list(s.nn_tile_ticks_hist)
        d["nn_wave_phase_ticks"] = list(s.nn_wave_phase_ticks)
        return d

    def local_bbox_ptr(self):
        return self._L.mp2p_hip_ctx_local_bbox_ptr(self._h)

    def gn_sums_ptr(self):
        return self._L.mp2p_hip_gn_sums_ptr(self._h)


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class GlobalMap:
    """NN index of one global point layer (replaces the layer's NearestNeighborsCapable)."""

    def __init__(self, ctx, x, y, z, cell_size=0.0, target_per_cell=0.0, max_levels=0,
                 device_ptrs=False, no_occupancy_bitmap=False):
        self.ctx = ctx
        L = ctx._L
        prm = _lib.MapParams(cell_size, target_per_cell, max_levels, int(no_occupancy_bitmap))
        h = C.c_void_p()
        if device_ptrs:
            n = int(device_ptrs)
            check(L.mp2p_hip_map_upload_device(ctx.handle, x, y, z, n, C.byref(prm), C.byref(h)),
                  ctx.handle)
            self.n = n
        else:
            x, y, z = _f32(x), _f32(y), _f32(z)
            assert x.size == y.size == z.size
            check(L.mp2p_hip_map_upload(ctx.handle, _fp(x), _fp(y), _fp(z), x.size, C.byref(prm),
                                        C.byref(h)), ctx.handle)
            self.n = x.size
        self._h = h
        self._fin = weakref.finalize(self, _free_child, ctx, L.mp2p_hip_map_free, h)

    @property
    def handle(self):
        return self._h

    def info(self):
        i = _lib.MapInfo()
        check(self.ctx._L.mp2p_hip_map_get_info(self.ctx.handle, self._h, C.byref(i)),
              self.ctx.handle)
        d = {k: getattr(i, k) for k, _ in i._fields_}
        d["bbox_min"] = list(i.bbox_min)
        d["bbox_max"] = list(i.bbox_max)
        return d

    def claims_ptr(self):
        return self.ctx._L.mp2p_hip_map_claims_ptr(self._h)


class LocalCloud:
    def __init__(self, ctx, x, y, z, device_ptrs=False):
        self.ctx = ctx
        L = ctx._L
        h = C.c_void_p()
        if device_ptrs:
            n = int(device_ptrs)
            check(L.mp2p_hip_cloud_upload_device(ctx.handle, x, y, z, n, C.byref(h)), ctx.handle)
            self.n = n
        else:
            x, y, z = _f32(x), _f32(y), _f32(z)
            assert x.size == y.size == z.size
            check(L.mp2p_hip_cloud_upload(ctx.handle, _fp(x), _fp(y), _fp(z), x.size, C.byref(h)),
                  ctx.handle)
            self.n = x.size
        self._h = h
        self._fin = weakref.finalize(self, _free_child, ctx, L.mp2p_hip_cloud_free, h)

    @property
    def handle(self):
        return self._h

    def set_visit_order(self, order):
        """order: distinct original indices, visited in that order (None/empty: all, ascending)"""
        if order is None or len(order) == 0:
            check(self.ctx._L.mp2p_hip_cloud_set_visit_order(self.ctx.handle, self._h, None, 0),
                  self.ctx.handle)
            self.n_visit = 0
            return
        o = np.ascontiguousarray(order, dtype=np.uint32)
        check(self.ctx._L.mp2p_hip_cloud_set_visit_order(self.ctx.handle, self._h, o.ctypes.data, o.size),
              self.ctx.handle)
        self.n_visit = int(o.size)


class DeviceMatchState:
    def __init__(self, ctx, n_global, n_local):
        self.ctx, self.n_global, self.n_local = ctx, n_global, n_local
        h = C.c_void_p()
        check(ctx._L.mp2p_hip_mstate_create(ctx.handle, n_global, n_local, C.byref(h)), ctx.handle)
        self._h = h
        self._fin = weakref.finalize(self, _free_child, ctx, ctx._L.mp2p_hip_mstate_free, h)

    @property
    def handle(self):
        return self._h

    def reset(self):
        check(self.ctx._L.mp2p_hip_mstate_reset(self.ctx.handle, self._h), self.ctx.handle)

    def download(self):
        g = np.zeros(max(1, self.n_global), np.uint8)
        l = np.zeros(max(1, self.n_local), np.uint8)
        u8 = C.POINTER(C.c_uint8)
        check(self.ctx._L.mp2p_hip_mstate_download(self.ctx.handle, self._h, g.ctypes.data_as(u8),
                                                   l.ctypes.data_as(u8)), self.ctx.handle)
        return g[:self.n_global], l[:self.n_local]

    def upload(self, global_taken=None, local_taken=None):
        u8 = C.POINTER(C.c_uint8)
        g = np.ascontiguousarray(global_taken, np.uint8) if global_taken is not None else None
        l = np.ascontiguousarray(local_taken, np.uint8) if local_taken is not None else None
        check(self.ctx._L.mp2p_hip_mstate_upload(
            self.ctx.handle, self._h, g.ctypes.data_as(u8) if g is not None else None,
            l.ctypes.data_as(u8) if l is not None else None), self.ctx.handle)


class DevicePairs:
    """Device-resident mp2p_icp::Pairings (pt2pt + pt2pl lists, potential_pairings)."""

    def __init__(self, ctx, cap_pt2pt, cap_pt2pl=0):
        self.ctx, self.cap_pt2pt, self.cap_pt2pl = ctx, int(cap_pt2pt), int(cap_pt2pl)
        h = C.c_void_p()
        check(ctx._L.mp2p_hip_pairs_create(ctx.handle, self.cap_pt2pt, self.cap_pt2pl, C.byref(h)),
              ctx.handle)
        self._h = h
        self._fin = weakref.finalize(self, _free_child, ctx, ctx._L.mp2p_hip_pairs_free, h)

    @property
    def handle(self):
        return self._h

    def clear(self):
        check(self.ctx._L.mp2p_hip_pairs_clear(self.ctx.handle, self._h), self.ctx.handle)

    def reserve(self, cap_pt2pt, cap_pt2pl):
        check(self.ctx._L.mp2p_hip_pairs_reserve(self.ctx.handle, self._h, int(cap_pt2pt),
                                                 int(cap_pt2pl)), self.ctx.handle)
        self.cap_pt2pt = max(self.cap_pt2pt, int(cap_pt2pt))
        self.cap_pt2pl = max(self.cap_pt2pl, int(cap_pt2pl))

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self.ctx._L.mp2p_hip_pairs_counts(self.ctx.handle, self._h, C.byref(a), C.byref(b),
                                                C.byref(c)), self.ctx.handle)
        return a.value, b.value, c.value

    def download_pt2pt(self):
        n, _, _ = self.counts()
        out = np.zeros(max(1, n), _lib.PAIR_PT2PT)
        got = C.c_size_t()
        check(self.ctx._L.mp2p_hip_pairs_download_pt2pt(self.ctx.handle, self._h, out.ctypes.data,
                                                        out.size, C.byref(got)), self.ctx.handle)
        return out[:got.value]

    def download_pt2pl(self):
        _, n, _ = self.counts()
        out = np.zeros(max(1, n), _lib.PAIR_PT2PL)
        idx = np.zeros(max(1, n), np.uint32)
        got = C.c_size_t()
        check(self.ctx._L.mp2p_hip_pairs_download_pt2pl(
            self.ctx.handle, self._h, out.ctypes.data, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
            out.size, C.byref(got)), self.ctx.handle)
        return out[:got.value], idx[:got.value]

    def upload(self, pt2pt=None, pt2pl=None):
        a = np.ascontiguousarray(pt2pt, _lib.PAIR_PT2PT) if pt2pt is not None else np.zeros(0, _lib.PAIR_PT2PT)
        b = np.ascontiguousarray(pt2pl, _lib.PAIR_PT2PL) if pt2pl is not None else np.zeros(0, _lib.PAIR_PT2PL)
        check(self.ctx._L.mp2p_hip_pairs_upload(self.ctx.handle, self._h,
                                                a.ctypes.data if a.size else None, a.size,
                                                b.ctypes.data if b.size else None, b.size),
              self.ctx.handle)


def _pairs_upload_lines_planes(self, pt2ln=None, pl2pl=None):
    """paired_pt2ln / paired_pl2pl (host-produced) for the Gauss-Newton solver; replaces both"""
    a = np.ascontiguousarray(pt2ln, _lib.PAIR_PT2LN) if pt2ln is not None else np.zeros(0, _lib.PAIR_PT2LN)
    b = np.ascontiguousarray(pl2pl, _lib.PAIR_PL2PL) if pl2pl is not None else np.zeros(0, _lib.PAIR_PL2PL)
    check(self.ctx._L.mp2p_hip_pairs_upload_lines_planes(self.ctx.handle, self._h,
                                                         a.ctypes.data if a.size else None, a.size,
                                                         b.ctypes.data if b.size else None, b.size),
          self.ctx.handle)


def _pairs_counts_lines_planes(self):
    a, b = C.c_uint64(), C.c_uint64()
    check(self.ctx._L.mp2p_hip_pairs_counts_lines_planes(self.ctx.handle, self._h, C.byref(a), C.byref(b)),
          self.ctx.handle)
    return a.value, b.value


DevicePairs.upload_lines_planes = _pairs_upload_lines_planes
DevicePairs.counts_lines_planes = _pairs_counts_lines_planes


def covariance(ctx, pairs, pose, finDif_xyz=1e-7, finDif_angles=1e-7):
    """mp2p_icp::covariance on device-resident pairings -> (cov 6x6, H 6x6, positive definite?)"""
    T = _pose(pose)
    H, cov = np.zeros(36), np.zeros(36)
    pd = C.c_int32()
    dp = C.POINTER(C.c_double)
    check(ctx._L.mp2p_hip_covariance(ctx.handle, pairs.handle, T.ctypes.data_as(dp), finDif_xyz,
                                     finDif_angles, H.ctypes.data_as(dp), cov.ctypes.data_as(dp),
                                     C.byref(pd)), ctx.handle)
    return cov.reshape(6, 6), H.reshape(6, 6), bool(pd.value)


def filter_decimate_voxels(ctx, x, y, z, resolution, method, flatten_to=None):
    """-> (xyz [m,3] float32, source index [m] uint32; 0xFFFFFFFF where the point is an average)"""
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = x.size
    prm = _lib.DecimateParams(float(resolution), int(method), int(flatten_to is not None),
                              float(flatten_to or 0.0))
    ox, oy, oz = (np.zeros(max(1, n), np.float32) for _ in range(3))
    src = np.zeros(max(1, n), np.uint32)
    m = C.c_size_t()
    check(ctx._L.mp2p_hip_filter_decimate_voxels(ctx.handle, _fp(x), _fp(y), _fp(z), n, C.byref(prm),
                                                 _fp(ox), _fp(oy), _fp(oz), src.ctypes.data, C.byref(m)),
          ctx.handle)
    return np.stack([ox[:m.value], oy[:m.value], oz[:m.value]], 1), src[:m.value].copy()


def match_pt2pt(ctx, gmap, cloud, pose, prm, mstate, pairs):
    T = _pose(pose)
    check(ctx._L.mp2p_hip_match_pt2pt(ctx.handle, gmap.handle, cloud.handle,
                                      T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                      mstate.handle if mstate is not None else None,
                                      pairs.handle), ctx.handle)


def match_pt2pt_phase1(ctx, gmap, cloud, pose, prm, mstate):
    T = _pose(pose)
    check(ctx._L.mp2p_hip_match_pt2pt_phase1(ctx.handle, gmap.handle, cloud.handle,
                                             T.ctypes.data_as(C.POINTER(C.c_double)),
                                             C.byref(prm),
                                             mstate.handle if mstate is not None else None),
          ctx.handle)


def match_pt2pt_phase2(ctx, gmap, cloud, prm, mstate, pairs):
    check(ctx._L.mp2p_hip_match_pt2pt_phase2(ctx.handle, gmap.handle, cloud.handle, C.byref(prm),
                                             mstate.handle if mstate is not None else None,
                                             pairs.handle), ctx.handle)


def exchange_pack(ctx, gmap, cloud, prm):
    """-> (device pointer of exch double[8], device pointer of the claim record list, its length)"""
    e, l, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    check(ctx._L.mp2p_hip_exchange_pack(ctx.handle, gmap.handle, cloud.handle, C.byref(prm),
                                        C.byref(e), C.byref(l), C.byref(n)), ctx.handle)
    return e.value, l.value, n.value


def exchange_unpack(ctx, gmap, gathered_ptr, n_records):
    check(ctx._L.mp2p_hip_exchange_unpack(ctx.handle, gmap.handle, gathered_ptr, n_records), ctx.handle)


def match_inlier_ratio(ctx, gmap, cloud, pose, prm, mstate, pairs):
    T = _pose(pose)
    check(ctx._L.mp2p_hip_match_inlier_ratio(ctx.handle, gmap.handle, cloud.handle,
                                             T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                             mstate.handle if mstate is not None else None,
                                             pairs.handle), ctx.handle)


def adaptive_search(ctx, gmap, cloud, pose, prm, mstate):
    """Matcher_Adaptive steps 1+2: neighbour lists (kept on the context) -> _lib.AdaptiveHist"""
    T = _pose(pose)
    h = _lib.AdaptiveHist()
    check(ctx._L.mp2p_hip_adaptive_search(ctx.handle, gmap.handle, cloud.handle,
                                          T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                          mstate.handle if mstate is not None else None, C.byref(h)),
          ctx.handle)
    return h


def adaptive_ci_high(ctx, hist, confidenceInterval):
    """the restated MRPT confidence limit (parity unpinned: see include/mp2p_hip.h)"""
    return float(ctx._L.mp2p_hip_adaptive_ci_high(C.byref(hist), float(confidenceInterval)))


def adaptive_select(ctx, gmap, cloud, prm, ci_high, mstate, pairs):
    check(ctx._L.mp2p_hip_adaptive_select(ctx.handle, gmap.handle, cloud.handle, C.byref(prm),
                                          float(ci_high), mstate.handle if mstate is not None else None,
                                          pairs.handle), ctx.handle)


def match_adaptive(ctx, gmap, cloud, pose, prm, mstate, pairs):
    """-> (ci_high, AdaptiveHist)"""
    T = _pose(pose)
    h = _lib.AdaptiveHist()
    ci = C.c_double(0.0)
    check(ctx._L.mp2p_hip_match_adaptive(ctx.handle, gmap.handle, cloud.handle,
                                         T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                         mstate.handle if mstate is not None else None, pairs.handle,
                                         C.byref(ci), C.byref(h)), ctx.handle)
    return ci.value, h


def match_pt2pl(ctx, gmap, cloud, pose, prm, mstate, pairs):
    T = _pose(pose)
    check(ctx._L.mp2p_hip_match_pt2pl(ctx.handle, gmap.handle, cloud.handle,
                                      T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                      mstate.handle if mstate is not None else None,
                                      pairs.handle), ctx.handle)


def gn_solve(ctx, pairs, pose0, prm):
    T = _pose(pose0)
    res = _lib.GNResult()
    check(ctx._L.mp2p_hip_gn_solve(ctx.handle, pairs.handle,
                                   T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm),
                                   C.byref(res)), ctx.handle)
    return res


def gn_result_to_dict(res):
    return dict(pose=np.array(res.pose), H=np.array(res.H).reshape(6, 6), g=np.array(res.g),
                cost=res.cost, iterations=res.iterations)


def timeline(ctx):
    """profiling level 4: ({start,end} ticks [n_tiles,2], [n_single_blocks,2]) of the last match call"""
    a, b = C.c_size_t(0), C.c_size_t(0)
    check(ctx._L.mp2p_hip_get_timeline(ctx.handle, None, 0, C.byref(a), C.byref(b)), ctx.handle)
    n = a.value + b.value
    buf = np.zeros((max(1, n), 2), np.uint64)
    if n:
        check(ctx._L.mp2p_hip_get_timeline(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(a),
                                           C.byref(b)), ctx.handle)
    return buf[:a.value], buf[a.value:n]


def horn_solve_wp(ctx, pairs, prm, point_weights=None):
    """optimal_tf_horn with WeightParameters (prm: _lib.HornParams) -> (pose, solved, n_outliers)"""
    keep = None
    prm.n_weight_blocks = 0
    if point_weights:
        cnt = (C.c_size_t * len(point_weights))(*[int(c) for c, _ in point_weights])
        ws = (C.c_double * len(point_weights))(*[float(w) for _, w in point_weights])
        prm.n_weight_blocks, prm.weight_block_count, prm.weight_block_w = len(point_weights), cnt, ws
        keep = (cnt, ws)
    res = _lib.HornResult()
    check(ctx._L.mp2p_hip_horn_solve_wp(ctx.handle, pairs.handle, C.byref(prm), C.byref(res)), ctx.handle)
    del keep
    return np.array(res.pose), bool(res.solved), int(res.n_outliers)


def horn_outlier_flags(ctx, n):
    """one byte per point pairing of the last Horn call: 1 = scale outlier"""
    out = np.zeros(max(1, n), np.uint8)
    check(ctx._L.mp2p_hip_horn_outlier_flags(ctx.handle, out.ctypes.data_as(C.POINTER(C.c_uint8)), int(n)),
          ctx.handle)
    return out[:n]


def pairs_pt2ln_pl_to_pt2pt(ctx, pairs_in, guess, pairs_out):
    """pt2ln_pl_to_pt2pt.cpp:47-113: pairs_out (cleared, another handle) receives the converted list"""
    T = _pose(guess)
    check(ctx._L.mp2p_hip_pairs_pt2ln_pl_to_pt2pt(ctx.handle, pairs_in.handle,
                                                  T.ctypes.data_as(C.POINTER(C.c_double)), pairs_out.handle),
          ctx.handle)


def horn_solve(ctx, pairs, w_pt2pt=1.0):
    T = np.zeros(12)
    ok = C.c_int32(0)
    check(ctx._L.mp2p_hip_horn_solve(ctx.handle, pairs.handle, float(w_pt2pt),
                                     T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ok)),
          ctx.handle)
    return T, bool(ok.value)
