"""Host-side mirror of the reference's Matcher plugin interface for the hot path
(same class names, YAML keys, argument meaning and error behaviour), calling the HIP library.

  Matcher                          mp2p_icp/src/Matcher.cpp:28-44, Matcher.h:80-109
  run_matchers                     Matcher.cpp:46-88
  MatchState / MatchContext        Matcher.h:30-70
  Matcher_Points_Base              Matcher_Points_Base.cpp:30-181
  Matcher_Points_DistanceThreshold Matcher_Points_DistanceThreshold.cpp:39-269
  Matcher_Point2Plane              Matcher_Point2Plane.cpp:35-114
  Matcher_Points_InlierRatio       Matcher_Points_InlierRatio.cpp:28-143
  Matcher_Adaptive                 Matcher_Adaptive.cpp:32-314
  Pairings                         Pairings.h:84-169, Pairings.cpp:123-147
"""
import math

import numpy as np

from . import _lib, core
from .metric_map import metric_map_t  # noqa: F401
from .parameterizable import Parameterizable


class Pairings:
    """mp2p_icp::Pairings.  The lists live in HBM (core.DevicePairs) so that a solver can
    consume them without a PCIe round trip; `paired_pt2pt` / `paired_pt2pl` materialise the
    host containers (byte-compatible with TMatchingPair / point_plane_pair_t) on demand."""

    def __init__(self, ctx=None, capacity_pt2pt=0, capacity_pt2pl=0):
        self.ctx = ctx
        self._dev = None
        self._cap = (int(capacity_pt2pt), int(capacity_pt2pl))
        self.point_weights = []  # [(count, weight)]  Pairings.h:111
        self._host_pt2pt = None
        self._host_pt2pl = None
        self._host_pl_idx = None
        self._ub = [0, 0]  # host-side upper bounds of the list lengths (no sync needed)
        # lists no matcher of this path produces (Pairings.h:94-99); the Gauss-Newton solver
        # uploads paired_pt2ln / paired_pl2pl, paired_ln2ln is not supported
        self.paired_pt2ln = np.zeros(0, _lib.PAIR_PT2LN)
        self.paired_pl2pl = np.zeros(0, _lib.PAIR_PL2PL)
        self.paired_ln2ln = []

    # -- device side -----------------------------------------------------------------------
    def _ensure_dev(self, ctx, cap_pt2pt, cap_pt2pl):
        if self._dev is None or self.ctx is not ctx:
            self.ctx = ctx
            self._dev = core.DevicePairs(ctx, max(cap_pt2pt, self._cap[0], 1),
                                         max(cap_pt2pl, self._cap[1], 0))
        elif self._dev.cap_pt2pt < cap_pt2pt or self._dev.cap_pt2pl < cap_pt2pl:
            self._dev.reserve(max(cap_pt2pt, self._dev.cap_pt2pt), max(cap_pt2pl, self._dev.cap_pt2pl))
        self._invalidate()
        return self._dev

    def _invalidate(self):
        self._host_pt2pt = self._host_pt2pl = self._host_pl_idx = None

    @property
    def device(self):
        return self._dev

    # -- reference interface -----------------------------------------------------------------
    @property
    def paired_pt2pt(self):
        if self._host_pt2pt is None:
            self._host_pt2pt = (self._dev.download_pt2pt() if self._dev is not None
                                else np.zeros(0, _lib.PAIR_PT2PT))
        return self._host_pt2pt

    @property
    def paired_pt2pl(self):
        if self._host_pt2pl is None:
            if self._dev is not None:
                self._host_pt2pl, self._host_pl_idx = self._dev.download_pt2pl()
            else:
                self._host_pt2pl = np.zeros(0, _lib.PAIR_PT2PL)
                self._host_pl_idx = np.zeros(0, np.uint32)
        return self._host_pt2pl

    @property
    def paired_pt2pl_local_idx(self):
        self.paired_pt2pl
        return self._host_pl_idx

    @property
    def potential_pairings(self):
        return self._dev.counts()[2] if self._dev is not None else 0

    def size(self):  # Pairings.cpp:143-147
        n = len(self.paired_pt2ln) + len(self.paired_pl2pl) + len(self.paired_ln2ln)
        if self._dev is None:
            return n
        a, b, _ = self._dev.counts()
        return a + b + n

    def empty(self):
        return self.size() == 0

    def contents_summary(self):
        a, b, _ = self._dev.counts() if self._dev is not None else (0, 0, 0)
        return f"{a} point-point, {b} point-plane" if (a or b) else "none"

    @staticmethod
    def from_host(ctx, pt2pt=None, pt2pl=None, point_weights=None, pt2ln=None, pl2pl=None):
        """Build a Pairings from host lists (a solver handed pairings it did not produce)."""
        n1 = 0 if pt2pt is None else len(pt2pt)
        n2 = 0 if pt2pl is None else len(pt2pl)
        p = Pairings(ctx, max(n1, 1), n2)
        p._ensure_dev(ctx, max(n1, 1), n2)
        p._ub = [n1, n2]
        p._dev.upload(pt2pt, pt2pl)
        p.point_weights = list(point_weights or [])
        if pt2ln is not None:
            p.paired_pt2ln = np.ascontiguousarray(pt2ln, _lib.PAIR_PT2LN)
        if pl2pl is not None:
            p.paired_pl2pl = np.ascontiguousarray(pl2pl, _lib.PAIR_PL2PL)
        return p


class MatchContext:  # Matcher.h:30-41
    def __init__(self, icpIteration=0):
        self.icpIteration = icpIteration


class MatchState:
    """Matcher.h:44-70: which global / local points are already paired, per layer."""

    def __init__(self, pcGlobal, pcLocal, ctx=None):
        self.ctx = ctx or core.default_context()
        self.pcGlobal, self.pcLocal = pcGlobal, pcLocal
        self._dev = {}

    def for_layers(self, gname, lname):
        key = (gname, lname)
        if key not in self._dev:
            # one device object per (global layer, local layer) pair carries both bit-fields;
            # bits of the same layer are shared between pairs through an upload when needed
            g, l = self.pcGlobal.layers[gname], self.pcLocal.layers[lname]
            ms = core.DeviceMatchState(self.ctx, g.size(), l.size())
            for (og, ol), other in self._dev.items():
                if og == gname or ol == lname:
                    gt, lt = other.download()
                    ms.upload(gt if og == gname else None, lt if ol == lname else None)
            self._dev[key] = ms
        return self._dev[key]

    def _sync_shared(self, gname, lname):
        """propagate marks to other (global,local) combinations sharing a layer"""
        if len(self._dev) <= 1:
            return
        src = self._dev[(gname, lname)]
        gt, lt = src.download()
        for (og, ol), other in self._dev.items():
            if (og, ol) == (gname, lname):
                continue
            if og == gname or ol == lname:
                other.upload(gt if og == gname else None, lt if ol == lname else None)


class Matcher(Parameterizable):
    """Matcher.cpp:28-44"""

    def __init__(self):
        super().__init__()
        self.runFromIteration = 0
        self.runUpToIteration = 0
        self.enabled = True
        self.ctx = None

    def initialize(self, params):
        params = params or {}
        self.runFromIteration = int(params.get("runFromIteration", self.runFromIteration))
        self.runUpToIteration = int(params.get("runUpToIteration", self.runUpToIteration))
        self.enabled = bool(params.get("enabled", self.enabled))

    def match(self, pcGlobal, pcLocal, localPose, mc, ms, out):
        mc = mc or MatchContext()
        if not self.enabled:
            return False
        if mc.icpIteration < self.runFromIteration:
            return False
        if self.runUpToIteration > 0 and mc.icpIteration > self.runUpToIteration:
            return False
        return self.impl_match(pcGlobal, pcLocal, localPose, mc, ms, out)


class Matcher_Points_Base(Matcher):
    """Matcher_Points_Base.cpp:30-181 (layer loop, parameters)."""

    def __init__(self):
        super().__init__()
        self.weight_pt2pt_layers = {}  # global -> {local: weight}
        self.maxLocalPointsPerLayer_ = 0
        self.localPointsSampleSeed_ = 0
        self.allowMatchAlreadyMatchedPoints_ = False
        self.allowMatchAlreadyMatchedGlobalPoints_ = False
        self.kdtree_leaf_max_points_ = None
        self.bounding_box_intersection_check_epsilon_ = 0.20
        # tuning of the HIP search (not reference parameters)
        self.initial_radius_cells = 0.0
        self.queries_per_wave = 0
        self.group_radius_factor = 0.0
        self.cell_budget = 0
        self.defer_radius_cells = 0.0
        self.disable_warm_start = False
        self.tile_order = False
        self.multi_search_radius_mode = True
        self.brick_budget = 0

    def initialize(self, params):
        super().initialize(params)
        params = params or {}
        if "pointLayerMatches" in params:
            self.weight_pt2pt_layers = {}
            seq = params["pointLayerMatches"]
            assert isinstance(seq, (list, tuple)), "pointLayerMatches must be a sequence"
            for e in seq:
                assert "global" in e and "local" in e
                self.weight_pt2pt_layers.setdefault(e["global"], {})[e["local"]] = float(
                    e.get("weight", 1.0))
        self.maxLocalPointsPerLayer_ = int(params.get("maxLocalPointsPerLayer", 0))
        self.localPointsSampleSeed_ = int(params.get("localPointsSampleSeed", 0))
        self.allowMatchAlreadyMatchedPoints_ = bool(
            params.get("allowMatchAlreadyMatchedPoints", self.allowMatchAlreadyMatchedPoints_))
        self.allowMatchAlreadyMatchedGlobalPoints_ = bool(
            params.get("allowMatchAlreadyMatchedGlobalPoints",
                       self.allowMatchAlreadyMatchedGlobalPoints_))
        v = int(params.get("kdtree_leaf_max_points", 0))
        if v > 0:
            self.kdtree_leaf_max_points_ = v  # accepted and ignored: there is no KD-tree
        self.bounding_box_intersection_check_epsilon_ = float(
            params.get("bounding_box_intersection_check_epsilon",
                       self.bounding_box_intersection_check_epsilon_))
        self.initial_radius_cells = float(params.get("hip_initial_radius_cells", 0.0))
        self.queries_per_wave = int(params.get("hip_queries_per_wave", 0))
        self.group_radius_factor = float(params.get("hip_group_radius_factor", 0.0))
        self.cell_budget = int(params.get("hip_cell_budget", 0))
        self.defer_radius_cells = float(params.get("hip_defer_radius_cells", 0.0))
        self.disable_warm_start = bool(params.get("hip_disable_warm_start", False))
        self.tile_order = bool(params.get("hip_tile_order", False))
        # pairingsPerPoint > 1: the shipped (TBB) build's nn_radius_search meaning by default
        # (Matcher_Points_DistanceThreshold.cpp:172-177); False = the sequential build's nn_multiple_search
        self.multi_search_radius_mode = bool(params.get("hip_multi_search_radius_mode", True))
        self.brick_budget = int(params.get("hip_brick_budget", 0))

    # maxLocalPointsPerLayer (Matcher_Points_Base.cpp:222-246): when the local layer is larger,
    # only idxs[0..maxLocalPoints) are visited, where idxs = iota(0..maxLocalPoints) shuffled by
    # mrpt::random::partial_shuffle with std::default_random_engine(seed) (seed 0 = the clock).
    # The result is a permutation of the FIRST maxLocalPoints indices: it changes the visiting
    # order (hence which claimant of a contested global point wins and the output order), not
    # the set.  MRPT's shuffle is not in the reference tree, so this mirror draws its own
    # permutation (numpy PCG64) -- "parity unpinned" for the order; set `visit_order_fn` to a
    # callable (n_local, max_points, seed) -> indices to plug the exact list in (the C++ adapter
    # passes MRPT's own list to mp2p_hip_cloud_set_visit_order).
    visit_order_fn = None

    def _visit_order(self, n_local):
        m = int(self.maxLocalPointsPerLayer_)
        if m == 0 or n_local <= m:
            return None
        seed = int(self.localPointsSampleSeed_)
        if self.visit_order_fn is not None:
            return np.asarray(self.visit_order_fn(n_local, m, seed), dtype=np.uint32)
        rng = np.random.default_rng(seed if seed != 0 else None)
        return rng.permutation(m).astype(np.uint32)

    def _apply_visit_order(self, lLayer, cloud):
        order = self._visit_order(lLayer.size())
        key = None if order is None else order.tobytes()
        if getattr(cloud, "_order_key", None) != key:
            cloud.set_visit_order(order)
            cloud._order_key = key
        return lLayer.size() if order is None else int(order.size)

    def impl_match(self, pcGlobal, pcLocal, localPose, mc, ms, out):
        out._reset(ms.ctx)  # out = Pairings()  (:37)
        return self._impl_match_append(pcGlobal, pcLocal, localPose, mc, ms, out)

    def _impl_match_append(self, pcGlobal, pcLocal, localPose, mc, ms, out):
        ctx = ms.ctx
        for glName in sorted(pcGlobal.layers):  # std::map order (:40)
            if self.weight_pt2pt_layers:
                if glName not in self.weight_pt2pt_layers:
                    continue
                localLayers = dict(self.weight_pt2pt_layers[glName])
            else:
                localLayers = {glName: None}
            for lcName in sorted(localLayers):
                w = localLayers[lcName]
                if lcName not in pcLocal.layers:
                    if w is None:
                        continue  # silently ignored (:74-78)
                    raise RuntimeError(f"Local pointcloud layer '{lcName}' not found matching "
                                       f"global layer '{glName}'")
                g, l = pcGlobal.layers[glName], pcLocal.layers[lcName]
                nBefore = out._n_pt2pt_hint() if w is not None else 0
                self.implMatchOneLayer(ctx, g, l, localPose, ms, glName, lcName, out)
                if w is not None:
                    nAfter = out._n_pt2pt_hint()
                    if nAfter != nBefore:
                        out.point_weights.append((nAfter - nBefore, w))  # :121-125
        return True


def _pairings_reset(self, ctx):
    self.ctx = ctx
    self.point_weights = []
    self.paired_pt2ln = np.zeros(0, _lib.PAIR_PT2LN)
    self.paired_pl2pl = np.zeros(0, _lib.PAIR_PL2PL)
    self.paired_ln2ln = []
    self._ub = [0, 0]
    if self._dev is not None:
        self._dev.clear()
    self._invalidate()


def _n_pt2pt_hint(self):
    if self._dev is None:
        return 0
    return self._dev.counts()[0]


Pairings._reset = _pairings_reset
Pairings._n_pt2pt_hint = _n_pt2pt_hint


class Matcher_Points_DistanceThreshold(Matcher_Points_Base):
    """Matcher_Points_DistanceThreshold.cpp:39-269"""

    def __init__(self):
        super().__init__()
        self.threshold = 0.50
        self.thresholdAngularDeg = 0.50
        self.pairingsPerPoint = 1

    def initialize(self, params):
        super().initialize(params)
        self.declare_parameter_req(params, "threshold")
        self.declare_parameter_req(params, "thresholdAngularDeg")
        self.declare_parameter_opt(params, "pairingsPerPoint", cast=int)

    def _params(self, local_index_offset=0):
        return _lib.Pt2PtParams(
            float(self.threshold), float(self.thresholdAngularDeg), int(self.pairingsPerPoint),
            int(self.allowMatchAlreadyMatchedPoints_),
            int(self.allowMatchAlreadyMatchedGlobalPoints_),
            float(self.bounding_box_intersection_check_epsilon_), int(local_index_offset),
            float(self.initial_radius_cells), int(self.queries_per_wave),
            float(self.group_radius_factor), int(self.cell_budget),
            float(self.defer_radius_cells), int(self.disable_warm_start), int(self.brick_budget),
            int(self.tile_order), int(self.multi_search_radius_mode))

    def implMatchOneLayer(self, ctx, gLayer, lLayer, localPose, ms, glName, lcName, out):
        self.checkAllParametersAreRealized()
        prm = self._params()
        gmap, cloud = gLayer.as_global(ctx), lLayer.as_local(ctx)
        n_visit = self._apply_visit_order(lLayer, cloud)
        out._ub[0] += n_visit * int(self.pairingsPerPoint)
        dev = out._ensure_dev(ctx, out._ub[0], out._ub[1])
        core.match_pt2pt(ctx, gmap, cloud, localPose, prm, ms.for_layers(glName, lcName), dev)
        ms._sync_shared(glName, lcName)


class Matcher_Points_InlierRatio(Matcher_Points_Base):
    """Matcher_Points_InlierRatio.cpp:28-143"""

    def __init__(self):
        super().__init__()
        self.inliersRatio = 0.80

    def initialize(self, params):
        super().initialize(params)
        if params is None or "inliersRatio" not in params:
            raise KeyError("Required parameter `inliersRatio` not an existing key in dictionary.")
        self.inliersRatio = float(params["inliersRatio"])

    def implMatchOneLayer(self, ctx, gLayer, lLayer, localPose, ms, glName, lcName, out):
        if not (0.0 < self.inliersRatio < 1.0):
            raise RuntimeError("ASSERT_GT_(inliersRatio, 0.0) / ASSERT_LT_(inliersRatio, 1.0)")
        prm = _lib.InlierRatioParams(float(self.inliersRatio), int(self.allowMatchAlreadyMatchedPoints_),
                                     int(self.allowMatchAlreadyMatchedGlobalPoints_),
                                     float(self.bounding_box_intersection_check_epsilon_))
        gmap, cloud = gLayer.as_global(ctx), lLayer.as_local(ctx)
        n_visit = self._apply_visit_order(lLayer, cloud)
        out._ub[0] += n_visit
        dev = out._ensure_dev(ctx, out._ub[0], out._ub[1])
        core.match_inlier_ratio(ctx, gmap, cloud, localPose, prm, ms.for_layers(glName, lcName), dev)
        ms._sync_shared(glName, lcName)


class Matcher_Adaptive(Matcher_Points_Base):
    """Matcher_Adaptive.cpp:32-314 (parameters Matcher_Adaptive.h:67-77).

    `threshold_from_histogram`: optional callable (hist dict) -> squared-distance limit, in place of
    the restated mrpt::math::confidenceIntervalsFromHistogram (un-vendored MRPT, parity unpinned);
    `last_ci_high` / `last_histogram` keep what the last call used."""

    def __init__(self, ConfidenceInterval=0.80, FirstToSecondDistanceMax=1.2, AbsoluteMaxSearchDistance=5.0):
        super().__init__()
        self.confidenceInterval = ConfidenceInterval
        self.firstToSecondDistanceMax = FirstToSecondDistanceMax
        self.absoluteMaxSearchDistance = AbsoluteMaxSearchDistance
        self.enableDetectPlanes = False
        self.maxPt2PtCorrespondences = 1
        self.planeSearchPoints = 8
        self.planeMinimumFoundPoints = 4
        self.planeMinimumDistance = 0.10
        self.planeEigenThreshold = 0.01
        self.minimumCorrDist = 0.1
        self.threshold_from_histogram = None
        self.last_ci_high = None
        self.last_histogram = None

    def initialize(self, params):
        super().initialize(params)
        params = params or {}
        for k in ("confidenceInterval", "firstToSecondDistanceMax", "absoluteMaxSearchDistance",
                  "enableDetectPlanes"):  # MCP_LOAD_REQ (:36-42)
            if k not in params:
                raise KeyError("Required parameter `%s` not an existing key in dictionary." % k)
        self.confidenceInterval = float(params["confidenceInterval"])
        self.firstToSecondDistanceMax = float(params["firstToSecondDistanceMax"])
        self.absoluteMaxSearchDistance = float(params["absoluteMaxSearchDistance"])
        self.minimumCorrDist = float(params.get("minimumCorrDist", self.minimumCorrDist))
        self.enableDetectPlanes = bool(params["enableDetectPlanes"])
        self.planeSearchPoints = int(params.get("planeSearchPoints", self.planeSearchPoints))
        self.planeMinimumFoundPoints = int(params.get("planeMinimumFoundPoints", self.planeMinimumFoundPoints))
        self.planeEigenThreshold = float(params.get("planeEigenThreshold", self.planeEigenThreshold))
        self.maxPt2PtCorrespondences = int(params.get("maxPt2PtCorrespondences", self.maxPt2PtCorrespondences))
        self.planeMinimumDistance = float(params.get("planeMinimumDistance", self.planeMinimumDistance))
        if not (0.0 < self.confidenceInterval < 1.0):  # :50-51
            raise RuntimeError("ASSERT_LT_(confidenceInterval, 1.0) / ASSERT_GT_(confidenceInterval, 0.0)")
        if self.planeSearchPoints < self.planeMinimumFoundPoints:  # :53
            raise RuntimeError("ASSERT_GE_(planeSearchPoints, planeMinimumFoundPoints)")
        if self.planeMinimumFoundPoints < 3:  # :54
            raise RuntimeError("ASSERT_GE_(planeMinimumFoundPoints, 3)")
        if not self.planeEigenThreshold > 0.0:  # :56
            raise RuntimeError("ASSERT_GT_(planeEigenThreshold, 0.0)")

    def _params(self):
        return _lib.AdaptiveParams(float(self.confidenceInterval), float(self.firstToSecondDistanceMax),
                                   float(self.absoluteMaxSearchDistance), float(self.minimumCorrDist),
                                   int(bool(self.enableDetectPlanes)), int(self.maxPt2PtCorrespondences),
                                   int(self.planeSearchPoints), int(self.planeMinimumFoundPoints),
                                   float(self.planeMinimumDistance), float(self.planeEigenThreshold),
                                   int(self.allowMatchAlreadyMatchedPoints_),
                                   int(self.allowMatchAlreadyMatchedGlobalPoints_),
                                   float(self.bounding_box_intersection_check_epsilon_))

    def implMatchOneLayer(self, ctx, gLayer, lLayer, localPose, ms, glName, lcName, out):
        prm = self._params()
        gmap, cloud = gLayer.as_global(ctx), lLayer.as_local(ctx)
        n_visit = self._apply_visit_order(lLayer, cloud)  # a subset is refused by the library
        out._ub[0] += n_visit * int(self.maxPt2PtCorrespondences)
        out._ub[1] += n_visit
        dev = out._ensure_dev(ctx, out._ub[0], out._ub[1])
        mst = ms.for_layers(glName, lcName)
        if self.threshold_from_histogram is None:
            ci, h = core.match_adaptive(ctx, gmap, cloud, localPose, prm, mst, dev)
        else:
            h = core.adaptive_search(ctx, gmap, cloud, localPose, prm, mst)
            if h.valid:
                ci = float(self.threshold_from_histogram(_hist_dict(h)))
                core.adaptive_select(ctx, gmap, cloud, prm, ci, mst, dev)
            else:  # nobody has a neighbour: the one-shot entry only counts potential_pairings
                ci, h = core.match_adaptive(ctx, gmap, cloud, localPose, prm, mst, dev)
        self.last_ci_high, self.last_histogram = ci, _hist_dict(h)
        ms._sync_shared(glName, lcName)


def _hist_dict(h):
    return dict(valid=bool(h.valid), minSqr=float(h.minSqr), maxSqr=float(h.maxSqr), count=int(h.count),
                bins=[int(b) for b in h.bins])


class Matcher_Point2Plane(Matcher_Points_Base):
    """Matcher_Point2Plane.cpp:35-114.  The neighbour search / plane fit parameters
    (searchRadius, knn, minimumPlanePoints, planeEigenThreshold) configure the
    NearestPlaneCapable side (tests/test-mp2p_matcher_pt2pl.cpp:75-79)."""

    def __init__(self):
        super().__init__()
        self.distanceThreshold = 0.50
        self.searchRadius = 0.50
        self.knn = 5
        self.minimumPlanePoints = 5
        self.planeEigenThreshold = 0.01

    def initialize(self, params):
        super().initialize(params)
        self.declare_parameter_req(params, "distanceThreshold")
        params = params or {}
        self.searchRadius = float(params.get("searchRadius", self.searchRadius))
        self.knn = int(params.get("knn", self.knn))
        self.minimumPlanePoints = int(float(params.get("minimumPlanePoints", self.minimumPlanePoints)))
        self.planeEigenThreshold = float(params.get("planeEigenThreshold", self.planeEigenThreshold))

    def implMatchOneLayer(self, ctx, gLayer, lLayer, localPose, ms, glName, lcName, out):
        self.checkAllParametersAreRealized()
        prm = _lib.Pt2PlParams(float(self.distanceThreshold), float(self.searchRadius),
                               int(self.knn), int(self.minimumPlanePoints),
                               float(self.planeEigenThreshold),
                               int(self.allowMatchAlreadyMatchedPoints_),
                               float(self.bounding_box_intersection_check_epsilon_),
                               float(self.initial_radius_cells), int(self.queries_per_wave))
        gmap, cloud = gLayer.as_global(ctx), lLayer.as_local(ctx)
        out._ub[1] += self._apply_visit_order(lLayer, cloud)
        dev = out._ensure_dev(ctx, out._ub[0], out._ub[1])
        core.match_pt2pl(ctx, gmap, cloud, localPose, prm, ms.for_layers(glName, lcName), dev)
        ms._sync_shared(glName, lcName)


def run_matchers(matchers, pcGlobal, pcLocal, local_wrt_global, mc=None, userProvidedMS=None,
                 ctx=None, out=None):
    """Matcher.cpp:46-88.  Every matcher appends to the aggregate (the copy overload of
    Pairings::push_back does not carry point_weights over: Pairings.cpp:123-131)."""
    mc = mc or MatchContext()
    ms = userProvidedMS or MatchState(pcGlobal, pcLocal, ctx)
    pairings = out if out is not None else Pairings(ms.ctx)
    pairings._reset(ms.ctx)
    anyRun = False
    for m in matchers:
        assert m is not None
        if not m.enabled or mc.icpIteration < m.runFromIteration or (
                m.runUpToIteration > 0 and mc.icpIteration > m.runUpToIteration):
            continue
        m._impl_match_append(pcGlobal, pcLocal, local_wrt_global, mc, ms, pairings)
        anyRun = True
    pairings.point_weights = []
    if not anyRun:
        import sys
        print("[mp2p_icp::run_matchers] WARNING: No active matcher actually ran on the two maps.",
              file=sys.stderr)
    return pairings


def DEG2RAD(d):
    return d * math.pi / 180.0
