"""GPU parity of Matcher_Adaptive (through the C ABI) against the CPU oracle: the same pt2pt and
pt2pl pairings in the same order, the same histogram and threshold, MatchState handling, the
split search / select entry points with a caller-provided threshold, and the error behaviour.
(The histogram -> threshold rule itself is MRPT's and restated on both sides: parity unpinned.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _scene(seed, n_g=40_000, n_l=6_000):
    """walls and a floor (planes) with clutter; a scan of it under a small pose"""
    rng = np.random.default_rng(seed)
    n3 = n_g // 4
    floor = np.column_stack([rng.uniform(-8, 8, n3), rng.uniform(-8, 8, n3), rng.normal(0, 0.004, n3)])
    wall = np.column_stack([rng.uniform(-8, 8, n3), np.full(n3, 8.0) + rng.normal(0, 0.004, n3), rng.uniform(0, 4, n3)])
    wall2 = np.column_stack([np.full(n3, -8.0) + rng.normal(0, 0.004, n3), rng.uniform(-8, 8, n3), rng.uniform(0, 4, n3)])
    clutter = rng.uniform([-8, -8, 0], [8, 8, 4], (n_g - 3 * n3, 3))
    g = np.vstack([floor, wall, wall2, clutter]).astype(np.float32)
    pick = rng.choice(n_g, n_l, replace=False)
    l = (g[pick] + rng.normal(0, 0.02, (n_l, 3))).astype(np.float32)
    l[:50] += 30.0                                   # some local points far from everything
    return g, l


def _same_pt2pt(hip, orc):
    assert len(hip) == len(orc), (len(hip), len(orc))
    assert np.array_equal(hip["localIdx"], orc["localIdx"])
    assert np.array_equal(hip["globalIdx"], orc["globalIdx"])
    assert np.array_equal(hip["errorSquareAfterTransformation"].view(np.uint32), orc["errSq"].view(np.uint32))
    assert np.array_equal(hip["local"], np.stack([orc["lx"], orc["ly"], orc["lz"]], 1))
    assert np.array_equal(hip["global"], np.stack([orc["gx"], orc["gy"], orc["gz"]], 1))


def _same_pt2pl(pairs, r):
    hip, idx = pairs.paired_pt2pl, pairs.paired_pt2pl_local_idx
    orc = r["pt2pl"]
    assert len(hip) == len(orc), (len(hip), len(orc))
    assert np.array_equal(idx, r["pl_local_idx"])
    assert np.array_equal(hip["pt_local"], np.stack([orc["lx"], orc["ly"], orc["lz"]], 1))
    # as tests/test_gpu_matcher_pt2pl.py: which points pair is exact, the plane to 1e-9
    assert np.allclose(hip["plane"], orc["plane"], rtol=0, atol=1e-9)
    assert np.allclose(hip["centroid"], orc["centroid"], rtol=0, atol=1e-9)


CASES = [
    dict(enableDetectPlanes=False, maxPt2PtCorrespondences=1),
    dict(enableDetectPlanes=False, maxPt2PtCorrespondences=3, firstToSecondDistanceMax=1.6),
    dict(enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8, planeMinimumFoundPoints=4,
         firstToSecondDistanceMax=2.0),
    dict(enableDetectPlanes=True, maxPt2PtCorrespondences=1, planeSearchPoints=12, planeMinimumFoundPoints=5),
    dict(enableDetectPlanes=True, maxPt2PtCorrespondences=5, planeSearchPoints=16, planeMinimumFoundPoints=3,
         confidenceInterval=0.95, minimumCorrDist=0.02),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_parity_vs_oracle(amd, oracle, case):
    g, l = _scene(70 + case)
    kw = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=1.5,
              minimumCorrDist=0.1)
    kw.update(CASES[case])
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Adaptive()
    m.initialize(kw)
    n_pl = 0
    for pose in (oracle.pose_from_xyzypr(0.03, -0.02, 0.01, 0.004, 0.0, -0.002), oracle.pose_identity(),
                 oracle.pose_from_xyzypr(0.4, 0.3, -0.1, 0.05, 0.01, 0.02)):
        r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree, **kw)
        pairs = amd.Pairings()
        ms = amd.MatchState(pcG, pcL)
        assert m.match(pcG, pcL, pose, amd.MatchContext(), ms, pairs)
        h = m.last_histogram
        assert h["valid"] and h["bins"] == r["hist"]["bins"].tolist() and h["count"] == r["hist"]["count"]
        assert np.float32(h["minSqr"]) == r["hist"]["minSq"] and np.float32(h["maxSqr"]) == r["hist"]["maxSq"]
        assert m.last_ci_high == r["ci_high"]
        _same_pt2pt(pairs.paired_pt2pt, r["pt2pt"])
        _same_pt2pl(pairs, r)
        assert pairs.potential_pairings == r["potential"] == l.shape[0] * kw["maxPt2PtCorrespondences"]
        gm, lm = ms.for_layers("raw", "raw").download()
        assert not gm.any()                                                   # global marks: never written
        assert set(np.flatnonzero(lm).tolist()) == set(r["pl_local_idx"].tolist()) | set(r["pt2pt"]["localIdx"].tolist())
        n_pl += len(r["pt2pl"])
    if kw["enableDetectPlanes"]:
        assert n_pl > 1000
    assert len(r["pt2pt"]) > 0


def test_match_state_reuse_and_given_threshold(amd, oracle):
    g, l = _scene(91)
    rng = np.random.default_rng(5)
    lt0 = (rng.random(l.shape[0]) < 0.3).astype(np.uint8)
    gt0 = (rng.random(g.shape[0]) < 0.3).astype(np.uint8)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    pose = oracle.pose_from_xyzypr(0.02, 0.01, -0.01, 0.002, 0.001, 0.0)
    for allowL, allowG, given in ((False, False, None), (True, False, None), (False, True, None), (False, False, 0.05)):
        kw = dict(confidenceInterval=0.9, firstToSecondDistanceMax=1.5, absoluteMaxSearchDistance=1.0,
                  enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8,
                  planeMinimumFoundPoints=4)
        lt, gt = lt0.copy(), gt0.copy()
        r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree,
                                  allowMatchAlreadyMatchedPoints=allowL, allowMatchAlreadyMatchedGlobalPoints=allowG,
                                  local_taken=lt, global_taken=gt, ci_high=given, **kw)
        m = amd.Matcher_Adaptive()
        m.initialize(dict(kw, allowMatchAlreadyMatchedPoints=allowL, allowMatchAlreadyMatchedGlobalPoints=allowG))
        if given is not None:
            seen = {}
            m.threshold_from_histogram = lambda h: (seen.update(h), given)[1]
        ms = amd.MatchState(pcG, pcL)
        ms.for_layers("raw", "raw").upload(gt0, lt0)
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, pose, amd.MatchContext(), ms, pairs)
        if given is not None:
            assert seen["bins"] == r["hist"]["bins"].tolist() and m.last_ci_high == given
        _same_pt2pt(pairs.paired_pt2pt, r["pt2pt"])
        _same_pt2pl(pairs, r)
        gm, lm = ms.for_layers("raw", "raw").download()
        assert np.array_equal(gm, gt0) and np.array_equal(gm, gt)
        assert np.array_equal(lm, lt)
        assert len(r["pt2pt"]) > 100 and len(r["pt2pl"]) > 100


def test_appends_after_another_matcher_and_errors(amd, oracle):
    g, l = _scene(93, 20_000, 3_000)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    pose = oracle.pose_identity()
    # pipeline: a distance-threshold matcher first, the adaptive matcher on what is left
    m1 = amd.Matcher_Points_DistanceThreshold()
    m1.initialize({"threshold": 0.03, "thresholdAngularDeg": 0.0})
    m2 = amd.Matcher_Adaptive()
    kw = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=2.0,
              enableDetectPlanes=True, planeSearchPoints=6, planeMinimumFoundPoints=4)
    m2.initialize(kw)
    pairs = amd.run_matchers([m1, m2], pcG, pcL, pose)
    lt = np.zeros(l.shape[0], np.uint8)
    gt = np.zeros(g.shape[0], np.uint8)
    w1, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.03, 0.0, tree=tree,
                               local_taken=lt, global_taken=gt)
    r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree,
                              local_taken=lt, global_taken=gt, **kw)
    assert len(w1) > 50 and len(r["pt2pt"]) + len(r["pt2pl"]) > 50
    _same_pt2pt(pairs.paired_pt2pt, np.concatenate([w1, r["pt2pt"]]))
    _same_pt2pl(pairs, r)
    # nobody within reach: no pairings, potential_pairings still counted, no error
    m = amd.Matcher_Adaptive()
    m.initialize(dict(kw, absoluteMaxSearchDistance=1e-5))
    p = amd.Pairings()
    assert m.match(pcG, pcL, oracle.pose_from_xyzypr(0.5, 0.5, 0.5, 0, 0, 0), amd.MatchContext(), amd.MatchState(pcG, pcL), p)
    assert p.size() == 0 and p.potential_pairings == l.shape[0] and not m.last_histogram["valid"]
    # empty layers
    e = amd.metric_map_t({"raw": amd.PointLayer(np.zeros((0, 3), np.float32))})
    p = amd.Pairings()
    m.match(e, pcL, pose, amd.MatchContext(), amd.MatchState(e, pcL), p)
    assert p.size() == 0 and p.potential_pairings == l.shape[0]
    # parameters: required keys and the asserts of initialize() (:36-56)
    with pytest.raises(KeyError):
        amd.Matcher_Adaptive().initialize({"confidenceInterval": 0.8})
    for bad in (dict(confidenceInterval=1.0), dict(planeMinimumFoundPoints=2), dict(planeSearchPoints=3),
                dict(planeEigenThreshold=0.0)):
        with pytest.raises(RuntimeError):
            amd.Matcher_Adaptive().initialize(dict(kw, **bad))
    # a visit list: the reference throws std::out_of_range (matchesPerLocal_.at(localIdx))
    m = amd.Matcher_Adaptive()
    m.initialize(dict(kw, maxLocalPointsPerLayer=100, localPointsSampleSeed=1))
    with pytest.raises(amd.Mp2pHipError):
        m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), amd.Pairings())
    # more than 16 neighbours per point is outside this implementation
    m = amd.Matcher_Adaptive()
    m.initialize(dict(kw, planeSearchPoints=20))
    with pytest.raises(amd.Mp2pHipError):
        m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), amd.Pairings())


def test_boxes_apart_but_neighbours_in_reach(amd, oracle):
    """Matcher_Adaptive.cpp:78-81 returns before any search when the boxes of the two layers, inflated by the epsilon only, do not
    meet -- although absoluteMaxSearchDistance would still reach across the gap: no histogram, no pairing, nothing counted
    (found by the pt2pl / adaptive fuzz campaign of round 5, seed 852: the histogram used to be reported valid)"""
    g, l = _scene(95, n_g=20_000, n_l=2_000)
    l = l[50:].copy()                      # (without the far points of the scene)
    l[:, 2] += (g[:, 2].max() - l[:, 2].min()) + 0.5   # the scan lifted 0.5 m above the map's box
    kw = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=1.5, minimumCorrDist=0.1,
              enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8, planeMinimumFoundPoints=4)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pose = oracle.pose_identity()
    # neighbours ARE in reach of the search radius
    low = np.argsort(l[:, 2])[:200]
    assert min(tree.knn((float(l[i, 0]), float(l[i, 1]), float(l[i, 2])), 1)[1][0] for i in low) < 1.5 * 1.5
    r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree, **kw)
    assert not r["hist"]["valid"] and len(r["pt2pt"]) == 0 and len(r["pt2pl"]) == 0
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    for split in (False, True):
        m = amd.Matcher_Adaptive()
        m.initialize(kw)
        called = []
        if split:
            m.threshold_from_histogram = lambda h: called.append(h) or 1.0
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        assert not m.last_histogram["valid"] and not called
        assert len(pairs.paired_pt2pt) == 0 and len(pairs.paired_pt2pl) == 0
        assert pairs.potential_pairings == r["potential"]


def test_layers_far_apart_return_before_any_search(amd, oracle):
    """round 6: when the ball that holds every transformed local point clears the map's box the library returns before
    any launch, as Matcher_Adaptive.cpp:78-81 does; the split entry points stay consistent: a select after such a search emits nothing
    (no stale lists of an earlier call are read), potential_pairings still grows (:69)"""
    from mp2p_icp_amd import _lib, core
    g, l = _scene(96, n_g=20_000, n_l=2_000)
    l = l[50:].copy()
    kw = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=1.5, minimumCorrDist=0.1,
              enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8, planeMinimumFoundPoints=4)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    near, far = oracle.pose_identity(), amd.se3.from_xyzypr(500.0, -300.0, 40.0, 0.3, 0.0, 0.0)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Adaptive()
    m.initialize(kw)
    for pose in (near, far, near, far):       # a real search first: its lists must not leak into the far call
        r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree, **kw)
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        assert m.last_histogram["valid"] == r["hist"]["valid"]
        assert len(pairs.paired_pt2pt) == len(r["pt2pt"]) and np.array_equal(pairs.paired_pt2pl_local_idx, r["pl_local_idx"])
        assert pairs.potential_pairings == r["potential"]
        if pose is far:
            assert not m.last_histogram["valid"] and pairs.empty()
    # the split entry points at the C ABI: a real search, then a search that returns early, then select -- which must not read the
    # first search's lists
    ctx = amd.Context(0)
    gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    dev = core.DevicePairs(ctx, 2 * l.shape[0], l.shape[0])
    prm = m._params()
    assert core.adaptive_search(ctx, gmap, cloud, near, prm, None).valid
    h = core.adaptive_search(ctx, gmap, cloud, far, prm, None)
    assert not h.valid
    core.adaptive_select(ctx, gmap, cloud, prm, 1.0, None, dev)
    n_pt, n_pl, pot = dev.counts()
    assert n_pt == 0 and n_pl == 0 and pot == l.shape[0] * kw["maxPt2PtCorrespondences"]
    with pytest.raises(_lib.Mp2pHipError):   # the (empty) lists are consumed: a second select has no search to refer to
        core.adaptive_select(ctx, gmap, cloud, prm, 1.0, None, dev)
