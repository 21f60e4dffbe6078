"""GPU parity of Matcher_Points_DistanceThreshold (HIP path, through the C ABI) against the CPU
oracle and the reference's own known-answer test.  Correspondence lists must be BIT-EXACT
(indices, order, coordinates, errorSquareAfterTransformation)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEG = math.pi / 180.0


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _maps(amd, g, l, **layer_kw):
    pcG = amd.metric_map_t({amd.PT_LAYER_RAW: amd.PointLayer(g, **layer_kw)})
    pcL = amd.metric_map_t({amd.PT_LAYER_RAW: amd.PointLayer(l)})
    return pcG, pcL


def _hip_match(amd, pcG, pcL, pose, params, ms=None):
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize(params)
    pairs = amd.Pairings()
    ms = ms or amd.MatchState(pcG, pcL)
    assert m.match(pcG, pcL, pose, amd.MatchContext(), ms, pairs)
    return pairs, ms


def _assert_same_pairs(hip_pairs, orc_pairs):
    assert len(hip_pairs) == len(orc_pairs), (len(hip_pairs), len(orc_pairs))
    if len(orc_pairs) == 0:
        return
    assert np.array_equal(hip_pairs["localIdx"], orc_pairs["localIdx"])
    assert np.array_equal(hip_pairs["globalIdx"], orc_pairs["globalIdx"])
    for k, (a, b, c) in (("global", ("gx", "gy", "gz")), ("local", ("lx", "ly", "lz"))):
        want = np.stack([orc_pairs[a], orc_pairs[b], orc_pairs[c]], 1)
        assert np.array_equal(hip_pairs[k].view(np.uint32), want.view(np.uint32)), k
    assert np.array_equal(hip_pairs["errorSquareAfterTransformation"].view(np.uint32),
                          orc_pairs["errSq"].view(np.uint32))


# ---- tests/test-mp2p_matcher_pt2pt.cpp:26-107 through the mirrored plugin interface ----------
def test_reference_known_answers(amd, oracle):
    from test_oracle_kat import KAT_POSES, _kat_global, _kat_local
    g, l = _kat_global(), _kat_local()
    pcG, pcL = _maps(amd, g, l)
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": 1.05, "thresholdAngularDeg": 0.001})
    assert abs(m.threshold - 1.05) < 1e-4 and abs(m.thresholdAngularDeg - 0.001) < 1e-4
    for pose6, expected in KAT_POSES:
        pairs = amd.Pairings()
        ms = amd.MatchState(pcG, pcL)
        m.match(pcG, pcL, amd.se3.from_xyzypr(*pose6), amd.MatchContext(), ms, pairs)
        got = [(int(p["localIdx"]), int(p["globalIdx"])) for p in pairs.paired_pt2pt]
        assert got == expected, (pose6, got)
        assert pairs.empty() == (len(expected) == 0)
        assert pairs.potential_pairings == 2


CASES = [
    # n_g, n_l, threshold, angular, seed, layer kwargs
    (5000, 1000, 0.5, 0.0, 1, {}),
    (5000, 1000, 2.0, 0.0, 2, {}),
    (20000, 4097, 0.3, 0.5, 3, {}),
    (20000, 63, 1.0, 0.0, 4, {}),
    (3000, 3000, 5.0, 0.0, 5, {}),           # r_max far larger than the voxels
    (50000, 10000, 0.2, 0.0, 6, dict(cell_size=0.05)),
    (50000, 10000, 0.7, 0.1, 7, dict(cell_size=3.0)),   # very coarse voxels
    (777, 129, 0.4, 0.0, 8, dict(target_per_cell=1.0)),
    (20000, 4097, 0.5, 0.2, 9, dict(no_occupancy_bitmap=1)),   # hash probes alone
    (20000, 4097, 0.5, 0.2, 9, dict(no_occupancy_bitmap=2)),   # bitmaps + hash, no dense voxel directory
    (20000, 4097, 0.5, 0.2, 9, dict(no_occupancy_bitmap=4)),   # directory for the coarse levels only
]


@pytest.mark.parametrize("n_g,n_l,thr,ang,seed,kw", CASES)
@pytest.mark.parametrize("q", [0, 64, 32, 16])   # 0 = the default tile size
def test_random_parity_vs_oracle(amd, oracle, n_g, n_l, thr, ang, seed, kw, q):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(n_l, n_g, seed, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l, **kw)
    for pose in (d["T_gt"], d["T_init"]):
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose,
                                       thr, ang, tree=tree)
        pairs, _ = _hip_match(amd, pcG, pcL, pose,
                              {"threshold": thr, "thresholdAngularDeg": ang,
                               "hip_queries_per_wave": q})
        _assert_same_pairs(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot


def test_brute_force_definition_small(amd, oracle):
    """against the brute-force oracle (THE definition), including exact duplicates -> ties"""
    rng = np.random.default_rng(11)
    g = rng.uniform(-3, 3, (600, 3)).astype(np.float32)
    g[100:140] = g[0:40]          # duplicated global points: lowest index must win
    l = np.concatenate([g[:300] + rng.normal(0, 0.01, (300, 3)).astype(np.float32), g[50:60]])
    pcG, pcL = _maps(amd, g, l)
    T = amd.se3.identity()
    for allow in (False, True):
        want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, 0.3, 0.0,
                                     allowMatchAlreadyMatchedGlobalPoints=allow, tree=None)
        pairs, _ = _hip_match(amd, pcG, pcL, T, {"threshold": 0.3, "thresholdAngularDeg": 0.0,
                                                 "allowMatchAlreadyMatchedGlobalPoints": allow})
        _assert_same_pairs(pairs.paired_pt2pt, want)


def test_allow_flags_and_match_state(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(3000, 8000, 21)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l)
    rng = np.random.default_rng(5)
    gt0 = (rng.random(g.shape[0]) < 0.3).astype(np.uint8)
    lt0 = (rng.random(l.shape[0]) < 0.3).astype(np.uint8)
    for allowL in (False, True):
        for allowG in (False, True):
            gt, lt = gt0.copy(), lt0.copy()
            want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2],
                                         d["T_init"], 0.6, 0.0, tree=tree,
                                         allowMatchAlreadyMatchedPoints=allowL,
                                         allowMatchAlreadyMatchedGlobalPoints=allowG,
                                         local_taken=lt, global_taken=gt)
            ms = amd.MatchState(pcG, pcL)
            ms.for_layers("raw", "raw").upload(gt0, lt0)
            pairs, ms = _hip_match(amd, pcG, pcL, d["T_init"],
                                   {"threshold": 0.6, "thresholdAngularDeg": 0.0,
                                    "allowMatchAlreadyMatchedPoints": allowL,
                                    "allowMatchAlreadyMatchedGlobalPoints": allowG}, ms)
            _assert_same_pairs(pairs.paired_pt2pt, want)
            g_after, l_after = ms.for_layers("raw", "raw").download()
            assert np.array_equal(g_after, gt) and np.array_equal(l_after, lt)


def test_edge_cases(amd, oracle):
    rng = np.random.default_rng(3)
    g = rng.uniform(-5, 5, (2000, 3)).astype(np.float32)
    l = rng.uniform(-5, 5, (100, 3)).astype(np.float32)
    prm = {"threshold": 0.5, "thresholdAngularDeg": 0.0}
    # empty local
    pcG, pcL = _maps(amd, g, np.zeros((0, 3), np.float32))
    pairs, _ = _hip_match(amd, pcG, pcL, amd.se3.identity(), prm)
    assert pairs.empty() and pairs.potential_pairings == 0
    # empty global: potential_pairings still counted (Matcher_Points_DistanceThreshold.cpp:64-67)
    pcG, pcL = _maps(amd, np.zeros((0, 3), np.float32), l)
    pairs, _ = _hip_match(amd, pcG, pcL, amd.se3.identity(), prm)
    assert pairs.empty() and pairs.potential_pairings == 100
    # disjoint bounding boxes -> early out, no pairs
    pcG, pcL = _maps(amd, g, l)
    pairs, _ = _hip_match(amd, pcG, pcL, amd.se3.from_xyzypr(100, 0, 0), prm)
    assert pairs.empty() and pairs.potential_pairings == 100
    # single global point / single local point
    pcG, pcL = _maps(amd, g[:1], g[:1] + np.float32(0.01))
    pairs, _ = _hip_match(amd, pcG, pcL, amd.se3.identity(), prm)
    assert len(pairs.paired_pt2pt) == 1 and pairs.paired_pt2pt[0]["globalIdx"] == 0
    # non-finite local point is never paired
    l2 = l.copy()
    l2[7] = np.nan
    pcG, pcL = _maps(amd, g, l2)
    l3 = l.copy()
    l3[7] = 1e6  # the oracle's stand-in for "cannot be paired"
    want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l3[:, 0], l3[:, 1], l3[:, 2],
                                 amd.se3.identity(), 0.5, 0.0)
    pairs, _ = _hip_match(amd, pcG, pcL, amd.se3.identity(), prm)
    assert 7 not in pairs.paired_pt2pt["localIdx"]
    _assert_same_pairs(pairs.paired_pt2pt, want)
    # invalid parameters raise (ASSERT_GT_(threshold, .0))
    with pytest.raises(amd.Mp2pHipError):
        _hip_match(amd, pcG, pcL, amd.se3.identity(), {"threshold": 0.0, "thresholdAngularDeg": 0.0})
    with pytest.raises(KeyError):
        amd.Matcher_Points_DistanceThreshold().initialize({"threshold": 1.0})


def test_formula_parameters(amd):
    """tests/test-mp2p_matcher_pt2pt_parameterizable.cpp:28-52"""
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": "MATCH_THRESHOLD*2.0", "thresholdAngularDeg": 0.0})
    with pytest.raises(RuntimeError):
        m.checkAllParametersAreRealized()
    ps = amd.ParameterSource()
    m.attachToParameterSource(ps)
    ps.updateVariable("MATCH_THRESHOLD", 0.75)
    ps.realize()
    m.checkAllParametersAreRealized()
    assert abs(m.threshold - 1.5) < 1e-12


def test_kitti_shape_c2_parity(amd, oracle):
    """BASELINE config 2 shape at reduced size: 120k-point scan vs 500k-point map."""
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(120_000, 500_000, 2001)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l)
    for pose in (d["T_gt"], d["T_init"]):
        want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose,
                                     2.0, 0.0, tree=tree, threads=8)
        pairs, _ = _hip_match(amd, pcG, pcL, pose, {"threshold": 2.0, "thresholdAngularDeg": 0.0})
        _assert_same_pairs(pairs.paired_pt2pt, want)


def test_warm_start_pose_sequence(amd, oracle):
    """Repeated calls on the same (map, cloud) seed every query with its previous nearest
    neighbour.  The lists must stay bit-exact for any pose sequence (small steps, big jumps,
    back again) and identical to a cold call, whatever the launch order of the tiles."""
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(20_000, 100_000, 77)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l)
    rng = np.random.default_rng(5)
    poses = [d["T_init"], d["T_gt"]]
    for _ in range(4):
        xi = np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 0.05, 3)])
        poses.append(amd.se3.compose(d["T_gt"], amd.se3.exp(xi)))
    poses += [amd.se3.compose(d["T_gt"], amd.se3.from_xyzypr(30.0, -20.0, 1.0, 1.0, 0, 0)),  # far away
              d["T_gt"], d["T_init"]]
    ms_w = amd.Matcher_Points_DistanceThreshold()
    ms_w.initialize({"threshold": 1.5, "thresholdAngularDeg": 0.05})
    ms_c = amd.Matcher_Points_DistanceThreshold()
    ms_c.initialize({"threshold": 1.5, "thresholdAngularDeg": 0.05, "hip_disable_warm_start": True})
    # longest-first launch order of the tiles (from the previous call's measured durations)
    ms_o = amd.Matcher_Points_DistanceThreshold()
    ms_o.initialize({"threshold": 1.5, "thresholdAngularDeg": 0.05, "hip_tile_order": True})
    for pose in poses:
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose,
                                       1.5, 0.05, tree=tree, threads=8)
        for m in (ms_w, ms_c, ms_o, ms_o):
            pairs = amd.Pairings()
            m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
            _assert_same_pairs(pairs.paired_pt2pt, want)
            assert pairs.potential_pairings == pot


@pytest.mark.parametrize("mode", ["nn_cert=2", "nn_cert=1,nn_cert_step_mm=3"])
def test_search_skip_certificate_pose_sequence(amd, oracle, mode, monkeypatch):
    """the point-to-point search's certificate (nn_query.hip, NNArgs::lb2nd): poses that creep from millimetres to 20
    micrometres on ONE context -- a growing share of the queries keeps its previous neighbour without a search, because
    every other map point is provably farther -- every call equal to the oracle's cold result, bit for bit (indices,
    coordinates, fp32 d2); local points taken in one call and free in the next, a 4 cm jump, and a return.  Both the
    always-tracking mode and the adaptive one (bounds tracked only after steps below 3 mm)."""
    from mp2p_icp_amd import _lib, core, synthetic
    monkeypatch.setenv("MP2P_HIP_TUNE", mode)
    d = synthetic.make_pair(30_000, 400_000, 91)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    rng = np.random.default_rng(3)
    poses, scales = [d["T_gt"]], [2e-3, 1e-3, 1e-3, 3e-4, 1e-4, 1e-4, 5e-5, 2e-5, 2e-5, 4e-2, 2e-5, 2e-5]
    for sc in scales:
        poses.append(amd.se3.compose(poses[-1], amd.se3.exp(np.concatenate([rng.normal(0, sc, 3), rng.normal(0, 0.03 * sc, 3)]))))
    poses.append(poses[3])  # back to an earlier pose
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, l.shape[0], 0)
    ms = core.DeviceMatchState(ctx, g.shape[0], l.shape[0])
    prm = _lib.Pt2PtParams(1.0, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    skipped = {}
    for k, pose in enumerate(poses):
        lt = np.zeros(l.shape[0], np.uint8)
        if k in (4, 7):
            lt[rng.choice(l.shape[0], 2000, replace=False)] = 1  # skipped by this call only
        gt = np.zeros(g.shape[0], np.uint8)
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 1.0, 0.0, tree=tree,
                                       local_taken=lt.copy(), global_taken=gt.copy())
        # the instrumented build (profiling 2) counts the queries that finish without a search; it reads the bounds the
        # previous (plain) call left but tracks none itself, so it is used for a look at selected poses only
        instr = k in (6, 10, 12)
        ctx.set_profiling(2 if instr else 0)
        ms.upload(gt, lt)
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, ms, pairs)
        _assert_same_pairs(pairs.download_pt2pt(), want)
        if instr:
            skipped[k] = ctx.stats()["nn_lane_skipped"]
    assert skipped[6] > 0.15 * l.shape[0], skipped   # 0.1 mm steps: a good share is certified
    assert skipped[10] < 0.8 * skipped[6], skipped    # the 4 cm jump itself: fewer are (a sparse neighbourhood keeps its room)
    assert skipped[12] > 0.15 * l.shape[0], skipped   # ... and two calls later the bounds are back


def test_outliers_with_nothing_in_reach_skip_later_calls(amd, oracle):
    """A local point metres from every surface (30 % of BASELINE config C5's layer) searched its whole ball at EVERY call:
    the bound it kept was the radius just covered.  The one-query kernel now looks for an empty cube of half-edge 2 r_max
    (else 1.5 r_max) around such a query -- a handful of coarse occupancy bits -- and the warm start lets it skip while it
    has moved less than the difference.  Lists bit-exact at every pose of a creeping sequence with a jump and a return;
    from the third call on the far points finish without a search."""
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.random_cloud_pair(6000, 150_000, 321, outlier_frac=0.0)
    g = d["glob"]
    rng = np.random.default_rng(9)
    l = d["local"].copy()
    far = rng.choice(l.shape[0], 3000, replace=False)
    l[far, 2] += rng.uniform(3.0, 15.0, far.size).astype(np.float32)  # metres above the surface: nothing within 0.8 m
    l[far[:500], 2] -= 2.2                                             # ... and some in the grey zone (0.8 .. 13 m)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    ctx = amd.Context(0)
    ctx.set_profiling(2)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, l.shape[0], 0)
    prm = _lib.Pt2PtParams(0.8, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    pose, skipped = d["T_init"], []
    steps = [0.02, 0.02, 0.01, 0.05, 0.01, 2.5, 0.01, -2.5, 0.01]  # creep, a 2.5 m jump (beyond any room), back
    for k in range(len(steps) + 1):
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.8, 0.0, tree=tree)
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
        _assert_same_pairs(pairs.download_pt2pt(), want)
        skipped.append(ctx.stats()["nn_lane_skipped"])
        if k < len(steps):
            pose = amd.se3.compose(pose, amd.se3.exp(np.array([steps[k], 0.3 * steps[k], 0.0, 0.0, 0.0, 0.002 * np.sign(steps[k])])))
    assert skipped[0] == 0                      # a first call has nothing to go by
    assert skipped[2] > 1500, skipped           # the far points have their room and skip ...
    assert skipped[6] < skipped[5], skipped     # ... the jump uses it up (they search again) ...
    assert skipped[9] > 1500, skipped           # ... and it is found again


@pytest.mark.parametrize("K", [2, 3, 5, 8, 13, 16])
@pytest.mark.parametrize("allow_global", [False, True])
@pytest.mark.parametrize("radius_mode", [True, False])
def test_pairings_per_point(amd, oracle, K, allow_global, radius_mode):
    """pairingsPerPoint > 1 (Matcher_Points_DistanceThreshold.cpp:242-265): the k nearest in
    ascending d2 up to the threshold, first claimant of a global point wins -- with the search of the
    shipped TBB build (nn_radius_search, :172-177; the default) and of the sequential build
    (nn_multiple_search, :246-248); thresholdAngularDeg > 0 tells them apart."""
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(3000, 12000, 40 + K, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    g[200:230] = g[0:30]  # duplicated global points: ties inside a k-list
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l)
    for pose in (d["T_gt"], d["T_init"]):
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose,
                                       0.6, 0.2, pairingsPerPoint=K, tree=tree,
                                       allowMatchAlreadyMatchedGlobalPoints=allow_global,
                                       multi_search_radius_mode=int(radius_mode))
        pairs, _ = _hip_match(amd, pcG, pcL, pose,
                              {"threshold": 0.6, "thresholdAngularDeg": 0.2, "pairingsPerPoint": K,
                               "allowMatchAlreadyMatchedGlobalPoints": allow_global,
                               "hip_multi_search_radius_mode": radius_mode})
        _assert_same_pairs(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot == l.shape[0] * K


def test_pairings_per_point_with_match_state(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(2000, 8000, 77, outlier_frac=0.05)
    g, l = d["glob"], d["local"]
    rng = np.random.default_rng(3)
    lt = (rng.random(l.shape[0]) < 0.2).astype(np.uint8)
    gt = (rng.random(g.shape[0]) < 0.2).astype(np.uint8)
    lt_o, gt_o = lt.copy(), gt.copy()
    want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_gt"], 0.5, 0.0,
                                 pairingsPerPoint=4, local_taken=lt_o, global_taken=gt_o)
    pcG, pcL = _maps(amd, g, l)
    ms = amd.MatchState(pcG, pcL)
    ms.for_layers("raw", "raw").upload(gt, lt)
    pairs, ms = _hip_match(amd, pcG, pcL, d["T_gt"], {"threshold": 0.5, "thresholdAngularDeg": 0.0,
                                                      "pairingsPerPoint": 4}, ms=ms)
    _assert_same_pairs(pairs.paired_pt2pt, want)
    g_after, l_after = ms.for_layers("raw", "raw").download()
    assert np.array_equal(g_after, gt_o) and np.array_equal(l_after, lt_o)


@pytest.mark.parametrize("K", [1, 3])
def test_max_local_points_visit_order(amd, oracle, K):
    """maxLocalPointsPerLayer (Matcher_Points_Base.cpp:222-246): a shuffled list of the first
    maxLocalPoints indices is visited in list order; box, winners and output order follow it."""
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(4000, 9000, 90 + K, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    l[3000:] += 50.0  # far-away tail: must not enter the bounding box of the visited points
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG, pcL = _maps(amd, g, l)
    order = np.random.default_rng(8).permutation(2500).astype(np.uint32)
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": 0.8, "thresholdAngularDeg": 0.0, "pairingsPerPoint": K,
                  "maxLocalPointsPerLayer": 2500, "localPointsSampleSeed": 5})
    m.visit_order_fn = lambda n, mx, seed: order
    for pose in (d["T_gt"], d["T_init"]):
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose,
                                       0.8, 0.0, pairingsPerPoint=K, tree=tree, idxs=order)
        pairs = amd.Pairings()
        m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        _assert_same_pairs(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot == 4000 * K  # :64: pcLocal.size(), not the visited subset
    # a later matcher without the limit sees the whole layer again
    want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_gt"],
                                 0.8, 0.0, tree=tree)
    pairs, _ = _hip_match(amd, pcG, pcL, d["T_gt"], {"threshold": 0.8, "thresholdAngularDeg": 0.0})
    _assert_same_pairs(pairs.paired_pt2pt, want)
    # the mirror's own permutation: seeded -> reproducible, a permutation of the first 2500
    m2 = amd.Matcher_Points_DistanceThreshold()
    m2.initialize({"threshold": 0.8, "thresholdAngularDeg": 0.0, "maxLocalPointsPerLayer": 2500,
                   "localPointsSampleSeed": 5})
    o1, o2 = m2._visit_order(4000), m2._visit_order(4000)
    assert np.array_equal(o1, o2) and sorted(o1.tolist()) == list(range(2500))
    assert m2._visit_order(2500) is None


@pytest.mark.parametrize("tune", ["pipelines=2", "mfma_scan=0", "tile_waves=5", "dir_budget_mb=0,claim_dedup=0,claim_peek=0", "nn_cert=2",
                                  "nn_cert=0,tile_bricks=0,hard_cand=0,empty_room=0", "nn_cert=2,pipelines=2,tile_cand_cap=2000",
                                  # round 5: the round-4 kernels (box rule, lane kernel + pending list), the selection behind the lane
                                  # kernel, the fused prologue on two pipelines, tiny budgets on the new path
                                  "tile_select=0", "nn_direct=0", "nn_direct=0,nn_cert=2,hard_cand=50", "pipelines=2,nn_cert=2",
                                  "tile_cand_cap=300,coop_max=0", "nn_direct=0,pipelines=2,tile_cand_cap=500",
                                  # all pending queries in one pass, always or never
                                  "grp_all_bricks=0", "pipelines=2,grp_all_bricks=64", "nn_direct=1,grp_all_bricks=0", "nn_direct=1,grp_all_bricks=200"])
def test_tune_knobs_do_not_change_the_lists(amd, oracle, tune, monkeypatch):
    """MP2P_HIP_TUNE is read once per context: every setting is a measurement aid that must compute the
    same lists (two search pipelines on two streams, exact scan instead of the matrix-pipe prefilter, another
    register budget, no voxel directory / claim shortcuts).  A warm sequence of three poses."""
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.random_cloud_pair(40_000, 300_000, 123, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    monkeypatch.setenv("MP2P_HIP_TUNE", tune)
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, l.shape[0], 0)
    prm = _lib.Pt2PtParams(0.8, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    rng = np.random.default_rng(5)
    pose = d["T_init"]
    for _ in range(3):
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.8, 0.0, tree=tree)
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
        _assert_same_pairs(pairs.download_pt2pt(), want)
        pose = amd.se3.compose(pose, amd.se3.exp(np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.004, 3)])))


