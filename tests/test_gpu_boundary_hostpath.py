"""The HOST path of the drop-in boundary: the reference-side adapter's per-call logic
(adapter/mp2p_hip_host.hpp, through adapter/hostpath_capi.cpp) against host containers -- packed
MatchState bit-fields in and out, 36-byte / 72-byte pair records out, a solver handed host Pairings --
checked against the CPU oracle, plus what it may and may not transfer per call."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _xyz(a):
    return a[:, 0], a[:, 1], a[:, 2]


def _same_pt2pt(got, want):
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(got["localIdx"], want["localIdx"])
    assert np.array_equal(got["globalIdx"], want["globalIdx"])
    assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32))
    assert np.array_equal(got["local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))
    assert np.array_equal(got["global"], np.stack([want["gx"], want["gy"], want["gz"]], 1))


def _pt2pt_prm(threshold, **kw):
    from mp2p_icp_amd import _lib
    p = _lib.Pt2PtParams()
    p.threshold, p.thresholdAngularDeg, p.pairingsPerPoint = threshold, 0.0, 1
    p.bounding_box_intersection_check_epsilon = 0.20
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _gn_prm(iters=3):
    from mp2p_icp_amd import _lib
    g = _lib.GNParams()
    g.maxInnerLoopIterations, g.minDelta, g.maxCost = iters, 1e-7, 0.0
    g.kernel, g.kernelParam = _lib.KERNEL_GEMANMCCLURE, 0.15
    g.w_pt2pt = g.w_pt2pl = 1.0
    return g


def test_icp_chain_through_host_containers(oracle):
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.make_pair(120_000, 500_000, 31)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    s = hostpath.Session(g, l)
    c0 = hostpath.counters()
    prm, gnp = _pt2pt_prm(1.5), _gn_prm()
    oprm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    pose_h, pose_o = d["T_init"].copy(), d["T_init"].copy()
    for it in range(4):
        s.begin_iteration()
        n = s.match_pt2pt(pose_h, prm, icp_iteration=it)
        lt, gt = np.zeros(l.shape[0], np.uint8), np.zeros(g.shape[0], np.uint8)
        want, pot = oracle.match_pt2pt(*_xyz(g), *_xyz(l), pose_o, 1.5, 0.0, tree=tree, local_taken=lt, global_taken=gt)
        got = s.pairs_pt2pt()
        assert n == len(got)
        _same_pt2pt(got, want)
        assert s.potential_pairings == pot
        # the host MatchState carries the marks of the emitted pairs (:116-120), set from the pair list
        assert np.array_equal(s.bits(1), lt.astype(bool)) and np.array_equal(s.bits(0), gt.astype(bool))
        pose_h, iters = s.solve_gn(pose_h, gnp)
        pose_o, *_ = oracle.optimal_tf_gauss_newton(want, None, None, pose_o, oprm)
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)
    c1 = hostpath.counters()
    # layers uploaded once, no MatchState transfer (the fields were clear), no Pairings upload (the
    # solver recognised the device-resident list by size + checksum)
    assert c1["map_uploads"] - c0["map_uploads"] == 1 and c1["cloud_uploads"] - c0["cloud_uploads"] == 1
    assert c1["mstate_uploads"] == c0["mstate_uploads"] and c1["pairings_uploads"] == c0["pairings_uploads"]
    s.close()


@pytest.mark.parametrize("allow_local,allow_global", [(False, False), (True, False), (False, True)])
def test_premarked_match_state_and_marks(oracle, allow_local, allow_global):
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.random_cloud_pair(20_000, 60_000, 11, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    rng = np.random.default_rng(4)
    lt0, gt0 = rng.random(l.shape[0]) < 0.3, rng.random(g.shape[0]) < 0.2
    tree = oracle.KDTree(*_xyz(g))
    s = hostpath.Session(g, l)
    c0 = hostpath.counters()
    s.begin_iteration()
    s.set_bits(1, lt0), s.set_bits(0, gt0)
    prm = _pt2pt_prm(0.8, allowMatchAlreadyMatchedPoints=int(allow_local),
                     allowMatchAlreadyMatchedGlobalPoints=int(allow_global))
    s.match_pt2pt(d["T_init"], prm)
    lt, gt = lt0.astype(np.uint8), gt0.astype(np.uint8)
    want, _ = oracle.match_pt2pt(*_xyz(g), *_xyz(l), d["T_init"], 0.8, 0.0, tree=tree, local_taken=lt, global_taken=gt,
                                 allowMatchAlreadyMatchedPoints=allow_local,
                                 allowMatchAlreadyMatchedGlobalPoints=allow_global)
    _same_pt2pt(s.pairs_pt2pt(), want)
    assert np.array_equal(s.bits(1), lt.astype(bool)) and np.array_equal(s.bits(0), gt.astype(bool))
    assert hostpath.counters()["mstate_uploads"] == c0["mstate_uploads"] + 1  # packed words went up once
    s.close()


def test_two_matchers_one_solver_and_foreign_pairings(oracle):
    """Matcher_Point2Plane then Matcher_Points_DistanceThreshold on ONE MatchState (one run_matchers
    call): the device list is continued, the solver finds it resident.  Pairings from elsewhere are
    uploaded."""
    from mp2p_icp_amd import _lib, hostpath, synthetic
    d = synthetic.make_pair(30_000, 300_000, 9)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    PL = dict(distanceThreshold=0.25, searchRadius=0.5, knn=6, minimumPlanePoints=5, planeEigenThreshold=0.05)
    plp = _lib.Pt2PlParams()
    plp.distanceThreshold, plp.searchRadius, plp.knn = PL["distanceThreshold"], PL["searchRadius"], PL["knn"]
    plp.minimumPlanePoints, plp.planeEigenThreshold = PL["minimumPlanePoints"], PL["planeEigenThreshold"]
    plp.bounding_box_intersection_check_epsilon = 0.20
    s = hostpath.Session(g, l)
    c0 = hostpath.counters()
    s.begin_iteration()
    s.match_pt2pl(d["T_init"], plp)
    s.match_pt2pt(d["T_init"], _pt2pt_prm(1.0))
    lt = np.zeros(l.shape[0], np.uint8)
    w_pl, w_idx, pot1 = oracle.match_pt2pl(*_xyz(g), *_xyz(l), d["T_init"], tree=tree, local_taken=lt, **PL)
    gt = np.zeros(g.shape[0], np.uint8)
    w_pt, pot2 = oracle.match_pt2pt(*_xyz(g), *_xyz(l), d["T_init"], 1.0, 0.0, tree=tree, local_taken=lt, global_taken=gt)
    _same_pt2pt(s.pairs_pt2pt(), w_pt)
    got_pl = s.pairs_pt2pl()
    assert len(got_pl) == len(w_pl) and np.allclose(got_pl["plane"], w_pl["plane"], rtol=0, atol=1e-9)
    assert np.array_equal(got_pl["pt_local"], np.stack([w_pl["lx"], w_pl["ly"], w_pl["lz"]], 1))
    assert s.potential_pairings == pot1 + pot2
    assert np.array_equal(s.bits(1), lt.astype(bool)) and np.array_equal(s.bits(0), gt.astype(bool))
    gnp = _gn_prm()
    oprm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    pose, _ = s.solve_gn(d["T_init"], gnp)
    want, *_ = oracle.optimal_tf_gauss_newton(w_pt, w_pl, None, d["T_init"], oprm)
    dt, dr = oracle.pose_err_split(pose, want)
    assert dt < 1e-5 and dr < 1e-5
    assert hostpath.counters()["pairings_uploads"] == c0["pairings_uploads"]
    # the token is consumed: the same solver call again uploads (nothing vouches for the device list)
    pose2, _ = s.solve_gn(d["T_init"], gnp)
    assert np.allclose(pose2, pose, atol=1e-12) and hostpath.counters()["pairings_uploads"] == c0["pairings_uploads"] + 1
    # pairings produced elsewhere
    sub = s.pairs_pt2pt()[::2].copy()
    s.set_pairings(sub, None)
    pose3, _ = s.solve_gn(d["T_init"], gnp)
    o = np.zeros(len(sub), oracle.PAIR_PT2PT)
    o["globalIdx"], o["localIdx"] = sub["globalIdx"], sub["localIdx"]
    o["gx"], o["gy"], o["gz"] = sub["global"].T
    o["lx"], o["ly"], o["lz"] = sub["local"].T
    o["errSq"] = sub["errorSquareAfterTransformation"]
    want3, *_ = oracle.optimal_tf_gauss_newton(o, None, None, d["T_init"], oprm)
    dt, dr = oracle.pose_err_split(pose3, want3)
    assert dt < 1e-5 and dr < 1e-5
    assert hostpath.counters()["pairings_uploads"] == c0["pairings_uploads"] + 2
    s.close()


def test_resident_list_fingerprint_and_strict_mode(oracle):
    """The solver takes the device-resident list only when the host list has its fingerprint (length, last
    record, the first 32 and every 64th record): the matcher's own list -> no upload; the same records in
    another order, a list cut short, a sampled record edited -> uploaded, and the solve is the solve of the
    HOST list.  Strict mode uploads always."""
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.make_pair(30_000, 300_000, 11)
    g, l = d["glob"], d["local"]
    s = hostpath.Session(g, l)
    gnp = _gn_prm()
    oprm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)

    def oracle_pose(P):
        o = np.zeros(len(P), oracle.PAIR_PT2PT)
        o["globalIdx"], o["localIdx"] = P["globalIdx"], P["localIdx"]
        o["gx"], o["gy"], o["gz"] = P["global"].T
        o["lx"], o["ly"], o["lz"] = P["local"].T
        o["errSq"] = P["errorSquareAfterTransformation"]
        return oracle.optimal_tf_gauss_newton(o, None, None, d["T_init"], oprm)[0]

    def fresh():
        s.begin_iteration()
        assert s.match_pt2pt(d["T_init"], _pt2pt_prm(1.0)) > 1000
        return s.pairs_pt2pt()

    def uploads():
        return hostpath.counters()["pairings_uploads"]

    P = fresh()
    u = uploads()
    pose, _ = s.solve_gn(d["T_init"], gnp)
    assert uploads() == u                                   # the matcher's own list
    dt, dr = oracle.pose_err_split(pose, oracle_pose(P))
    assert dt < 1e-5 and dr < 1e-5
    for name, edit in (("reversed", lambda Q: Q[::-1].copy()),
                       ("cut short", lambda Q: Q[:-1].copy()),
                       ("two records swapped at a sampled position", lambda Q: _swap(Q, 64, 65)),
                       ("last record edited", lambda Q: _poke(Q, len(Q) - 1))):
        P = fresh()
        Q = edit(P)
        s.set_pairings(Q, None)
        u = uploads()
        pose, _ = s.solve_gn(d["T_init"], gnp)
        assert uploads() == u + 1, name
        dt, dr = oracle.pose_err_split(pose, oracle_pose(Q))
        assert dt < 1e-5 and dr < 1e-5, name
    hostpath.set_strict(True)
    try:
        P = fresh()
        u = uploads()
        pose, _ = s.solve_gn(d["T_init"], gnp)
        assert uploads() == u + 1
        dt, dr = oracle.pose_err_split(pose, oracle_pose(P))
        assert dt < 1e-5 and dr < 1e-5
    finally:
        hostpath.set_strict(False)
    s.close()


def _swap(Q, i, j):
    Q = Q.copy()
    Q[[i, j]] = Q[[j, i]]
    return Q


def _poke(Q, i):
    Q = Q.copy()
    Q["local"][i] += np.float32(0.25)
    Q["localIdx"][i] ^= 1
    return Q


def test_layer_change_detection(oracle):
    """a layer edited in place between two aligns: found at ICP iteration 0 -- a layer first seen at its address is
    hashed in full, one verified before is re-verified on every 61st point, so an edit of >= 61 consecutive points (any
    bulk edit: motion compensation, a filter pass) is always seen; MP2P_HIP_HOST_STRICT=1 / release_layers() force the
    full hash.  Later iterations of one align only pay the 1024-point sample."""
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.random_cloud_pair(5000, 20000, 3)
    g, l = d["glob"], d["local"].copy()
    s = hostpath.Session(g, l)
    prm = _pt2pt_prm(0.8)
    s.begin_iteration()
    s.match_pt2pt(d["T_init"], prm, icp_iteration=0)
    c0 = hostpath.counters()
    s.begin_iteration()
    s.match_pt2pt(d["T_init"], prm, icp_iteration=0)  # unchanged: nothing re-uploaded
    assert hostpath.counters() == c0
    before = s.pairs_pt2pt()
    # an interior run that the 1024-point sample does not see, moved far away
    s._l[0][1237:1237 + 61] += 50.0
    lm = np.stack(s._l, 1)
    tree = oracle.KDTree(*_xyz(g))
    s.begin_iteration()
    s.match_pt2pt(d["T_init"], prm, icp_iteration=0)
    assert hostpath.counters()["cloud_uploads"] == c0["cloud_uploads"] + 1
    want, _ = oracle.match_pt2pt(*_xyz(g), *_xyz(lm), d["T_init"], 0.8, 0.0, tree=tree)
    _same_pt2pt(s.pairs_pt2pt(), want)
    assert len(before) != len(want) or not np.array_equal(before["localIdx"], want["localIdx"])
    s.close()


@pytest.mark.timeout(600)
def test_host_path_cost_at_full_size(oracle):
    """1 M x 10 M: a step through host containers (fresh MatchState, pairs into a host vector, marks,
    solver finding the list resident) stays close to the device-resident step (loose regression bound; the measured ratio is in bench.py); the lists are the
    device-resident path's lists."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import core, hostpath, synthetic
    d = synthetic.make_pair(1_000_000, 10_000_000, 1)
    g, l = d["glob"], d["local"]
    s = hostpath.Session(g, l)
    prm, gnp = _pt2pt_prm(2.0), _gn_prm()
    ctx = core.default_context()
    gmap = core.GlobalMap(ctx, *_xyz(g))
    cloud = core.LocalCloud(ctx, *_xyz(l))
    pairs = core.DevicePairs(ctx, l.shape[0], 0)

    def chain(step, n=8):
        pose, ts = d["T_init"].copy(), []
        for it in range(n):
            t0 = time.perf_counter()
            pose = step(pose, it)
            ts.append(time.perf_counter() - t0)
        return pose, float(np.median(ts[2:]))

    trace = []

    def host_step(pose, it):
        s.begin_iteration()
        n = s.match_pt2pt(pose, prm, icp_iteration=it)
        out = s.solve_gn(pose, gnp)[0]
        trace.append((it, n, hostpath.counters()["pairings_uploads"], s.last_ms(),
                      {k: round(v, 3) for k, v in hostpath.stage_ms().items()}))
        return out

    def dev_step(pose, it):
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
        return np.array(core.gn_solve(ctx, pairs, pose, gnp).pose)

    pose_d, t_dev = chain(dev_step)
    c0 = hostpath.counters()
    pose_h, t_host = chain(host_step)
    c1 = hostpath.counters()
    assert np.allclose(pose_h, pose_d, atol=1e-9)
    print("\n[host path] (iteration, pairs, pairings uploads so far, (match ms, solve ms)):", trace)
    assert c1["mstate_uploads"] == c0["mstate_uploads"] and c1["pairings_uploads"] == c0["pairings_uploads"], trace
    assert c1["map_uploads"] - c0["map_uploads"] <= 1 and c1["cloud_uploads"] - c0["cloud_uploads"] <= 1
    print(f"\n[host path] device-resident step {t_dev * 1e3:.3f} ms, host-container step {t_host * 1e3:.3f} ms")
    # (the wall-time bound lives under `-m perf`: tests/test_perf_bounds.py sets MP2P_PERF_ASSERTS; bench.py host_boundary reports the ratio)
    if os.environ.get("MP2P_PERF_ASSERTS") == "1":
        assert t_host < 3.0 * t_dev + 1e-3, (t_host, t_dev)
    s.close()


# ---- the SURVEY 8(f) classes through the adapter's host layer --------------------------------
def test_inlier_ratio_through_host_containers(oracle):
    from mp2p_icp_amd import _lib, hostpath, synthetic
    d = synthetic.random_cloud_pair(5000, 20000, 61, outlier_frac=0.2)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    s = hostpath.Session(g, l)
    for allow_global in (0, 1):
        prm = _lib.InlierRatioParams()
        prm.inliersRatio, prm.allowMatchAlreadyMatchedGlobalPoints = 0.5, allow_global
        prm.bounding_box_intersection_check_epsilon = 0.20
        for it, pose in enumerate((d["T_init"], d["T_gt"])):
            s.begin_iteration()
            n = s.match_inlier_ratio(pose, prm, icp_iteration=it)
            want, pot = oracle.match_inlier_ratio(*_xyz(g), *_xyz(l), pose, 0.5, allowMatchAlreadyMatchedGlobalPoints=bool(allow_global), tree=tree)
            got = s.pairs_pt2pt()
            assert n == len(want)
            _same_pt2pt(got, want)
            assert s.potential_pairings == pot
            # marks: every emitted pair, whatever the re-use flag (Matcher_Points_InlierRatio.cpp:127-131)
            assert set(np.flatnonzero(s.bits(1)).tolist()) == set(want["localIdx"].tolist())
            assert set(np.flatnonzero(s.bits(0)).tolist()) == set(want["globalIdx"].tolist())
    s.close()


def test_adaptive_through_host_containers(oracle):
    from mp2p_icp_amd import _lib, hostpath
    from test_gpu_matcher_adaptive import _scene
    g, l = _scene(72)
    kw = dict(confidenceInterval=0.8, firstToSecondDistanceMax=2.0, absoluteMaxSearchDistance=1.5, minimumCorrDist=0.1,
              enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8, planeMinimumFoundPoints=4)
    tree = oracle.KDTree(*_xyz(g))
    prm = _lib.AdaptiveParams()
    for k, v in kw.items():
        setattr(prm, k, v)
    prm.planeMinimumDistance, prm.planeEigenThreshold, prm.bounding_box_intersection_check_epsilon = 0.10, 0.01, 0.20
    s = hostpath.Session(g, l)
    for it, pose in enumerate((oracle.pose_from_xyzypr(0.03, -0.02, 0.01, 0.004, 0.0, -0.002), oracle.pose_identity())):
        r = oracle.match_adaptive(*_xyz(g), *_xyz(l), pose, tree=tree, **kw)
        s.begin_iteration()
        n_pt, n_pl, ci = s.match_adaptive(pose, prm, icp_iteration=it)
        assert ci == r["ci_high"]
        assert n_pt == len(r["pt2pt"]) and n_pl == len(r["pt2pl"]) and n_pl > 1000
        _same_pt2pt(s.pairs_pt2pt(), r["pt2pt"])
        got = s.pairs_pt2pl()
        assert np.allclose(got["plane"], r["pt2pl"]["plane"], rtol=0, atol=1e-9)
        assert np.array_equal(got["pt_local"], np.stack([r["pt2pl"]["lx"], r["pt2pl"]["ly"], r["pt2pl"]["lz"]], 1))
        assert s.potential_pairings == r["potential"]
        # local marks for both kinds, global marks never (Matcher_Adaptive.cpp:260, 289-293)
        assert set(np.flatnonzero(s.bits(1)).tolist()) == set(r["pl_local_idx"].tolist()) | set(r["pt2pt"]["localIdx"].tolist())
        assert not s.bits(0).any()
        # ... and the solver recognises the device-resident lists of BOTH kinds
        c0 = hostpath.counters()["pairings_uploads"]
        s.solve_gn(pose, _gn_prm())
        assert hostpath.counters()["pairings_uploads"] == c0
    s.close()


@pytest.mark.parametrize("method", [0, 1, 2])
def test_filter_decimate_through_host_containers(oracle, method):
    from mp2p_icp_amd import _lib, hostpath
    rng = np.random.default_rng(40 + method)
    pts = rng.normal(0, 3.0, (200_000, 3)).astype(np.float32)
    want, wsrc = oracle.filter_decimate_voxels(pts[:, 0], pts[:, 1], pts[:, 2], 0.25, method)
    s = hostpath.Session(pts[:10], pts)
    prm = _lib.DecimateParams()
    prm.voxel_filter_resolution, prm.decimate_method = 0.25, method
    got, src = s.filter_decimate_local(prm)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(src, wsrc)
    s.close()


def test_layer_cache_is_bounded_and_reseen_layers_are_cheap(oracle):
    """ADVICE r2 (unbounded cache) and VERDICT r2 #8: fresh layers per scan must not pile up in HBM; a layer seen
    again at ICP iteration 0 is verified on a stride, not hashed in full"""
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.random_cloud_pair(20_000, 60_000, 5)
    g = d["glob"]
    prm = _pt2pt_prm(0.8)
    base = hostpath.cache(max_layers=3)
    sessions = []
    for k in range(6):  # six "scans": six different local layers (different buffers), one map
        l = (d["local"] + np.float32(0.001 * k)).astype(np.float32)
        s = hostpath.Session(g, l)
        s.begin_iteration()
        s.match_pt2pt(d["T_init"], prm, icp_iteration=0)
        sessions.append(s)  # keep the buffers alive: distinct addresses
    c = hostpath.cache()
    assert c["evictions"] - base["evictions"] >= 2 and c["layers"] <= 3 + 3 + 1
    # the first session's cloud was evicted: matching it again uploads it again, results unchanged
    up0 = hostpath.counters()["cloud_uploads"]
    sessions[0].begin_iteration()
    n0 = sessions[0].match_pt2pt(d["T_init"], prm, icp_iteration=0)
    assert hostpath.counters()["cloud_uploads"] == up0 + 1 and n0 > 0
    # a re-seen layer at iteration 0 is hashed in full again (the default: the live layer decides) ...
    f0 = hostpath.cache()
    sessions[0].begin_iteration()
    sessions[0].match_pt2pt(d["T_init"], prm, icp_iteration=0)
    f1 = hostpath.cache()
    assert f1["full_checks"] == f0["full_checks"] + 2 and f1["reseen_checks"] == f0["reseen_checks"]
    # ... and on a stride only for a host that vouches for its layers (MP2P_HIP_HOST_TRUST_RESEEN=1)
    hostpath.set_trust_reseen(True)
    try:
        sessions[0].begin_iteration()
        sessions[0].match_pt2pt(d["T_init"], prm, icp_iteration=0)  # the stride's print is taken here
        f1 = hostpath.cache()
        sessions[0].begin_iteration()
        sessions[0].match_pt2pt(d["T_init"], prm, icp_iteration=0)
        f2 = hostpath.cache()
        assert f2["full_checks"] == f1["full_checks"] and f2["reseen_checks"] >= f1["reseen_checks"] + 2
    finally:
        hostpath.set_trust_reseen(False)
    # explicit release
    sessions[0].release_layers()
    assert hostpath.cache()["layers"] < f1["layers"]
    for s in sessions:
        s.close()


def test_quality_paired_ratio_counts_only_and_leaves_the_resident_list(oracle):
    """mp2p_icp_hip::QualityEvaluator_PairedRatio (QualityEvaluator_PairedRatio.cpp:45-73, reuse_icp_pairings = false):
    the private matcher on a fresh MatchState with global re-use allowed (:33-38); quality = pairs / potential pairings.
    Only the two counts come back, and the list the iteration's own matcher left on the device is still what the solver
    finds afterwards (no Pairings upload)."""
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.make_pair(60_000, 400_000, 17)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    s = hostpath.Session(g, l)
    prm, gnp = _pt2pt_prm(1.5), _gn_prm()
    qprm = _pt2pt_prm(0.6, allowMatchAlreadyMatchedGlobalPoints=1)
    pose = d["T_init"].copy()
    for it in range(3):
        s.begin_iteration()
        n = s.match_pt2pt(pose, prm, icp_iteration=it)
        got = s.pairs_pt2pt().copy()
        bits = (s.bits(0).copy(), s.bits(1).copy())
        # a quality checkpoint between the matcher and the solver (ICP.cpp:259-283 runs it after the solver; here it
        # is placed where it could do the most damage)
        q, hard, npairs = s.quality_paired_ratio(pose, qprm, absolute_minimum_pairing_ratio=0.20)
        want, pot = oracle.match_pt2pt(*_xyz(g), *_xyz(l), pose, 0.6, 0.0, tree=tree, allowMatchAlreadyMatchedGlobalPoints=True)
        assert npairs == len(want) and pot == l.shape[0]
        assert q == len(want) / pot and hard == (q < 0.20)
        # nothing of the running iteration was touched
        assert n == len(s.pairs_pt2pt()) and np.array_equal(s.pairs_pt2pt(), got)
        assert np.array_equal(s.bits(0), bits[0]) and np.array_equal(s.bits(1), bits[1])
        c0 = hostpath.counters()["pairings_uploads"]
        pose, _ = s.solve_gn(pose, gnp)
        assert hostpath.counters()["pairings_uploads"] == c0
    # a threshold nothing passes: quality 0, discarded
    q, hard, npairs = s.quality_paired_ratio(pose, _pt2pt_prm(1e-4, allowMatchAlreadyMatchedGlobalPoints=1))
    assert npairs == 0 and q == 0.0 and hard
    s.close()


def test_copy_out_24_bytes_per_pair_equals_the_record_form(oracle, monkeypatch):
    """round 6: the point pairings come back as index arrays + {global point, squared error} (24 bytes per pair on the link), the
    36-byte records assembled on the host with `local` read from the caller's own layer arrays
    (mp2p_hip_pairs_copy_pt2pt_begin_soa).  Byte-identical to the record form (MP2P_HIP_HOST_COPY_SOA=0) and to the oracle, over a
    pose chain whose lists are long enough for the staged path (> 256 KB)."""
    from mp2p_icp_amd import hostpath, synthetic
    d = synthetic.make_pair(60_000, 400_000, 23)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    prm = _pt2pt_prm(1.5, allowMatchAlreadyMatchedGlobalPoints=1)   # (every local point keeps its pair: a list beyond 256 KB)
    lists = {}
    for form in ("1", "0"):
        monkeypatch.setenv("MP2P_HIP_HOST_COPY_SOA", form)
        s = hostpath.Session(g, l)
        pose, got = d["T_init"].copy(), []
        for it in range(3):
            s.begin_iteration()
            n = s.match_pt2pt(pose, prm, icp_iteration=it)
            got.append(s.pairs_pt2pt().copy())
            assert n == len(got[-1]) and n * 36 > 256 * 1024
            pose = s.solve_gn(pose, _gn_prm())[0]
        lists[form] = got
        s.close()
    want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_init"], 1.5, 0.0, tree=tree,
                                 allowMatchAlreadyMatchedGlobalPoints=True)
    _same_pt2pt(lists["1"][0], want)
    for a, b in zip(lists["1"], lists["0"]):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("tune", ["", "copy_chunk_kb=64", "copy_chunk_kb=4"])
def test_copy_pt2pt_begin_soa_entry_point(oracle, monkeypatch, tune):
    """the C entry point itself (one chunk, many chunks shared by the helper thread and the caller; a range that starts inside the
    list): records byte-equal to the plain download; local arrays that do not cover a pairing's localIdx are reported by _end"""
    import ctypes as C
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    if tune:
        monkeypatch.setenv("MP2P_HIP_TUNE", tune)
    d = synthetic.make_pair(60_000, 400_000, 24)
    g, l = d["glob"], d["local"]
    ctx = amd.Context(0)
    gmap, cloud = core.GlobalMap(ctx, *_xyz(g)), core.LocalCloud(ctx, *_xyz(l))
    pairs = core.DevicePairs(ctx, l.shape[0], 0)
    core.match_pt2pt(ctx, gmap, cloud, d["T_init"], _pt2pt_prm(1.5, allowMatchAlreadyMatchedGlobalPoints=1), None, pairs)
    want = pairs.download_pt2pt()
    n = len(want)
    assert n * 36 > 256 * 1024
    L = ctx._L
    raw = L.mp2p_hip_host_alloc(ctx.handle, 8 * n)
    li, gi = C.cast(raw, C.POINTER(C.c_uint32)), C.cast(raw + 4 * n, C.POINTER(C.c_uint32))
    lx, ly, lz = (np.ascontiguousarray(l[:, k]) for k in range(3))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    for first in (0, 777):
        m = n - first
        out = np.zeros(m, _lib.PAIR_PT2PT)
        _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin_soa(ctx.handle, pairs.handle, first, m, out.ctypes.data, li, gi, fp(lx), fp(ly), fp(lz),
                                                         l.shape[0], 0), ctx.handle)
        _lib.check(L.mp2p_hip_pairs_copy_wait_idx(ctx.handle), ctx.handle)
        assert np.array_equal(np.ctypeslib.as_array(li, (m,)), want["localIdx"][first:])
        _lib.check(L.mp2p_hip_pairs_copy_end(ctx.handle), ctx.handle)
        assert out.tobytes() == want[first:].tobytes()
    out = np.zeros(n, _lib.PAIR_PT2PT)
    _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin_soa(ctx.handle, pairs.handle, 0, n, out.ctypes.data, li, gi, fp(lx), fp(ly), fp(lz), 100, 0),
               ctx.handle)
    assert L.mp2p_hip_pairs_copy_end(ctx.handle) == _lib.ERR_INVALID
    assert b"localIdx" in L.mp2p_hip_last_error(ctx.handle)
    L.mp2p_hip_host_free(ctx.handle, raw)
