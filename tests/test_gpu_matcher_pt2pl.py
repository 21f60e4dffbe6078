"""GPU parity of Matcher_Point2Plane (K5) against the CPU oracle's declared nn_search_pt2pl
semantics, the committed golden vectors and the expectations of the reference's (disabled)
tests/test-mp2p_matcher_pt2pl.cpp.  Which local points are paired must match exactly; plane
coefficients / centroids to 1e-9 (fp64 eigen-solver on both sides)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


@pytest.fixture(autouse=True, params=["box", "ball"])
def pl_kernel(request, monkeypatch):
    """every test of this module with BOTH search kernels of Matcher_Point2Plane: the box-rule tile kernel of rounds 2-5
    (pt2pl_tile_kernel) and round 6's ball-rule / matrix-pipe kernel (pt2pl_seltile_kernel, nn_pl_seltile.hip).  The library
    picks by the layer's size (the second one above 524 288 queries); the knob pl_select forces one.  Read when a context is
    created (the tests on core.Context), and set on the shared context of the Matcher classes."""
    from mp2p_icp_amd import core
    monkeypatch.setenv("MP2P_HIP_TUNE", "pl_select=%d" % (request.param == "ball"))
    core.default_context().set_tune("pl_select=%d" % (request.param == "ball"))  # (the Matcher classes' shared context)
    yield request.param
    core.default_context().set_tune("pl_select=-1")


def _match(amd, g, l, pose, params, ms=None, layer_kw=None):
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g, **(layer_kw or {}))})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Point2Plane()
    m.initialize(params)
    pairs = amd.Pairings()
    ms = ms or amd.MatchState(pcG, pcL)
    assert m.match(pcG, pcL, pose, amd.MatchContext(), ms, pairs)
    return pairs, ms


def _check(pairs, want_pl, want_idx):
    got, gidx = pairs.paired_pt2pl, pairs.paired_pt2pl_local_idx
    assert len(got) == len(want_pl), (len(got), len(want_pl))
    assert np.array_equal(gidx, want_idx)
    if len(got):
        assert np.allclose(got["plane"], want_pl["plane"], rtol=0, atol=1e-9)
        assert np.allclose(got["centroid"], want_pl["centroid"], rtol=0, atol=1e-9)
        want_l = np.stack([want_pl["lx"], want_pl["ly"], want_pl["lz"]], 1)
        assert np.array_equal(got["pt_local"], want_l)


def test_reference_pt2pl_expectations(amd):
    """tests/test-mp2p_matcher_pt2pl.cpp:75-131"""
    from test_oracle_kat import PT2PL_MATCH_PRM, _kat_local, pt2pl_kat_global
    g, l = pt2pl_kat_global(), _kat_local()
    P = dict(PT2PL_MATCH_PRM)

    def run(p6):
        pairs, _ = _match(amd, g, l, amd.se3.from_xyzypr(*p6), P)
        return pairs

    assert run((0, 0, 0, 0, 0, 0)).empty()
    assert len(run((0, 5, 0, 0, 0, 0)).paired_pt2pl) == 1
    p = run((8.04, 0, 0, 0, 0, 0)).paired_pt2pl
    assert len(p) == 1
    assert np.allclose(p[0]["pt_local"], (2, 0, 0), atol=1e-3)
    assert np.allclose(p[0]["centroid"], (10, 0, 0), atol=0.01)
    assert np.allclose(p[0]["plane"], (1, 0, 0, -10), atol=1e-3)
    assert len(run((18.053, 0.05, 0.03, 0, 0, 0)).paired_pt2pl) == 0


def test_golden_pt2pl(amd):
    gold = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))
    dt, sr, knn, mpp, pet = gold["pt2pl_params"]
    pairs, _ = _match(amd, gold["pt2pl_glob"], gold["pt2pl_local"], amd.se3.identity(),
                      dict(distanceThreshold=float(dt), searchRadius=float(sr), knn=int(knn),
                           minimumPlanePoints=int(mpp), planeEigenThreshold=float(pet)))
    _check(pairs, gold["pt2pl_pairs"], gold["pt2pl_local_idx"])
    assert pairs.potential_pairings == gold["pt2pl_potential"][0]


@pytest.mark.parametrize("variant", [0, 1, 2, 4])   # index variants (mp2p_hip_map_params::no_occupancy_bitmap)
@pytest.mark.parametrize("knn,minpts,radius,eig", [(5, 5, 0.4, 0.05), (8, 6, 0.6, 0.02),
                                                   (12, 5, 0.3, 0.1), (16, 10, 0.8, 0.05)])
def test_random_parity_vs_oracle(amd, oracle, knn, minpts, radius, eig, variant):
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(6000, 60000, 77 + knn)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    for pose in (d["T_gt"], d["T_init"]):
        want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2],
                                             pose, 0.25, radius, knn, minpts, eig, tree=tree)
        pairs, _ = _match(amd, g, l, pose, dict(distanceThreshold=0.25, searchRadius=radius, knn=knn,
                                                minimumPlanePoints=minpts, planeEigenThreshold=eig),
                          layer_kw=dict(no_occupancy_bitmap=variant))
        _check(pairs, want, widx)
        assert pairs.potential_pairings == pot


def test_local_taken_and_state(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(3000, 40000, 5)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    rng = np.random.default_rng(1)
    lt0 = (rng.random(l.shape[0]) < 0.4).astype(np.uint8)
    P = dict(distanceThreshold=0.3, searchRadius=0.5, knn=6, minimumPlanePoints=5, planeEigenThreshold=0.05)
    for allow in (False, True):
        lt = lt0.copy()
        want, widx, _ = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2],
                                           d["T_gt"], tree=tree, local_taken=lt,
                                           allowMatchAlreadyMatchedPoints=allow, **P)
        pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
        pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
        ms = amd.MatchState(pcG, pcL)
        ms.for_layers("raw", "raw").upload(None, lt0)
        m = amd.Matcher_Point2Plane()
        m.initialize(dict(P, allowMatchAlreadyMatchedPoints=allow))
        pairs = amd.Pairings()
        m.match(pcG, pcL, d["T_gt"], amd.MatchContext(), ms, pairs)
        _check(pairs, want, widx)
        _, l_after = ms.for_layers("raw", "raw").download()
        assert np.array_equal(l_after, lt)


def test_pt2pl_then_pt2pt_pipeline_and_gn(amd, oracle):
    """run_matchers with both matchers (tests/test-mp2p_matcher_pt2pl.cpp:136-167) and a GN solve
    over the mixed pairings."""
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(5000, 50000, 9)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    P = dict(distanceThreshold=0.3, searchRadius=0.6, knn=6, minimumPlanePoints=5, planeEigenThreshold=0.05)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    for allow in (True, False):
        mpl = amd.Matcher_Point2Plane()
        mpl.initialize(P)
        mpt = amd.Matcher_Points_DistanceThreshold()
        mpt.initialize({"threshold": 0.3, "thresholdAngularDeg": 0.0, "allowMatchAlreadyMatchedPoints": allow})
        pairs = amd.run_matchers([mpl, mpt], pcG, pcL, d["T_init"])
        lt = np.zeros(l.shape[0], np.uint8)
        gt = np.zeros(g.shape[0], np.uint8)
        wpl, widx, _ = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2],
                                          d["T_init"], tree=tree, local_taken=lt, **P)
        wpt, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_init"],
                                    0.3, 0.0, tree=tree, allowMatchAlreadyMatchedPoints=allow,
                                    local_taken=lt, global_taken=gt)
        _check(pairs, wpl, widx)
        got = pairs.paired_pt2pt
        assert np.array_equal(got["localIdx"], wpt["localIdx"]) and np.array_equal(got["globalIdx"], wpt["globalIdx"])
        s = amd.Solver_GaussNewton()
        s.initialize({"maxIterations": 4, "robustKernel": "RobustKernel::Cauchy", "robustKernelParam": 0.2})
        sc = amd.SolverContext()
        sc.guessRelativePose = d["T_init"]
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        To, *_ = oracle.optimal_tf_gauss_newton(wpt, wpl, None, d["T_init"],
                                                oracle.make_gn_params(4, kernel=oracle.KERNEL_CAUCHY, kernelParam=0.2))
        dt, dr = oracle.pose_err_split(out.optimalPose, To)
        assert dt < 1e-5 and dr < 1e-5


def test_pt2pl_max_local_points_visit_order(amd, oracle):
    """Matcher_Point2Plane.cpp:58-59, 81: the same visit list as the point matcher"""
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(5000, 50000, 19)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    order = np.random.default_rng(2).permutation(3000).astype(np.uint32)
    P = dict(distanceThreshold=0.3, searchRadius=0.5, knn=6, minimumPlanePoints=5, planeEigenThreshold=0.05)
    want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2],
                                         d["T_gt"], tree=tree, idxs=order, **P)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Point2Plane()
    m.initialize(dict(P, maxLocalPointsPerLayer=3000, localPointsSampleSeed=1))
    m.visit_order_fn = lambda n, mx, seed: order
    pairs = amd.Pairings()
    assert m.match(pcG, pcL, d["T_gt"], amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
    _check(pairs, want, widx)
    assert pairs.potential_pairings == pot == l.shape[0]  # :54: pcLocal.size(), not the visited subset


@pytest.mark.parametrize("knn,radius", [(5, 0.4), (8, 0.6), (16, 0.8)])
def test_warm_start_pose_sequence(amd, oracle, knn, radius):
    """the search's warm start (start radius = previous k-th distance + displacement): a sequence of poses on ONE
    context / map / cloud -- small steps, a jump, a return -- every call equal to the oracle's cold result; and
    MP2P_HIP_TUNE pl_warm=0 (own context) gives the same lists"""
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.make_pair(6000, 60000, 177 + knn)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    rng = np.random.default_rng(knn)
    poses = [d["T_init"]]
    for k in range(5):
        step = np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.002, 3)]) * (30.0 if k == 2 else 1.0)  # k = 2: a jump
        poses.append(amd.se3.compose(poses[-1], amd.se3.exp(step)))
    poses.append(d["T_init"])  # and back
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    prm = _lib.Pt2PlParams()
    prm.distanceThreshold, prm.searchRadius, prm.knn, prm.minimumPlanePoints, prm.planeEigenThreshold = 0.25, radius, knn, 5, 0.05
    prm.bounding_box_intersection_check_epsilon = 0.20
    for pose in poses:
        want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.25, radius, knn, 5, 0.05, tree=tree)
        pairs.clear()
        core.match_pt2pl(ctx, gmap, cloud, pose, prm, None, pairs)
        got, gidx = pairs.download_pt2pl()
        assert len(got) == len(want), (len(got), len(want))
        assert np.array_equal(gidx, widx)
        if len(got):
            assert np.allclose(got["plane"], want["plane"], rtol=0, atol=1e-9)
            assert np.allclose(got["centroid"], want["centroid"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("knn,radius", [(5, 0.4), (8, 0.6)])
def test_certificate_pose_sequence(amd, oracle, knn, radius):
    """the search's certificate (nn_pt2pl.hip, PlArgs::lb_io): poses that creep from millimetre to 50-micrometre steps on
    ONE context -- a growing share of the queries skips the search because their previous neighbours are provably
    still the nearest -- every call equal to the oracle's cold result, bit for bit; a local point that is taken in one
    call and free in the next goes through the search again; a jump invalidates everything"""
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.make_pair(8000, 400000, 277 + knn)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    rng = np.random.default_rng(knn)
    poses, scales = [d["T_gt"]], [1e-3, 1e-3, 3e-4, 1e-4, 5e-5, 5e-5, 3e-2, 5e-5]
    for sc in scales:
        poses.append(amd.se3.compose(poses[-1], amd.se3.exp(np.concatenate([rng.normal(0, sc, 3), rng.normal(0, 0.1 * sc, 3)]))))
    ctx = amd.Context(0)
    ctx.set_profiling(1)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    ms = core.DeviceMatchState(ctx, g.shape[0], l.shape[0])
    prm = _lib.Pt2PlParams()
    prm.distanceThreshold, prm.searchRadius, prm.knn, prm.minimumPlanePoints, prm.planeEigenThreshold = 0.25, radius, knn, 5, 0.05
    prm.bounding_box_intersection_check_epsilon = 0.20
    certified = []
    for k, pose in enumerate(poses):
        taken = np.zeros(l.shape[0], np.uint8)
        if k in (2, 5):
            taken[rng.choice(l.shape[0], 500, replace=False)] = 1  # these are skipped by this call only
        ms.upload(np.zeros(g.shape[0], np.uint8), taken)
        want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.25, radius, knn, 5, 0.05,
                                             tree=tree, local_taken=taken.copy())
        pairs.clear()
        core.match_pt2pl(ctx, gmap, cloud, pose, prm, ms, pairs)
        st = ctx.stats()
        certified.append(st["pl_certified"])
        got, gidx = pairs.download_pt2pl()
        assert len(got) == len(want) and len(want) > 500, (k, len(got), len(want))
        assert np.array_equal(gidx, widx), k
        assert np.allclose(got["plane"], want["plane"], rtol=0, atol=1e-9) and np.allclose(got["centroid"], want["centroid"], rtol=0, atol=1e-9), k
    per_call = np.diff([0] + certified)
    assert per_call[0] == 0                      # nothing to go by at the first call
    assert per_call[4] > 0.3 * l.shape[0], per_call  # 0.1 mm steps: a good share of the queries is certified
    assert per_call[7] < per_call[5], per_call   # the 3 cm jump leaves (next to) nothing certified


def test_certificate_far_from_the_origin(amd, oracle):
    """the certificate's margins are tied to the MAP (slack = 2^-20 of its extent / largest coordinate)
    while the rounding of a computed distance is tied to the COORDINATES.  A map 1.4 km from the origin and 500 m wide
    (fp32 spacing 0.12 mm there, slack 1.3 mm): creeping poses, every call equal to the oracle's cold result."""
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.make_pair(6000, 300000, 911)
    rng = np.random.default_rng(4)
    off = np.array([1000.0, 1000.0, 20.0])
    g = d["glob"].astype(np.float64) + off
    # a few far points make the layer 500 m wide (the slack follows the extent)
    far = np.stack([rng.uniform(off[0] - 250, off[0] + 250, 2000), rng.uniform(off[1] - 250, off[1] + 250, 2000),
                    rng.uniform(off[2] - 5, off[2] + 30, 2000)], 1)
    g = np.ascontiguousarray(np.concatenate([g, far]).astype(np.float32))
    l = d["local"]
    T0 = d["T_gt"].copy()
    T0[9:12] += off
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    poses, scales = [T0], [1e-3, 5e-4, 2e-4, 1e-4, 1e-4, 5e-5, 5e-5, 2e-5, 2e-5, 1e-3]
    for sc in scales:
        poses.append(amd.se3.compose(poses[-1], amd.se3.exp(np.concatenate([rng.normal(0, sc, 3), rng.normal(0, 0.05 * sc, 3)]))))
    ctx = amd.Context(0)
    ctx.set_profiling(1)
    gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    assert gmap.info()["bbox_max"][0] - gmap.info()["bbox_min"][0] > 450.0
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    certified = []
    for k, pose in enumerate(poses):
        n = _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree)
        assert n > 300, (k, n)
        certified.append(ctx.stats()["pl_certified"])
    per_call = np.diff([0] + certified)
    assert per_call[0] == 0 and per_call[1:-1].sum() > 0, per_call  # the rule still fires out there -- and never wrongly


@pytest.mark.timeout(900)
def test_certificate_long_creep(amd, oracle):
    """2 000 calls creeping 20 micrometres each in ONE direction (4 cm in all: neighbour lists change
    many times on the way).  A certified query keeps its previous bound minus the step and slack / 4 -- the bound decays
    call after call until a search refreshes it; it must never certify a list that the cold search would not return.
    Checked against the oracle at every 20th call, at the 30 last calls, and wherever the certified share jumps."""
    from mp2p_icp_amd import core, synthetic
    d = synthetic.make_pair(3000, 400000, 577)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    step = amd.se3.exp(np.array([1.4e-5, 1.0e-5, 1.0e-5, 1.5e-7, -1.0e-7, 2.0e-7]))  # 20.5 um + 0.27 urad per call
    ctx = amd.Context(0)
    ctx.set_profiling(1)
    gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    prm = _pl_prm()
    pose, N = d["T_gt"], 2000
    cert_prev, checked, cert_total = 0, 0, 0
    for k in range(N):
        if k % 20 == 0 or k >= N - 30:
            assert _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree) > 100, k
            checked += 1
        else:
            pairs.clear()
            core.match_pt2pl(ctx, gmap, cloud, pose, prm, None, pairs)
        c = ctx.stats()["pl_certified"]
        cert_total, cert_prev = c, c
        pose = amd.se3.compose(pose, step)
    assert checked >= 129
    # the certificate carried a good part of the work (otherwise this test would not exercise the decay at all)
    assert cert_total > 50 * N, cert_total


def _pl_prm(radius=0.4, knn=5):
    from mp2p_icp_amd import _lib
    prm = _lib.Pt2PlParams()
    prm.distanceThreshold, prm.searchRadius, prm.knn, prm.minimumPlanePoints, prm.planeEigenThreshold = 0.25, radius, knn, 5, 0.05
    prm.bounding_box_intersection_check_epsilon = 0.20
    return prm


def _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree, radius=0.4, knn=5, idxs=None):
    want, widx, _ = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, 0.25, radius, knn, 5, 0.05,
                                       tree=tree, idxs=idxs)
    pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, pose, _pl_prm(radius, knn), None, pairs)
    got, gidx = pairs.download_pt2pl()
    assert np.array_equal(gidx, widx), (len(gidx), len(widx))
    if len(got):
        assert np.allclose(got["plane"], want["plane"], rtol=0, atol=1e-9)
    return len(widx)


@pytest.mark.parametrize("n_local", [1, 7, 9, 64, 65, 257, 513])
def test_certificate_tiny_layers(amd, oracle, n_local):
    """query lists, their padding and the certificate on layers of a few points (one partial block, one partial tile)"""
    from mp2p_icp_amd import core, synthetic
    d = synthetic.make_pair(2000, 150000, 31)
    g, l = d["glob"], np.ascontiguousarray(d["local"][:: max(1, 2000 // n_local)][:n_local])
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    ctx = amd.Context(0)
    gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    pose = d["T_gt"]
    for k in range(4):
        _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree)
        pose = amd.se3.compose(pose, amd.se3.exp(np.array([2e-4, -1e-4, 1e-4, 0, 0, 2e-5])))


def test_hard_class_overflow_and_visit_order_changes(amd, oracle, monkeypatch):
    """every query in the hard class (threshold 1): the hard list overflows its capacity (an eighth of the layer) and the
    rest stays in the easy class; then a visit list that changes between calls -- a point that was not visited has no
    certificate and goes through the search"""
    from mp2p_icp_amd import core, synthetic
    monkeypatch.setenv("MP2P_HIP_TUNE", os.environ["MP2P_HIP_TUNE"] + ",pl_hard_cand=1,pl_sel_hard_cand=1")
    d = synthetic.make_pair(6000, 300000, 41)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    ctx = amd.Context(0)
    gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    pose, n = d["T_gt"], 0
    for k in range(4):
        n += _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree)
        pose = amd.se3.compose(pose, amd.se3.exp(np.array([3e-4, 1e-4, -1e-4, 0, 0, 3e-5])))
    assert n > 1000
    rng = np.random.default_rng(5)
    for k in range(4):
        order = rng.permutation(l.shape[0])[: 2500 + 500 * k].astype(np.uint32)
        cloud.set_visit_order(order)
        _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree, idxs=order)
        pose = amd.se3.compose(pose, amd.se3.exp(np.array([1e-4, 1e-4, 0, 0, 0, 1e-5])))
    cloud.set_visit_order(None)
    _pl_equal(oracle, core, ctx, gmap, cloud, pairs, g, l, pose, tree)


@pytest.mark.timeout(900)
def test_large_layer_hard_first_order_does_not_change_the_lists(amd, oracle):
    """round 6: above 524 288 queries the ball-rule kernel serves its hard class with single waves -- the class is only the dispatch
    ORDER (the tiles that staged the most at the previous call first).  A 600 000-point layer along a warm chain: Morton order
    (pl_sel_hard_large=0), the default threshold, a threshold every tile exceeds (the hard list, n / 16 entries, overflows and the
    rest stays in the easy class) -- the same lists from every one, and the oracle's at the last pose."""
    from mp2p_icp_amd import core, synthetic
    d = synthetic.make_scan_union_pair(600_000, 2_000_000, 77, map_scan_points=400_000)
    g, l = d["glob"], d["local"]
    assert l.shape[0] > 524_288
    rng = np.random.default_rng(3)
    poses = [d["T_init"]]
    for step in (0.05, 0.004, 0.0004):
        poses.append(amd.se3.compose(poses[-1], amd.se3.exp(np.concatenate([rng.normal(0, step, 3), rng.normal(0, 0.1 * step, 3)]))))
    ref = None
    for knob in ("pl_sel_hard_large=0", "pl_sel_hard_large=1500", "pl_sel_hard_large=1", "pl_sel_hard_large=1,pl_cert_step_mm=0"):
        ctx = amd.Context(0)
        ctx.set_tune(knob)
        gmap, cloud = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2]), core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
        pairs = core.DevicePairs(ctx, 0, l.shape[0])
        got = []
        for pose in poses:
            pairs.clear()
            core.match_pt2pl(ctx, gmap, cloud, pose, _pl_prm(0.4, 5), None, pairs)
            rec, idx = pairs.download_pt2pl()
            got.append((idx.copy(), rec["plane"].copy(), rec["centroid"].copy()))
        if ref is None:
            ref = got
            tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
            want, widx, _ = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], poses[-1], 0.25, 0.4, 5, 5, 0.05, tree=tree)
            assert np.array_equal(got[-1][0], widx) and len(widx) > 50_000
            assert np.allclose(got[-1][1], want["plane"], rtol=0, atol=1e-9)
        else:
            for k, (a, b) in enumerate(zip(ref, got)):
                assert np.array_equal(a[0], b[0]), (knob, k)
                assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (knob, k)
