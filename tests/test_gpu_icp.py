"""End-to-end: the ICP::align-shaped driver over the HIP matcher + solver
(tests/test-mp2p_icp_algos.cpp:52-233: bunny decimated x10, random pose within +-15 % bbox /
+-10 deg, threshold = 0.40*max_dim, thresholdAngularDeg = 0, 100 iterations, |log SE3 err| < 0.1),
and iteration-by-iteration agreement with the same loop run on the CPU oracle."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _bunny():
    pts = np.loadtxt(gzip.open(os.path.join(HERE, "golden", "bunny_decim.xyz.gz"))).astype(np.float32)
    return pts[::10].copy()  # decimation 10 (test-mp2p_icp_algos.cpp:62-72)


@pytest.mark.parametrize("solver_name", ["GaussNewton", "Horn"])
def test_bunny_icp_converges(oracle, solver_name):
    import mp2p_icp_amd as amd
    pts = _bunny()
    max_dim = float((pts.max(0) - pts.min(0)).max())
    rng = np.random.default_rng(1234)
    for rep in range(5):
        t = rng.uniform(-0.15, 0.15, 3) * max_dim
        r = np.deg2rad(rng.uniform(-10, 10, 3))
        gt = amd.se3.from_xyzypr(*t, *r)
        R, tt = amd.se3.Rt(gt)
        # global = gt (+) local  => pose of local w.r.t. global is gt
        glob = (pts.astype(np.float64) @ R.T + tt).astype(np.float32)
        pcG = amd.metric_map_t({"raw": amd.PointLayer(glob)})
        pcL = amd.metric_map_t({"raw": amd.PointLayer(pts)})
        m = amd.Matcher_Points_DistanceThreshold()
        m.initialize({"threshold": 0.40 * max_dim, "thresholdAngularDeg": 0.0})
        if solver_name == "GaussNewton":
            s = amd.Solver_GaussNewton()
            s.initialize({"maxIterations": 10})
        else:
            s = amd.Solver_Horn()
            s.initialize({})
        icp = amd.ICP()
        icp.set_matchers([m])
        icp.set_solvers([s])
        res = icp.align(pcL, pcG, amd.se3.identity(), amd.Parameters(maxIterations=100))
        err = float(np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(res.optimal_tf, gt))))
        assert err < 0.1, (rep, err, res.nIterations)
        assert res.terminationReason in (amd.IterTermReason.Stalled, amd.IterTermReason.MaxIterations)


def test_icp_iterations_track_the_oracle(oracle):
    """same outer loop on the oracle: identical pair lists and poses (1e-5) at every iteration"""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(8000, 80000, 42, max_t=0.2, max_r_deg=1.0)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": 1.0, "thresholdAngularDeg": 0.0})
    s = amd.Solver_GaussNewton()
    s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure", "robustKernelParam": 0.15})
    pose_h = d["T_init"].copy()
    pose_o = d["T_init"].copy()
    prm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    for it in range(8):
        pairs = amd.run_matchers([m], pcG, pcL, pose_h, amd.MatchContext(it))
        want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose_o, 1.0, 0.0, tree=tree)
        got = pairs.paired_pt2pt
        # poses agree to ~1e-12, far below fp32 resolution of the transformed points: the
        # correspondence lists stay identical along the whole chain
        assert np.array_equal(got["localIdx"], want["localIdx"]), it
        assert np.array_equal(got["globalIdx"], want["globalIdx"]), it
        sc = amd.SolverContext()
        sc.guessRelativePose = pose_h
        sc.icpIteration = it
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        pose_h = out.optimalPose
        pose_o, *_ = oracle.optimal_tf_gauss_newton(want, None, None, pose_o, prm)
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)


@pytest.mark.parametrize("planes", [False, True])
def test_kitti_shaped_pipeline_tracks_the_oracle(oracle, planes):
    """demos/icp-settings-kitti.yaml: Matcher_Points_DistanceThreshold (2.0 m) for the first
    iterations, Matcher_Adaptive (0.75 / 1.2 / 2.0 m) afterwards, Gauss-Newton 3 x GemanMcClure 0.15;
    with plane detection the solver gets point and plane pairings.  Same lists and poses as the
    oracle loop at every iteration."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(6000, 60000, 77, max_t=0.15, max_r_deg=0.5)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m1 = amd.Matcher_Points_DistanceThreshold()
    m1.initialize({"threshold": 2.0, "thresholdAngularDeg": 0.0, "runFromIteration": 0, "runUpToIteration": 2})
    A = dict(confidenceInterval=0.75, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=2.0,
             enableDetectPlanes=planes, planeSearchPoints=8, planeMinimumFoundPoints=4, planeMinimumDistance=0.3)
    m2 = amd.Matcher_Adaptive()
    m2.initialize(dict(A, runFromIteration=3, runUpToIteration=0))
    s = amd.Solver_GaussNewton()
    s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure", "robustKernelParam": 0.15})
    prm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    pose_h, pose_o = d["T_init"].copy(), d["T_init"].copy()
    n_pl = 0
    for it in range(7):
        pairs = amd.run_matchers([m1, m2], pcG, pcL, pose_h, amd.MatchContext(it))
        if it <= 2:
            want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose_o, 2.0, 0.0, tree=tree)
            want_pl = None
        else:
            r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose_o, tree=tree, **A)
            want, want_pl = r["pt2pt"], r["pt2pl"]
            assert np.array_equal(pairs.paired_pt2pl_local_idx, r["pl_local_idx"]), it
            assert m2.last_ci_high == r["ci_high"], it
            n_pl += len(want_pl)
        got = pairs.paired_pt2pt
        assert np.array_equal(got["localIdx"], want["localIdx"]), it
        assert np.array_equal(got["globalIdx"], want["globalIdx"]), it
        sc = amd.SolverContext()
        sc.guessRelativePose = pose_h
        sc.icpIteration = it
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        pose_h = out.optimalPose
        pose_o, *_ = oracle.optimal_tf_gauss_newton(want, want_pl, None, pose_o, prm)
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)
    assert (n_pl > 500) == planes
    dt, dr = oracle.pose_err_split(pose_h, d["T_gt"])
    assert dt < 0.1 and dr < 0.01, (dt, dr)


def test_quality_and_covariance_of_align(oracle):
    """ICP.cpp:316-337: Results::quality from QualityEvaluator_PairedRatio (both modes) and the
    covariance of the final pairings; a quality checkpoint aborts a hopeless registration."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(4000, 12000, 13, outlier_frac=0.1)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(d["glob"])})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(d["local"])})

    def make(thr):
        icp = amd.ICP()
        m = amd.Matcher_Points_DistanceThreshold()
        m.initialize({"threshold": thr, "thresholdAngularDeg": 0.0})
        s = amd.Solver_GaussNewton()
        s.initialize({"maxIterations": 3})
        icp.set_matchers([m])
        icp.set_solvers([s])
        return icp
    icp = make(0.8)
    res = icp.align(pcL, pcG, d["T_init"], amd.Parameters(maxIterations=15))
    P = res.finalPairings
    assert res.quality == pytest.approx(P.size() / P.potential_pairings)          # reuse_icp_pairings
    o = np.zeros(len(P.paired_pt2pt), oracle.PAIR_PT2PT)
    hp = P.paired_pt2pt
    o["gx"], o["gy"], o["gz"] = hp["global"].T
    o["lx"], o["ly"], o["lz"] = hp["local"].T
    want, _, ok = oracle.covariance(o, None, None, None, res.optimal_tf)
    assert ok and np.allclose(res.optimal_tf_cov, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    # a fresh matcher that may pair a global point several times
    q = amd.QualityEvaluator_PairedRatio()
    q.initialize({"reuse_icp_pairings": False, "threshold": 0.1, "thresholdAngularDeg": 0.0})
    quality, hard = q.evaluate(pcG, pcL, res.optimal_tf, None)
    tree = oracle.KDTree(d["glob"][:, 0], d["glob"][:, 1], d["glob"][:, 2])
    w, pot = oracle.match_pt2pt(d["glob"][:, 0], d["glob"][:, 1], d["glob"][:, 2], d["local"][:, 0], d["local"][:, 1],
                                d["local"][:, 2], res.optimal_tf, 0.1, 0.0, tree=tree,
                                allowMatchAlreadyMatchedGlobalPoints=True)
    assert quality == pytest.approx(len(w) / pot) and hard == (quality < 0.20)
    # hopeless start + a checkpoint at iteration 2
    far = amd.se3.compose(d["T_gt"], amd.se3.from_xyzypr(40.0, 40.0, 5.0, 1.0, 0.0, 0.0))
    res = make(0.8).align(pcL, pcG, far, amd.Parameters(maxIterations=15, quality_checkpoints={2: 0.5}))
    assert res.terminationReason in (amd.IterTermReason.QualityCheckpointFailed, amd.IterTermReason.NoPairings)


def test_batch_of_independent_pairs(oracle):
    """SURVEY.md 8e (i) / BASELINE config C4 in miniature: BatchRegistration deals scan pairs to
    ranks (one here) and registers each with ICP.align; every registration improves the guess and
    equals the same call made directly."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import synthetic
    from mp2p_icp_amd.distributed import BatchRegistration

    def make():
        icp = amd.ICP()
        m = amd.Matcher_Points_DistanceThreshold()
        m.initialize({"threshold": 1.0, "thresholdAngularDeg": 0.0})
        s = amd.Solver_GaussNewton()
        s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure", "robustKernelParam": 0.15})
        icp.set_matchers([m])
        icp.set_solvers([s])
        return icp

    scenes = [synthetic.make_pair(5000, 40000, 300 + b, max_t=0.2, max_r_deg=1.0) for b in range(3)]

    def align(b):
        d = scenes[b]
        res = make().align(amd.metric_map_t({"raw": amd.PointLayer(d["local"])}),
                           amd.metric_map_t({"raw": amd.PointLayer(d["glob"])}), d["T_init"],
                           amd.Parameters(maxIterations=30))
        return res.optimal_tf, res.nIterations, res.quality

    reg = BatchRegistration(len(scenes))
    assert reg.owned() == [0, 1, 2]
    table = reg.run(align)
    for b, d in enumerate(scenes):
        pose, iters, _ = align(b)
        assert np.array_equal(table[b, :12], pose) and table[b, 12] == iters
        # (convergence is the bunny tests' subject: point-to-point ICP slides along the street of
        # this synthetic scene; here: the batch equals the direct calls)
        assert table[b, 12] >= 1 and np.isfinite(table[b]).all()
