"""BASELINE.json configs C2 .. C5 at their stated sizes on the HIP path (through the C ABI), against
the CPU oracle on the same inputs.  Scenes follow SURVEY.md section 8d: the map is a union of
consecutive scans, voxel-thinned to the exact size (mp2p_icp_amd.synthetic.make_scan_union_pair);
seeds as listed there (C2 2001, C3 3001, C4 4000 + pair, C5 5001).

  C2  ~120 k-point scan vs 2 M-point map, Matcher_Points_DistanceThreshold (2.0 m) + Solver_Horn,
      chained for 6 outer iterations: identical pair lists and poses (1e-5) at every iteration.
  C3  ~120 k-point scan vs 10 M-point map, Matcher_Point2Plane (knn 5, radius 0.4) + Solver_GaussNewton
      on the plane pairings, chained for 5 iterations: the same local points paired, planes 1e-9,
      poses 1e-5 -- the FULL oracle on every query (its searches spread over the host's threads).
  C4  a batch of 8 independent 1 M x 1 M pairs on one rank (BatchRegistration; 64 over 8 ranks is the
      same code, see tests/test_distributed_gloo.py): pair lists of the first iteration and the pose
      after three against the oracle for two of them, the batch table against direct calls for all.
  C5  5 M x 5 M, 30 % uniform outliers in the local layer, two matchers in one run_matchers call
      (Matcher_Point2Plane, then Matcher_Points_DistanceThreshold on the points it left), both
      pairing kinds into ONE Gauss-Newton, with GemanMcClure and with Cauchy.  "Welsch" of
      BASELINE.json has no upstream semantics (robust_kernels.h:33-43 knows None, GemanMcClure,
      Cauchy): it is not run.

The oracle's per-query searches are spread over the host's threads (orc_match_pt2pt_mt_ms,
orc_match_pt2pl_mt); their gathering and the unique-global filter stay sequential, so the lists are
the sequential loop's lists (checked against it on CPU in tests/test_oracle_kat.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
THREADS = max(1, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _xyz(a):
    return a[:, 0], a[:, 1], a[:, 2]


def _same_pt2pt(got, want):
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(got["localIdx"], want["localIdx"])
    assert np.array_equal(got["globalIdx"], want["globalIdx"])
    assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32))
    assert np.array_equal(got["local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))
    assert np.array_equal(got["global"], np.stack([want["gx"], want["gy"], want["gz"]], 1))


def _same_pt2pl(pairs, want, widx):
    got, gidx = pairs.paired_pt2pl, pairs.paired_pt2pl_local_idx
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(gidx, widx)
    assert np.allclose(got["plane"], want["plane"], rtol=0, atol=1e-9)
    assert np.allclose(got["centroid"], want["centroid"], rtol=0, atol=1e-9)
    assert np.array_equal(got["pt_local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))


# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(600)
def test_c2_scan_vs_2M_map_pt2pt_horn_chain(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.make_scan_union_pair(120_000, 2_000_000, 2001, map_scan_points=120_000)
    g, l = d["glob"], d["local"]
    assert g.shape[0] == 2_000_000 and l.shape[0] > 100_000
    tree = oracle.KDTree(*_xyz(g))
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": 2.0, "thresholdAngularDeg": 0.0})  # demos/icp-settings-kitti.yaml:42
    s = amd.Solver_Horn()
    s.initialize({})
    pose_h, pose_o = d["T_init"].copy(), d["T_init"].copy()
    e0 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(pose_h, d["T_gt"])))
    for it in range(6):
        pairs = amd.run_matchers([m], pcG, pcL, pose_h, amd.MatchContext(it))
        want, pot = oracle.match_pt2pt(*_xyz(g), *_xyz(l), pose_o, 2.0, 0.0, tree=tree, threads=THREADS)
        _same_pt2pt(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot == l.shape[0]
        assert len(want) > 10_000  # (the unique-global filter drops most claimants of a shared map point)
        sc = amd.SolverContext()
        sc.guessRelativePose, sc.icpIteration = pose_h, it
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        pose_h = out.optimalPose
        pose_o, ok = oracle.optimal_tf_horn(want)
        assert ok
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)
    e1 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(pose_h, d["T_gt"])))
    assert e1 < e0, (e0, e1)  # the chain moves towards the ground truth


# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_c3_scan_vs_10M_map_pt2pl_gauss_newton_chain(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.make_scan_union_pair(120_000, 10_000_000, 3001, map_scan_points=1_000_000)
    g, l = d["glob"], d["local"]
    assert g.shape[0] == 10_000_000 and l.shape[0] > 100_000
    tree = oracle.KDTree(*_xyz(g))
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    # distanceThreshold 0.4, knn 5 (demos/icp-settings-example1.yaml:40-42)
    P = dict(distanceThreshold=0.4, searchRadius=0.4, knn=5, minimumPlanePoints=5, planeEigenThreshold=0.05)
    m = amd.Matcher_Point2Plane()
    m.initialize(P)
    s = amd.Solver_GaussNewton()
    s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure", "robustKernelParam": 0.15})
    prm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    pose_h, pose_o = d["T_init"].copy(), d["T_init"].copy()
    e0 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(pose_h, d["T_gt"])))
    for it in range(5):
        pairs = amd.run_matchers([m], pcG, pcL, pose_h, amd.MatchContext(it))
        want, widx, pot = oracle.match_pt2pl(*_xyz(g), *_xyz(l), pose_o, tree=tree, threads=THREADS, **P)
        _same_pt2pl(pairs, want, widx)
        assert pairs.potential_pairings == pot == l.shape[0]
        assert len(want) > 1000
        sc = amd.SolverContext()
        sc.guessRelativePose, sc.icpIteration = pose_h, it
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        pose_h = out.optimalPose
        pose_o, *_ = oracle.optimal_tf_gauss_newton(None, want, None, pose_o, prm)
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)
    e1 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(pose_h, d["T_gt"])))
    assert e1 < e0, (e0, e1)


# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_c4_batch_of_1M_pairs_on_one_rank(amd, oracle):
    from mp2p_icp_amd import synthetic
    from mp2p_icp_amd.distributed import BatchRegistration
    N_PAIRS, N = 8, 1_000_000
    prm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)

    def scene(b):  # 1 M-point scan vs 1 M-point map (seed 4000 + pair id)
        return synthetic.make_scan_union_pair(N, N, 4000 + b, map_scan_points=N, max_t=0.3, max_r_deg=1.5)

    def make_icp():
        icp = amd.ICP()
        m = amd.Matcher_Points_DistanceThreshold()
        m.initialize({"threshold": 1.0, "thresholdAngularDeg": 0.0})
        s = amd.Solver_GaussNewton()
        s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure", "robustKernelParam": 0.15})
        icp.set_matchers([m])
        icp.set_solvers([s])
        return icp, m, s

    checked = []

    def align(b):
        d = scene(b)
        g, l = d["glob"], d["local"]
        pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
        pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
        icp, m, s = make_icp()
        res = icp.align(pcL, pcG, d["T_init"], amd.Parameters(maxIterations=3, minAbsStep_trans=0.0, minAbsStep_rot=0.0))
        e0 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(d["T_init"], d["T_gt"])))
        e1 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(res.optimal_tf, d["T_gt"])))
        assert e1 < e0, (b, e0, e1)
        if b in (0, 5):  # the same three iterations on the oracle
            tree = oracle.KDTree(*_xyz(g))
            pose = d["T_init"].copy()
            first = None
            for it in range(3):
                want, _ = oracle.match_pt2pt(*_xyz(g), *_xyz(l), pose, 1.0, 0.0, tree=tree, threads=THREADS)
                first = want if first is None else first
                pose, *_ = oracle.optimal_tf_gauss_newton(want, None, None, pose, prm, threads=THREADS)
            pairs = amd.run_matchers([m], pcG, pcL, d["T_init"], amd.MatchContext(0))
            _same_pt2pt(pairs.paired_pt2pt, first)
            dt, dr = oracle.pose_err_split(res.optimal_tf, pose)
            assert dt < 1e-5 and dr < 1e-5, (b, dt, dr)
            checked.append(b)
        return res.optimal_tf, res.nIterations, res.quality

    reg = BatchRegistration(N_PAIRS)
    assert reg.owned() == list(range(N_PAIRS))
    table = reg.run(align)
    assert checked == [0, 5]
    assert np.isfinite(table).all() and (table[:, 12] == 3).all()


# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("kernel", ["GemanMcClure", "Cauchy"])
def test_c5_mixed_pairings_outlier_heavy_5M(amd, oracle, kernel):
    from mp2p_icp_amd import synthetic
    N = 5_000_000
    d = synthetic.make_scan_union_pair(N, N, 5001, map_scan_points=1_000_000, outlier_frac=0.30)
    g, l = d["glob"], d["local"]
    assert g.shape[0] == N and l.shape[0] > 0.9 * N
    tree = oracle.KDTree(*_xyz(g))
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    PL = dict(distanceThreshold=0.25, searchRadius=0.4, knn=5, minimumPlanePoints=5, planeEigenThreshold=0.05)
    m_pl = amd.Matcher_Point2Plane()
    m_pl.initialize(PL)
    m_pt = amd.Matcher_Points_DistanceThreshold()
    m_pt.initialize({"threshold": 1.0, "thresholdAngularDeg": 0.0})
    kid = {"GemanMcClure": oracle.KERNEL_GEMANMCCLURE, "Cauchy": oracle.KERNEL_CAUCHY}[kernel]
    s = amd.Solver_GaussNewton()
    s.initialize({"maxIterations": 3, "robustKernel": f"RobustKernel::{kernel}", "robustKernelParam": 0.15})
    prm = oracle.make_gn_params(3, kernel=kid, kernelParam=0.15)
    pose_h, pose_o = d["T_init"].copy(), d["T_init"].copy()
    for it in range(2):
        pairs = amd.run_matchers([m_pl, m_pt], pcG, pcL, pose_h, amd.MatchContext(it))
        # oracle: the plane matcher marks the local points it pairs (:109); the point matcher skips them
        lt = np.zeros(l.shape[0], np.uint8)
        w_pl, w_idx, pot_pl = oracle.match_pt2pl(*_xyz(g), *_xyz(l), pose_o, tree=tree, local_taken=lt,
                                                 threads=THREADS, **PL)
        gt = np.zeros(g.shape[0], np.uint8)
        w_pt, pot_pt = oracle.match_pt2pt(*_xyz(g), *_xyz(l), pose_o, 1.0, 0.0, tree=tree, local_taken=lt,
                                          global_taken=gt, threads=THREADS)
        _same_pt2pl(pairs, w_pl, w_idx)
        _same_pt2pt(pairs.paired_pt2pt, w_pt)
        assert pairs.potential_pairings == pot_pl + pot_pt == 2 * l.shape[0]
        assert len(w_pl) > 10_000 and len(w_pt) > 10_000
        assert not np.intersect1d(w_idx, w_pt["localIdx"]).size  # a local point is paired once
        # the 30 % outliers pair rarely (uniform in the scan's box: most are far from every surface)
        sc = amd.SolverContext()
        sc.guessRelativePose, sc.icpIteration = pose_h, it
        out = amd.OptimalTF_Result()
        assert s.optimal_pose(pairs, out, sc)
        pose_h = out.optimalPose
        pose_o, *_ = oracle.optimal_tf_gauss_newton(w_pt, w_pl, None, pose_o, prm, threads=THREADS)
        dt, dr = oracle.pose_err_split(pose_h, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (kernel, it, dt, dr)
    e0 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(d["T_init"], d["T_gt"])))
    e1 = np.linalg.norm(amd.se3.log(amd.se3.inverse_compose(pose_h, d["T_gt"])))
    assert e1 < e0, (e0, e1)
