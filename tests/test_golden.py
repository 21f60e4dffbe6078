"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from
the CPU oracle): the oracle must keep reproducing them (CPU), and the HIP path must match them
(GPU) -- bit-exact for correspondences, 1e-5 m / 1e-5 rad for poses."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def test_oracle_reproduces_golden(oracle, gold):
    g, l = gold["pt2pt_glob"], gold["pt2pt_local"]
    for tag, T in (("a", gold["pt2pt_T_init"]), ("b", gold["pt2pt_T_gt"])):
        thr, ang = gold[f"pt2pt_{tag}_params"]
        pairs, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, thr, ang)
        assert pairs.tobytes() == gold[f"pt2pt_{tag}_pairs"].tobytes()
        assert pot == gold[f"pt2pt_{tag}_potential"][0]
    prm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    T, it, H, gg = oracle.optimal_tf_gauss_newton(gold["pt2pt_a_pairs"], None, None,
                                                  gold["pt2pt_T_init"], prm)
    assert np.allclose(T, gold["gn_a_pose"], atol=1e-12) and it == gold["gn_a_iters"][0]
    g2, l2 = gold["pt2pl_glob"], gold["pt2pl_local"]
    dt, sr, knn, mpp, pet = gold["pt2pl_params"]
    pl, idx, pot = oracle.match_pt2pl(g2[:, 0], g2[:, 1], g2[:, 2], l2[:, 0], l2[:, 1], l2[:, 2],
                                      oracle.pose_identity(), dt, sr, int(knn), int(mpp), pet)
    assert np.array_equal(idx, gold["pt2pl_local_idx"])
    assert np.allclose(pl["plane"], gold["pt2pl_pairs"]["plane"], atol=1e-12)


def test_bunny_fixture_is_the_reference_file():
    import gzip
    pts = np.loadtxt(gzip.open(os.path.join(HERE, "golden", "bunny_decim.xyz.gz")))
    assert pts.shape == (10642, 3)


@pytest.mark.gpu
def test_hip_matches_golden(oracle, gold):
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib
    g, l = gold["pt2pt_glob"], gold["pt2pt_local"]
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    for tag, T in (("a", gold["pt2pt_T_init"]), ("b", gold["pt2pt_T_gt"])):
        thr, ang = gold[f"pt2pt_{tag}_params"]
        m = amd.Matcher_Points_DistanceThreshold()
        m.initialize({"threshold": float(thr), "thresholdAngularDeg": float(ang)})
        pairs = amd.Pairings()
        m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        want = gold[f"pt2pt_{tag}_pairs"]
        got = pairs.paired_pt2pt
        assert len(got) == len(want)
        assert np.array_equal(got["localIdx"], want["localIdx"])
        assert np.array_equal(got["globalIdx"], want["globalIdx"])
        assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32))
        if tag == "a":
            s = amd.Solver_GaussNewton()
            s.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::GemanMcClure",
                          "robustKernelParam": 0.15})
            sc = amd.SolverContext()
            sc.guessRelativePose = T
            out = amd.OptimalTF_Result()
            assert s.optimal_pose(pairs, out, sc)
            dt_, dr_ = oracle.pose_err_split(out.optimalPose, gold["gn_a_pose"])
            assert dt_ < 1e-5 and dr_ < 1e-5
