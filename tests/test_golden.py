"""Committed golden vectors (tests/golden/golden_v1.npz and golden_v2.npz, made by
tests/golden/make_golden.py from the CPU oracle): the oracle must keep reproducing them (CPU), and
the HIP path must match them (GPU) -- bit-exact for correspondences, 1e-5 m / 1e-5 rad for poses."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


@pytest.fixture(scope="module")
def gold2():
    return np.load(os.path.join(HERE, "golden", "golden_v2.npz"))


ADAPTIVE_KEYS = sorted(["confidenceInterval", "firstToSecondDistanceMax", "absoluteMaxSearchDistance",
                        "minimumCorrDist", "enableDetectPlanes", "maxPt2PtCorrespondences", "planeSearchPoints",
                        "planeMinimumFoundPoints", "planeMinimumDistance", "planeEigenThreshold"])
ADAPTIVE_INT = ("maxPt2PtCorrespondences", "planeSearchPoints", "planeMinimumFoundPoints")


def _adaptive_kw(gold2):
    kw = dict(zip(ADAPTIVE_KEYS, gold2["adaptive_params"].tolist()))
    for k in ADAPTIVE_INT:
        kw[k] = int(kw[k])
    kw["enableDetectPlanes"] = bool(kw["enableDetectPlanes"])
    return kw


def test_oracle_reproduces_golden_v2(oracle, gold, gold2):
    """the components of SURVEY.md 8f: adaptive and inlier-ratio matchers, Horn with
    WeightParameters, pt2ln_pl_to_pt2pt, covariance, voxel decimation"""
    g, l, T = gold["pt2pl_glob"], gold["pt2pl_local"], gold2["pose"]
    r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, **_adaptive_kw(gold2))
    assert r["pt2pt"].tobytes() == gold2["adaptive_pt2pt"].tobytes()
    assert np.array_equal(r["pl_local_idx"], gold2["adaptive_pl_idx"])
    assert np.allclose(r["pt2pl"]["plane"], gold2["adaptive_pt2pl"]["plane"], atol=1e-12)
    assert r["ci_high"] == gold2["adaptive_ci_high"][0]
    h = gold2["adaptive_hist"]
    assert (r["hist"]["minSq"], r["hist"]["maxSq"], r["hist"]["count"]) == (np.float32(h[0]), np.float32(h[1]), h[2])
    assert np.array_equal(r["hist"]["bins"], h[3:].astype(np.uint64))
    ir, _ = oracle.match_inlier_ratio(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, 0.6)
    assert ir.tobytes() == gold2["inlier_ratio_pairs"].tobytes()
    pt = gold2["horn_pairs"]
    Th, rc, fl = oracle.optimal_tf_horn_wp(pt, None, use_scale_outlier_detector=True, scale_outlier_threshold=1.15,
                                           point_weights=[(300, 2.0), (len(pt), 0.5)])
    assert rc == 1 and np.allclose(Th, gold2["horn_pose"], atol=1e-12)
    assert np.array_equal(np.flatnonzero(fl), gold2["horn_outliers"])
    assert oracle.pt2ln_pl_to_pt2pt(gold2["adaptive_pt2pl"], None, T).tobytes() == gold2["converted_pairs"].tobytes()
    cov, H, ok = oracle.covariance(gold2["adaptive_pt2pt"], gold2["adaptive_pt2pl"], None, None, T)
    assert ok and np.allclose(H, gold2["cov_H"], rtol=1e-12) and np.allclose(cov, gold2["cov"], rtol=1e-9)
    for name, method in (("first", 0), ("closest", 1), ("average", 2)):
        xyz, src = oracle.filter_decimate_voxels(g[:, 0], g[:, 1], g[:, 2], 0.5, method)
        assert np.array_equal(xyz, gold2[f"decimate_{name}_xyz"]) and np.array_equal(src, gold2[f"decimate_{name}_src"])


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


def _hip_pt2pt(o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PT2PT)
    h["globalIdx"], h["localIdx"] = o["globalIdx"], o["localIdx"]
    h["global"] = np.stack([o["gx"], o["gy"], o["gz"]], 1)
    h["local"] = np.stack([o["lx"], o["ly"], o["lz"]], 1)
    h["errorSquareAfterTransformation"] = o["errSq"]
    return h


def _hip_pt2pl(o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PT2PL)
    h["plane"], h["centroid"] = o["plane"], o["centroid"]
    h["pt_local"] = np.stack([o["lx"], o["ly"], o["lz"]], 1)
    return h


def _same_pt2pt(got, want):
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(got["localIdx"], want["localIdx"])
    assert np.array_equal(got["globalIdx"], want["globalIdx"])
    assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32))
    assert np.array_equal(got["global"], np.stack([want["gx"], want["gy"], want["gz"]], 1))
    assert np.array_equal(got["local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))


@pytest.mark.gpu
def test_hip_matches_golden_v2(oracle, gold, gold2):
    import mp2p_icp_amd as amd
    from mp2p_icp_amd.solver import WeightParameters, optimal_tf_horn, pt2ln_pl_to_pt2pt
    g, l, T = gold["pt2pl_glob"], gold["pt2pl_local"], gold2["pose"]
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    # adaptive matcher
    m = amd.Matcher_Adaptive()
    m.initialize(_adaptive_kw(gold2))
    pairs = amd.Pairings()
    assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
    _same_pt2pt(pairs.paired_pt2pt, gold2["adaptive_pt2pt"])
    assert np.array_equal(pairs.paired_pt2pl_local_idx, gold2["adaptive_pl_idx"])
    assert np.allclose(pairs.paired_pt2pl["plane"], gold2["adaptive_pt2pl"]["plane"], rtol=0, atol=1e-9)
    assert m.last_ci_high == gold2["adaptive_ci_high"][0]
    assert m.last_histogram["bins"] == gold2["adaptive_hist"][3:].astype(np.uint64).tolist()
    # inlier-ratio matcher
    m = amd.Matcher_Points_InlierRatio()
    m.initialize({"inliersRatio": 0.6})
    pairs = amd.Pairings()
    assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
    _same_pt2pt(pairs.paired_pt2pt, gold2["inlier_ratio_pairs"])
    # Horn with the scale outlier detector and weight blocks
    pt = gold2["horn_pairs"]
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _hip_pt2pt(pt), point_weights=[(300, 2.0), (len(pt), 0.5)])
    w = WeightParameters()
    w.use_scale_outlier_detector, w.scale_outlier_threshold = True, 1.15
    out = amd.OptimalTF_Result()
    assert optimal_tf_horn(p, w, out)
    dt, dr = oracle.pose_err_split(out.optimalPose, gold2["horn_pose"])
    assert dt < 1e-5 and dr < 1e-5
    assert out.outliers == gold2["horn_outliers"].tolist()
    # pt2ln_pl_to_pt2pt of the adaptive matcher's plane pairings
    p = amd.Pairings.from_host(ctx, None, _hip_pt2pl(gold2["adaptive_pt2pl"]))
    sc = amd.SolverContext()
    sc.guessRelativePose = T
    got = pt2ln_pl_to_pt2pt(p, sc).paired_pt2pt
    want = gold2["converted_pairs"]
    assert len(got) == len(want)
    assert np.array_equal(got["global"], np.stack([want["gx"], want["gy"], want["gz"]], 1))
    assert np.array_equal(got["local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))
