"""GPU parity of Matcher_Points_InlierRatio (through the C ABI) against the CPU oracle: the same
pairs in the same (multimap) order, incl. equal distances (later-visited point first), pre-marked
MatchState, global re-use, a visit list, and the reference's own use of the matcher:
tests/test-mp2p_icp_algos.cpp runs ICP with it on the bunny and asserts convergence < 0.1."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _same(hip, orc):
    assert len(hip) == len(orc), (len(hip), len(orc))
    assert np.array_equal(hip["localIdx"], orc["localIdx"])
    assert np.array_equal(hip["globalIdx"], orc["globalIdx"])
    assert np.array_equal(hip["errorSquareAfterTransformation"].view(np.uint32), orc["errSq"].view(np.uint32))
    assert np.array_equal(hip["local"], np.stack([orc["lx"], orc["ly"], orc["lz"]], 1))
    assert np.array_equal(hip["global"], np.stack([orc["gx"], orc["gy"], orc["gz"]], 1))


@pytest.mark.parametrize("ratio", [0.1, 0.5, 0.83])
@pytest.mark.parametrize("allow_global", [False, True])
def test_parity_vs_oracle(amd, oracle, ratio, allow_global):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(5000, 20000, 61, outlier_frac=0.2)   # outliers: unbounded searches
    g, l = d["glob"], d["local"]
    l[100:160] = l[0:60]                      # equal local points -> equal d2: multimap tie order
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Points_InlierRatio()
    m.initialize({"inliersRatio": ratio, "allowMatchAlreadyMatchedGlobalPoints": allow_global})
    for pose in (d["T_init"], d["T_gt"], d["T_init"]):
        want, pot = oracle.match_inlier_ratio(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, ratio,
                                              allowMatchAlreadyMatchedGlobalPoints=allow_global, tree=tree)
        pairs = amd.Pairings()
        ms = amd.MatchState(pcG, pcL)
        assert m.match(pcG, pcL, pose, amd.MatchContext(), ms, pairs)
        _same(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot == l.shape[0]
        if not allow_global:
            assert np.unique(want["globalIdx"]).size == len(want)
        e = want["errSq"]
        assert np.all(np.diff(e) >= 0)                                  # ascending distance
        # marks: every emitted pair, whatever the re-use flag (:136-138)
        gm, lm = ms.for_layers("raw", "raw").download()
        assert set(np.flatnonzero(lm).tolist()) == set(want["localIdx"].tolist())
        assert set(np.flatnonzero(gm).tolist()) == set(want["globalIdx"].tolist())


def test_match_state_visit_list_and_errors(amd, oracle):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(3000, 9000, 62, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    rng = np.random.default_rng(1)
    lt0 = (rng.random(l.shape[0]) < 0.3).astype(np.uint8)
    gt0 = (rng.random(g.shape[0]) < 0.3).astype(np.uint8)
    order = rng.permutation(2000).astype(np.uint32)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    for allowL in (False, True):
        lt, gt = lt0.copy(), gt0.copy()
        want, pot = oracle.match_inlier_ratio(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_gt"], 0.6,
                                              allowMatchAlreadyMatchedPoints=allowL, tree=tree,
                                              local_taken=lt, global_taken=gt, idxs=order)
        m = amd.Matcher_Points_InlierRatio()
        m.initialize({"inliersRatio": 0.6, "allowMatchAlreadyMatchedPoints": allowL,
                      "maxLocalPointsPerLayer": 2000, "localPointsSampleSeed": 3})
        m.visit_order_fn = lambda n, mx, seed: order
        ms = amd.MatchState(pcG, pcL)
        ms.for_layers("raw", "raw").upload(gt0, lt0)
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, d["T_gt"], amd.MatchContext(), ms, pairs)
        _same(pairs.paired_pt2pt, want)
        assert pairs.potential_pairings == pot == l.shape[0]          # :53: the whole layer's size
        gm, lm = ms.for_layers("raw", "raw").download()
        assert np.array_equal(gm, gt) and np.array_equal(lm, lt)
    # every local point already paired: the reference asserts nTotal > 0
    ms = amd.MatchState(pcG, pcL)
    ms.for_layers("raw", "raw").upload(np.zeros(g.shape[0], np.uint8), np.ones(l.shape[0], np.uint8))
    m = amd.Matcher_Points_InlierRatio()
    m.initialize({"inliersRatio": 0.5})
    with pytest.raises(amd.Mp2pHipError):
        m.match(pcG, pcL, d["T_gt"], amd.MatchContext(), ms, amd.Pairings())
    with pytest.raises(KeyError):
        amd.Matcher_Points_InlierRatio().initialize({})


@pytest.mark.parametrize("solver_name", ["GaussNewton", "Horn"])
def test_bunny_icp_with_inlier_ratio(amd, oracle, solver_name):
    """tests/test-mp2p_icp_algos.cpp:132-141: ICP | Solver | Matcher_Points_InlierRatio, decimation 10,
    random pose within 15 % of the bounding box / 10 degrees, ASSERT_LT_(err_se3, 0.1)"""
    with gzip.open(os.path.join(HERE, "golden", "bunny_decim.xyz.gz"), "rt") as f:
        pts = np.loadtxt(f, dtype=np.float32)[:, :3][::10]
    rng = np.random.default_rng(1234)
    size = pts.max(0) - pts.min(0)
    for rep in range(3):
        gt = amd.se3.from_xyzypr(*(rng.uniform(-0.15, 0.15, 3) * size), *np.radians(rng.uniform(-10, 10, 3)))
        R, t = gt[:9].reshape(3, 3), gt[9:]
        reg = ((pts.astype(np.float64) - t) @ R).astype(np.float32)      # changeCoordinatesReference(-gt)
        icp = amd.ICP()
        s = amd.Solver_GaussNewton() if solver_name == "GaussNewton" else amd.Solver_Horn()
        s.initialize({"maxIterations": 10} if solver_name == "GaussNewton" else {})
        m = amd.Matcher_Points_InlierRatio()
        m.initialize({"inliersRatio": 0.80})
        icp.set_solvers([s])
        icp.set_matchers([m])
        res = icp.align(amd.metric_map_t({"raw": amd.PointLayer(reg)}), amd.metric_map_t({"raw": amd.PointLayer(pts)}),
                        amd.se3.identity(), amd.Parameters(maxIterations=100))
        err = amd.se3.log(amd.se3.inverse_compose(res.optimal_tf, gt))
        assert np.linalg.norm(err) < 0.1, (rep, np.linalg.norm(err))
