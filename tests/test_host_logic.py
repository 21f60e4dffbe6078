"""Host-side mirror of the reference's plugin interface, exercised without a GPU: YAML-style
parameter parsing and the asserts of initialize(), gating by ICP iteration, formula parameters,
SE(3) helpers against the oracle, the sharding helpers.  (What needs device memory is covered by
the -m gpu tests.)"""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def test_se3_helpers_match_the_oracle(amd, oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        p6 = np.concatenate([rng.uniform(-5, 5, 3), rng.uniform(-3, 3, 1), rng.uniform(-1.5, 1.5, 1), rng.uniform(-3, 3, 1)])
        A, B = amd.se3.from_xyzypr(*p6), oracle.pose_from_xyzypr(*p6)
        assert np.allclose(A, B, atol=1e-15)
        assert np.allclose(amd.se3.to_xyzypr(A), oracle.pose_to_xyzypr(B), atol=1e-12)
        xi = rng.normal(0, 0.7, 6)
        E = amd.se3.exp(xi)
        assert np.allclose(E, oracle.se3_exp(xi), atol=1e-14)
        assert np.allclose(amd.se3.log(E), oracle.se3_log(E), atol=1e-12)
        assert np.allclose(amd.se3.log(E), xi, atol=1e-9)
        C = amd.se3.compose(A, E)
        assert np.allclose(C, oracle.pose_compose(A, E), atol=1e-14)
        assert np.allclose(amd.se3.inverse(C), oracle.pose_inverse(C), atol=1e-14)
        assert np.allclose(amd.se3.compose(C, amd.se3.inverse(C)), amd.se3.identity(), atol=1e-12)
        assert np.allclose(amd.se3.inverse_compose(C, A), amd.se3.compose(amd.se3.inverse(A), C), atol=1e-12)
    # tiny rotations take the series branch
    assert np.allclose(amd.se3.log(amd.se3.exp(np.array([1, 2, 3, 1e-9, -2e-9, 1e-10]))),
                       [1, 2, 3, 1e-9, -2e-9, 1e-10], atol=1e-12)


def test_matcher_gating_and_parameters(amd):
    calls = []

    class Probe(amd.Matcher):
        def impl_match(self, *a):
            calls.append(a[3].icpIteration)
            return True

    m = Probe()
    m.initialize({"runFromIteration": 2, "runUpToIteration": 4})
    for it in range(7):
        m.match(None, None, None, amd.MatchContext(it), None, None)
    assert calls == [2, 3, 4]                                   # Matcher.cpp:40-42
    m.initialize({"enabled": False})
    assert m.match(None, None, None, amd.MatchContext(3), None, None) is False
    m.initialize({})                                            # MCP_LOAD_OPT keeps the member values
    assert m.enabled is False and m.runFromIteration == 2
    m = Probe()                                                 # defaults: 0 / 0 = every iteration
    calls.clear()
    for it in (0, 100):
        m.match(None, None, None, amd.MatchContext(it), None, None)
    assert calls == [0, 100]

    d = amd.Matcher_Points_DistanceThreshold()
    with pytest.raises(KeyError):                               # threshold REQ (:43)
        d.initialize({"thresholdAngularDeg": 0.0})
    with pytest.raises(KeyError):                               # thresholdAngularDeg REQ (:44)
        d.initialize({"threshold": 1.0})
    d.initialize({"threshold": 1.5, "thresholdAngularDeg": 0.1, "pairingsPerPoint": 3, "maxLocalPointsPerLayer": 100,
                  "localPointsSampleSeed": 7, "allowMatchAlreadyMatchedGlobalPoints": True,
                  "pointLayerMatches": [{"global": "raw", "local": "decimated", "weight": 2.0},
                                        {"global": "raw", "local": "other"}],
                  "bounding_box_intersection_check_epsilon": 0.5, "kdtree_leaf_max_points": 20,
                  "hip_queries_per_wave": 16, "hip_tile_order": True})
    assert (d.threshold, d.thresholdAngularDeg, d.pairingsPerPoint) == (1.5, 0.1, 3)
    assert d.weight_pt2pt_layers == {"raw": {"decimated": 2.0, "other": 1.0}}
    assert d.maxLocalPointsPerLayer_ == 100 and d.localPointsSampleSeed_ == 7
    assert d.allowMatchAlreadyMatchedGlobalPoints_ and not d.allowMatchAlreadyMatchedPoints_
    assert d.bounding_box_intersection_check_epsilon_ == 0.5 and d.queries_per_wave == 16 and d.tile_order
    p = d._params(local_index_offset=123)
    assert (p.threshold, p.pairingsPerPoint, p.local_index_offset, p.queries_per_wave, p.tile_order) == (1.5, 3, 123, 16, 1)

    a = amd.Matcher_Adaptive()
    ok = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2, absoluteMaxSearchDistance=5.0,
              enableDetectPlanes=True)
    a.initialize(ok)
    assert (a.planeSearchPoints, a.planeMinimumFoundPoints, a.maxPt2PtCorrespondences, a.minimumCorrDist) == (8, 4, 1, 0.1)
    for missing in ok:
        with pytest.raises(KeyError):
            amd.Matcher_Adaptive().initialize({k: v for k, v in ok.items() if k != missing})
    for bad in (dict(confidenceInterval=0.0), dict(confidenceInterval=1.0), dict(planeSearchPoints=3),
                dict(planeMinimumFoundPoints=2), dict(planeEigenThreshold=0.0)):
        with pytest.raises(RuntimeError):
            amd.Matcher_Adaptive().initialize(dict(ok, **bad))
    q = a._params()
    assert (q.confidenceInterval, q.enableDetectPlanes, q.planeSearchPoints) == (0.8, 1, 8)

    with pytest.raises(KeyError):
        amd.Matcher_Points_InlierRatio().initialize({})
    with pytest.raises(KeyError):
        amd.Matcher_Point2Plane().initialize({"searchRadius": 1.0})


def test_formula_parameters_follow_the_parameter_source(amd):
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": "2.0 * exp(-0.5 * ICP_ITERATION) + 0.1", "thresholdAngularDeg": 0})
    with pytest.raises(RuntimeError):                           # not realised yet (…DistanceThreshold.cpp:55)
        m.checkAllParametersAreRealized()
    src = amd.ParameterSource()
    m.attachToParameterSource(src)
    for it in (0, 1, 4):
        src.updateVariable("ICP_ITERATION", it)
        src.realize()
        m.checkAllParametersAreRealized()
        assert m.threshold == pytest.approx(2.0 * np.exp(-0.5 * it) + 0.1)
    c = amd.Matcher_Points_DistanceThreshold()
    c.initialize({"threshold": "sqrt(4.0)", "thresholdAngularDeg": "0.5*2"})    # constants: at once
    c.checkAllParametersAreRealized()
    assert (c.threshold, c.thresholdAngularDeg) == (2.0, 1.0)


def test_solver_gating_and_parameters(amd):
    from mp2p_icp_amd.solver import Solver, WeightParameters
    runs = []

    class Probe(Solver):
        def impl_optimal_pose(self, pairings, out, sc):
            runs.append(sc.icpIteration)
            return True

    s = Probe()
    s.initialize({"runFromIteration": 1, "runUpToIteration": 2, "runUntilTranslationCorrectionSmallerThan": 0.01})
    sc = amd.SolverContext()
    for it, step in enumerate([None, 0.5, 0.2, 0.001]):
        sc.icpIteration = it
        sc.lastIcpStepIncrement = None if step is None else np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, step, 0, 0.0])
        s.optimal_pose(None, amd.OptimalTF_Result(), sc)
    assert runs == [1, 2]                                       # Solver.cpp:40-46
    s.initialize({"runUntilTranslationCorrectionSmallerThan": 0.01})
    runs.clear()
    sc = amd.SolverContext()
    for it, step in enumerate([0.5, 0.005, 0.5]):
        sc.icpIteration = it
        sc.lastIcpStepIncrement = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, step, 0, 0.0])
        s.optimal_pose(None, amd.OptimalTF_Result(), sc)
    assert runs == [0]                                          # once below the limit: finished for good (:48-62)

    g = amd.Solver_GaussNewton()
    with pytest.raises(KeyError):
        g.initialize({})
    with pytest.raises(ValueError):
        g.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::Huber"})
    with pytest.raises(KeyError):                               # PairWeights.cpp:26-34: all five required
        g.initialize({"maxIterations": 3, "pair_weights": {"pt2pt": 1.0}})
    g.initialize({"maxIterations": 3, "robustKernel": "RobustKernel::Cauchy", "robustKernelParam": "0.1*3",
                  "pair_weights": dict(pt2pt=2, pt2pl=3, pt2ln=4, ln2ln=5, pl2pl=6)})
    p = g.gn_params(amd.SolverContext(), point_weights=[(10, 0.5), (20, 2.0)])
    assert (p.maxInnerLoopIterations, p.kernel, p.w_pt2pt, p.w_pt2pl, p.w_pt2ln, p.w_pl2pl) == (3, 2, 2.0, 3.0, 4.0, 6.0)
    assert p.kernelParam == pytest.approx(0.3) and p.n_weight_blocks == 2 and p.minDelta == 1e-7
    assert (p.weight_block_count[1], p.weight_block_w[1]) == (20, 2.0)
    with pytest.raises(ValueError):
        g.gn_params(amd.SolverContext(), point_weights=[(1, 1.0)] * 33)   # MP2P_HIP_MAX_WEIGHT_BLOCKS = 32
    assert g.gn_params(amd.SolverContext(), point_weights=[(1, 1.0)] * 32).n_weight_blocks == 32

    w = WeightParameters()
    w.load_from({"use_scale_outlier_detector": True, "robust_kernel": "RobustKernel::GemanMcClure",
                 "robust_kernel_param": 0.7})
    w.currentEstimateForRobust = amd.se3.from_xyzypr(1, 2, 3, 0.1, 0.2, 0.3)
    L = w.to_lib()
    assert (L.use_scale_outlier_detector, L.scale_outlier_threshold, L.robust_kernel, L.robust_kernel_param,
            L.has_current_estimate) == (1, 1.2, 1, 0.7, 1)
    assert np.allclose(list(L.current_estimate), w.currentEstimateForRobust)
    with pytest.raises(KeyError):
        WeightParameters().load_from({"use_scale_outlier_detector": False})
    with pytest.raises(ValueError):
        WeightParameters().load_from({"use_scale_outlier_detector": False, "robust_kernel": "Tukey"})


def test_sharding_helpers(amd):
    from mp2p_icp_amd.distributed import BatchRegistration, shard_range
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    reg = BatchRegistration(7)
    assert reg.owned() == list(range(7)) and (reg.rank, reg.world) == (0, 1)
    t = reg.run(lambda b: (amd.se3.from_xyzypr(b, 0, 0), b + 1, 0.5))
    assert t.shape == (7, 14) and t[3, 9] == 3.0 and t[3, 12] == 4 and (t[:, 13] == 0.5).all()


def test_synthetic_inputs_are_reproducible(amd):
    from mp2p_icp_amd import synthetic
    a = synthetic.make_pair(2000, 8000, 11)
    b = synthetic.make_pair(2000, 8000, 11)
    c = synthetic.make_pair(2000, 8000, 12)
    assert np.array_equal(a["glob"], b["glob"]) and np.array_equal(a["local"], b["local"])
    assert np.array_equal(a["T_init"], b["T_init"]) and not np.array_equal(a["local"], c["local"])
    assert a["glob"].dtype == np.float32 and a["glob"].shape == (8000, 3) and a["local"].shape == (2000, 3)


def test_weight_block_bounds_algorithm():
    """horn.hip replays the reference's sequential weight-block cursor (visit_correspondences.h:
    113-119: one block per VISITED pairing at most, restart at the index it moved at) as "first
    pairing of every block" found by forward scans; the scan rule, restated here, must give every
    pairing the block the sequential cursor gives it -- for any pattern of skipped pairings,
    zero-length blocks and exhausted block lists."""
    rng = np.random.default_rng(0)

    def sequential(n, counts, skipped):
        cur, start, out = 0, 0, {}
        for i in range(n):
            if skipped[i]:
                continue
            if i >= start + counts[cur]:
                cur += 1
                if cur >= len(counts):
                    return None
                start = i
            out[i] = cur
        return out

    def by_bounds(n, counts, skipped):          # horn_block_bounds_kernel + the lookup of horn_cov_kernel
        nb, start, b = len(counts), 0, 0
        bounds = [0] + [n] * (nb - 1)
        while True:
            s = start + counts[b]
            if b > 0 and s <= start:
                s = start + 1
            found = next((i for i in range(s, n) if not skipped[i]), n)
            if found >= n:
                break
            if b + 1 >= nb:
                return None
            b, start = b + 1, found
            bounds[b] = found
        out = {}
        for i in range(n):
            if not skipped[i]:
                k = 0
                while k + 1 < nb and i >= bounds[k + 1]:
                    k += 1
                out[i] = k
        return out

    for _ in range(3000):
        n, nb = int(rng.integers(1, 60)), int(rng.integers(1, 6))
        counts = [int(rng.integers(0, 25)) for _ in range(nb)]
        skipped = rng.random(n) < rng.choice([0.0, 0.2, 0.6])
        assert sequential(n, counts, skipped) == by_bounds(n, counts, skipped), (n, counts, skipped)
