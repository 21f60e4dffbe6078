"""GPU parity of the Horn solver (through the C ABI) against the CPU oracle: optimal_tf_horn with
WeightParameters (pair weights, point_weights blocks, plane-to-plane normals, robust kernels,
scale outlier detector incl. the flagged indices) shaped after tests/test-mp2p_optimal_tf_algos.cpp,
pt2ln_pl_to_pt2pt, and Solver_Horn on pairings that hold point-to-plane / point-to-line entries.
Tolerance: pose 1e-5 m / 1e-5 rad (north_star), converted pairs bit-exact."""
import numpy as np
import pytest

from test_gpu_gn import _close, _to_hip_pl2pl, _to_hip_pt2ln, _to_hip_pt2pl, _to_hip_pt2pt
from test_oracle_kat import horn_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _wp(amd, **kw):
    from mp2p_icp_amd.solver import WeightParameters
    w = WeightParameters()
    for k, v in kw.items():
        if k.startswith("w_"):
            setattr(w.pair_weights, k[2:], v)
        else:
            setattr(w, k, v)
    return w


def _okw(kw):
    """the same settings for oracle.optimal_tf_horn_wp"""
    m = {"currentEstimateForRobust": "current_estimate"}
    return {m.get(k, k): v for k, v in kw.items()}


CASES = [
    dict(),
    dict(w_pt2pt=0.3, w_pl2pl=4.0),
    dict(robust_kernel=1, robust_kernel_param=1.0, est="gt"),
    dict(robust_kernel=2, robust_kernel_param=0.5, est="identity", w_pl2pl=2.0),
    dict(use_scale_outlier_detector=True, scale_outlier_threshold=1.2),
    dict(use_scale_outlier_detector=True, scale_outlier_threshold=1.05, robust_kernel=1, est="gt",
         blocks=[(3000, 0.5), (0, 9.0), (5000, 2.0), (100000, 1.0)]),
    dict(blocks=[(1000, 3.0), (19000, 0.25)]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_optimal_tf_horn_weight_parameters(amd, oracle, case):
    from mp2p_icp_amd.solver import optimal_tf_horn
    c = dict(CASES[case])
    gt, pt, pl = horn_scene(oracle, 40 + case, n_pt=20_000, n_pl=200, noise=0.02, outliers=1500)
    blocks = c.pop("blocks", None)
    est = c.pop("est", None)
    if est is not None:
        c["currentEstimateForRobust"] = gt if est == "gt" else oracle.pose_identity()
    To, rc, fl = oracle.optimal_tf_horn_wp(pt, pl, point_weights=blocks, **_okw(c))
    assert rc == 1
    p = amd.Pairings.from_host(amd.default_context(), _to_hip_pt2pt(amd, pt), point_weights=blocks,
                               pl2pl=_to_hip_pl2pl(pl))
    out = amd.OptimalTF_Result()
    assert optimal_tf_horn(p, _wp(amd, **c), out)
    assert _close(oracle, out.optimalPose, To), (out.optimalPose, To)
    assert out.outliers == np.flatnonzero(fl).tolist()
    if c.get("use_scale_outlier_detector"):
        assert 500 < len(out.outliers) < 10_000
        assert oracle.pose_err(out.optimalPose, gt) < 0.1
    else:
        assert out.outliers == []


def test_horn_errors_and_small_inputs(amd, oracle):
    from mp2p_icp_amd.solver import optimal_tf_horn
    ctx = amd.default_context()
    gt, pt, pl = horn_scene(oracle, 60, n_pt=300, n_pl=10)
    out = amd.OptimalTF_Result()
    # fewer than 3 pairings: not solved (optimal_tf_horn.cpp:98)
    assert not optimal_tf_horn(amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt[:2])), _wp(amd), out)
    assert optimal_tf_horn(amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt[:1]), pl2pl=_to_hip_pl2pl(pl[:2])),
                           _wp(amd), out)
    To, rc, _ = oracle.optimal_tf_horn_wp(pt[:1], pl[:2])
    assert rc == 1 and _close(oracle, out.optimalPose, To)
    # where the reference throws
    full = lambda: amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt), pl2pl=_to_hip_pl2pl(pl))  # noqa: E731
    for bad in (dict(w_pl2pl=0.0), dict(robust_kernel=2), dict(w_pt2pt=0.0, w_ln2ln=0.0, w_pl2pl=0.0),
                dict(w_pt2pt=-1.0)):
        with pytest.raises(amd.Mp2pHipError):
            optimal_tf_horn(full(), _wp(amd, **bad), out)
    with pytest.raises(amd.Mp2pHipError):       # no point pairings: eval_centroids_robust asserts
        optimal_tf_horn(amd.Pairings.from_host(ctx, None, pl2pl=_to_hip_pl2pl(pl)), _wp(amd), out)
    with pytest.raises(amd.Mp2pHipError):       # blocks cover fewer pairs than the list
        optimal_tf_horn(amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt), point_weights=[(10, 1.0)]), _wp(amd), out)
    # YAML: WeightParameters.cpp:47-55
    s = amd.Solver_Horn()
    s.initialize({"pairingsWeightParameters": {"use_scale_outlier_detector": True, "scale_outlier_threshold": 1.3,
                                               "robust_kernel": "RobustKernel::Cauchy", "robust_kernel_param": 0.4,
                                               "pair_weights": dict(pt2pt=2, pt2pl=1, pt2ln=1, ln2ln=3, pl2pl=4)}})
    w = s.pairingsWeightParameters
    assert (w.use_scale_outlier_detector, w.scale_outlier_threshold, w.robust_kernel, w.robust_kernel_param,
            w.pair_weights.pt2pt, w.pair_weights.pl2pl) == (True, 1.3, 2, 0.4, 2.0, 4.0)
    with pytest.raises(KeyError):
        amd.Solver_Horn().initialize({"pairingsWeightParameters": {"scale_outlier_threshold": 1.3}})


def _plane_line_pairings(oracle, seed, n_pl, n_ln):
    rng = np.random.default_rng(seed)
    gt = oracle.pose_from_xyzypr(0.4, -0.3, 0.2, 0.06, -0.03, 0.04)
    R, t = gt[:9].reshape(3, 3), gt[9:]
    pl = np.zeros(n_pl, oracle.PAIR_PT2PL)
    lp = rng.uniform(-8, 8, (n_pl, 3))
    nrm = rng.normal(size=(n_pl, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    on = lp @ R.T + t                                        # the point, in the global frame, lies on its plane
    pl["plane"][:, :3], pl["plane"][:, 3] = nrm, -(nrm * on).sum(1)
    pl["centroid"] = on + rng.normal(0, 0.1, (n_pl, 3))
    pl["lx"], pl["ly"], pl["lz"] = lp.astype(np.float32).T
    if n_pl > 20:
        pl[10:14] = pl[0]                                    # equal distances: multimap order
    ln = np.zeros(n_ln, oracle.PAIR_PT2LN)
    lq = rng.uniform(-8, 8, (n_ln, 3))
    ln["director"] = rng.normal(size=(n_ln, 3))
    ln["pbase"] = lq @ R.T + t + ln["director"] * rng.normal(size=(n_ln, 1))
    ln["lx"], ln["ly"], ln["lz"] = lq.T
    return gt, pl, ln


@pytest.mark.parametrize("n_pl,n_ln", [(5000, 800), (3000, 0), (0, 700), (2, 2), (1, 0)])
def test_pt2ln_pl_to_pt2pt_and_solver_horn(amd, oracle, n_pl, n_ln):
    from mp2p_icp_amd.solver import pt2ln_pl_to_pt2pt
    gt, pl, ln = _plane_line_pairings(oracle, 70 + n_pl % 7, n_pl, n_ln)
    guess = oracle.pose_from_xyzypr(0.1, 0.1, -0.1, 0.01, 0.02, -0.01)
    want = oracle.pt2ln_pl_to_pt2pt(pl, ln, guess)
    ctx = amd.default_context()
    # the input's own point pairings must not survive the conversion (pt2ln_pl_to_pt2pt.cpp:49)
    extra = np.zeros(5, oracle.PAIR_PT2PT)
    extra["gx"] = 100.0
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, extra), _to_hip_pt2pl(amd, pl) if n_pl else None,
                               pt2ln=_to_hip_pt2ln(ln) if n_ln else None)
    sc = amd.SolverContext()
    sc.guessRelativePose = guess
    got = pt2ln_pl_to_pt2pt(p, sc).paired_pt2pt
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(got["global"], np.stack([want["gx"], want["gy"], want["gz"]], 1))
    assert np.array_equal(got["local"], np.stack([want["lx"], want["ly"], want["lz"]], 1))
    assert not got["globalIdx"].any() and not got["localIdx"].any()
    if n_pl + n_ln > 100:
        assert 3 <= len(want) < n_pl + n_ln
    # Solver_Horn on such pairings = optimal_tf_horn on the converted list (Solver_Horn.cpp:51-58)
    s = amd.Solver_Horn()
    s.initialize({})
    out = amd.OptimalTF_Result()
    ok = s.optimal_pose(p, out, sc)
    To, rc, _ = oracle.optimal_tf_horn_wp(want, None)
    assert ok == (rc == 1)
    if ok:
        assert _close(oracle, out.optimalPose, To)
    # iterating it converges towards the ground truth when the geometry allows
    if n_pl >= 3000 and n_ln:
        pose = guess
        for _ in range(12):
            sc.guessRelativePose = pose
            assert s.optimal_pose(p, out, sc)
            pose = out.optimalPose
        assert oracle.pose_err(pose, gt) < oracle.pose_err(guess, gt)
