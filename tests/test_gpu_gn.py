"""GPU parity of Solver_GaussNewton (HIP K6/K7/K8 through the C ABI) against the reference's
known-answer tests and the CPU oracle.  Tolerance: solved pose within 1e-5 m / 1e-5 rad of the
oracle on identical inputs (BASELINE.json north_star); normal equations to 1e-9 relative."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEG = math.pi / 180.0
TOL_T, TOL_R = 1e-5, 1e-5


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _to_hip_pt2pt(amd, o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PT2PT)
    h["globalIdx"], h["localIdx"] = o["globalIdx"], o["localIdx"]
    h["global"] = np.stack([o["gx"], o["gy"], o["gz"]], 1)
    h["local"] = np.stack([o["lx"], o["ly"], o["lz"]], 1)
    h["errorSquareAfterTransformation"] = o["errSq"]
    return h


def _to_hip_pt2pl(amd, o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PT2PL)
    h["plane"], h["centroid"] = o["plane"], o["centroid"]
    h["pt_local"] = np.stack([o["lx"], o["ly"], o["lz"]], 1)
    return h


def _solve(amd, pt=None, pl=None, T0=None, solver_params=None, prior=None, point_weights=None):
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt) if pt is not None else None,
                               _to_hip_pt2pl(amd, pl) if pl is not None else None,
                               point_weights=point_weights)
    s = amd.Solver_GaussNewton()
    s.initialize(solver_params or {"maxIterations": 25})
    sc = amd.SolverContext()
    sc.guessRelativePose = T0 if T0 is not None else amd.se3.identity()
    sc.prior = prior
    out = amd.OptimalTF_Result()
    assert s.optimal_pose(p, out, sc)
    return out


def _close(oracle, A, B):
    dt, dr = oracle.pose_err_split(A, B)
    return dt < TOL_T and dr < TOL_R


# ---- tests/test-mp2p_optimize_pt2pl.cpp ------------------------------------------------------
def test_optimize_pt2pl_kat(amd, oracle):
    from test_oracle_kat import PT2PL_POSES, make_pt2pl_kat
    for pose in PT2PL_POSES:
        gt = oracle.pose_from_xyzypr(*pose)
        pt, pl = make_pt2pl_kat(oracle, gt)
        out = _solve(amd, pt, pl)
        assert oracle.pose_err(out.optimalPose, gt) < 1e-3, pose           # reference assertion
        To, it, H, g = oracle.optimal_tf_gauss_newton(pt, pl, None, oracle.pose_identity(),
                                                      oracle.make_gn_params(25))
        assert _close(oracle, out.optimalPose, To), pose                   # parity with the oracle
        assert out.gn["iterations"] == it


# ---- tests/test-mp2p_optimize_with_prior.cpp --------------------------------------------------
def test_optimize_with_prior_kat(amd, oracle):
    from test_oracle_kat import PRIOR_GT, make_prior_kat
    gt = oracle.pose_from_xyzypr(*PRIOR_GT[0])
    pt = make_prior_kat(oracle, gt)
    mean6 = (2.0, 3.0, 4.0, 10 * DEG, 10 * DEG, 10 * DEG)
    mean = oracle.pose_from_xyzypr(*mean6)
    out = _solve(amd, pt)
    assert oracle.pose_err(out.optimalPose, gt) < 1e-3
    for case, rng in ((1, range(0, 3)), (2, range(3, 6))):
        ci = np.zeros((6, 6))
        for i in rng:
            ci[i, i] = 100.0
        out = _solve(amd, pt, prior=amd.PosePrior(mean, ci))
        got = amd.se3.to_xyzypr(out.optimalPose)
        for i in rng:
            assert abs(got[i] - mean6[i]) < 0.05
        To, *_ = oracle.optimal_tf_gauss_newton(
            pt, None, None, oracle.pose_identity(),
            oracle.make_gn_params(25, prior_mean=mean, prior_cov_inv=ci))
        dt, dr = oracle.pose_err_split(out.optimalPose, To)
        assert dt < 1e-4 and dr < 1e-4  # both sides differentiate the prior numerically


KERNELS = [("RobustKernel::None", 0, 1.0), ("RobustKernel::GemanMcClure", 1, 0.15),
           ("RobustKernel::Cauchy", 2, 0.5)]


@pytest.mark.parametrize("kname,kid,kparam", KERNELS)
@pytest.mark.parametrize("n", [3, 100, 4099, 200_000])
def test_random_pt2pt_parity(amd, oracle, kname, kid, kparam, n):
    rng = np.random.default_rng(1234 + n)
    gt = oracle.pose_from_xyzypr(*rng.uniform(-1, 1, 3), *(rng.uniform(-8, 8, 3) * DEG))
    l = rng.uniform(-20, 20, (n, 3))
    R = gt[:9].reshape(3, 3)
    g = l @ R.T + gt[9:] + rng.normal(0, 0.03, (n, 3))
    n_out = n // 10
    if n_out:
        g[rng.choice(n, n_out, replace=False)] += rng.uniform(-2, 2, (n_out, 3))
    pt = np.zeros(n, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    pt["localIdx"] = np.arange(n)
    T0 = oracle.pose_compose(gt, oracle.pose_from_xyzypr(0.2, -0.1, 0.15, 1 * DEG, -2 * DEG, 1.5 * DEG))
    for iters in (1, 3, 10):
        sp = {"maxIterations": iters, "robustKernel": kname, "robustKernelParam": kparam,
              "pair_weights": {"pt2pt": 1.3, "pt2pl": 0.7, "pt2ln": 1, "ln2ln": 1, "pl2pl": 1}}
        out = _solve(amd, pt, None, T0, sp)
        prm = oracle.make_gn_params(iters, kernel=kid, kernelParam=kparam, w_pt2pt=1.3, w_pt2pl=0.7)
        To, it, H, gg = oracle.optimal_tf_gauss_newton(pt, None, None, T0, prm)
        assert _close(oracle, out.optimalPose, To), (iters, oracle.pose_err_split(out.optimalPose, To))
        assert out.gn["iterations"] == it
        if iters == 1:  # normal equations of the (only) linearisation
            assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
            assert np.allclose(out.gn["g"], gg, rtol=1e-9, atol=1e-9 * np.abs(gg).max())


@pytest.mark.parametrize("kname,kid,kparam", KERNELS[:2])
def test_random_mixed_pt2pt_pt2pl_parity(amd, oracle, kname, kid, kparam):
    rng = np.random.default_rng(77)
    n1, n2 = 5000, 7001
    gt = oracle.pose_from_xyzypr(0.3, -0.2, 0.1, 2 * DEG, -1 * DEG, 3 * DEG)
    R = gt[:9].reshape(3, 3)
    l1 = rng.uniform(-15, 15, (n1, 3))
    g1 = l1 @ R.T + gt[9:] + rng.normal(0, 0.02, (n1, 3))
    pt = np.zeros(n1, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l1.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g1.T.astype(np.float32)
    l2 = rng.uniform(-15, 15, (n2, 3))
    w2 = l2 @ R.T + gt[9:]
    nrm = rng.normal(size=(n2, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scale = rng.uniform(0.5, 2.0, n2)  # non-unit normals are legal (errorTerms.cpp:126-134)
    pl = np.zeros(n2, oracle.PAIR_PT2PL)
    d = -(nrm * (w2 + rng.normal(0, 0.02, (n2, 3)))).sum(1)
    pl["plane"] = np.concatenate([nrm, d[:, None]], 1) * scale[:, None]
    pl["centroid"] = w2
    pl["lx"], pl["ly"], pl["lz"] = l2.T.astype(np.float32)
    T0 = oracle.pose_identity()
    for iters in (1, 4):
        sp = {"maxIterations": iters, "robustKernel": kname, "robustKernelParam": kparam}
        out = _solve(amd, pt, pl, T0, sp)
        To, it, H, gg = oracle.optimal_tf_gauss_newton(
            pt, pl, None, T0, oracle.make_gn_params(iters, kernel=kid, kernelParam=kparam))
        assert _close(oracle, out.optimalPose, To), oracle.pose_err_split(out.optimalPose, To)
        if iters == 1:
            assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
            assert np.allclose(out.gn["g"], gg, rtol=1e-9, atol=1e-9 * np.abs(gg).max())


def test_point_weight_blocks(amd, oracle):
    """Pairings::point_weights (optimal_tf_gauss_newton.cpp:159-167), single inner iteration
    (the reference's cursor is not reset between iterations)."""
    rng = np.random.default_rng(9)
    n = 3000
    l = rng.uniform(-10, 10, (n, 3))
    g = l + rng.normal(0, 0.05, (n, 3)) + np.array([0.1, 0.05, -0.02])
    pt = np.zeros(n, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    blocks = [(1000, 0.5), (1500, 2.0), (500, 1.25)]
    out = _solve(amd, pt, None, None, {"maxIterations": 1}, point_weights=blocks)
    To, it, H, gg = oracle.optimal_tf_gauss_newton(pt, None, None, oracle.pose_identity(),
                                                   oracle.make_gn_params(1, weight_blocks=blocks))
    assert _close(oracle, out.optimalPose, To)
    assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())


def test_point_weight_blocks_over_several_inner_iterations(amd, oracle):
    """several inner iterations: block b covers its pairings at EVERY iteration (the oracle's
    reset_weight_cursor_each_iter mode; the reference's own cursor runs off its list from the second
    iteration on -- DESIGN.md section 2); an empty block is refused"""
    rng = np.random.default_rng(10)
    n = 4000
    l = rng.uniform(-10, 10, (n, 3))
    g = l + rng.normal(0, 0.05, (n, 3)) + np.array([0.2, -0.1, 0.05])
    pt = np.zeros(n, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    blocks = [(700, 0.25), (2300, 3.0), (1000, 1.0)]
    for iters in (2, 4):
        out = _solve(amd, pt, None, None, {"maxIterations": iters, "robustKernel": "RobustKernel::Cauchy",
                                           "robustKernelParam": 0.5}, point_weights=blocks)
        To, it, H, gg = oracle.optimal_tf_gauss_newton(
            pt, None, None, oracle.pose_identity(),
            oracle.make_gn_params(iters, kernel=oracle.KERNEL_CAUCHY, kernelParam=0.5, weight_blocks=blocks,
                                  reset_weight_cursor_each_iter=1))
        assert _close(oracle, out.optimalPose, To)
        assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    with pytest.raises(amd.Mp2pHipError):
        _solve(amd, pt, None, None, {"maxIterations": 1}, point_weights=[(1000, 0.5), (0, 2.0), (3000, 1.0)])


def test_convergence_flags_and_empty(amd, oracle):
    # exact pairs at the linearisation point: cost 0 -> break before solving (:344-346)
    from test_oracle_kat import make_prior_kat
    pt = make_prior_kat(oracle, oracle.pose_identity())
    out = _solve(amd, pt, None, None, {"maxIterations": 25})
    assert out.gn["iterations"] == 1 and np.allclose(out.optimalPose, amd.se3.identity())
    with pytest.raises(KeyError):
        amd.Solver_GaussNewton().initialize({})


def test_horn_parity(amd, oracle):
    rng = np.random.default_rng(5)
    gt = oracle.pose_from_xyzypr(1, -2, 0.5, 0.3, -0.1, 0.2)
    n = 50_000
    l = rng.uniform(-10, 10, (n, 3))
    R = gt[:9].reshape(3, 3)
    g = l @ R.T + gt[9:] + rng.normal(0, 0.02, (n, 3))
    pt = np.zeros(n, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt))
    s = amd.Solver_Horn()
    s.initialize({})
    out = amd.OptimalTF_Result()
    assert s.optimal_pose(p, out, amd.SolverContext())
    To, ok = oracle.optimal_tf_horn(pt)
    assert ok and _close(oracle, out.optimalPose, To)


# ---- paired_pt2ln / paired_pl2pl in the solver ---------------------------------------------------
def _to_hip_pt2ln(o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PT2LN)
    h["ln_base"], h["ln_director"] = o["pbase"], o["director"]
    h["pt_local"] = np.stack([o["lx"], o["ly"], o["lz"]], 1)
    return h


def _to_hip_pl2pl(o):
    from mp2p_icp_amd import _lib
    h = np.zeros(len(o), _lib.PAIR_PL2PL)
    for k in ("pl_global", "c_global", "pl_local", "c_local"):
        h[k] = o[k]
    return h


def _solve_all(amd, pt, pl, ln, pp, T0, solver_params):
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt) if pt is not None else None,
                               _to_hip_pt2pl(amd, pl) if pl is not None else None,
                               pt2ln=_to_hip_pt2ln(ln) if ln is not None else None,
                               pl2pl=_to_hip_pl2pl(pp) if pp is not None else None)
    s = amd.Solver_GaussNewton()
    s.initialize(solver_params)
    sc = amd.SolverContext()
    sc.guessRelativePose = T0
    out = amd.OptimalTF_Result()
    assert s.optimal_pose(p, out, sc)
    return out, p


def test_optimize_pt2ln_kat(amd, oracle):
    """tests/test-mp2p_optimize_pt2ln.cpp:25-76 through the HIP solver"""
    from test_oracle_kat import PT2PL_POSES
    for pose in PT2PL_POSES:
        gt = oracle.pose_from_xyzypr(*pose)
        ln = np.zeros(3, oracle.PAIR_PT2LN)
        specs = [((1, 0, 0), (0.5, 0, 0)), ((0, 1, 0), (0, 0.4, 0)), ((0, 0, 1), (0, 0, 0.2))]
        for i, (d, gp) in enumerate(specs):
            ln[i]["pbase"] = (0, 0, 0)
            ln[i]["director"] = d
            ln[i]["lx"], ln[i]["ly"], ln[i]["lz"] = oracle.pose_inverse_compose_point(gt, gp)
        out, _ = _solve_all(amd, None, None, ln, None, oracle.pose_identity(), {"maxIterations": 25})
        assert oracle.pose_err(out.optimalPose, gt) < 1e-3, pose          # reference assertion
        To, it, *_ = oracle.optimal_tf_gauss_newton(None, None, ln, oracle.pose_identity(),
                                                    oracle.make_gn_params(25))
        assert _close(oracle, out.optimalPose, To), pose
        assert out.gn["iterations"] == it


@pytest.mark.parametrize("kname,kid,kparam", KERNELS)
def test_all_term_kinds_parity(amd, oracle, kname, kid, kparam):
    """pt2pt + pt2pl + pt2ln + pl2pl together, non-unit directors / normals, pair weights"""
    from test_oracle_kat import _random_pt_pl_pairs
    rng = np.random.default_rng(123)
    gt = oracle.pose_from_xyzypr(0.2, -0.1, 0.05, 1.5 * DEG, -2 * DEG, 1 * DEG)
    R, t = gt[:9].reshape(3, 3), gt[9:]
    pt, pp = _random_pt_pl_pairs(oracle, rng, gt, 3000, 500, noise=0.01)
    pt["lx"], pt["ly"], pt["lz"] = [pt[k].astype(np.float32) for k in ("lx", "ly", "lz")]
    pp["pl_local"][:, :3] += rng.normal(0, 0.01, (500, 3))              # noisy, non-unit normals
    pp["pl_global"] *= rng.uniform(0.5, 1.5, (500, 1))
    n_ln = 800
    ln = np.zeros(n_ln, oracle.PAIR_PT2LN)
    base = rng.uniform(-10, 10, (n_ln, 3))
    u = rng.normal(size=(n_ln, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    on_line = base + u * rng.uniform(-5, 5, (n_ln, 1))
    loc = (on_line - t) @ R + rng.normal(0, 0.01, (n_ln, 3))
    ln["pbase"], ln["director"] = base, u * rng.uniform(0.9, 1.1, (n_ln, 1))
    ln["lx"], ln["ly"], ln["lz"] = loc.T
    n_pl = 700
    l2 = rng.uniform(-10, 10, (n_pl, 3))
    nrm = rng.normal(size=(n_pl, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    w2 = l2 @ R.T + t
    pl = np.zeros(n_pl, oracle.PAIR_PT2PL)
    pl["plane"] = np.concatenate([nrm, -(nrm * (w2 + rng.normal(0, 0.01, (n_pl, 3)))).sum(1)[:, None]], 1)
    pl["centroid"] = w2
    pl["lx"], pl["ly"], pl["lz"] = l2.T.astype(np.float32)
    weights = {"pt2pt": 1.2, "pt2pl": 0.8, "pt2ln": 1.5, "ln2ln": 1.0, "pl2pl": 2.5}
    T0 = oracle.pose_identity()
    for iters in (1, 5):
        sp = {"maxIterations": iters, "robustKernel": kname, "robustKernelParam": kparam,
              "pair_weights": weights}
        out, p = _solve_all(amd, pt, pl, ln, pp, T0, sp)
        To, it, H, g = oracle.optimal_tf_gauss_newton(
            pt, pl, ln, T0, oracle.make_gn_params(iters, kernel=kid, kernelParam=kparam, w_pt2pt=1.2,
                                                  w_pt2pl=0.8, w_pt2ln=1.5, w_pl2pl=2.5), pl2pl=pp)
        assert _close(oracle, out.optimalPose, To), oracle.pose_err_split(out.optimalPose, To)
        assert out.gn["iterations"] == it
        if iters == 1:
            assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
            assert np.allclose(out.gn["g"], g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    assert p.size() == 3000 + n_pl + n_ln + 500
    assert p.device.counts_lines_planes() == (n_ln, 500)
    # lines and planes alone (no point lists at all)
    out, _ = _solve_all(amd, None, None, ln, pp, T0, {"maxIterations": 6})
    To, *_ = oracle.optimal_tf_gauss_newton(None, None, ln, T0, oracle.make_gn_params(6), pl2pl=pp)
    assert _close(oracle, out.optimalPose, To)


# ---- mp2p_icp::covariance (covariance.cpp:29-141) ---------------------------------------------------
def test_covariance_vs_oracle(amd, oracle):
    from test_oracle_kat import _random_pt_pl_pairs
    rng = np.random.default_rng(31)
    gt = oracle.pose_from_xyzypr(3.0, -2.0, 0.7, 20 * DEG, -5 * DEG, 3 * DEG)
    R, t = gt[:9].reshape(3, 3), gt[9:]
    pt, pp = _random_pt_pl_pairs(oracle, rng, gt, 20_000, 300, noise=0.02)
    pt["lx"], pt["ly"], pt["lz"] = [pt[k].astype(np.float32) for k in ("lx", "ly", "lz")]
    n_pl = 5000
    l2 = rng.uniform(-10, 10, (n_pl, 3))
    nrm = rng.normal(size=(n_pl, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    w2 = l2 @ R.T + t
    pl = np.zeros(n_pl, oracle.PAIR_PT2PL)
    pl["plane"] = np.concatenate([nrm, -(nrm * w2).sum(1)[:, None]], 1) * rng.uniform(0.5, 2, (n_pl, 1))
    pl["centroid"] = w2
    pl["lx"], pl["ly"], pl["lz"] = l2.T.astype(np.float32)
    n_ln = 700
    ln = np.zeros(n_ln, oracle.PAIR_PT2LN)
    base, u = rng.uniform(-10, 10, (n_ln, 3)), rng.normal(size=(n_ln, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    ln["pbase"], ln["director"] = base, u
    ln["lx"], ln["ly"], ln["lz"] = ((base + u * rng.uniform(-5, 5, (n_ln, 1)) - t) @ R).T
    ctx = amd.default_context()
    cases = [(pt, None, None, None), (None, pl, None, None), (pt, pl, ln, pp)]
    for a, b, c, d in cases:
        p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, a) if a is not None else None,
                                   _to_hip_pt2pl(amd, b) if b is not None else None,
                                   pt2ln=_to_hip_pt2ln(c) if c is not None else None,
                                   pl2pl=_to_hip_pl2pl(d) if d is not None else None)
        cov = amd.covariance(p, gt)
        want, H, ok = oracle.covariance(a, b, c, d, gt)
        assert ok
        from mp2p_icp_amd import core
        _, Hg, pd = core.covariance(ctx, p.device, gt)
        assert pd
        assert np.allclose(Hg, H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
        assert np.allclose(cov, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
        assert np.allclose(cov, cov.T, rtol=0, atol=1e-12 * np.abs(cov).max())
    # point pairs alone: the translation block of H is N * I whatever the pose
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt))
    from mp2p_icp_amd import core
    _, Hg, _ = core.covariance(ctx, p.device, gt)
    assert np.allclose(Hg[:3, :3], len(pt) * np.eye(3), rtol=1e-6, atol=1e-3)
    # no pairings -> 1e6 * I (covariance.cpp:33-39)
    assert np.array_equal(amd.covariance(amd.Pairings(ctx), gt), 1e6 * np.eye(6))
    # rank-deficient (plane normals only constrain the rotation): flagged, not a garbage inverse
    p = amd.Pairings.from_host(ctx, pl2pl=_to_hip_pl2pl(pp))
    cov, Hg, pd = core.covariance(ctx, p.device, gt)
    wc, wH, wok = oracle.covariance(None, None, None, pp, gt)
    assert pd == wok


def test_more_than_eight_weight_blocks(amd, oracle):
    """up to MP2P_HIP_MAX_WEIGHT_BLOCKS = 32 `point_weights` blocks (one per layer pair that produced pairings,
    Matcher_Points_Base.cpp:121-125; rounds 1-4 took 8): Gauss-Newton over 1 and 4 inner iterations and Horn against the oracle,
    33 blocks refused"""
    from mp2p_icp_amd.solver import WeightParameters, optimal_tf_horn
    rng = np.random.default_rng(77)
    n = 6000
    l = rng.uniform(-10, 10, (n, 3))
    g = l + rng.normal(0, 0.05, (n, 3)) + np.array([0.15, -0.1, 0.05])
    pt = np.zeros(n, oracle.PAIR_PT2PT)
    pt["lx"], pt["ly"], pt["lz"] = l.T.astype(np.float32)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    for nb in (9, 20, 32):
        cut = np.sort(rng.choice(np.arange(1, n), nb - 1, replace=False))
        blocks = [(int(c), float(w)) for c, w in zip(np.diff(np.concatenate([[0], cut, [n]])), rng.choice([0.25, 0.5, 1.0, 2.0, 4.0], nb))]
        for iters in (1, 4):
            out = _solve(amd, pt, None, None, {"maxIterations": iters, "robustKernel": "RobustKernel::GemanMcClure",
                                               "robustKernelParam": 0.3}, point_weights=blocks)
            To, it, H, gg = oracle.optimal_tf_gauss_newton(
                pt, None, None, oracle.pose_identity(),
                oracle.make_gn_params(iters, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.3, weight_blocks=blocks,
                                      reset_weight_cursor_each_iter=1))
            assert _close(oracle, out.optimalPose, To), (nb, iters)
            assert np.allclose(out.gn["H"], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
        # Horn with the same blocks
        To, rc, fl = oracle.optimal_tf_horn_wp(pt, None, point_weights=blocks)
        res = amd.OptimalTF_Result()
        p = amd.Pairings.from_host(amd.default_context(), _to_hip_pt2pt(amd, pt), point_weights=blocks)
        assert optimal_tf_horn(p, WeightParameters(), res) and rc == 1
        assert _close(oracle, res.optimalPose, To), nb
    with pytest.raises((ValueError, amd.Mp2pHipError)):
        _solve(amd, pt, None, None, {"maxIterations": 1}, point_weights=[(1, 1.0)] * 32 + [(n - 32, 1.0)])
