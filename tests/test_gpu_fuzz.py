"""Seeded differential fuzzing of the matchers (distance threshold, point-to-plane, inlier ratio,
adaptive) against the CPU oracle: random geometries
(clusters, exact lattices with many equal distances, collinear and coplanar sets, duplicates, large
coordinate offsets, tiny extents), random thresholds / angular thresholds / pairingsPerPoint /
voxel sizes / bitmap on-off / tile sizes, and pose sequences (warm start).  Lists must be bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(rng, kind, n):
    if kind == "uniform":
        return rng.uniform(-5, 5, (n, 3))
    if kind == "clusters":
        c = rng.uniform(-20, 20, (max(2, n // 200), 3))
        return c[rng.integers(0, len(c), n)] + rng.normal(0, 0.3, (n, 3))
    if kind == "lattice":                       # exact ties everywhere
        k = max(2, int(round(n ** (1 / 3))))
        g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.25
        return g[rng.permutation(len(g))[:n]]
    if kind == "plane":
        p = rng.uniform(-10, 10, (n, 3))
        p[:, 2] = 0.0
        return p
    if kind == "line":
        p = np.zeros((n, 3))
        p[:, 0] = rng.uniform(-50, 50, n)
        return p
    if kind == "far_offset":
        return rng.uniform(-3, 3, (n, 3)) + np.array([4000.0, -2500.0, 300.0])
    if kind == "tiny":
        return rng.uniform(-1e-3, 1e-3, (n, 3))
    raise ValueError(kind)


KINDS = ["uniform", "clusters", "lattice", "plane", "line", "far_offset", "tiny"]


# (MP2P_FUZZ_PT_SEEDS=a:b runs another range of seeds; campaigns: profiles/r05_fuzz_campaign.txt)
_PT_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_PT_SEEDS", "0:36").split(":")])


@pytest.mark.parametrize("seed", _PT_SEEDS)
def test_fuzz_pt2pt(oracle, seed):
    import mp2p_icp_amd as amd
    rng = np.random.default_rng(1000 + seed)
    kind = KINDS[seed % len(KINDS)]
    n_g = int(rng.integers(50, 30000))
    n_l = int(rng.integers(1, 6000))
    g = _cloud(rng, kind, n_g).astype(np.float32)
    scale = float(np.ptp(g, axis=0).max()) or 1.0
    src = g[rng.integers(0, len(g), n_l)].astype(np.float64)
    l = (src + rng.normal(0, 0.01 * scale, (n_l, 3)) * rng.integers(0, 2)).astype(np.float32)
    if rng.random() < 0.5:                       # some far outliers
        k = max(1, n_l // 10)
        l[:k] += (rng.uniform(-1, 1, (k, 3)) * scale * 3).astype(np.float32)
    if rng.random() < 0.3:
        g[: len(g) // 10] = g[len(g) // 10: 2 * (len(g) // 10)][: len(g) // 10]   # duplicates
    thr = float(rng.choice([0.02, 0.1, 0.5, 2.0]) * scale / 10.0)
    ang = float(rng.choice([0.0, 0.0, 0.3]))
    K = int(rng.choice([1, 1, 1, 2, 5]))
    layer_kw = {}
    if rng.random() < 0.3:
        layer_kw["cell_size"] = float(rng.choice([0.02, 0.2, 1.5]) * scale / 10.0)
    if rng.random() < 0.3:
        layer_kw["no_occupancy_bitmap"] = int(rng.choice([1, 2, 4]))   # index variants (mp2p_hip_map_params)
    params = {"threshold": thr, "thresholdAngularDeg": ang, "pairingsPerPoint": K,
              "hip_queries_per_wave": int(rng.choice([0, 16, 32, 64])),
              "allowMatchAlreadyMatchedGlobalPoints": bool(rng.random() < 0.3)}
    # K > 1: the nn_radius_search meaning (TBB build, the default here) or nn_multiple_search (sequential build)
    radius_mode = bool((seed // len(KINDS)) % 2 == 0)
    params["hip_multi_search_radius_mode"] = radius_mode
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g, **layer_kw)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize(params)
    T = amd.se3.identity()
    for step in range(4):                        # a pose sequence: warm start between the calls
        xi = np.concatenate([rng.normal(0, 0.02 * scale, 3), rng.normal(0, 0.02, 3)]) * (step > 0)
        T = amd.se3.compose(T, amd.se3.exp(xi))
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, thr, ang,
                                       pairingsPerPoint=K, tree=tree, multi_search_radius_mode=int(radius_mode),
                                       allowMatchAlreadyMatchedGlobalPoints=params["allowMatchAlreadyMatchedGlobalPoints"])
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        got = pairs.paired_pt2pt
        info = (kind, n_g, n_l, thr, ang, K, layer_kw, params, step)
        assert len(got) == len(want), info
        assert np.array_equal(got["localIdx"], want["localIdx"]), info
        assert np.array_equal(got["globalIdx"], want["globalIdx"]), info
        assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32)), info
        assert pairs.potential_pairings == pot


# (MP2P_FUZZ_PL_SEEDS=a:b runs another range of seeds: the end-of-round campaign of round 5 ran 14:214 after the directory of the
#  k-nearest search was changed to 256 voxels per round trip)
_PL_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_PL_SEEDS", "0:14").split(":")])


@pytest.mark.parametrize("seed", _PL_SEEDS)
def test_fuzz_pt2pl_and_inlier_ratio(oracle, seed):
    import mp2p_icp_amd as amd
    rng = np.random.default_rng(5000 + seed)
    kind = KINDS[seed % len(KINDS)]
    n_g = int(rng.integers(200, 20000))
    n_l = int(rng.integers(1, 3000))
    g = _cloud(rng, kind, n_g).astype(np.float32)
    scale = float(np.ptp(g, axis=0).max()) or 1.0
    l = (g[rng.integers(0, len(g), n_l)].astype(np.float64) + rng.normal(0, 0.01 * scale, (n_l, 3))).astype(np.float32)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    T = amd.se3.exp(np.concatenate([rng.normal(0, 0.01 * scale, 3), rng.normal(0, 0.01, 3)]))
    # point-to-plane
    P = dict(distanceThreshold=0.05 * scale, searchRadius=float(rng.choice([0.03, 0.1])) * scale,
             knn=int(rng.choice([5, 6, 9, 16])), minimumPlanePoints=5,
             planeEigenThreshold=float(rng.choice([0.01, 0.1])))
    want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, tree=tree, **P)
    m = amd.Matcher_Point2Plane()
    m.initialize(P)
    pairs = amd.Pairings()
    assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
    assert np.array_equal(pairs.paired_pt2pl_local_idx, widx), (kind, n_g, n_l, P)
    if len(widx):
        s = max(1.0, float(np.abs(want["plane"]).max()))
        assert np.allclose(pairs.paired_pt2pl["plane"], want["plane"], rtol=0, atol=1e-9 * s)
    assert pairs.potential_pairings == pot
    # inlier ratio
    ratio = float(rng.choice([0.2, 0.5, 0.9]))
    want, pot = oracle.match_inlier_ratio(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, ratio, tree=tree)
    m = amd.Matcher_Points_InlierRatio()
    m.initialize({"inliersRatio": ratio})
    pairs = amd.Pairings()
    assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
    got = pairs.paired_pt2pt
    assert len(got) == len(want), (kind, n_g, n_l, ratio)
    assert np.array_equal(got["localIdx"], want["localIdx"]) and np.array_equal(got["globalIdx"], want["globalIdx"])
    assert pairs.potential_pairings == pot
    # adaptive matcher (k-NN lists, histogram, plane / point selection)
    A = dict(confidenceInterval=float(rng.choice([0.5, 0.8, 0.95])), firstToSecondDistanceMax=float(rng.choice([1.1, 2.0])),
             absoluteMaxSearchDistance=float(rng.choice([0.05, 0.3])) * scale, minimumCorrDist=0.002 * scale,
             enableDetectPlanes=bool(rng.random() < 0.6), maxPt2PtCorrespondences=int(rng.choice([1, 2, 4])),
             planeSearchPoints=int(rng.choice([5, 8, 14])), planeMinimumFoundPoints=int(rng.choice([3, 5])),
             planeMinimumDistance=0.05 * scale, planeEigenThreshold=float(rng.choice([0.01, 0.1])))
    for pose in (T, amd.se3.identity()):
        r = oracle.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose, tree=tree, **A)
        m = amd.Matcher_Adaptive()
        m.initialize(A)
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, pose, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        info = (kind, n_g, n_l, A)
        assert m.last_histogram["valid"] == r["hist"]["valid"], info
        if r["hist"]["valid"]:
            assert m.last_histogram["bins"] == r["hist"]["bins"].tolist(), info
            assert m.last_ci_high == r["ci_high"], info
        got = pairs.paired_pt2pt
        assert len(got) == len(r["pt2pt"]), info
        assert np.array_equal(got["localIdx"], r["pt2pt"]["localIdx"]), info
        assert np.array_equal(got["globalIdx"], r["pt2pt"]["globalIdx"]), info
        assert np.array_equal(pairs.paired_pt2pl_local_idx, r["pl_local_idx"]), info
        assert pairs.potential_pairings == r["potential"]


# (MP2P_FUZZ_SEEDS=a:b runs another range of seeds: the end-of-round campaign of round 4 ran 48..448, all equal)
_R4_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_SEEDS", "0:48").split(":")])


@pytest.mark.parametrize("seed", _R4_SEEDS)
def test_fuzz_round4_search_paths(oracle, seed, monkeypatch):
    """the point-to-point search's round-4 paths under random knobs, a fresh context per case: brick lists in the tile
    kernel (also with a budget of 1 or 8 bricks: the coarse dense fallback), tiny candidate budgets (passes cut short in
    flight, queries handed on with partial bounds), cost classes with a threshold everything / nothing exceeds, the
    search-skip certificate in all three modes, the empty-room bound; pose sequences from 10^-5 to 0.3 of the scene with
    jumps, local points taken in some calls, large and tiny thresholds.  Lists bit-exact against the oracle at every call."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core
    rng = np.random.default_rng(77000 + seed)
    kind = KINDS[seed % len(KINDS)]
    n_g = int(rng.integers(2000, 120000))
    n_l = int(rng.integers(64, 12000))
    g = _cloud(rng, kind, n_g).astype(np.float32)
    scale = float(np.ptp(g, axis=0).max()) or 1.0
    src = g[rng.integers(0, len(g), n_l)].astype(np.float64)
    off = rng.normal(0, 1, (n_l, 3)) * scale * float(rng.choice([0.0, 0.002, 0.02, 0.1]))
    l = (src + off).astype(np.float32)
    if rng.random() < 0.6:                       # far outliers: nothing in reach
        k = max(1, n_l // int(rng.choice([3, 10])))
        l[:k] += (rng.uniform(-1, 1, (k, 3)) * scale * float(rng.choice([0.5, 3.0]))).astype(np.float32)
    thr = float(rng.choice([0.005, 0.03, 0.1, 0.4, 1.5]) * scale)
    tune = ",".join([f"nn_cert={int(rng.choice([0, 1, 2, 2]))}", f"nn_cert_step_mm={int(rng.choice([1, 50, 100000]))}",
                     f"tile_cand_cap={int(rng.choice([6144, 600, 100]))}", f"tile_brick_budget={int(rng.choice([512, 8, 1]))}",
                     f"hard_cand={int(rng.choice([1700, 50, 0]))}", f"tile_bricks={int(rng.choice([1, 1, 0]))}",
                     f"empty_room={int(rng.choice([1, 1, 0]))}", f"coop_max={int(rng.choice([4, 0]))}",
                     # round 5 (drawn after the clouds and the threshold: those are the ones of round 4's campaign)
                     f"tile_select={int(rng.choice([1, 1, 1, 0]))}", f"nn_direct={int(rng.choice([1, 0, 0]))}",
                     f"grp_all_bricks={int(rng.choice([6, 0, 100]))}",
                     # round 6: the one-query kernel's wider pass for a query with nothing within the threshold (its bound only)
                     f"far_pass={int(rng.choice([1, 0]))}"])
    monkeypatch.setenv("MP2P_HIP_TUNE", tune)
    layer_kw = {}
    if rng.random() < 0.3:
        layer_kw["cell_size"] = float(rng.choice([0.005, 0.03, 0.2]) * scale)
    if rng.random() < 0.2:
        layer_kw["no_occupancy_bitmap"] = int(rng.choice([1, 2, 4]))
    allow_g = int(rng.random() < 0.3)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2], **layer_kw)
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, n_l, 0)
    ms = core.DeviceMatchState(ctx, n_g, n_l)
    prm = _lib.Pt2PtParams(thr, 0.0, 1, 0, allow_g, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    T = amd.se3.exp(np.concatenate([rng.normal(0, 0.01 * scale, 3), rng.normal(0, 0.01, 3)]))
    for step in range(7):
        s = float(rng.choice([1e-5, 1e-4, 1e-3, 1e-2, 0.3])) * (step > 0)
        T = amd.se3.compose(T, amd.se3.exp(np.concatenate([rng.normal(0, s * scale, 3), rng.normal(0, s, 3)])))
        lt = np.zeros(n_l, np.uint8)
        if rng.random() < 0.3:
            lt[rng.integers(0, n_l, max(1, n_l // 7))] = 1
        gt = np.zeros(n_g, np.uint8)
        want, pot = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, thr, 0.0, tree=tree,
                                       allowMatchAlreadyMatchedGlobalPoints=bool(allow_g), local_taken=lt.copy(), global_taken=gt.copy())
        ms.upload(gt, lt)
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, T, prm, ms, pairs)
        got = pairs.download_pt2pt()
        info = (kind, n_g, n_l, thr, tune, layer_kw, allow_g, step)
        assert len(got) == len(want), info
        assert np.array_equal(got["localIdx"], want["localIdx"]), info
        assert np.array_equal(got["globalIdx"], want["globalIdx"]), info
        assert np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32)), info


# ---- Solver_GaussNewton (optimal_tf_gauss_newton.cpp:60-330): random mixes of the four pairing kinds the device holds, robust kernels,
#      pair weights, weight blocks, priors, iteration counts; pose within 1e-5 m / 1e-5 rad of the oracle on the same lists
#      (MP2P_FUZZ_GN_SEEDS=a:b runs another range of seeds)
_GN_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_GN_SEEDS", "0:24").split(":")])


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


@pytest.mark.parametrize("seed", _GN_SEEDS)
def test_fuzz_gauss_newton(oracle, seed):
    import mp2p_icp_amd as amd
    from test_gpu_gn import KERNELS, _to_hip_pl2pl, _to_hip_pt2ln, _to_hip_pt2pl, _to_hip_pt2pt
    rng = np.random.default_rng(9000 + seed)
    gt = oracle.pose_from_xyzypr(*rng.uniform(-1, 1, 3), *rng.uniform(-0.15, 0.15, 3))
    R, t = gt[:9].reshape(3, 3), gt[9:]
    ext = float(rng.choice([2.0, 10.0, 60.0]))
    noise = float(rng.choice([0.0, 0.002, 0.02])) * ext / 10.0
    n_pt, n_pl, n_ln, n_pp = [int(rng.integers(0, m)) if rng.random() < 0.7 else 0 for m in (4000, 1500, 800, 400)]
    if n_pt + n_pl + n_ln < 12:
        n_pt += 40                                           # enough terms for a well-posed system
    # point-to-point (a share of gross outliers for the robust kernels)
    g = rng.uniform(-ext, ext, (n_pt, 3))
    loc = (g - t) @ R + rng.normal(0, noise, (n_pt, 3))
    bad = rng.random(n_pt) < float(rng.choice([0.0, 0.05]))
    loc[bad] += rng.uniform(-1, 1, (int(bad.sum()), 3)) * ext * 0.2
    pt = np.zeros(n_pt, oracle.PAIR_PT2PT)
    pt["gx"], pt["gy"], pt["gz"] = g.T.astype(np.float32)
    pt["lx"], pt["ly"], pt["lz"] = loc.T.astype(np.float32)
    pt["globalIdx"] = pt["localIdx"] = np.arange(n_pt)
    # point-to-plane
    w2 = rng.uniform(-ext, ext, (n_pl, 3))
    nrm = _unit(rng, n_pl) * rng.uniform(0.8, 1.2, (n_pl, 1))
    pl = np.zeros(n_pl, oracle.PAIR_PT2PL)
    pl["plane"] = np.concatenate([nrm, -(nrm * (w2 + rng.normal(0, noise, (n_pl, 3)))).sum(1)[:, None]], 1)
    pl["centroid"] = w2
    pl["lx"], pl["ly"], pl["lz"] = ((w2 - t) @ R).T.astype(np.float32)
    # point-to-line
    base = rng.uniform(-ext, ext, (n_ln, 3))
    u = _unit(rng, n_ln)
    on_line = base + u * rng.uniform(-ext / 2, ext / 2, (n_ln, 1))
    ln = np.zeros(n_ln, oracle.PAIR_PT2LN)
    ln["pbase"], ln["director"] = base, u * rng.uniform(0.9, 1.1, (n_ln, 1))
    ln["lx"], ln["ly"], ln["lz"] = ((on_line - t) @ R + rng.normal(0, noise, (n_ln, 3))).T
    # plane-to-plane (normals)
    cg, ng = rng.uniform(-ext, ext, (n_pp, 3)), _unit(rng, n_pp)
    cl, nl = (cg - t) @ R, ng @ R
    pp = np.zeros(n_pp, oracle.PAIR_PL2PL)
    pp["pl_global"] = np.concatenate([ng, -(ng * cg).sum(1)[:, None]], 1) * rng.uniform(0.5, 1.5, (n_pp, 1))
    pp["c_global"] = cg
    pp["pl_local"] = np.concatenate([nl + rng.normal(0, 0.005, (n_pp, 3)), -(nl * cl).sum(1)[:, None]], 1)
    pp["c_local"] = cl
    kname, kid, kparam = KERNELS[int(rng.integers(0, len(KERNELS)))]
    kparam = float(kparam * rng.choice([0.5, 1.0, 4.0]) * ext / 10.0)
    iters = int(rng.integers(1, 9))
    wts = {k: float(rng.choice([0.5, 1.0, 2.5])) for k in ("pt2pt", "pt2pl", "pt2ln", "pl2pl")}
    T0 = oracle.pose_from_xyzypr(*(np.array(amd.se3.to_xyzypr(gt)) + np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.02, 3)])))
    blocks, reset = None, 0
    if n_pt >= 10 and rng.random() < 0.3:                    # Pairings::point_weights over the pt2pt list
        cut = sorted(rng.choice(np.arange(1, n_pt), size=min(int(rng.integers(1, 32)), n_pt - 1), replace=False).tolist())
        cnt = np.diff([0] + cut + [n_pt]).tolist()
        blocks, reset = [(int(c), float(rng.choice([0.25, 1.0, 3.0]))) for c in cnt], 1
    prior, pm, pci = None, None, None
    if rng.random() < 0.25:
        pm = oracle.pose_from_xyzypr(*(np.array(amd.se3.to_xyzypr(gt)) + rng.normal(0, 0.01, 6)))
        pci = np.diag(rng.choice([0.0, 10.0, 1000.0], 6))
        prior = amd.PosePrior(pm, pci)
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt) if n_pt else None, _to_hip_pt2pl(amd, pl) if n_pl else None,
                               pt2ln=_to_hip_pt2ln(ln) if n_ln else None, pl2pl=_to_hip_pl2pl(pp) if n_pp else None,
                               point_weights=blocks)
    s = amd.Solver_GaussNewton()
    s.initialize({"maxIterations": iters, "robustKernel": kname, "robustKernelParam": kparam,
                  "pair_weights": dict(wts, ln2ln=1.0)})
    sc = amd.SolverContext()
    sc.guessRelativePose, sc.prior = T0, prior
    out = amd.OptimalTF_Result()
    assert s.optimal_pose(p, out, sc)
    To, it, H, gg = oracle.optimal_tf_gauss_newton(
        pt if n_pt else None, pl if n_pl else None, ln if n_ln else None, T0,
        oracle.make_gn_params(iters, kernel=kid, kernelParam=kparam, w_pt2pt=wts["pt2pt"], w_pt2pl=wts["pt2pl"], w_pt2ln=wts["pt2ln"],
                              w_pl2pl=wts["pl2pl"], prior_mean=pm, prior_cov_inv=pci, weight_blocks=blocks,
                              reset_weight_cursor_each_iter=reset), pl2pl=pp if n_pp else None)
    dt, dr = oracle.pose_err_split(out.optimalPose, To)
    info = (seed, n_pt, n_pl, n_ln, n_pp, kname, kparam, iters, wts, blocks, prior is not None, ext, noise)
    tol = 1e-4 if prior is not None else 1e-5             # (both sides differentiate the prior numerically: test_gpu_gn.py)
    assert oracle.pose_err(To, T0) > 1e-6, info             # (the comparison is not between two untouched guesses)
    assert dt < tol * max(1.0, ext / 10.0) and dr < tol, (dt, dr, info)


# ---- Matcher_Point2Plane over pose SEQUENCES on one matcher instance: the warm start (start radius from the previous k-th distance) and
#      the skip certificate (a query whose previous list is certainly still its k nearest is not searched) only act from the second
#      call on; steps from 1e-5 to 0.3 of the scene, back-and-forth jumps, repeated poses
#      (MP2P_FUZZ_PLSEQ_SEEDS=a:b runs another range of seeds)
_PLSEQ_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_PLSEQ_SEEDS", "0:16").split(":")])


@pytest.mark.parametrize("seed", _PLSEQ_SEEDS)
def test_fuzz_pt2pl_pose_sequences(oracle, seed):
    import mp2p_icp_amd as amd
    rng = np.random.default_rng(7000 + seed)
    kind = KINDS[seed % len(KINDS)]
    n_g = int(rng.integers(300, 40000))
    n_l = int(rng.integers(1, 4000))
    g = _cloud(rng, kind, n_g).astype(np.float32)
    scale = float(np.ptp(g, axis=0).max()) or 1.0
    l = (g[rng.integers(0, len(g), n_l)].astype(np.float64) + rng.normal(0, 0.01 * scale, (n_l, 3))).astype(np.float32)
    if rng.random() < 0.4:                       # some far outliers
        k = max(1, n_l // 8)
        l[:k] += (rng.uniform(-1, 1, (k, 3)) * scale * 2).astype(np.float32)
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    pcG = amd.metric_map_t({"raw": amd.PointLayer(g)})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(l)})
    P = dict(distanceThreshold=float(rng.choice([0.02, 0.05, 0.2])) * scale, searchRadius=float(rng.choice([0.03, 0.1, 0.3])) * scale,
             knn=int(rng.choice([5, 6, 9, 16])), minimumPlanePoints=int(rng.choice([3, 5])),
             planeEigenThreshold=float(rng.choice([0.01, 0.1])))
    m = amd.Matcher_Point2Plane()
    m.initialize(P)
    # round 6: odd seeds on the ball-rule / matrix-pipe search kernel, even ones on the box-rule kernel (the library's own choice goes
    # by the layer's size: these layers are all small); a campaign forces one through MP2P_HIP_TUNE=pl_select=...
    from mp2p_icp_amd import core
    forced = "pl_select" in os.environ.get("MP2P_HIP_TUNE", "")
    if not forced:
        core.default_context().set_tune("pl_select=%d" % (seed % 2))
    # ... and the certificate read / its margin staged after steps below 10 mm only (the default), always, after any step below 1 mm or
    # 100 m: calls that skip it leave bounds without the margin, which the next small step's call certifies from
    core.default_context().set_tune("pl_cert_step_mm=%d" % [10, 0, 1, 100000][(seed // 2) % 4])
    try:
        _pl_pose_sequence(amd, oracle, rng, g, l, tree, pcG, pcL, P, m, scale, seed, kind, n_g, n_l)
    finally:
        core.default_context().set_tune("pl_cert_step_mm=10")
        if not forced:
            core.default_context().set_tune("pl_select=-1")


def _pl_pose_sequence(amd, oracle, rng, g, l, tree, pcG, pcL, P, m, scale, seed, kind, n_g, n_l):
    # (poses about the cloud's centre: a rotation about the origin of a far-offset cloud is a jump of its own)
    c = g.mean(0).astype(np.float64)
    to_c, from_c = amd.se3.exp(np.concatenate([-c, np.zeros(3)])), amd.se3.exp(np.concatenate([c, np.zeros(3)]))
    xi = np.concatenate([rng.normal(0, 0.01 * scale, 3), rng.normal(0, 0.01, 3)])
    seen = []
    for call in range(7):
        step = float(rng.choice([0.0, 1e-5, 1e-3, 1e-2, 0.3]))
        if seen and rng.random() < 0.2:
            xi = seen[int(rng.integers(0, len(seen)))].copy()           # back to an earlier pose
        else:
            xi = xi + np.concatenate([rng.normal(0, step * scale, 3), rng.normal(0, step, 3)])
        seen.append(xi.copy())
        T = amd.se3.compose(from_c, amd.se3.compose(amd.se3.exp(xi), to_c))
        want, widx, pot = oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, tree=tree, **P)
        pairs = amd.Pairings()
        assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        info = (seed, call, kind, n_g, n_l, P, step)
        assert np.array_equal(pairs.paired_pt2pl_local_idx, widx), info
        if len(widx):
            s = max(1.0, float(np.abs(want["plane"]).max()))
            assert np.allclose(pairs.paired_pt2pl["plane"], want["plane"], rtol=0, atol=1e-9 * s), info
            assert np.allclose(pairs.paired_pt2pl["centroid"], want["centroid"], rtol=0, atol=1e-9 * s), info
        assert pairs.potential_pairings == pot, info


# ---- Matcher_Points_Base::impl_match over random LAYER configurations (Matcher_Points_Base.cpp:30-130): 1-3 global x 1-3 local layers of
#      random sizes, default same-name matching or a random `pointLayerMatches` list (weights on some entries, entries naming layers a map
#      lacks), the MatchState flags; pair order, `point_weights` blocks and `potential_pairings` against the oracle run layer pair by layer
#      pair in std::map order with the MatchState bits carried along (MP2P_FUZZ_LAYER_SEEDS=a:b runs another range of seeds)
_LAYER_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_LAYER_SEEDS", "0:12").split(":")])


@pytest.mark.parametrize("seed", _LAYER_SEEDS)
def test_fuzz_layer_configurations(oracle, seed):
    import mp2p_icp_amd as amd
    from test_gpu_multilayer import _oracle_layers, _same
    rng = np.random.default_rng(11000 + seed)
    kind = KINDS[seed % (len(KINDS) - 1)]                                  # (not "tiny")
    base = _cloud(rng, kind, int(rng.integers(2000, 20000))).astype(np.float32)
    scale = float(np.ptp(base, axis=0).max()) or 1.0
    names = ["a", "b", "c", "d"]
    G = {n: base[rng.random(len(base)) < rng.uniform(0.2, 0.9)] for n in rng.choice(names, int(rng.integers(1, 4)), replace=False)}
    G = {k: (v if len(v) else base[:5]) for k, v in G.items()}
    L = {}
    for n in rng.choice(names, int(rng.integers(1, 4)), replace=False):
        k = int(rng.integers(1, 1500))
        L[n] = (base[rng.integers(0, len(base), k)].astype(np.float64) + rng.normal(0, 0.01 * scale, (k, 3))).astype(np.float32)
    thr = float(rng.choice([0.02, 0.05, 0.2])) * scale
    allow_l, allow_g = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    T = amd.se3.exp(np.concatenate([rng.normal(0, 0.005 * scale, 3), np.zeros(3)]))
    params = {"threshold": thr, "thresholdAngularDeg": 0.0, "allowMatchAlreadyMatchedPoints": allow_l,
              "allowMatchAlreadyMatchedGlobalPoints": allow_g}
    throws = False
    if rng.random() < 0.3:                                                  # default: every global layer against its namesake
        plan = [(n, n, None) for n in sorted(G)]                            # (a missing namesake is skipped, :74-78)
    else:
        cfg, seen = [], set()
        for _ in range(int(rng.integers(1, 6))):
            gn = str(rng.choice(names))
            ln = str(rng.choice(names if rng.random() < 0.15 else sorted(L)))   # (now and then a layer the local map lacks)
            if (gn, ln) in seen:
                continue
            seen.add((gn, ln))
            e = {"global": gn, "local": ln}
            if rng.random() < 0.6:
                e["weight"] = float(rng.choice([0.5, 1.0, 2.0]))
            cfg.append(e)
        params["pointLayerMatches"] = cfg
        # the loop runs over the global map's layers in name order, then that layer's entries in local-name order (:40-67); every
        # configured entry carries a weight (1 when not given), and one naming a local layer the map lacks throws (:79-86)
        plan = [(e["global"], e["local"], e.get("weight", 1.0)) for e in sorted(cfg, key=lambda e: (e["global"], e["local"]))
                if e["global"] in G]
        throws = any(ln not in L for _, ln, _ in plan)
    pcG = amd.metric_map_t({k: amd.PointLayer(v) for k, v in G.items()})
    pcL = amd.metric_map_t({k: amd.PointLayer(v) for k, v in L.items()})
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize(params)
    pairs = amd.Pairings()
    info = (seed, kind, {k: len(v) for k, v in G.items()}, {k: len(v) for k, v in L.items()}, params)
    if throws:
        with pytest.raises(RuntimeError, match="not found"):
            m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs)
        return
    assert m.match(pcG, pcL, T, amd.MatchContext(), amd.MatchState(pcG, pcL), pairs), info
    want, blocks, pot = _oracle_layers(oracle, G, L, T, thr, plan, allow_local=allow_l, allow_global=allow_g)
    _same(pairs.paired_pt2pt, want)
    assert pairs.point_weights == blocks, (pairs.point_weights, blocks, info)
    assert pairs.potential_pairings == pot, info


# ---- optimal_tf_horn with WeightParameters (optimal_tf_horn.cpp:77-252, visit_correspondences.h:38-212): random scenes, pair weights,
#      robust kernels with an estimate, the scale outlier detector, up to 32 point_weights blocks (incl. blocks that run out: the reference
#      throws); pose 1e-5, flagged outliers equal (MP2P_FUZZ_HORN_SEEDS=a:b runs another range of seeds)
_HORN_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_HORN_SEEDS", "0:16").split(":")])


@pytest.mark.parametrize("seed", _HORN_SEEDS)
def test_fuzz_horn(oracle, seed):
    import mp2p_icp_amd as amd
    from mp2p_icp_amd.solver import WeightParameters, optimal_tf_horn
    from test_gpu_gn import _to_hip_pl2pl, _to_hip_pt2pt
    from test_oracle_kat import horn_scene
    rng = np.random.default_rng(13000 + seed)
    n_pt, n_pl = int(rng.integers(3, 6000)), int(rng.integers(0, 300)) if rng.random() < 0.5 else 0
    n_out = int(n_pt * rng.choice([0.0, 0.05, 0.2]))
    gt, pt, pl = horn_scene(oracle, 14000 + seed, n_pt=n_pt, n_pl=n_pl, noise=float(rng.choice([0.0, 0.01, 0.05])), outliers=n_out)
    kw = dict(w_pt2pt=float(rng.choice([0.3, 1.0, 2.0])), w_pl2pl=float(rng.choice([0.5, 1.0, 4.0])))
    if rng.random() < 0.4:
        kw.update(use_scale_outlier_detector=True, scale_outlier_threshold=float(rng.choice([1.05, 1.2, 1.5])))
    if rng.random() < 0.4:
        kw.update(robust_kernel=int(rng.choice([1, 2])), robust_kernel_param=float(rng.choice([0.3, 1.0])),
                  currentEstimateForRobust=gt if rng.random() < 0.5 else oracle.pose_identity())
    blocks = None
    if n_pt >= 40 and rng.random() < 0.4:
        cut = np.sort(rng.choice(np.arange(1, n_pt), int(rng.integers(1, 32)), replace=False))
        cnt = np.diff(np.concatenate([[0], cut, [n_pt]])).tolist()
        if rng.random() < 0.15:
            cnt[-1] = max(1, cnt[-1] // 2)                                  # blocks that cover fewer pairs than the list
        blocks = [(int(c), float(rng.choice([0.25, 1.0, 3.0]))) for c in cnt]
    okw = {{"currentEstimateForRobust": "current_estimate"}.get(k, k): v for k, v in kw.items()}
    To, rc, fl = oracle.optimal_tf_horn_wp(pt, pl if n_pl else None, point_weights=blocks, **okw)
    w = WeightParameters()
    for k, v in kw.items():
        setattr(w.pair_weights, k[2:], v) if k.startswith("w_") else setattr(w, k, v)
    p = amd.Pairings.from_host(amd.default_context(), _to_hip_pt2pt(amd, pt), point_weights=blocks,
                               pl2pl=_to_hip_pl2pl(pl) if n_pl else None)
    out = amd.OptimalTF_Result()
    info = (seed, n_pt, n_pl, n_out, kw.keys(), None if blocks is None else len(blocks), rc)
    if rc == -1:                                                            # where the reference throws
        with pytest.raises((amd.Mp2pHipError, RuntimeError, ValueError)):
            optimal_tf_horn(p, w, out)
        return
    ok = optimal_tf_horn(p, w, out)
    assert bool(ok) == (rc == 1), info
    if rc == 1:
        dt, dr = oracle.pose_err_split(out.optimalPose, To)
        assert dt < 1e-5 and dr < 1e-5, (dt, dr, info)
        assert out.outliers == np.flatnonzero(fl).tolist(), info


# ---- FilterDecimateVoxels (FilterDecimateVoxels.cpp, PointCloudToVoxelGrid.cpp:57-92): random geometries (far offsets, lattices whose
#      points sit on voxel faces, duplicates, both signs), resolutions, the three deterministic methods, flatten_to; points and source
#      indices bit for bit (MP2P_FUZZ_DECIM_SEEDS=a:b runs another range of seeds)
_DECIM_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_DECIM_SEEDS", "0:14").split(":")])


@pytest.mark.parametrize("seed", _DECIM_SEEDS)
def test_fuzz_filter_decimate(oracle, seed):
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import core
    rng = np.random.default_rng(15000 + seed)
    kind = KINDS[seed % len(KINDS)]
    pts = _cloud(rng, kind, int(rng.integers(1, 40000))).astype(np.float32)
    if rng.random() < 0.5:
        pts -= pts.mean(0).astype(np.float32)                               # both signs: the cells that touch 0
    if rng.random() < 0.3 and len(pts) > 20:
        pts[len(pts) // 2:len(pts) // 2 + len(pts) // 10] = pts[:len(pts) // 10]   # duplicates
    scale = float(np.ptp(pts, axis=0).max()) or 1.0
    res = float(rng.choice([0.003, 0.02, 0.1, 0.5])) * scale
    method = int(rng.integers(0, 3))
    flat = float(rng.normal(0, scale)) if rng.random() < 0.25 else None
    want, wsrc = oracle.filter_decimate_voxels(pts[:, 0], pts[:, 1], pts[:, 2], res, method, flatten_to=flat)
    xyz, src = core.filter_decimate_voxels(amd.default_context(), pts[:, 0], pts[:, 1], pts[:, 2], res, method, flatten_to=flat)
    info = (seed, kind, len(pts), res, method, flat)
    assert xyz.shape == want.shape, info
    assert np.array_equal(xyz.view(np.uint32), want.view(np.uint32)), info
    assert np.array_equal(src, wsrc), info


# ---- the HOST path of the boundary (adapter/mp2p_hip_host.hpp through libmp2p_hip_hostpath.so): packed MatchState words in, pair records
#      and marks out, over layer sizes that are no multiple of 64, random pre-marked bits (none / sparse / dense / all), the allow flags,
#      and TWO matcher calls in one ICP iteration (the second sees the marks of the first); lists and bit-fields against the oracle
#      (MP2P_FUZZ_HOST_SEEDS=a:b runs another range of seeds)
_HOST_SEEDS = range(*[int(v) for v in os.environ.get("MP2P_FUZZ_HOST_SEEDS", "0:12").split(":")])


@pytest.mark.parametrize("seed", _HOST_SEEDS)
def test_fuzz_host_path_match_state(oracle, seed):
    from mp2p_icp_amd import hostpath
    from test_gpu_boundary_hostpath import _pt2pt_prm, _same_pt2pt, _xyz
    import mp2p_icp_amd as amd
    rng = np.random.default_rng(17000 + seed)
    kind = KINDS[seed % (len(KINDS) - 1)]
    g = _cloud(rng, kind, int(rng.integers(70, 30000))).astype(np.float32)
    scale = float(np.ptp(g, axis=0).max()) or 1.0
    n_l = int(rng.integers(1, 5000))
    l = (g[rng.integers(0, len(g), n_l)].astype(np.float64) + rng.normal(0, 0.01 * scale, (n_l, 3))).astype(np.float32)
    dens = [float(rng.choice([0.0, 0.0, 0.02, 0.5, 1.0])) for _ in range(2)]
    lt0, gt0 = rng.random(n_l) < dens[0], rng.random(len(g)) < dens[1]
    tree = oracle.KDTree(*_xyz(g))
    T = amd.se3.exp(np.concatenate([rng.normal(0, 0.005 * scale, 3), np.zeros(3)]))
    s = hostpath.Session(g, l)
    try:
        s.begin_iteration()
        if lt0.any():
            s.set_bits(1, lt0)
        if gt0.any():
            s.set_bits(0, gt0)
        lt, gt = lt0.astype(np.uint8), gt0.astype(np.uint8)
        want_all = []
        for call in range(2):
            thr = float(rng.choice([0.02, 0.05, 0.2])) * scale
            al, ag = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
            s.match_pt2pt(T, _pt2pt_prm(thr, allowMatchAlreadyMatchedPoints=int(al), allowMatchAlreadyMatchedGlobalPoints=int(ag)))
            want, _ = oracle.match_pt2pt(*_xyz(g), *_xyz(l), T, thr, 0.0, tree=tree, local_taken=lt, global_taken=gt,
                                         allowMatchAlreadyMatchedPoints=al, allowMatchAlreadyMatchedGlobalPoints=ag)
            want_all.append(want)
            info = (seed, call, kind, len(g), n_l, dens, thr, al, ag)
            got = s.pairs_pt2pt()                                           # the session's list: both calls' pairs, in call order
            _same_pt2pt(got, np.concatenate(want_all))
            assert np.array_equal(s.bits(1), lt.astype(bool)) and np.array_equal(s.bits(0), gt.astype(bool)), info
    finally:
        s.close()
