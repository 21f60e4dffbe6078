"""The per-call temporaries of the solvers, matchers and filters live in scratch owned by the context: after the first
call of a given size, a steady-state call makes no device allocation (mp2p_hip_debug_alloc_count stays put).  Covers the
call sites that used to hipMalloc/hipFree per call: pt2ln_pl_to_pt2pt + Solver_Horn (horn.hip), Matcher_Points_InlierRatio,
Matcher_Adaptive, FilterDecimateVoxels, covariance; plus the decimation time of 1 M points."""
import os
import time

import numpy as np
import pytest

from test_gpu_gn import _to_hip_pt2ln, _to_hip_pt2pl, _to_hip_pt2pt
from test_gpu_horn import _plane_line_pairings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _allocs(amd):
    from mp2p_icp_amd import _lib
    return int(_lib.load().mp2p_hip_debug_alloc_count())


def _steady(amd, fn, warm=2, reps=3):
    for _ in range(warm):
        fn()
    a0 = _allocs(amd)
    for _ in range(reps):
        fn()
    return _allocs(amd) - a0


def test_solver_horn_with_planes_and_lines_allocates_nothing(amd, oracle):
    gt, pl, ln = _plane_line_pairings(oracle, 71, 5000, 800)
    ctx = amd.default_context()
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, np.zeros(0, oracle.PAIR_PT2PT)), _to_hip_pt2pl(amd, pl),
                               pt2ln=_to_hip_pt2ln(ln))
    sc = amd.SolverContext()
    s = amd.Solver_Horn()
    s.initialize({})
    out = amd.OptimalTF_Result()
    state = {"pose": oracle.pose_from_xyzypr(0.1, 0.1, -0.1, 0.01, 0.02, -0.01)}

    def it():
        sc.guessRelativePose = state["pose"]
        assert s.optimal_pose(p, out, sc)
        state["pose"] = out.optimalPose

    assert _steady(amd, it) == 0
    assert oracle.pose_err(state["pose"], gt) < 0.05


def test_matchers_allocate_nothing_in_steady_state(amd):
    from mp2p_icp_amd import synthetic
    d = synthetic.random_cloud_pair(20000, 200000, 5)
    pcG = amd.metric_map_t({"raw": amd.PointLayer(d["glob"])})
    pcL = amd.metric_map_t({"raw": amd.PointLayer(d["local"])})
    for m, prm in ((amd.Matcher_Points_InlierRatio(), {"inliersRatio": 0.6}),
                   (amd.Matcher_Adaptive(), dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.2,
                                                 absoluteMaxSearchDistance=2.0, enableDetectPlanes=True,
                                                 planeSearchPoints=6, planeMinimumFoundPoints=4))):
        m.initialize(prm)
        pairs = amd.Pairings()
        poses = [d["T_init"], d["T_gt"]]
        k = [0]

        def it():
            ms = amd.MatchState(pcG, pcL)
            assert m.match(pcG, pcL, poses[k[0] % 2], amd.MatchContext(), ms, pairs)
            k[0] += 1

        # MatchState carries two device bit-fields of its own (2 allocations per object): count those out
        for _ in range(2):
            it()
        ms_only0 = _allocs(amd)
        amd.MatchState(pcG, pcL).for_layers("raw", "raw")
        per_ms = _allocs(amd) - ms_only0
        a0 = _allocs(amd)
        for _ in range(3):
            it()
        assert _allocs(amd) - a0 == 3 * per_ms, (type(m).__name__, _allocs(amd) - a0, per_ms)
        assert len(pairs.paired_pt2pt) > 1000


def test_covariance_and_decimation_allocate_nothing(amd, oracle):
    from mp2p_icp_amd import core
    ctx = amd.default_context()
    gt, pl, ln = _plane_line_pairings(oracle, 72, 3000, 0)
    pt = np.zeros(2000, oracle.PAIR_PT2PT)
    rng = np.random.default_rng(3)
    lp = rng.uniform(-5, 5, (2000, 3)).astype(np.float32)
    R, t = gt[:9].reshape(3, 3), gt[9:]
    gp = (lp @ R.T + t).astype(np.float32)
    pt["lx"], pt["ly"], pt["lz"] = lp.T
    pt["gx"], pt["gy"], pt["gz"] = gp.T
    p = amd.Pairings.from_host(ctx, _to_hip_pt2pt(amd, pt), _to_hip_pt2pl(amd, pl))
    assert _steady(amd, lambda: amd.covariance(p, gt)) == 0

    pts = rng.uniform(-40, 40, (1_000_000, 3)).astype(np.float32)
    for method in (0, 1, 2):
        assert _steady(amd, lambda: core.filter_decimate_voxels(ctx, pts[:, 0], pts[:, 1], pts[:, 2], 0.5, method), 1, 2) == 0


def test_decimation_of_one_million_points_device_time(amd):
    """device-resident entry (the one the sensor pipeline calls): 1 M points, voxel 0.5 m -- the bar is 1 ms of device
    time per call at steady state; asserted with slack for a busy box (measured value in DESIGN.md)"""
    import torch
    from mp2p_icp_amd import _lib, core
    import ctypes as C
    ctx = amd.default_context()
    L = _lib.load()
    n = 1_000_000
    g = torch.Generator(device="cpu").manual_seed(5)
    xyz = (torch.rand((3, n), generator=g) * 80.0 - 40.0).to("cuda")
    out = torch.empty((3, n), dtype=torch.float32, device="cuda")
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    prm = _lib.DecimateParams(0.5, 0, 0, 0.0)
    m = C.c_size_t()
    fp = lambda tns, i: C.cast(tns[i].data_ptr(), C.POINTER(C.c_float))  # noqa: E731

    def call():
        core.check(L.mp2p_hip_filter_decimate_voxels_device(ctx.handle, fp(xyz, 0), fp(xyz, 1), fp(xyz, 2), n, C.byref(prm),
                                                            fp(out, 0), fp(out, 1), fp(out, 2),
                                                            C.cast(src.data_ptr(), C.POINTER(C.c_uint32)), C.byref(m)), ctx.handle)

    for _ in range(3):
        call()
    a0 = _allocs(amd)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps * 1e3
    assert _allocs(amd) == a0
    assert 100_000 < m.value < n
    print(f"decimate 1M points (FirstPoint, 0.5 m): {dt:.3f} ms per call (wall, incl. the count read-back)")
    # a timing bound in a parity suite must not decide the run on a busy box (3.4 ms was seen once in round 4 against the
    # usual 0.3-0.5): the allocation count above is the regression check; the wall-time bound lives under `-m perf`
    # (tests/test_perf_bounds.py sets MP2P_PERF_ASSERTS)
    if os.environ.get("MP2P_PERF_ASSERTS") == "1":
        assert dt < 3.0, dt
