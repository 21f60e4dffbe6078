"""GPU checks of the split (multi-GPU) entry points on one GPU: phase1/phase2 == match, the
begin/accumulate/step/end loop == gn_solve, and the exchange buffers (claim words, local
bounding box, normal-equation sums) are reachable as torch tensors for RCCL collectives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_split_equals_fused_and_buffers_are_torch_visible(oracle):
    import torch
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    from mp2p_icp_amd.distributed import HipBackend, ShardedRegistration

    d = synthetic.random_cloud_pair(30_000, 120_000, 31, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    ctx = amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    prm = _lib.Pt2PtParams(0.7, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    gnp = _lib.GNParams()
    gnp.maxInnerLoopIterations = 4
    gnp.minDelta, gnp.maxCost = 1e-7, 0.0
    gnp.kernel, gnp.kernelParam = _lib.KERNEL_CAUCHY, 0.3
    gnp.w_pt2pt = gnp.w_pt2pl = 1.0

    # fused reference
    p1 = core.DevicePairs(ctx, l.shape[0], 0)
    core.match_pt2pt(ctx, gmap, cloud, d["T_init"], prm, None, p1)
    a = p1.download_pt2pt()
    r1 = core.gn_solve(ctx, p1, d["T_init"], gnp)

    # split path, with the collectives replaced by reads of the exchange buffers
    p2 = core.DevicePairs(ctx, l.shape[0], 0)
    be = HipBackend(ctx, gmap, cloud, prm, gnp, p2)
    be.phase1(d["T_init"])
    torch.cuda.synchronize()
    bbox = be.bbox.cpu().numpy()
    tx, ty, tz, bmin, bmax = oracle.transform_local_to_global(l[:, 0], l[:, 1], l[:, 2], d["T_init"])
    assert np.array_equal(bbox[:3], bmin) and np.array_equal(bbox[3:], bmax)   # exact fp32 box
    claims = be.claims.cpu().numpy()
    assert claims.shape[0] == g.shape[0] and (claims < 0).all()  # int64 view: MIN-reducible
    be.phase2()
    b = p2.download_pt2pt()
    assert np.array_equal(a, b)

    be.gn_begin(d["T_init"])
    for _ in range(be.max_inner):
        be.gn_accumulate()
        torch.cuda.synchronize()
        s = be.sums.cpu().numpy()
        assert s.shape == (48,) and np.isfinite(s).all()
        be.gn_step()
    pose, iters = be.gn_end()
    assert iters == r1.iterations
    assert np.allclose(pose, np.array(r1.pose), rtol=0, atol=1e-12)

    # the same through ShardedRegistration at world size 1
    reg = ShardedRegistration(be, None)
    pose2, _ = reg.step(d["T_init"])
    assert np.allclose(pose2, np.array(r1.pose), rtol=0, atol=1e-12)

    # against the oracle
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], d["T_init"],
                                 0.7, 0.0, tree=tree)
    assert np.array_equal(a["localIdx"], want["localIdx"]) and np.array_equal(a["globalIdx"], want["globalIdx"])
    To, *_ = oracle.optimal_tf_gauss_newton(want, None, None, d["T_init"],
                                            oracle.make_gn_params(4, kernel=oracle.KERNEL_CAUCHY, kernelParam=0.3))
    dt, dr = oracle.pose_err_split(pose, To)
    assert dt < 1e-5 and dr < 1e-5
