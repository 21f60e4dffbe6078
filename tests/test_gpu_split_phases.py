"""GPU checks of the split (multi-GPU) entry points on one GPU: phase1/phase2 == match, the
begin/accumulate/step/end loop == gn_solve, the exchange buffers (bounding box + record count,
claim records, normal-equation sums) are reachable as torch tensors for RCCL collectives, and
two shards exchanged by hand on one GPU reproduce the unsharded oracle."""
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
    exch, recs = be.exchange_pack()
    torch.cuda.synchronize()
    e = exch.cpu().numpy()
    tx, ty, tz, bmin, bmax = oracle.transform_local_to_global(l[:, 0], l[:, 1], l[:, 2], d["T_init"])
    assert np.array_equal((-e[:3]).astype(np.float32), bmin) and np.array_equal(e[3:6].astype(np.float32), bmax)
    assert np.array_equal(-e[:3], bmin.astype(np.float64))          # the fp32 box, exactly
    r = recs.cpu().numpy()
    k = int(e[6])
    assert r.shape[0] == l.shape[0] and (r[k:] == -1).all() and (r[:k] >= 0).all()
    assert sorted((r[:k] & 0xFFFFFFFF).tolist()) == sorted(a["localIdx"].tolist())  # one GPU: records = pairs
    be.exchange_unpack(None)
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


def _manual_exchange(torch, backs):
    """what ShardedRegistration.match does between the phases, for in-process 'ranks'"""
    packs = [b.exchange_pack() for b in backs]
    red = torch.stack([p[0] for p in packs]).max(0).values           # all-reduce MAX
    n_max = int(red[6].item())
    cap = -(-max(n_max, 1) // 1024) * 1024
    sends = []
    for _, recs in packs:
        send = recs[:cap]
        if send.numel() < cap:
            pad = recs.new_full((cap,), -1)
            pad[:send.numel()] = send
            send = pad
        sends.append(send.clone())
    gathered = torch.cat(sends)                                        # all-gather
    for (exch, _), b in zip(packs, backs):
        exch.copy_(red)
        b.exchange_unpack(gathered if b.uses_claims else None)


@pytest.mark.parametrize("world,own_stream", [(2, False), (3, True)])
def test_two_shards_on_one_gpu_match_the_oracle(oracle, world, own_stream):
    """own_stream False: torch's default (null) stream, which Context passes on as
    hipStreamLegacy; True: a dedicated torch stream made current (what bench.py does)."""
    import torch
    if own_stream:
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            _two_shards(oracle, world)
        ts.synchronize()
    else:
        _two_shards(oracle, world)


def _two_shards(oracle, world):
    import torch
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    from mp2p_icp_amd.distributed import HipBackend, shard_range

    d = synthetic.make_pair(20_001, 80_000, 5)
    g, l = d["glob"], d["local"]
    thr = 1.0
    gnp = _lib.GNParams()
    gnp.maxInnerLoopIterations = 3
    gnp.minDelta, gnp.maxCost = 1e-7, 0.0
    gnp.kernel, gnp.kernelParam = _lib.KERNEL_GEMANMCCLURE, 0.2
    gnp.w_pt2pt = gnp.w_pt2pl = 1.0
    backs, keep = [], []
    for r in range(world):
        ctx = amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        b0, e0 = shard_range(l.shape[0], r, world)
        gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
        cloud = core.LocalCloud(ctx, l[b0:e0, 0], l[b0:e0, 1], l[b0:e0, 2])
        prm = _lib.Pt2PtParams(thr, 0.0, 1, 0, 0, 0.20, b0, 0.0, 0, 0.0, 0, 0.0, 0)
        pairs = core.DevicePairs(ctx, e0 - b0, 0)
        backs.append(HipBackend(ctx, gmap, cloud, prm, gnp, pairs))
        keep.append((ctx, gmap, cloud, pairs))
    tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
    gno = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.2)
    pose = d["T_init"].copy()
    pose_o = d["T_init"].copy()
    for it in range(3):
        for b in backs:
            b.phase1(pose)
        _manual_exchange(torch, backs)
        for b in backs:
            b.phase2()
        got = np.concatenate([b.pairs.download_pt2pt() for b in backs])
        want, _ = oracle.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], pose_o,
                                     thr, 0.0, tree=tree, threads=8)
        assert len(got) == len(want), (it, len(got), len(want))
        assert np.array_equal(got["localIdx"], want["localIdx"])
        assert np.array_equal(got["globalIdx"], want["globalIdx"])
        assert len(set(got["globalIdx"].tolist())) == len(got)        # unique across shards
        # Gauss-Newton with the sums all-reduced by hand
        for b in backs:
            b.gn_begin(pose)
        for _ in range(3):
            for b in backs:
                b.gn_accumulate()
            tot = torch.stack([b.sums for b in backs]).sum(0)          # all-reduce SUM
            for b in backs:
                b.sums.copy_(tot)
                b.gn_step()
        outs = [b.gn_end() for b in backs]
        for p, _ in outs[1:]:
            assert np.array_equal(p, outs[0][0])                       # every rank: the same pose
        pose = outs[0][0]
        pose_o, *_ = oracle.optimal_tf_gauss_newton(want, None, None, pose_o, gno)
        dt, dr = oracle.pose_err_split(pose, pose_o)
        assert dt < 1e-5 and dr < 1e-5, (it, dt, dr)
