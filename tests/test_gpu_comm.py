"""RCCL inside the C boundary (include/mp2p_hip.h, "multi-GPU"): mp2p_hip_comm_init + mp2p_hip_step_sharded.

One GPU is what this box has, so:
  * the real RCCL route runs with ONE rank (ncclCommInitRank(nranks = 1), the collectives are skipped):
    the sharded step must equal the plain match + solve;
  * the exchange logic of mp2p_hip_step_sharded with TWO and THREE ranks runs through
    mp2p_hip_comm_init_hooks: one context (own stream) per rank on the same GPU, one thread per rank,
    the two collectives done by the test -- RCCL itself refuses two ranks on one device.  Concatenated
    in rank order the shards' pair lists must be the unsharded list (= the oracle's), all ranks must
    agree on the pose bit for bit, and the pose must equal the unsharded solve's to 1e-9.
The same protocol over real inter-process collectives is covered by tests/test_distributed_gloo.py."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _xyz(a):
    return a[:, 0], a[:, 1], a[:, 2]


def _prms(offset=0, threshold=1.5, allow_global=0):
    from mp2p_icp_amd import _lib
    p = _lib.Pt2PtParams()
    p.threshold, p.thresholdAngularDeg, p.pairingsPerPoint = threshold, 0.0, 1
    p.allowMatchAlreadyMatchedGlobalPoints = allow_global
    p.bounding_box_intersection_check_epsilon = 0.20
    p.local_index_offset = offset
    g = _lib.GNParams()
    g.maxInnerLoopIterations, g.minDelta, g.maxCost = 3, 1e-7, 0.0
    g.kernel, g.kernelParam, g.w_pt2pt, g.w_pt2pl = _lib.KERNEL_GEMANMCCLURE, 0.15, 1.0, 1.0
    return p, g


def _step_sharded(ctx, gmap, cloud, pose, prm, gnp, pairs):
    from mp2p_icp_amd import _lib
    T = np.ascontiguousarray(pose, dtype=np.float64)
    res, redone = _lib.GNResult(), C.c_int32(0)
    _lib.check(ctx._L.mp2p_hip_step_sharded(ctx.handle, gmap.handle, cloud.handle, T.ctypes.data_as(C.POINTER(C.c_double)),
                                            C.byref(prm), C.byref(gnp), pairs.handle, C.byref(res), C.byref(redone)),
               ctx.handle)
    return np.array(res.pose), int(redone.value)


def test_one_rank_over_real_rccl(oracle):
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    d = synthetic.make_pair(40_000, 300_000, 21)
    g, l = d["glob"], d["local"]
    ctx = amd.Context(0)
    gmap, cloud = core.GlobalMap(ctx, *_xyz(g)), core.LocalCloud(ctx, *_xyz(l))
    pairs, ref = core.DevicePairs(ctx, l.shape[0], 0), core.DevicePairs(ctx, l.shape[0], 0)
    prm, gnp = _prms()
    idb = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(ctx._L.mp2p_hip_comm_get_unique_id(idb))
    assert any(idb)
    assert ctx._L.mp2p_hip_comm_size(ctx.handle) == 0
    _lib.check(ctx._L.mp2p_hip_comm_init(ctx.handle, idb, 0, 1), ctx.handle)
    assert ctx._L.mp2p_hip_comm_size(ctx.handle) == 1 and ctx._L.mp2p_hip_comm_rank(ctx.handle) == 0
    with pytest.raises(_lib.Mp2pHipError):  # one communicator per context
        _lib.check(ctx._L.mp2p_hip_comm_init(ctx.handle, idb, 0, 1), ctx.handle)
    pose = d["T_init"].copy()
    for it in range(3):
        got, redone = _step_sharded(ctx, gmap, cloud, pose, prm, gnp, pairs)
        ref.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, ref)
        want = np.array(core.gn_solve(ctx, ref, pose, gnp).pose)
        assert redone == 0 and np.array_equal(pairs.download_pt2pt(), ref.download_pt2pt())
        assert np.allclose(got, want, rtol=0, atol=1e-12)
        pose = got
    _lib.check(ctx._L.mp2p_hip_comm_destroy(ctx.handle), ctx.handle)
    assert ctx._L.mp2p_hip_comm_size(ctx.handle) == 0


class _Exchange:
    """the collectives of `world` in-process ranks (one thread each) on device buffers"""

    def __init__(self, world, torch, dev):
        self.world, self.torch, self.dev = world, torch, dev
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = {"allreduce": 0, "allgather": 0}

    def hooks(self, rank):
        from mp2p_icp_amd import _lib
        from mp2p_icp_amd.distributed import _DevArray
        torch, ex = self.torch, self

        def allreduce(user, buf, n, op, stream):
            try:
                t = torch.as_tensor(_DevArray(buf, n, "<f8"), device=ex.dev)
                torch.cuda.synchronize()
                ex.slots[rank] = t
                ex.bar.wait()
                vals = torch.stack(list(ex.slots))
                r = vals.max(0).values if op else vals.sum(0)  # the same order on every rank
                torch.cuda.synchronize()
                ex.bar.wait()
                t.copy_(r)
                torch.cuda.synchronize()
                ex.bar.wait()
                if rank == 0:
                    ex.calls["allreduce"] += 1
                return 0
            except Exception:  # pragma: no cover
                ex.bar.abort()
                return 1

        def allgather(user, send, recv, n, stream):
            try:
                s = torch.as_tensor(_DevArray(send, n, "<i8"), device=ex.dev)
                r = torch.as_tensor(_DevArray(recv, n * ex.world, "<i8"), device=ex.dev)
                torch.cuda.synchronize()
                ex.slots[rank] = s
                ex.bar.wait()
                r.copy_(torch.cat(list(ex.slots)))
                torch.cuda.synchronize()
                ex.bar.wait()
                if rank == 0:
                    ex.calls["allgather"] += 1
                return 0
            except Exception:  # pragma: no cover
                ex.bar.abort()
                return 1

        return _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather)


@pytest.mark.parametrize("world,allow_global", [(2, 0), (3, 0), (2, 1), (4, 0), (8, 0)])
def test_sharded_step_through_hooks(oracle, world, allow_global):
    import torch
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    from mp2p_icp_amd.distributed import shard_range
    # most local points win a global point of their own: tens of thousands of claim records per rank,
    # enough to outgrow a length predicted from an iteration that had none
    d = synthetic.random_cloud_pair(90_000, 120_000, 8, outlier_frac=0.05)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    dev = torch.device("cuda", 0)
    ex = _Exchange(world, torch, dev)
    # a pose sequence whose record lists GROW from one call to the next (far guess first): the
    # predicted list length is exceeded at least once and the iteration is redone
    far = amd.se3.compose(d["T_gt"], amd.se3.from_xyzypr(0.0, 0.0, 60.0, 0.0, 0.0, 0.0))  # above everything: no records
    seq = [far, d["T_init"], None, None]  # None: continue from the previous result
    out = [None] * world
    keep = []

    def run(rank):
        try:
            ctx = amd.Context(0)  # own stream
            b, e = shard_range(l.shape[0], rank, world)
            gmap, cloud = core.GlobalMap(ctx, *_xyz(g)), core.LocalCloud(ctx, *_xyz(l[b:e]))
            pairs = core.DevicePairs(ctx, e - b, 0)
            prm, gnp = _prms(offset=b, allow_global=allow_global)
            ar, ag = ex.hooks(rank)
            keep.append((ar, ag))
            _lib.check(ctx._L.mp2p_hip_comm_init_hooks(ctx.handle, rank, world, ar, ag, None), ctx.handle)
            res, pose = [], None
            for p0 in seq:
                pose = p0 if p0 is not None else pose
                start = pose.copy()
                pose, redone = _step_sharded(ctx, gmap, cloud, pose, prm, gnp, pairs)
                res.append((start, pairs.download_pt2pt(), pose.copy(), redone))
            out[rank] = res
        except Exception as exn:  # pragma: no cover
            ex.bar.abort()
            out[rank] = exn

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        assert out[r] is not None
    oprm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    n_redone = 0
    for k in range(len(seq)):
        start = out[0][k][0]
        got = np.concatenate([out[r][k][1] for r in range(world)])
        want, _ = oracle.match_pt2pt(*_xyz(g), *_xyz(l), start, 1.5, 0.0, tree=tree,
                                     allowMatchAlreadyMatchedGlobalPoints=bool(allow_global))
        assert len(got) == len(want), (k, len(got), len(want))
        assert np.array_equal(got["localIdx"], want["localIdx"]) and np.array_equal(got["globalIdx"], want["globalIdx"])
        for r in range(1, world):
            assert np.array_equal(out[r][k][2], out[0][k][2])  # every rank solved the same system
            assert np.array_equal(out[r][k][0], out[0][k][0])
        if len(want):
            To, *_ = oracle.optimal_tf_gauss_newton(want, None, None, start, oprm)
            dt, dr = oracle.pose_err_split(out[0][k][2], To)
            assert dt < 1e-5 and dr < 1e-5, (k, dt, dr)
        else:  # nothing paired: the normal equations are empty and the pose stays
            assert np.array_equal(out[0][k][2], start)
        n_redone += out[0][k][3]
    if allow_global:
        assert ex.calls["allgather"] == 0 and n_redone == 0  # no unique-global filter: nothing to gather
    else:
        assert ex.calls["allgather"] >= len(seq) and n_redone >= 1
    assert ex.calls["allreduce"] >= len(seq) * 4  # box + three Gauss-Newton sums per step


def _step_sharded_pt2pl(ctx, gmap, cloud, pose, prm, offset, gnp, pairs):
    from mp2p_icp_amd import _lib
    T = np.ascontiguousarray(pose, dtype=np.float64)
    res = _lib.GNResult()
    _lib.check(ctx._L.mp2p_hip_step_sharded_pt2pl(ctx.handle, gmap.handle, cloud.handle, T.ctypes.data_as(C.POINTER(C.c_double)),
                                                  C.byref(prm), offset, C.byref(gnp), pairs.handle, C.byref(res)), ctx.handle)
    return np.array(res.pose)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_sharded_pt2pl_step_through_hooks(oracle, world):
    """mp2p_hip_step_sharded_pt2pl (Matcher_Point2Plane + Gauss-Newton, BASELINE config C3 sharded): a chain of three
    iterations on `world` in-process ranks (world = 1: no communicator) against the UNSHARDED oracle chain -- the
    shards' pair lists concatenated in rank order are the oracle's (whole-layer local indices, planes 1e-9), every rank
    ends every iteration with the same pose bit for bit, the pose within 1e-5 of the oracle's"""
    import torch
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    from mp2p_icp_amd.distributed import shard_range
    d = synthetic.make_pair(30_000, 200_000, 33)
    g, l = d["glob"], d["local"]
    tree = oracle.KDTree(*_xyz(g))
    dev = torch.device("cuda", 0)
    ex = _Exchange(world, torch, dev)
    n_it = 3
    out = [None] * world
    keep = []

    def prms():
        pl = _lib.Pt2PlParams()
        pl.distanceThreshold, pl.searchRadius, pl.knn, pl.minimumPlanePoints, pl.planeEigenThreshold = 0.3, 0.5, 5, 5, 0.05
        pl.bounding_box_intersection_check_epsilon = 0.20
        gn = _lib.GNParams()
        gn.maxInnerLoopIterations, gn.minDelta, gn.maxCost = 3, 1e-7, 0.0
        gn.kernel, gn.kernelParam, gn.w_pt2pt, gn.w_pt2pl = _lib.KERNEL_GEMANMCCLURE, 0.15, 1.0, 1.0
        return pl, gn

    def run(rank):
        try:
            ctx = amd.Context(0)
            b, e = shard_range(l.shape[0], rank, world)
            gmap, cloud = core.GlobalMap(ctx, *_xyz(g)), core.LocalCloud(ctx, *_xyz(l[b:e]))
            pairs = core.DevicePairs(ctx, 0, e - b)
            pl, gn = prms()
            if world > 1:
                ar, ag = ex.hooks(rank)
                keep.append((ar, ag))
                _lib.check(ctx._L.mp2p_hip_comm_init_hooks(ctx.handle, rank, world, ar, ag, None), ctx.handle)
            res, pose = [], d["T_init"].copy()
            for _ in range(n_it):
                start = pose.copy()
                pose = _step_sharded_pt2pl(ctx, gmap, cloud, pose, pl, b, gn, pairs)
                rec, idx = pairs.download_pt2pl()
                res.append((start, rec, idx, pose.copy(), pairs.counts()[2]))
            out[rank] = res
        except Exception as exn:  # pragma: no cover
            ex.bar.abort()
            out[rank] = exn

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        assert out[r] is not None
    oprm = oracle.make_gn_params(3, kernel=oracle.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    for k in range(n_it):
        start = out[0][k][0]
        rec = np.concatenate([out[r][k][1] for r in range(world)])
        idx = np.concatenate([out[r][k][2] for r in range(world)])
        want, widx, pot = oracle.match_pt2pl(*_xyz(g), *_xyz(l), start, 0.3, 0.5, 5, 5, 0.05, tree=tree)
        assert len(rec) == len(want) and np.array_equal(idx, widx), (k, len(rec), len(want))
        assert np.allclose(rec["plane"], want["plane"], rtol=0, atol=1e-9)
        assert sum(out[r][k][4] for r in range(world)) == pot  # potential_pairings: every shard adds its own points
        for r in range(1, world):
            assert np.array_equal(out[r][k][3], out[0][k][3]) and np.array_equal(out[r][k][0], start)
        To, *_ = oracle.optimal_tf_gauss_newton(None, want, None, start, oprm)
        dt, dr = oracle.pose_err_split(out[0][k][3], To)
        assert dt < 1e-5 and dr < 1e-5, (k, dt, dr)
    if world > 1:
        assert ex.calls["allgather"] == 0                      # no unique-global filter: nothing to gather
        assert ex.calls["allreduce"] == n_it * (1 + 3)         # the box + three Gauss-Newton sums per step
