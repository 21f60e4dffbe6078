"""Multi-GPU form of the hot path: ONE local layer sharded over the ranks of a
torch.distributed process group (one process per GPU, backend "nccl" = RCCL over xGMI), the
global layer and its index replicated on every GPU (10 M points = 160 MB of 288 GB).

Exchange steps per outer ICP iteration:
  1. bounding box of the transformed local layer : all-reduce MIN/MAX of 3+3 floats
  2. unique-global filter                         : all-reduce MIN of the claim words
     (int64, one per global point; "lowest whole-layer local index wins" across ranks),
     skipped when allowMatchAlreadyMatchedGlobalPoints is set
  3. per Gauss-Newton inner iteration             : all-reduce SUM of 48 doubles
     (17 pt2pt + 28 pt2pl normal-equation sums); every rank then solves the same 6x6 system
     and retracts redundantly -- no broadcast.
The reference has no distributed code at all; results are identical to the
single-GPU path up to fp64 re-association of the sums.

The class is written against a small backend protocol so that the *exchange logic* can be
exercised with gloo on CPU (tests/test_distributed_gloo.py plugs the CPU oracle in); the only
product backend is HipBackend below.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous slice of the local layer owned by `rank` (keeps output order by rank)"""
    b = (n * rank) // world
    e = (n * (rank + 1)) // world
    return b, e


class _DevArray:
    """exposes a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class HipBackend:
    """This rank's shard on its GPU, through libmp2p_hip."""

    def __init__(self, ctx, gmap, cloud, pt2pt_params, gn_params, pairs):
        import torch
        from . import core  # noqa: F401
        self.torch = torch
        self.ctx, self.gmap, self.cloud, self.pairs = ctx, gmap, cloud, pairs
        self.prm, self.gn_prm = pt2pt_params, gn_params
        dev = torch.device("cuda", ctx.device)
        self.claims = torch.as_tensor(_DevArray(gmap.claims_ptr(), gmap.n, "<i8"), device=dev)
        self.bbox = torch.as_tensor(_DevArray(ctx.local_bbox_ptr(), 6, "<f4"), device=dev)
        self._sums = None
        self.uses_claims = not bool(pt2pt_params.allowMatchAlreadyMatchedGlobalPoints)

    def phase1(self, pose):
        from . import core
        self.pairs.clear()
        core.match_pt2pt_phase1(self.ctx, self.gmap, self.cloud, pose, self.prm, None)

    def phase2(self):
        from . import core
        core.match_pt2pt_phase2(self.ctx, self.gmap, self.cloud, self.prm, None, self.pairs)

    def gn_begin(self, pose):
        import ctypes as C
        from ._lib import check
        T = np.ascontiguousarray(pose, dtype=np.float64)
        check(self.ctx._L.mp2p_hip_gn_begin(self.ctx.handle, self.pairs.handle,
                                            T.ctypes.data_as(C.POINTER(C.c_double)),
                                            C.byref(self.gn_prm)), self.ctx.handle)
        if self._sums is None:
            dev = self.torch.device("cuda", self.ctx.device)
            self._sums = self.torch.as_tensor(_DevArray(self.ctx.gn_sums_ptr(), 48, "<f8"), device=dev)

    def gn_accumulate(self):
        from ._lib import check
        check(self.ctx._L.mp2p_hip_gn_accumulate(self.ctx.handle), self.ctx.handle)

    @property
    def sums(self):
        return self._sums

    def gn_step(self):
        from ._lib import check
        check(self.ctx._L.mp2p_hip_gn_step(self.ctx.handle), self.ctx.handle)

    def gn_end(self):
        import ctypes as C
        from . import _lib
        res = _lib.GNResult()
        _lib.check(self.ctx._L.mp2p_hip_gn_end(self.ctx.handle, C.byref(res)), self.ctx.handle)
        return np.array(res.pose), int(res.iterations)

    def gn_solve_fused(self, pose):
        from . import core
        res = core.gn_solve(self.ctx, self.pairs, pose, self.gn_prm)
        return np.array(res.pose), int(res.iterations)

    def n_pairs(self):
        return self.pairs.counts()[0]

    @property
    def max_inner(self):
        return int(self.gn_prm.maxInnerLoopIterations)


class ShardedRegistration:
    """match (pt2pt) + Gauss-Newton over a local layer sharded across the process group."""

    def __init__(self, backend, dist=None, group=None):
        self.b = backend
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist is not None else 1

    def _allreduce(self, t, op):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t, op=op, group=self.group)

    def match(self, pose):
        b = self.b
        b.phase1(pose)
        if self.world > 1:
            R = self.dist.ReduceOp
            self._allreduce(b.bbox[:3], R.MIN)
            self._allreduce(b.bbox[3:], R.MAX)
            if b.uses_claims:
                self._allreduce(b.claims, R.MIN)
        b.phase2()

    def solve(self, pose):
        b = self.b
        if self.world == 1 and hasattr(b, "gn_solve_fused"):
            return b.gn_solve_fused(pose)  # no all-reduce seam: one launch per inner iteration less
        b.gn_begin(pose)
        for _ in range(b.max_inner):
            b.gn_accumulate()
            if self.world > 1:
                self._allreduce(b.sums, self.dist.ReduceOp.SUM)
            b.gn_step()
        return b.gn_end()

    def step(self, pose):
        """one outer ICP iteration; returns (new pose, inner iterations)"""
        self.match(pose)
        return self.solve(pose)
