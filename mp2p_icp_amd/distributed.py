"""Multi-GPU form of the hot path: ONE local layer sharded over the ranks of a
torch.distributed process group (one process per GPU, backend "nccl" = RCCL over xGMI), the
global layer and its index replicated on every GPU (10 M points = 160 MB of 288 GB).

Exchange steps per outer ICP iteration:
  1. all-reduce MAX of 8 doubles {-min xyz, max xyz of the transformed local layer, number of
     claim records, 0}: the whole layer's bounding box and the longest record list in one go
  2. unique-global filter: all-gather of the claim RECORDS (uint64 = sorted global position <<
     32 | whole-layer local index) of the local points that survived their own rank's filter;
     every rank applies the foreign records with atomicMin ("lowest whole-layer local index
     wins").  Records number at most the distinct global points a rank hits (about a tenth of
     its shard on the bench scene), so this moves ~1 MB per rank where an all-reduce MIN of one
     word per global point would move 80 MB.  Skipped when allowMatchAlreadyMatchedGlobalPoints.
  3. per Gauss-Newton inner iteration             : all-reduce SUM of 48 doubles
     (17 pt2pt + 28 pt2pl normal-equation sums); every rank then solves the same 6x6 system
     and retracts redundantly -- no broadcast.
The reference has no distributed code at all; results are identical to the
single-GPU path up to fp64 re-association of the sums.

The class is written against a small backend protocol so that the *exchange logic* can be
exercised with gloo on CPU (tests/test_distributed_gloo.py plugs the CPU oracle in); the only
product backend is HipBackend below.
"""
import os

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
        self.dev = torch.device("cuda", ctx.device)
        self._sums = None
        self._gather = None
        self._views = None
        # (Pt2PlParams has no such field: Matcher_Point2Plane pairs every local point on its own)
        self.uses_claims = not bool(getattr(pt2pt_params, "allowMatchAlreadyMatchedGlobalPoints", 1))

    def phase1(self, pose):
        from . import core
        self.pairs.clear()
        core.match_pt2pt_phase1(self.ctx, self.gmap, self.cloud, pose, self.prm, None)

    def exchange_pack(self):
        """-> (exch: f64[8] device tensor, records: i64[n_l] device tensor padded with -1)"""
        from . import core
        e, l, n = core.exchange_pack(self.ctx, self.gmap, self.cloud, self.prm)
        key = (e, l, n)
        if self._views is None or self._views[0] != key:  # the buffers only move when they grow
            torch = self.torch
            exch = torch.as_tensor(_DevArray(e, 8, "<f8"), device=self.dev)
            recs = torch.as_tensor(_DevArray(l, max(1, n), "<i8"), device=self.dev)
            self._views = (key, exch, recs[:n])
        return self._views[1], self._views[2]

    def gather_buffer(self, n):
        if self._gather is None or self._gather.numel() < n:
            self._gather = self.torch.empty(max(n, 1), dtype=self.torch.int64, device=self.dev)
        return self._gather[:n]

    def exchange_unpack(self, gathered):
        from . import core
        if gathered is None:
            core.exchange_unpack(self.ctx, self.gmap, None, 0)
        else:
            core.exchange_unpack(self.ctx, self.gmap, gathered.data_ptr(), gathered.numel())

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

    # ---- RCCL inside the C boundary (include/mp2p_hip.h, "multi-GPU") -------------------------------
    def init_native_comm(self, dist, group=None):
        """One communicator per context, created from an ncclUniqueId that rank 0 draws and the process
        group carries to the others (128 bytes, the only use of torch.distributed on this route).  From
        then on step_native() runs the whole sharded iteration -- both all-reduces and the all-gather --
        inside libmp2p_hip on the context's stream."""
        import ctypes as C
        from . import _lib
        torch = self.torch
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # The decision is COLLECTIVE: a rank that falls back on its own would call torch.distributed
        # collectives while the others sit in ncclCommInitRank / mp2p_hip_step_sharded -- a deadlock.  Rank 0
        # broadcasts a status byte with the id; after the init every rank contributes a success flag to an
        # all-reduce MIN; the native route is taken only if every rank made it, otherwise every rank destroys
        # what it created and all fall back together.
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        dev = self.dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        # pre-flight: mp2p_hip_comm_init can fail on ONE rank before it reaches ncclCommInitRank (librccl
        # not loadable there, a communicator already present) while its peers block inside it.  Every rank therefore
        # first proves, locally, that it would get that far -- mp2p_hip_comm_available dlopens librccl, resolves its symbols
        # and looks at the context, without side effects (drawing a throw-away unique id left a bootstrap
        # listener thread and socket behind on every rank) -- and nobody calls comm_init unless all ranks can.
        pre = 1 if self.ctx._L.mp2p_hip_comm_available(self.ctx.handle) == 0 else 0
        pf = torch.tensor([pre], dtype=torch.int32, device=dev)
        dist.all_reduce(pf, op=dist.ReduceOp.MIN, group=group)
        if int(pf.item()) != 1:
            raise RuntimeError("the native communicator cannot be created on every rank (librccl not loadable, or a "
                               "communicator already present on some rank)")
        ok0 = 1
        if rank == 0:
            try:
                _lib.check(self.ctx._L.mp2p_hip_comm_get_unique_id(buf))
            except Exception:
                ok0 = 0
        t = torch.tensor([ok0] + list(buf), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        ok = raw[0] == 1
        if ok:
            idb = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(raw[1:])
            try:
                _lib.check(self.ctx._L.mp2p_hip_comm_init(self.ctx.handle, idb, rank, world), self.ctx.handle)
            except Exception:
                ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            if ok:
                self.ctx._L.mp2p_hip_comm_destroy(self.ctx.handle)
            raise RuntimeError("the native communicator could not be created on every rank")
        self.native = True

    def init_hooks_comm(self, dist, group=None):
        """The library's sharded step (mp2p_hip_step_sharded[_pt2pl]) over a process group that is NOT RCCL -- gloo, i.e. the
        one-GPU test boxes where N ranks share a device and RCCL refuses two ranks per GPU: the step's collectives go through
        mp2p_hip_comm_init_hooks (include/mp2p_hip.h) to this process group, staged through the host.  Same code inside the
        library as with RCCL from the collectives' call sites outwards; a transport for tests, not for speed."""
        import ctypes as C
        from . import _lib
        torch = self.torch
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = self.dev

        def allreduce(user, buf, n, op, stream):
            try:
                t = torch.as_tensor(_DevArray(buf, n, "<f8"), device=dev)
                torch.cuda.synchronize(dev)
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM, group=group)
                t.copy_(h)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:  # pragma: no cover  (the library reports MP2P_HIP_ERR_COMM; say why)
                import traceback
                traceback.print_exc()
                return 1

        def allgather(user, send, recv, n, stream):
            try:
                s_ = torch.as_tensor(_DevArray(send, n, "<i8"), device=dev)
                r_ = torch.as_tensor(_DevArray(recv, n * world, "<i8"), device=dev)
                torch.cuda.synchronize(dev)
                parts = [torch.empty(n, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(parts, s_.cpu(), group=group)
                r_.copy_(torch.cat(parts))
                torch.cuda.synchronize(dev)
                return 0
            except Exception:  # pragma: no cover  (the library reports MP2P_HIP_ERR_COMM; say why)
                import traceback
                traceback.print_exc()
                return 1

        self._hooks = (_lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather))  # kept alive with the backend
        _lib.check(self.ctx._L.mp2p_hip_comm_init_hooks(self.ctx.handle, rank, world, self._hooks[0], self._hooks[1], None),
                   self.ctx.handle)
        self.native = True

    def step_native(self, pose):
        import ctypes as C
        from . import _lib
        T = np.ascontiguousarray(pose, dtype=np.float64)
        res, redone = _lib.GNResult(), C.c_int32(0)
        _lib.check(self.ctx._L.mp2p_hip_step_sharded(self.ctx.handle, self.gmap.handle, self.cloud.handle,
                                                     T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.prm),
                                                     C.byref(self.gn_prm), self.pairs.handle, C.byref(res),
                                                     C.byref(redone)), self.ctx.handle)
        self.redone_steps = getattr(self, "redone_steps", 0) + int(redone.value)
        return np.array(res.pose), int(res.iterations)

    @property
    def max_inner(self):
        return int(self.gn_prm.maxInnerLoopIterations)


class HipPlaneBackend(HipBackend):
    """This rank's shard for Matcher_Point2Plane + Gauss-Newton (BASELINE config C3 sharded).  The matcher pairs every
    local point on its own, so the shards exchange only the layer's bounding box and the normal-equation sums -- both
    inside libmp2p_hip (mp2p_hip_step_sharded_pt2pl): this backend offers the native route only."""

    def __init__(self, ctx, gmap, cloud, pt2pl_params, gn_params, pairs, local_index_offset=0):
        HipBackend.__init__(self, ctx, gmap, cloud, pt2pl_params, gn_params, pairs)
        self.uses_claims = False
        self.native_only = True
        self.offset = int(local_index_offset)

    def step_native(self, pose):
        import ctypes as C
        from . import _lib
        T = np.ascontiguousarray(pose, dtype=np.float64)
        res = _lib.GNResult()
        _lib.check(self.ctx._L.mp2p_hip_step_sharded_pt2pl(self.ctx.handle, self.gmap.handle, self.cloud.handle,
                                                           T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.prm),
                                                           self.offset, C.byref(self.gn_prm), self.pairs.handle,
                                                           C.byref(res)), self.ctx.handle)
        return np.array(res.pose), int(res.iterations)

    def phase1(self, pose):
        raise RuntimeError("HipPlaneBackend runs inside libmp2p_hip only (mp2p_hip_comm_init / one rank)")

    def n_pairs(self):
        return self.pairs.counts()[1]


class ShardedRegistration:
    """match (pt2pt) + Gauss-Newton over a local layer sharded across the process group.

    Two routes to the same exchange steps: the product route runs them INSIDE libmp2p_hip
    (mp2p_hip_step_sharded over RCCL; chosen when the backend offers it and the process group is nccl),
    this class's own match()/solve() run them through torch.distributed collectives -- the protocol the
    gloo tests exercise on CPU with the oracle as the per-rank compute, and the fallback when the native
    communicator cannot be created."""

    def __init__(self, backend, dist=None, group=None, native=None):
        self.b = backend
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist is not None else 1
        if native is None:
            native = os.environ.get("MP2P_HIP_NATIVE_COMM", "1") != "0"
        if (native and self.world > 1 and getattr(backend, "native_only", False) and hasattr(backend, "init_hooks_comm")
                and dist.get_backend(group) != "nccl"):
            backend.init_hooks_comm(dist, group)  # (test boxes: the library's sharded step over gloo)
        elif (native and self.world > 1 and hasattr(backend, "init_native_comm")
                and dist.get_backend(group) == "nccl"):
            try:
                backend.init_native_comm(dist, group)
            except Exception as ex:  # stay on the torch.distributed route
                if getattr(backend, "native_only", False):
                    raise  # HipPlaneBackend: no torch.distributed route to fall back to -- the failure must not hide until step()
                import sys
                print(f"[mp2p_icp_amd] native RCCL communicator unavailable ({ex}); using torch.distributed",
                      file=sys.stderr)

    def _allreduce(self, t, op):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t, op=op, group=self.group)

    CAP_QUANTUM = 4096  # record lists are exchanged in multiples of this many records

    def _round_cap(self, n):
        q = self.CAP_QUANTUM
        return max(q, -(-int(n) // q) * q)

    def match(self, pose, predicted_cap=None):
        """predicted_cap: length of the record lists to all-gather, guessed by the caller (step()
        uses the previous iteration's count + 25 %) so that no host round trip sits between the
        two match phases; None = read the exact count back (one 8-byte device-to-host sync).
        Returns (exch tensor or None, capacity used): the caller checks exch[6] <= capacity."""
        b = self.b
        b.phase1(pose)
        exch, cap = None, 0
        if self.world > 1:
            exch, recs = b.exchange_pack()
            self._allreduce(exch, self.dist.ReduceOp.MAX)
            gathered = None
            if b.uses_claims:
                if predicted_cap is None:
                    cap = self._round_cap(int(exch[6].item()))  # the one host round trip
                else:
                    cap = int(predicted_cap)
                send = recs[:cap]
                if send.numel() < cap:  # a shorter shard than the list length: pad
                    pad = recs.new_full((cap,), -1)
                    pad[:send.numel()] = send
                    send = pad
                gathered = b.gather_buffer(self.world * cap)
                self.dist.all_gather_into_tensor(gathered, send, group=self.group)
            b.exchange_unpack(gathered)
        b.phase2()
        return exch, cap

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
        """one outer ICP iteration; returns (new pose, inner iterations).  Between the phases of
        the matcher nothing waits for the host: the record-list length comes from the previous
        iteration (+25 %); the true length is checked after the solve, when the stream is idle
        anyway, and the (rare) iteration whose lists did not fit is redone with the exact length."""
        if hasattr(self.b, "step_native") and (self.world == 1 or getattr(self.b, "native", False)):
            return self.b.step_native(pose)  # one call into libmp2p_hip (RCCL inside when sharded)
        if self.world == 1:
            self.match(pose)
            return self.solve(pose)
        guess = getattr(self, "_cap_guess", None)
        exch, cap = self.match(pose, predicted_cap=guess)
        out = self.solve(pose)
        if self.b.uses_claims:
            n_max = int(exch[6].item())
            if guess is not None and n_max > cap:
                exch, cap = self.match(pose)  # exact length this time
                out = self.solve(pose)
                self.redone_steps = getattr(self, "redone_steps", 0) + 1
            self._cap_guess = self._round_cap(n_max * 1.25 + 1024)
        return out


class BatchRegistration:
    """SURVEY.md section 8e, way (i): a batch of INDEPENDENT scan pairs (BASELINE config C4: 64 x
    (1 M vs 1 M) on 8 GPUs).  Pair b belongs to rank b mod world; every rank registers its own
    pairs with the single-GPU path, no collective on the data path.  `gather` hands every rank
    all results with ONE all-reduce SUM of a [B, 14] fp64 table (pose, iterations, quality) in
    which a rank fills only its own rows -- 7 kB for 64 pairs.

    `align_fn(pair_index) -> (pose[12], n_iterations, quality)` is the per-pair registration
    (in the product: an mp2p_icp_amd.ICP.align call on this rank's GPU)."""

    ROW = 14

    def __init__(self, n_pairs, dist=None, group=None, device="cpu"):
        self.n_pairs, self.dist, self.group, self.device = int(n_pairs), dist, group, device
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.world = dist.get_world_size(group) if dist is not None else 1

    def owned(self):
        """pair indices of this rank, in processing order"""
        return list(range(self.rank, self.n_pairs, self.world))

    def run(self, align_fn, gather=True):
        import torch
        table = torch.zeros((self.n_pairs, self.ROW), dtype=torch.float64)
        for b in self.owned():
            pose, iters, quality = align_fn(b)
            table[b, :12] = torch.as_tensor(np.asarray(pose, dtype=np.float64))
            table[b, 12], table[b, 13] = float(iters), float(quality)
        if gather and self.dist is not None and self.world > 1:
            t = table.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            table = t.cpu()
        return table.numpy()
