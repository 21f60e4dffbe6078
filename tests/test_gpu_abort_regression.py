"""Regression tests of the process abort of round 3 (GPUTEST_r03: SIGABRT inside mp2p_hip_map_upload of the first test
that followed the host-path tests).

Cause, as captured on the GPU box (profiles/r04_abort_*): the pair copy-out page-locked the CALLER's destination with
hipHostRegister / hipHostUnregister -- an unaligned sub-range of the malloc heap -- and a later pageable
hipMemcpyAsync of the runtime from neighbouring heap memory (numpy buffers of a map upload on a NEW context) faulted
on the GPU ("Memory access fault by GPU node ... on address <host heap address>"), which the runtime answers with
abort().  The library now stages through its own page-locked buffer (mp2p::CopyStage) and registers nothing.

A GPU fault kills the process, so these tests either pass or take the run down -- which is the point."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pt2pt_prm(thr, allow_global=0):
    from mp2p_icp_amd import _lib
    p = _lib.Pt2PtParams()
    p.threshold, p.thresholdAngularDeg, p.pairingsPerPoint = thr, 0.0, 1
    p.bounding_box_intersection_check_epsilon = 0.20
    p.allowMatchAlreadyMatchedGlobalPoints = allow_global  # 1: every local point keeps its pair (a long list)
    return p


def test_hostpath_sessions_alternating_with_fresh_contexts(oracle):
    """200 x { a host-path Session whose pair list is large enough for the staged copy-out (> 256 KB), with layer
    eviction and release_layers; then a FRESH context + map upload from fresh numpy buffers of the size that faulted
    (3 x 1.2 MB) + a match on it }.  Results are checked against the first round's."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import core, hostpath, synthetic
    d = synthetic.random_cloud_pair(20_000, 60_000, 5)
    g = d["glob"]
    prm = _pt2pt_prm(0.8)
    big = synthetic.make_pair(40_000, 300_000, 21)
    hostpath.cache(max_layers=3)
    want_n, want_pairs, want_big = None, None, None
    for k in range(200):
        # fresh buffers every round: distinct heap addresses, freed at the end of the round
        l = (d["local"] + np.float32(0.001 * (k % 7))).astype(np.float32)
        s = hostpath.Session(g.copy(), l)
        s.begin_iteration()
        n = s.match_pt2pt(d["T_init"], prm, icp_iteration=0)
        assert n * 36 > (256 << 10), "the list must take the staged path"
        got = s.pairs_pt2pt().copy()
        if k % 7 == 0:
            if want_pairs is None:
                want_n, want_pairs = n, got
            assert n == want_n and np.array_equal(got, want_pairs)
        if k % 3 == 0:
            s.release_layers()
        s.close()
        del s, l, got
        # what test_gpu_comm.py::test_one_rank_over_real_rccl did when the process died
        gb, lb = big["glob"], big["local"]
        ctx = amd.Context(0)
        gmap = core.GlobalMap(ctx, gb[:, 0], gb[:, 1], gb[:, 2])
        cloud = core.LocalCloud(ctx, lb[:, 0], lb[:, 1], lb[:, 2])
        pairs = core.DevicePairs(ctx, lb.shape[0], 0)
        core.match_pt2pt(ctx, gmap, cloud, big["T_init"], _pt2pt_prm(1.5), None, pairs)
        got_big = pairs.download_pt2pt()
        if want_big is None:
            want_big = got_big
        assert np.array_equal(got_big, want_big)
        del pairs, cloud, gmap, ctx
        if k % 16 == 0:
            gc.collect()


@pytest.mark.parametrize("tune", ["copy_chunk_kb=64,copy_stage_mb=1", "copy_chunk_kb=4"])
def test_staged_copy_out_in_many_chunks_and_rounds(oracle, tune, monkeypatch):
    """the staged copy-out with a staging buffer SMALLER than the list (several rounds) and with tiny chunks (hundreds of
    events, the helper thread and the calling thread sharing them): the records that reach the caller's container are
    the device list, byte for byte -- through the split copy (begin / wait_idx / end) and through the one-call copy"""
    import ctypes as C
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    monkeypatch.setenv("MP2P_HIP_TUNE", tune)
    d = synthetic.make_pair(60_000, 300_000, 33)
    g, l = d["glob"], d["local"]
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, l.shape[0], 0)
    core.match_pt2pt(ctx, gmap, cloud, d["T_init"], _pt2pt_prm(1.5, allow_global=1), None, pairs)
    want = pairs.download_pt2pt()  # plain pageable download
    n = len(want)
    assert n * 36 > (1 << 20) + (256 << 10)
    L = ctx._L
    for first in (0, 1234):
        m = n - first
        # one-call copy
        out = np.zeros(m, dtype=want.dtype)
        _lib.check(L.mp2p_hip_pairs_copy_pt2pt(ctx.handle, pairs.handle, first, m, out.ctypes.data_as(C.c_void_p)), ctx.handle)
        assert np.array_equal(out, want[first:])
        # split copy: index arrays first, records while the caller works
        out2 = np.zeros(m, dtype=want.dtype)
        li, gi = np.zeros(m, np.uint32), np.zeros(m, np.uint32)
        _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin(ctx.handle, pairs.handle, first, m, out2.ctypes.data_as(C.c_void_p),
                                                       li.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                       gi.ctypes.data_as(C.POINTER(C.c_uint32))), ctx.handle)
        _lib.check(L.mp2p_hip_pairs_copy_wait_idx(ctx.handle), ctx.handle)
        assert np.array_equal(li, want["localIdx"][first:]) and np.array_equal(gi, want["globalIdx"][first:])
        _lib.check(L.mp2p_hip_pairs_copy_end(ctx.handle), ctx.handle)
        assert np.array_equal(out2, want[first:])
    # a begin that is never ended is finished by the next begin / by the context's destruction
    out3 = np.zeros(n, dtype=want.dtype)
    li, gi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    args = (ctx.handle, pairs.handle, 0, n, out3.ctypes.data_as(C.c_void_p), li.ctypes.data_as(C.POINTER(C.c_uint32)),
            gi.ctypes.data_as(C.POINTER(C.c_uint32)))
    _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin(*args), ctx.handle)
    _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin(*args), ctx.handle)
    _lib.check(L.mp2p_hip_pairs_copy_end(ctx.handle), ctx.handle)
    assert np.array_equal(out3, want)
    _lib.check(L.mp2p_hip_pairs_copy_pt2pt_begin(*args), ctx.handle)
    del pairs, cloud, gmap, ctx  # destroyed with a copy open
    gc.collect()
