#!/usr/bin/env python3
"""Occupancy over time of the two search kernels on the bench workload (profiling level 4):
how many workgroups are resident in each slice of the kernel's duration, the share of the time
spent below half of the peak residency (the tail), and the duration distribution of the tiles.
usage: timeline_probe.py [n_local n_global [scene a|b]]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mp2p_icp_amd as amd  # noqa: E402
from mp2p_icp_amd import _lib, core  # noqa: E402

n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_g = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
scene = sys.argv[3] if len(sys.argv) > 3 else "b"
d = bench.build_inputs(n_l, n_g, 1, 0, 1, scene)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, n_l, 0)
prm = _lib.Pt2PtParams(2.0, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
chain = amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.3, -0.3, 0.05, 0.0, 0.0, 0.03])))
chain_prev = amd.se3.compose(chain, amd.se3.exp(np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))


def profile(rec, label, slices=20):
    rec = rec[rec[:, 1] > 0].astype(np.int64)
    if not len(rec):
        return {label: "no records"}
    t0, t1 = rec[:, 0].min(), rec[:, 1].max()
    span = max(1, t1 - t0)
    edges = t0 + span * np.arange(slices + 1) / slices
    resident = []
    for k in range(slices):  # workgroup-time inside the slice / slice length
        a, b = edges[k], edges[k + 1]
        ov = np.clip(np.minimum(rec[:, 1], b) - np.maximum(rec[:, 0], a), 0, None).sum()
        resident.append(round(float(ov) / float(b - a), 1))
    dur = (rec[:, 1] - rec[:, 0]) / 100.0  # us
    peak = max(resident)
    return {label: dict(span_us=round(span / 100.0, 1), workgroups=int(len(rec)), resident_per_slice=resident,
                        share_of_time_below_half_peak=round(sum(1 for r in resident if r < 0.5 * peak) / slices, 2),
                        mean_resident=round(float(np.mean(resident)), 1),
                        dur_us=dict(mean=round(float(dur.mean()), 1), p50=round(float(np.percentile(dur, 50)), 1),
                                    p90=round(float(np.percentile(dur, 90)), 1), p99=round(float(np.percentile(dur, 99)), 1),
                                    max=round(float(dur.max()), 1)))}


for name, warm, pose in (("init", d["T_gt"], d["T_init"]), ("gt", d["T_init"], d["T_gt"]), ("chain", chain_prev, chain)):
    ctx.set_profiling(0)
    pairs.clear()
    core.match_pt2pt(ctx, gmap, cloud, warm, prm, None, pairs)
    ctx.set_profiling(4)
    pairs.clear()
    core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
    ms = ctx.stats()["ms_nn"]
    tiles, singles = core.timeline(ctx)
    ctx.set_profiling(0)
    out = dict(pose=name, ms_nn=round(ms, 3))
    raw = tiles.astype(np.uint64)
    cand, npass, rounds = (raw[:, 0] >> np.uint64(40)).astype(np.int64) * 32, ((raw[:, 1] >> np.uint64(40)) & np.uint64(255)).astype(np.int64), (raw[:, 1] >> np.uint64(48)).astype(np.int64)
    tiles = (raw & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
    tt = tiles
    out.update(profile(tiles, "tile_kernel"))
    dur = (tt[:, 1] - tt[:, 0]) / 100.0
    top = np.argsort(-dur)[:12]
    t0_ = tt[tt[:, 1] > 0][:, 0].min()
    out["longest_tiles"] = [dict(block=int(i), start_us=round(float(tt[i, 0] - t0_) / 100.0, 1), dur_us=round(float(dur[i]), 1), cand=int(cand[i]),
                                 passes=int(npass[i]), brick_rounds=int(rounds[i])) for i in top]
    ok = tt[:, 1] > 0
    for name_, v in (("cand", cand), ("passes", npass), ("brick_rounds", rounds)):
        out["corr_dur_" + name_] = round(float(np.corrcoef(dur[ok], v[ok])[0, 1]), 3)
    out["tiles_by_passes"] = {int(k): [int((npass[ok] == k).sum()), round(float(dur[ok][npass[ok] == k].mean()), 1)] for k in np.unique(npass[ok])[:12]}
    nz = tt[:2048][tt[:2048, 1] > 0]
    out["first_2048_blocks"] = dict(n=int(len(nz)), dur_mean=round(float(((nz[:, 1] - nz[:, 0]) / 100.0).mean()), 1) if len(nz) else 0,
                                   dur_max=round(float(((nz[:, 1] - nz[:, 0]) / 100.0).max()), 1) if len(nz) else 0)
    out.update(profile(singles, "single_kernel"))
    print(json.dumps(out), flush=True)
