#!/usr/bin/env python3
"""Speed-of-light cuts of pt2pl_seltile_kernel on the C3 workload (round 6): the same grid, poses and warm start, the kernel cut
after a phase (MP2P_HIP_TUNE knob pl_sol through set_tune, profiling on; results of those calls are invalid).  Per cut: kernel time
(ms_nn), mean / p50 / max tile duration from the per-tile timeline.  usage: pl_sol_probe.py [n_local] [pl_waves]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = synthetic.make_scan_union_pair(n_l, 10_000_000, 3001, map_scan_points=1_000_000)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
n_l = l.shape[0]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, 1, n_l)
prm = _lib.Pt2PlParams()
prm.distanceThreshold, prm.searchRadius, prm.knn, prm.minimumPlanePoints, prm.planeEigenThreshold = 0.4, 0.4, 5, 5, 0.05
prm.bounding_box_intersection_check_epsilon = 0.20
chain = amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.05, -0.04, 0.01, 0.0, 0.0, 0.004])))
chain_prev = amd.se3.compose(chain, amd.se3.exp(np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))
ctx.set_tune(f"pl_waves={waves}")
out = {}
for sol in (0, 1, 2, 3, 4):
    ctx.set_profiling(0)
    ctx.set_tune("pl_warm=0")
    pairs.clear(); core.match_pt2pl(ctx, gmap, cloud, chain_prev, prm, None, pairs)   # cold, exact: the state of a normal call
    ctx.set_tune("pl_warm=1")
    pairs.clear(); core.match_pt2pl(ctx, gmap, cloud, chain, prm, None, pairs)
    ctx.set_profiling(4)
    ctx.set_tune(f"pl_sol={sol}")
    pairs.clear(); core.match_pt2pl(ctx, gmap, cloud, chain_prev, prm, None, pairs)
    ms = ctx.stats()["ms_nn"]
    rec, _ = core.timeline(ctx)
    ctx.set_tune("pl_sol=0")
    flat = rec.reshape(-1)
    n_grid = len(rec) * 2 // 3
    r = flat[:2 * n_grid].reshape(-1, 2)
    r = r[r[:, 1] > 0].astype(np.int64)
    dur = (r[:, 1] - r[:, 0]) / 100.0
    out[sol] = dict(ms=round(ms, 3), tiles=int(len(r)), mean_us=round(float(dur.mean()), 1), p50_us=round(float(np.percentile(dur, 50)), 1),
                    p99_us=round(float(np.percentile(dur, 99)), 1), max_us=round(float(dur.max()), 1), sum_ms=round(float(dur.sum()) / 1000, 2))
    print(sol, out[sol], flush=True)
print(json.dumps(out))
