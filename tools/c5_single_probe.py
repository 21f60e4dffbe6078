#!/usr/bin/env python3
"""What the one-query kernel does on configuration C5 (5 M x 5 M, 30 % uniform outliers): deferred queries, their passes / cells /
candidates, ticks, per chain position (instrumented point matcher calls after the plane matcher).  usage: c5_single_probe.py [n]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
d = synthetic.make_scan_union_pair(N, N, 5001, map_scan_points=1_000_000, outlier_frac=0.30)
g, l = d["glob"], d["local"]
ctx = amd.Context(0)
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
n_l = l.shape[0]
pairs = core.DevicePairs(ctx, n_l, n_l)
ms = core.DeviceMatchState(ctx, g.shape[0], n_l)
pt = _lib.Pt2PtParams(); pt.threshold, pt.thresholdAngularDeg, pt.pairingsPerPoint = 1.0, 0.0, 1
pt.bounding_box_intersection_check_epsilon = 0.20
pl = _lib.Pt2PlParams(); pl.distanceThreshold = 0.25
pl.searchRadius, pl.knn, pl.minimumPlanePoints, pl.planeEigenThreshold = 0.4, 5, 5, 0.05
pl.bounding_box_intersection_check_epsilon = 0.20
gnp = _lib.GNParams(); gnp.maxInnerLoopIterations, gnp.minDelta, gnp.maxCost = 3, 1e-7, 0.0
gnp.kernel, gnp.kernelParam, gnp.w_pt2pt, gnp.w_pt2pl = _lib.KERNEL_CAUCHY, 0.15, 1.0, 1.0
pose = d["T_init"].copy()
for k in range(6):
    ms.reset(); pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, pose, pl, ms, pairs)
    ctx.set_profiling(2)
    core.match_pt2pt(ctx, gmap, cloud, pose, pt, ms, pairs)
    st = ctx.stats()
    ctx.set_profiling(0)
    keep = {k2: int(st[k2]) for k2 in ("nn_queries", "nn_lane_pending", "nn_lane_skipped", "nn_lane_searched", "nn_tiles", "nn_single_queries", "nn_single_passes",
                                       "nn_single_cells", "nn_single_candidates", "nn_single_max_candidates", "nn_single_ticks_sum", "nn_single_ticks_max", "nn_single_max_cells")}
    keep["ms_nn"] = round(st["ms_nn"], 3)
    print(k, json.dumps(keep), flush=True)
    pose = np.array(core.gn_solve(ctx, pairs, pose, gnp).pose)
