#!/usr/bin/env python3
"""Search time per position of the bench chain (restart from the perturbed guess every 10 steps) for the plane matcher of C3
or the point matcher of the default line, with the warm start dropped at chosen positions (set_tune pl_warm / nn_warm... = 0 for
that call).  usage: pos_probe.py c3|n1 [cold positions, e.g. 0 or 0,1] [cycles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
cold = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] != "-" else []
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 4
extra = sys.argv[4] if len(sys.argv) > 4 else ""
ctx = amd.Context(0)
if which == "c3":
    d = synthetic.make_scan_union_pair(120_000, 10_000_000, 3001, map_scan_points=1_000_000)
else:
    d = bench.build_inputs(1_000_000, 10_000_000, 1, 0, 1, "b")
g, l = d["glob"], d["local"]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
n_l = l.shape[0]
pairs = core.DevicePairs(ctx, n_l, n_l)
pt = _lib.Pt2PtParams(); pt.threshold, pt.thresholdAngularDeg, pt.pairingsPerPoint = 2.0, 0.0, 1
pt.bounding_box_intersection_check_epsilon = 0.20
pl = _lib.Pt2PlParams(); pl.distanceThreshold = 0.4
pl.searchRadius, pl.knn, pl.minimumPlanePoints, pl.planeEigenThreshold = 0.4, 5, 5, 0.05
pl.bounding_box_intersection_check_epsilon = 0.20
gnp = _lib.GNParams(); gnp.maxInnerLoopIterations, gnp.minDelta, gnp.maxCost = 3, 1e-7, 0.0
gnp.kernel, gnp.kernelParam, gnp.w_pt2pt, gnp.w_pt2pl = _lib.KERNEL_GEMANMCCLURE, 0.15, 1.0, 1.0
if extra:
    ctx.set_tune(extra)
CY = 10
nn = np.zeros((cycles, CY)); wall = np.zeros((cycles, CY)); disp = np.zeros((cycles, CY))
ctx.set_profiling(1)
prev = None
for c in range(cycles + 1):
    pose = d["T_init"].copy()
    for k in range(CY):
        if k in cold:
            if which == "c3":
                ctx.set_tune("pl_warm=0")
            pt.disable_warm_start = 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pairs.clear()
        if which == "c3":
            core.match_pt2pl(ctx, gmap, cloud, pose, pl, None, pairs)
        else:
            core.match_pt2pt(ctx, gmap, cloud, pose, pt, None, pairs)
        new = np.array(core.gn_solve(ctx, pairs, pose, gnp).pose)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if k in cold:
            if which == "c3":
                ctx.set_tune("pl_warm=1")
            pt.disable_warm_start = 0
        if c > 0:
            nn[c - 1, k] = ctx.stats()["ms_nn"]; wall[c - 1, k] = (t1 - t0) * 1e3
            if prev is not None:
                disp[c - 1, k] = np.linalg.norm(np.asarray(pose)[:3, 3] - np.asarray(prev)[:3, 3]) if np.asarray(pose).shape == (4, 4) else 0.0
        prev = pose
        pose = new
print(f"{which} cold={cold} {extra}: search ms per position", np.round(nn.mean(0), 3).tolist(), "mean", round(float(nn.mean()), 4),
      "| wall ms per position", np.round(wall.mean(0), 3).tolist(), "mean", round(float(wall.mean()), 4))
