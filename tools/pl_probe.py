#!/usr/bin/env python3
"""Time Matcher_Point2Plane (K5) + Gauss-Newton on BASELINE config 3 shape.
usage: pl_probe.py [n_local n_global [r0_cells [a|b]]]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core
import bench

n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
n_g = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
if len(sys.argv) > 4 and sys.argv[4] == "b":  # SURVEY.md 8d scene: map = voxel-thinned union of scans (the C3 test's)
    from mp2p_icp_amd import synthetic
    d = synthetic.make_scan_union_pair(n_l, n_g, 3001, map_scan_points=1_000_000)
else:
    d = bench.build_inputs(n_l, n_g, 3001, 0, 1)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, 1, n_l)
gnp = _lib.GNParams(); gnp.maxInnerLoopIterations = 3; gnp.minDelta = 1e-7; gnp.kernel = 1; gnp.kernelParam = 0.15
gnp.w_pt2pt = gnp.w_pt2pl = 1.0
R0 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
for knn, rad in ((5, 0.4), (7, 0.5), (12, 0.8)):
    prm = _lib.Pt2PlParams(0.4, rad, knn, 5, 0.05, 0, 0.20, R0, 0)
    for name, pose in (("init", d["T_init"]), ("gt", d["T_gt"])):
        ts, tg = [], []
        for _ in range(5):
            ctx.set_profiling(1)
            pairs.clear()
            core.match_pt2pl(ctx, gmap, cloud, pose, prm, None, pairs)
            t0 = time.perf_counter()
            res = core.gn_solve(ctx, pairs, pose, gnp)
            st = ctx.stats()
            ts.append(st["ms_nn"]); tg.append(st["ms_gn"])
        ctx.set_profiling(2)
        pairs.clear()
        core.match_pt2pl(ctx, gmap, cloud, pose, prm, None, pairs)
        st = ctx.stats()
        ctx.set_profiling(0)
        nt = max(1, st["nn_tiles"])
        print(json.dumps(dict(tiles=st["nn_tiles"], passes_avg=round(st["nn_passes"] / nt, 2), max_pass=st["nn_max_passes_one_tile"],
                              cand_avg=round(st["nn_candidates_tested"] / nt), max_cand=st["nn_max_candidates_one_tile"],
                              cells_avg=round(st["nn_cells_visited"] / nt), us_avg=round(st["nn_tile_ticks_sum"] / nt / 100, 1),
                              us_max=round(st["nn_tile_ticks_max"] / 100, 1))).replace("{", "{\"knn_dbg\": 1, "))
        print(json.dumps(dict(knn=knn, radius=rad, pose=name, n_l=n_l, n_g=n_g, ms_match=round(float(np.median(ts)), 3),
                              ms_gn=round(float(np.median(tg)), 3), r0=R0, pairs=pairs.counts()[1])), flush=True)
