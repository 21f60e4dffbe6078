#!/usr/bin/env python3
"""pairingsPerPoint = K matcher (the k-NN search without the plane fit) vs Matcher_Point2Plane timing."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core
import bench
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = bench.build_inputs(n_l, 10_000_000, 3001, 0, 1)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
for K, rad in ((5, 0.4), (12, 0.8)):
    pairs = core.DevicePairs(ctx, n_l * K, n_l)
    p1 = _lib.Pt2PtParams(rad, 0.0, K, 0, 1, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
    p2 = _lib.Pt2PlParams(0.4, rad, K, 5, 0.05, 0, 0.20, 0.0, 0)
    for name, fn, prm in (("pt2pt_knn", core.match_pt2pt, p1), ("pt2pl", core.match_pt2pl, p2)):
        ts = []
        ctx.set_profiling(1)
        for _ in range(5):
            pairs.clear()
            fn(ctx, gmap, cloud, d["T_gt"], prm, None, pairs)
            ts.append(ctx.stats()["ms_nn"])
        ctx.set_profiling(0)
        print(json.dumps(dict(kernel=name, K=K, radius=rad, n_l=n_l, ms=round(float(np.median(ts)), 3), counts=pairs.counts())), flush=True)
