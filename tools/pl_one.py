#!/usr/bin/env python3
"""One Matcher_Point2Plane configuration, a few calls (for rocprofv3 --pmc).  usage: pl_one.py [n_local]"""
import os, sys
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
pairs = core.DevicePairs(ctx, 1, n_l)
prm = _lib.Pt2PlParams(0.4, 0.4, 5, 5, 0.05, 0, 0.20, 0.0, 0)
for _ in range(3):
    pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, d["T_gt"], prm, None, pairs)
ctx.sync()
print(pairs.counts())
