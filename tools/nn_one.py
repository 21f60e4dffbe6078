#!/usr/bin/env python3
"""Run the matcher a few times on the bench workload (for rocprofv3).  args: pose(gt|init|chain) reps [json cfg]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core
import bench
pose_name = sys.argv[1] if len(sys.argv) > 1 else "gt"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
n_l, n_g = cfg.get("n_l", 1_000_000), cfg.get("n_g", 10_000_000)
d = bench.build_inputs(n_l, n_g, 1, 0, 1, cfg.get("scene", "b"))  # the headline scene
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, n_l, 0)
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2], cell_size=cfg.get("cell", 0.0), target_per_cell=cfg.get("tpc", 0.0))
prm = _lib.Pt2PtParams(2.0, 0.0, 1, 0, 0, 0.20, 0, cfg.get("r0", 0.0), cfg.get("q", 0), cfg.get("grp", 0.0), cfg.get("budget", 0), cfg.get("defer", 0.0), int(cfg.get("cold", 0)), int(cfg.get("bricks", 0)))
pose = d["T_gt"] if pose_name == "gt" else d["T_init"]
# "chain": a typical mid-chain ICP iteration (0.4 m off, moving a few mm per call), alternating two poses
chain = amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.3, -0.3, 0.05, 0.0, 0.0, 0.03])))
chain_prev = amd.se3.compose(chain, amd.se3.exp(np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))
gnp = _lib.GNParams(); gnp.maxInnerLoopIterations = 3; gnp.minDelta = 1e-7; gnp.kernel = 1; gnp.kernelParam = 0.15; gnp.w_pt2pt = gnp.w_pt2pl = 1.0
for k in range(reps):
    pairs.clear()
    if pose_name == "chain":
        pose = chain if (k & 1) else chain_prev
    core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
    core.gn_solve(ctx, pairs, pose, gnp)
ctx.sync()
print("pairs", pairs.counts())
