#!/usr/bin/env python3
"""Replays tests/test_gpu_matcher_pt2pl.py::test_certificate_pose_sequence[8-0.6] with one feature off at a time and
reports which local points differ from the oracle at every call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
knn, radius = 8, 0.6
d = synthetic.make_pair(8000, 400000, 277 + knn)
g, l = d["glob"], d["local"]
tree = oracle.KDTree(g[:, 0], g[:, 1], g[:, 2])
rng = np.random.default_rng(knn)
poses, scales = [d["T_gt"]], [1e-3, 1e-3, 3e-4, 1e-4, 5e-5, 5e-5, 3e-2, 5e-5]
for sc in scales:
    poses.append(amd.se3.compose(poses[-1], amd.se3.exp(np.concatenate([rng.normal(0, sc, 3), rng.normal(0, 0.1 * sc, 3)]))))
takens = []
for k in range(len(poses)):
    t = np.zeros(l.shape[0], np.uint8)
    if k in (2, 5):
        t[rng.choice(l.shape[0], 500, replace=False)] = 1
    takens.append(t)
wants = [oracle.match_pt2pl(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], p, 0.25, radius, knn, 5, 0.05, tree=tree, local_taken=t.copy())[1]
         for p, t in zip(poses, takens)]
for tune in ("", "pl_cert_margin_mm=40"):
    os.environ["MP2P_HIP_TUNE"] = tune
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 0, l.shape[0])
    ms = core.DeviceMatchState(ctx, g.shape[0], l.shape[0])
    prm = _lib.Pt2PlParams()
    prm.distanceThreshold, prm.searchRadius, prm.knn, prm.minimumPlanePoints, prm.planeEigenThreshold = 0.25, radius, knn, 5, 0.05
    prm.bounding_box_intersection_check_epsilon = 0.20
    out = []
    for k, pose in enumerate(poses):
        ms.upload(np.zeros(g.shape[0], np.uint8), takens[k])
        pairs.clear()
        core.match_pt2pl(ctx, gmap, cloud, pose, prm, ms, pairs)
        got, gidx = pairs.download_pt2pl()
        miss = sorted(set(wants[k].tolist()) - set(gidx.tolist())), sorted(set(gidx.tolist()) - set(wants[k].tolist()))
        out.append((k, len(gidx), len(wants[k]), miss[0][:5], miss[1][:5]))
    print(repr(tune), [o for o in out if o[3] or o[4]] or "all equal")
