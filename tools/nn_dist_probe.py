#!/usr/bin/env python3
"""Distribution of the true nearest-neighbour distance of the bench's scan points (scene B) at a mid-chain pose: which
share of the queries lies beyond the deferral radius, and which has nothing inside the threshold at all (CPU kd-tree on
a sample).  usage: nn_dist_probe.py [scene]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mp2p_icp_amd as amd
from scipy.spatial import cKDTree
scene = sys.argv[1] if len(sys.argv) > 1 else "b"
d = bench.build_inputs(1_000_000, 10_000_000, 1, 0, 1, scene)
g, l = d["glob"], d["local"]
tree = cKDTree(g[::1])
rng = np.random.default_rng(0)
idx = rng.choice(l.shape[0], 50_000, replace=False)
for name, T in (("gt", d["T_gt"]), ("chain", amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.3, -0.3, 0.05, 0.0, 0.0, 0.03])))), ("init", d["T_init"])):
    R, t = np.asarray(T)[:9].reshape(3, 3), np.asarray(T)[9:]
    q = l[idx].astype(np.float64) @ R.T + t
    dd, _ = tree.query(q, k=2, workers=-1)
    d1, d2 = dd[:, 0], dd[:, 1]
    print(name, "NN distance percentiles [m]", np.round(np.percentile(d1, [10, 50, 75, 90, 95, 99]), 3),
          "share beyond 0.67 m: %.3f, beyond 1 m: %.3f, beyond 2 m (no pair): %.3f" % ((d1 > 0.672).mean(), (d1 > 1.0).mean(), (d1 > 2.0).mean()),
          "| of those beyond 0.67 m: no pair %.3f, median gap to the 2nd nearest %.4f m" % ((d1[d1 > 0.672] > 2.0).mean(), np.median((d2 - d1)[(d1 > 0.672) & (d1 < 2.0)])))
