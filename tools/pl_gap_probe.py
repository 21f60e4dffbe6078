#!/usr/bin/env python3
"""CPU statistics of the C3 scene behind the point-to-plane certificate's hit rates (scipy kd-tree, ~9 minutes on this
container, no GPU): how many queries have full k-lists, where the k-th neighbour sits, how large the gap to the
(k+1)-th is -- the room the certificate has -- and, for the short lists, how far the nearest point beyond the search
radius is.  Output of the round-3 run: profiles/r03_c3_scene_statistics.txt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp2p_icp_amd import synthetic, se3
from scipy.spatial import cKDTree
t0 = time.time()
d = synthetic.make_scan_union_pair(120_000, 10_000_000, 3001, map_scan_points=1_000_000)
print("scene generated in %.0f s:" % (time.time() - t0), d["local"].shape, d["glob"].shape, flush=True)
g, l = d["glob"].astype(np.float64), d["local"]
tree = cKDTree(g)
T = np.asarray(se3.compose(d["T_gt"], se3.exp(np.array([0.05, -0.04, 0.01, 0, 0, 0.004]))))
R, t = T[:9].reshape(3, 3), T[9:]
q = l.astype(np.float64) @ R.T + t
dd, _ = tree.query(q, k=6, workers=-1)
n_in = (dd[:, :5] <= 0.4).sum(1)
print("queries", len(q), "full lists (5 within 0.4 m):", float((n_in == 5).mean()), "none within:", float((n_in == 0).mean()),
      "1-4:", float(((n_in > 0) & (n_in < 5)).mean()))
full = n_in == 5
gap = dd[full, 5] - dd[full, 4]
print("full lists: 5th distance percentiles 10/50/90 [m]", np.round(np.percentile(dd[full, 4], [10, 50, 90]), 3),
      "gap to the 6th, percentiles 10/25/50/75/90 [m]", np.round(np.percentile(gap, [10, 25, 50, 75, 90]), 4))
k6 = dd[~full]
nxt = np.array([row[row > 0.4].min() if (row > 0.4).any() else np.inf for row in k6])
print("short lists: nearest point beyond the radius, percentiles 10/25/50/75/90 [m]",
      np.round(np.percentile(nxt[np.isfinite(nxt)], [10, 25, 50, 75, 90]), 3))
