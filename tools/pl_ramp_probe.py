#!/usr/bin/env python3
"""How pt2pl_seltile_kernel fills the chip: start times of its tiles (profiling level 4) by grid position.  usage: pl_ramp_probe.py [n_local]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
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
for k in range(4):
    pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, chain if (k & 1) else chain_prev, prm, None, pairs)
ctx.set_profiling(1)
pairs.clear(); core.match_pt2pl(ctx, gmap, cloud, chain, prm, None, pairs)
ms_plain = ctx.stats()["ms_nn"]
ctx.set_profiling(4)
pairs.clear()
core.match_pt2pl(ctx, gmap, cloud, chain_prev, prm, None, pairs)
ms = ctx.stats()["ms_nn"]
rec, _ = core.timeline(ctx)
ctx.set_profiling(0)
flat = rec.reshape(-1)
n_grid = len(rec) * 2 // 3
r = flat[:2 * n_grid].reshape(-1, 2).astype(np.int64)
ids = np.nonzero(r[:, 1] > 0)[0]
r = r[ids]
t0 = r[:, 0].min()
st = (r[:, 0] - t0) / 100.0
en = (r[:, 1] - t0) / 100.0
order = np.argsort(st)
out = dict(ms_events_plain=round(ms_plain, 3), ms_events_timeline_build=round(ms, 3), tiles=int(len(r)), span_us=round(float(en.max()), 1),
           start_us_of_kth_started={str(k): round(float(st[order[k]]), 1) for k in (100, 500, 1000, 2000, 3000, 4000, 6000, 10000, 20000, 30000) if k < len(r)},
           first_3000_grid_ids=dict(start_p50=round(float(np.median(st[ids < 3000])), 1), start_max=round(float(st[ids < 3000].max()), 1),
                                    dur_mean=round(float((en - st)[ids < 3000].mean()), 1)),
           dur_mean_us=round(float((en - st).mean()), 1), sum_wave_ms=round(float((en - st).sum()) / 1e3, 1))
print(json.dumps(out))
