#!/usr/bin/env python3
"""Occupancy over time and tile durations of pt2pl_tile_kernel on the C3 workload (profiling level 4: {start, end}
ticks per tile, no atomics).  usage: pl_timeline.py [n_local] [step scale] ; MP2P_HIP_TUNE as usual"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
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
chain_prev = amd.se3.compose(chain, amd.se3.exp(scale * np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))
for k in range(4):
    pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, chain if (k & 1) else chain_prev, prm, None, pairs)
ctx.set_profiling(4)
pairs.clear()
core.match_pt2pl(ctx, gmap, cloud, chain_prev, prm, None, pairs)
ms = ctx.stats()["ms_nn"]
rec, _ = core.timeline(ctx)
ctx.set_profiling(0)
flat = rec.reshape(-1)
n_grid = (len(flat) // 3) if len(flat) % 3 == 0 else int(round(len(flat) / 3))
n_grid = len(rec) * 2 // 3
info = flat[2 * n_grid:2 * n_grid + n_grid]
rec = flat[:2 * n_grid].reshape(-1, 2)
n_cb = (n_l + 1023) // 1024
Q = 8 if n_l <= 2000000 else 32
# the grid: [hard tiles][easy tiles]; ball-rule kernel (layers above 524 288 queries): 32-query tiles in both classes, hard list n / 16;
# box-rule kernel: 8-query tiles, hard tiles of 4, hard list min(n / 8, 32 768) entries (launch_match_pt2pl)
if n_l > 524288:
    n_hard_grid = (max(n_l // 16, 64) // 32 * 32) // 32
else:
    n_hard_grid = (min(max(n_l // 8, 64), 32768) // 32 * 32) // 4
is_hard = (np.arange(len(rec)) < n_hard_grid)[rec[:, 1] > 0]
info = info[rec[:, 1] > 0]
rec = rec[rec[:, 1] > 0].astype(np.int64)
t0, t1 = rec[:, 0].min(), rec[:, 1].max()
span = max(1, t1 - t0)
slices = 20
edges = t0 + span * np.arange(slices + 1) / slices
resident = []
for k in range(slices):
    a, b = edges[k], edges[k + 1]
    ov = np.clip(np.minimum(rec[:, 1], b) - np.maximum(rec[:, 0], a), 0, None).sum()
    resident.append(round(float(ov) / float(b - a), 1))
dur = (rec[:, 1] - rec[:, 0]) / 100.0
order = np.argsort(-dur)[:8]
hard_info = dict(hard_grid=int(n_hard_grid), hard_tiles_run=int(is_hard.sum()),
                 easy_first_start_us=(round(float((rec[~is_hard, 0] - t0).min()) / 100.0, 1) if (~is_hard).any() else None),
                 top100_by_duration_in_hard_class=int(is_hard[np.argsort(-dur)[:100]].sum()))
print(json.dumps(hard_info))
print(json.dumps(dict(ms_search_fit=round(ms, 3), span_us=round(span / 100.0, 1), tiles=int(len(rec)), resident_per_slice=resident,
                      dur_us=dict(mean=round(float(dur.mean()), 1), p50=round(float(np.percentile(dur, 50)), 1),
                                  p90=round(float(np.percentile(dur, 90)), 1), p99=round(float(np.percentile(dur, 99)), 1),
                                  max=round(float(dur.max()), 1)),
                      hard_tiles=int(is_hard.sum()), hard_dur_us=(dict(mean=round(float(dur[is_hard].mean()), 1), max=round(float(dur[is_hard].max()), 1),
                                                                   last_end_us=round(float((rec[is_hard, 1] - t0).max()) / 100.0, 1)) if is_hard.any() else None),
                      slowest=[dict(us=round(float(dur[i]), 1), start_us=round(float(rec[i, 0] - t0) / 100.0, 1), passes=int(info[i] >> 48),
                                    voxels=int((info[i] >> 32) & 0xFFFF), cand=int(info[i] & 0xFFFFFFFF)) for i in order],
                      corr=dict(dur_vs_cand=round(float(np.corrcoef(dur, (info & 0xFFFFFFFF).astype(np.float64))[0, 1]), 3),
                                dur_vs_passes=round(float(np.corrcoef(dur, (info >> 48).astype(np.float64))[0, 1]), 3),
                                us_per_pass=round(float(dur.sum() / max(1, (info >> 48).sum())), 1),
                                cand_mean=round(float((info & 0xFFFFFFFF).mean()), 1), passes_mean=round(float((info >> 48).mean()), 2)),
                      wave_seconds_over_span_times_5120=round(float(dur.sum()) / (span / 100.0 * 5120), 3))))
