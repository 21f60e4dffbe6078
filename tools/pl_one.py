#!/usr/bin/env python3
"""Matcher_Point2Plane on the C3 workload (120 k-pt scan vs 10 M-pt map; [n_local] for another scan size), a few calls at
mid-chain poses that move a few mm per call -- the command tools/gpu_pmc.sh profiles.  usage: pl_one.py [n_local] [reps] [step scale]
(step scale 1 = 4.5 mm + 1 mrad between consecutive calls -- an early iteration; 0.05 = a late one)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
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
ctx.set_profiling(1)
ms = []
for k in range(reps):
    pairs.clear()
    core.match_pt2pl(ctx, gmap, cloud, chain if (k & 1) else chain_prev, prm, None, pairs)
    ctx.sync()
    ms.append(ctx.stats()["ms_nn"])
st = ctx.stats()
print("pairs", pairs.counts(), "search+fit ms per call", [round(m, 3) for m in ms], "certified", st["pl_certified"], "searched", st["pl_searched"])
# device counters of one more call (profiling level 2): passes, staged candidates, voxels and ticks per tile
ctx.set_profiling(2)
pairs.clear()
core.match_pt2pl(ctx, gmap, cloud, chain if (reps & 1) else chain_prev, prm, None, pairs)
st = ctx.stats()
ctx.set_profiling(0)
print("instrumented call: ms", round(st["ms_nn"], 3), "tile duration histogram (log2 of 10 ns ticks):",
      {i: c for i, c in enumerate(st["nn_tile_ticks_hist"]) if c}, "slowest tile: us", (st["nn_single_max_candidates"] >> 40) / 100,
      "passes", (st["nn_single_max_candidates"] >> 32) & 255, "candidates", st["nn_single_max_candidates"] & 0xFFFFFFFF)
nt = max(1, st["nn_tiles"])
print(dict(tiles=st["nn_tiles"], passes_avg=round(st["nn_passes"] / nt, 2), max_pass=st["nn_max_passes_one_tile"],
           cand_avg=round(st["nn_candidates_tested"] / nt), max_cand=st["nn_max_candidates_one_tile"],
           cells_avg=round(st["nn_cells_visited"] / nt), us_avg=round(st["nn_tile_ticks_sum"] / nt / 100, 1),
           us_max=round(st["nn_tile_ticks_max"] / 100, 1)))
# round 6 (pt2pl_seltile_kernel): chain iterations of the queue flushes, queued hits, flushes -- per tile
print(dict(flush_iters_per_tile=round(st["nn_coop_passes"] / nt, 1), hits_per_tile=round(st["nn_single_queries"] / nt, 1),
           flushes_per_tile=round(st["nn_single_passes"] / nt, 2)))
print(dict(us_stage_per_tile=round(st["nn_single_cells"] / nt / 100, 1), us_prefilter_per_tile=round(st["nn_single_candidates"] / nt / 100, 1),
           us_flush_per_tile=round(st["nn_single_ticks_sum"] / nt / 100, 1)))
print(dict(us_passes_per_tile=round(st["nn_single_ticks_max"] / nt / 100, 1), us_merge_per_tile=round(st["nn_single_max_passes"] / nt / 100, 1)))
ph = st["nn_wave_phase_ticks"]
print(dict(slowest_tile=dict(hits=ph[0] & 0xFFFFFFFF, chain_iters=ph[1] & 0xFFFFFFFF, enqueue_iters=ph[2] & 0xFFFFFFFF, prefilter_pos=ph[3] & 0xFFFFFFFF, blocks=ph[4] & 0xFFFFFFFF),
           per_tile=dict(enqueue_iters=round(ph[5] / nt, 1), prefilter_pos=round(st["nn_wave_inserts"] / nt, 1), blocks=round(st["nn_wave_rounds"] / nt, 1))))
