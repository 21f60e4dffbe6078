#!/usr/bin/env python3
"""Probe the NN search kernel on the bench workload: per-tile time distribution and counters
for a few tuning points.  usage: nn_probe.py [n_local n_global]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core
import bench

n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_g = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
d = bench.build_inputs(n_l, n_g, 1, 0, 1)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, n_l, 0)
configs = [dict(), dict(q=32), dict(q=16), dict(q=16, defer=4.0), dict(q=32, grp=3.0), dict(q=16, r0=1.5), dict(q=16, tpc=4.0)]
if len(sys.argv) > 3:
    configs = json.loads(sys.argv[3])
maps = {}
for cfg in configs:
    tpc = cfg.get("tpc", 0.0); cell = cfg.get("cell", 0.0)
    key = (tpc, cell)
    if key not in maps:
        maps[key] = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2], cell_size=cell, target_per_cell=tpc)
    gmap = maps[key]
    prm = _lib.Pt2PtParams(2.0, 0.0, 1, 0, 0, 0.20, 0, cfg.get("r0", 0.0), cfg.get("q", 0), cfg.get("grp", 0.0), cfg.get("budget", 0), cfg.get("defer", 0.0), int(cfg.get("cold", 0)), int(cfg.get("bricks", 0)))
    chain = amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.3, -0.3, 0.05, 0.0, 0.0, 0.03])))
    chain_prev = amd.se3.compose(chain, amd.se3.exp(np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))
    other = {"init": d["T_gt"], "gt": d["T_init"], "chain": chain_prev}
    for name, pose in (("init", d["T_init"]), ("gt", d["T_gt"]), ("chain", chain)):
        ctx.set_profiling(1)
        ts = []; ts_s = []
        for _ in range(5):
            # the warm start of the timed call comes from the OTHER pose (a 0.5 m-class jump), not
            # from an identical call
            ctx.set_profiling(0); pairs.clear(); core.match_pt2pt(ctx, gmap, cloud, other[name], prm, None, pairs); ctx.set_profiling(1)
            pairs.clear()
            core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
            ts.append(ctx.stats()["ms_nn"]); ts_s.append(ctx.stats()["ms_nn_single"])
        ctx.set_profiling(0); pairs.clear(); core.match_pt2pt(ctx, gmap, cloud, other[name], prm, None, pairs)
        ctx.set_profiling(2)
        pairs.clear()
        core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
        st = ctx.stats()
        ctx.set_profiling(0)
        hist = st["nn_tile_ticks_hist"]
        nt = max(1, st["nn_tiles"])
        print(json.dumps(dict(cfg=cfg, pose=name, cell=round(gmap.info()["cell_size"], 3), ms_nn=round(float(np.median(ts)), 3), ms_single=round(float(np.median(ts_s)), 3),
              tiles=st["nn_tiles"], passes_per_tile=round(st["nn_passes"] / nt, 2), deferred=st["nn_coop_passes"], single=dict(q=st["nn_single_queries"], passes=st["nn_single_passes"], cells=st["nn_single_cells"], cand=st["nn_single_candidates"], maxcand=st["nn_single_max_candidates"], us_avg=round(st["nn_single_ticks_sum"] / max(1, st["nn_single_queries"]) / 100.0, 1), us_max=round(st["nn_single_ticks_max"] / 100.0, 1), maxpass=st["nn_single_max_passes"], maxcells=st["nn_single_max_cells"]),
              cand_per_tile=round(st["nn_candidates_tested"] / nt, 1), cells_per_tile=round(st["nn_cells_visited"] / nt, 1),
              touched=st["nn_points_staged"], max_cand=st["nn_max_candidates_one_tile"], max_pass=st["nn_max_passes_one_tile"],
              tile_us_avg=round(st["nn_tile_ticks_sum"] / nt / 100.0, 2), tile_us_max=round(st["nn_tile_ticks_max"] / 100.0, 1),
              hist_log2_10ns={i: h for i, h in enumerate(hist) if h}, pairs=pairs.counts()[0])), flush=True)
