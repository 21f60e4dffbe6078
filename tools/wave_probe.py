#!/usr/bin/env python3
"""Per-wave records of nn_wave_kernel on the bench workload (profiling level 4): duration against the
widest voxel list, staging rounds, passes of each wave; the slowest waves.
usage: wave_probe.py [scene a|b] [pose chain|init]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mp2p_icp_amd as amd  # noqa: E402
from mp2p_icp_amd import _lib, core  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "a"
only = sys.argv[2] if len(sys.argv) > 2 else ""
d = bench.build_inputs(1_000_000, 10_000_000, 1, 0, 1, scene)
ctx = amd.Context(0)
g, l = d["glob"], d["local"]
if os.environ.get("PROBE_NL"):
    l = np.ascontiguousarray(l[: int(os.environ["PROBE_NL"])])  # a prefix (scan order: one sector): few waves, no contention
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
pairs = core.DevicePairs(ctx, l.shape[0], 0)
prm = _lib.Pt2PtParams(2.0, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
chain = amd.se3.compose(d["T_gt"], amd.se3.exp(np.array([0.3, -0.3, 0.05, 0.0, 0.0, 0.03])))
chain_prev = amd.se3.compose(chain, amd.se3.exp(np.array([0.004, 0.002, 0.0, 0.0, 0.0, 0.001])))
M40 = np.uint64((1 << 40) - 1)
for name, warm, pose in (("chain", chain_prev, chain), ("init_after_gt", d["T_gt"], d["T_init"])):
    if only and name != only:
        continue
    ctx.set_profiling(0)
    pairs.clear()
    core.match_pt2pt(ctx, gmap, cloud, warm, prm, None, pairs)
    ctx.set_profiling(4)
    pairs.clear()
    core.match_pt2pt(ctx, gmap, cloud, pose, prm, None, pairs)
    ms = ctx.stats()["ms_nn"]
    tiles, singles = core.timeline(ctx)
    ctx.set_profiling(0)
    n_w = (l.shape[0] + 63) // 64
    rec = tiles.reshape(-1)[:8 * n_w].reshape(n_w, 8)
    phases = rec[:, 2:8].astype(np.float64) / 100.0  # us: prologue, window, directory, staging, tests, emit
    start, end = (rec[:, 0] & M40).astype(np.int64), (rec[:, 1] & M40).astype(np.int64)
    info = (rec[:, 0] >> np.uint64(40)).astype(np.int64)
    nu, rounds, passes, flags = info & 255, (info >> 8) & 255, (info >> 16) & 15, (info >> 20) & 3
    dur = (end - start) / 100.0
    out = dict(scene=scene, pose=name, ms_nn=round(ms, 3), span_us=round((end.max() - start.min()) / 100.0, 1),
               dur_us={k: round(float(np.percentile(dur, p)), 1) for k, p in (("p50", 50), ("p90", 90), ("p99", 99), ("p99.9", 99.9), ("max", 100))},
               mean_dur=round(float(dur.mean()), 1))
    by = {}
    for lab, arr, edges in (("nu", nu, [0, 1, 8, 16, 32, 64, 128, 200, 256]), ("rounds", rounds, [0, 1, 2, 3, 5, 9, 17, 33, 256]),
                            ("passes", passes, [0, 1, 2, 3, 4, 16])):
        rows = []
        for a, b in zip(edges[:-1], edges[1:]):
            m = (arr >= a) & (arr < b)
            if m.any():
                rows.append((f"[{a},{b})", int(m.sum()), round(float(dur[m].mean()), 1), round(float(dur[m].sum() / 1e3), 1)))
        by[lab] = rows
    out["by(count, mean_us, total_ms)"] = by
    names = ("prologue", "window", "directory", "staging", "tests", "emit")
    out["phase_us_mean"] = {n: round(float(phases[:, i].mean()), 2) for i, n in enumerate(names)}
    light = (nu < 16) & (rounds <= 1) & (passes <= 1)
    out["phase_us_mean_light_waves"] = {n: round(float(phases[light, i].mean()), 2) for i, n in enumerate(names)}
    out["light_waves"] = [int(light.sum()), round(float(dur[light].mean()), 2)]
    heavy = dur >= np.percentile(dur, 99)
    out["phase_us_mean_slowest_1pct"] = {n: round(float(phases[heavy, i].mean()), 2) for i, n in enumerate(names)}
    out["flags(ovf,toobig) waves"] = [int((flags & 1).astype(bool).sum()), int((flags & 2).astype(bool).sum())]
    top = np.argsort(-dur)[:15]
    out["slowest(wave, us, nu, rounds, passes, flags)"] = [(int(i), round(float(dur[i]), 1), int(nu[i]), int(rounds[i]), int(passes[i]), int(flags[i])) for i in top]
    # start order: is the tail made of late starters or of long runners?
    t0 = start.min()
    late = np.argsort(-end)[:10]
    out["last_to_finish(wave, start_us, dur_us)"] = [(int(i), round(float((start[i] - t0) / 100.0), 1), round(float(dur[i]), 1)) for i in late]
    print(json.dumps(out), flush=True)
