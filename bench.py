#!/usr/bin/env python3
"""bench.py -- headline benchmark of the mp2p_icp hot path on MI355X.

Metric (BASELINE.json): ICP iterations/s (+ matched pairs/s) on "1 M-point local vs 10 M-point
global".  One STEP = one outer ICP iteration = Matcher_Points_DistanceThreshold (transform +
exact NN + threshold + unique-global filter + ordered compaction) followed by
Solver_GaussNewton (3 inner iterations, GemanMcClure 0.15, demos/icp-settings-kitti.yaml:32-34),
with the point layers, the NN index and the pairings resident in HBM.  The pose chain is the
real ICP chain (step s starts from the pose step s-1 produced, restarting from the perturbed
initial guess every 10 steps), and every step ends with the 96-byte pose read-back the
reference's outer loop needs for its termination test.

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by torch.distributed.run (one rank per GPU); the local layer is sharded
(weak scaling: every rank holds its own 1 M-point slice of an N x 1 M-point local layer), the
10 M-point global layer is replicated, and the exchange steps of mp2p_icp_amd/distributed.py
run over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CYCLE = 10  # steps per ICP run before the pose restarts from the initial guess


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(n_local, n_global, seed, rank, world):
    from mp2p_icp_amd import synthetic, se3
    cache = f"/tmp/mp2p_bench_{n_local}_{n_global}_{seed}_{rank}_{world}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        return dict(local=z["local"], glob=z["glob"], T_gt=z["T_gt"], T_init=z["T_init"])
    scene = synthetic.Scene(seed)
    n_rings, n_az = synthetic.rings_for(int(n_local * 1.25))
    sensor = (scene.length * 0.5, 0.0, 0.0)
    yaw = 0.05
    # rank r scans with its own noise stream: its slice of the N x 1M-point local layer
    loc = scene.scan(sensor, yaw, n_rings, n_az, seed + 1 + 1000 * rank)
    if loc.shape[0] > n_local:
        loc = loc[np.linspace(0, loc.shape[0] - 1, n_local).astype(np.int64)]
    glob = scene.sample_map(n_global, seed + 2)  # identical on every rank
    T_gt = se3.from_xyzypr(sensor[0], sensor[1], sensor[2], yaw, 0.0, 0.0)
    T_init = se3.compose(T_gt, se3.from_xyzypr(*synthetic.perturbation(seed + 3)))
    d = dict(local=np.ascontiguousarray(loc), glob=glob, T_gt=T_gt, T_init=T_init)
    try:
        np.savez(cache, **d)
    except Exception:
        pass
    return d


def cpu_baseline(d, threshold, gn_iters, kernel_param, sample, cores):
    """The oracle (CPU port of the reference's algorithm class: exact KD-tree + GN), timed on
    this host's cores on a bounded sample of the same workload."""
    import oracle as orc
    g, l = d["glob"], d["local"]
    t0 = time.time()
    tree = orc.KDTree(g[:, 0], g[:, 1], g[:, 2])
    t_build = time.time() - t0
    n_s = min(sample, l.shape[0])
    ls = l[np.linspace(0, l.shape[0] - 1, n_s).astype(np.int64)]
    best = None
    for pose in (d["T_init"], d["T_gt"]):
        t0 = time.time()
        pairs, _ = orc.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], ls[:, 0], ls[:, 1], ls[:, 2], pose,
                                   threshold, 0.0, tree=tree, threads=cores)
        t_match = time.time() - t0
        prm = orc.make_gn_params(gn_iters, kernel=orc.KERNEL_GEMANMCCLURE, kernelParam=kernel_param)
        t0 = time.time()
        orc.optimal_tf_gauss_newton(pairs, None, None, pose, prm, threads=cores)
        t_solve = time.time() - t0
        best = (t_match, t_solve, len(pairs)) if best is None else (
            best[0] + t_match, best[1] + t_solve, best[2] + len(pairs))
    scale = l.shape[0] / n_s
    t_iter = (best[0] + best[1]) / 2 * scale  # mean of the hard (initial) and easy (converged) pose
    return {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{n_s} of {l.shape[0]} local points (uniform subsample) vs the full "
                      f"{g.shape[0]}-point map; mean of initial-guess and converged pose; "
                      f"KD-tree build {t_build:.1f}s excluded (amortised per map)",
            "pairs_per_s": best[2] / 2 * scale / t_iter}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-local", type=int, default=1_000_000)
    ap.add_argument("--n-global", type=int, default=10_000_000)
    ap.add_argument("--threshold", type=float, default=2.0)
    ap.add_argument("--gn-iters", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--q", type=int, default=0, help="queries per wave (tuning)")
    ap.add_argument("--r0", type=float, default=0.0, help="initial radius in cells (tuning)")
    ap.add_argument("--grp", type=float, default=0.0, help="group radius factor (tuning)")
    ap.add_argument("--bricks", type=int, default=0, help="brick budget of the deferred-query kernel (0 = default)")
    ap.add_argument("--no-bitmap", action="store_true", help="build the map without occupancy bitmaps")
    ap.add_argument("--no-events", action="store_true", help="timed loop without hipEvents (overhead probe)")
    ap.add_argument("--cold", action="store_true", help="disable the warm start from the previous iteration")
    ap.add_argument("--tile-order", action="store_true", help="launch the tiles longest-first (by the previous iteration's durations) instead of in Morton order")
    ap.add_argument("--defer", type=float, default=0.0, help="defer radius in cells (tuning)")
    ap.add_argument("--budget", type=int, default=0, help="voxel budget per search box (tuning)")
    ap.add_argument("--cell", type=float, default=0.0, help="voxel edge [m] (0 = automatic)")
    ap.add_argument("--target-per-cell", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: mp2p_icp_amd has no CPU fallback")
    # test hook: MP2P_BENCH_SHARE_GPU=1 runs all ranks on GPU 0 over gloo, to exercise the N>1
    # code path on a one-GPU box (numbers are then meaningless; RCCL refuses two ranks per GPU)
    share_gpu = os.environ.get("MP2P_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core
    from mp2p_icp_amd.distributed import HipBackend, ShardedRegistration

    t0 = time.time()
    d = build_inputs(args.n_local, args.n_global, args.seed, rank, world)
    log(f"[bench r{rank}] inputs ready in {time.time() - t0:.1f}s: local {d['local'].shape}, "
        f"global {d['glob'].shape}")

    # one HIP stream shared with torch so that RCCL collectives are ordered with our kernels: a
    # dedicated torch stream made current (torch's default stream is the null stream, raw value 0)
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    ctx = amd.Context(local_rank, stream=stream)
    g, l = d["glob"], d["local"]
    t0 = time.time()
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2], cell_size=args.cell,
                          target_per_cell=args.target_per_cell, no_occupancy_bitmap=args.no_bitmap)
    info = gmap.info()
    t_index = time.time() - t0
    t0 = time.time()
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    t_cloud = time.time() - t0
    log(f"[bench r{rank}] index: cell {info['cell_size']:.3f} m, {info['n_levels']} levels, "
        f"{info['n_cells_level0']} voxels, {info['device_bytes'] / 1e6:.0f} MB, build "
        f"{info['build_ms']:.1f} ms (upload+build {t_index * 1e3:.0f} ms); cloud {t_cloud * 1e3:.0f} ms")

    n_l = l.shape[0]
    prm = _lib.Pt2PtParams(args.threshold, 0.0, 1, 0, 0, 0.20, rank * n_l, args.r0, args.q, args.grp, args.budget, args.defer, int(args.cold), args.bricks,
                           int(args.tile_order))
    gnp = _lib.GNParams()
    gnp.maxInnerLoopIterations = args.gn_iters
    gnp.minDelta, gnp.maxCost = 1e-7, 0.0
    gnp.kernel, gnp.kernelParam = _lib.KERNEL_GEMANMCCLURE, 0.15
    gnp.w_pt2pt = gnp.w_pt2pl = 1.0
    pairs = core.DevicePairs(ctx, n_l, 0)
    reg = ShardedRegistration(HipBackend(ctx, gmap, cloud, prm, gnp, pairs), dist)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    state = {"pose": d["T_init"].copy(), "s": 0}

    def one_step():
        if state["s"] % CYCLE == 0:
            state["pose"] = d["T_init"].copy()
        state["pose"], _ = reg.step(state["pose"])  # ends with the pose read-back (sync)
        state["s"] += 1

    for _ in range(args.warmup):
        one_step()
    # ---- timed region: exactly K steps between two barrier+synchronize brackets --------------
    # two hipEvents per step around the search kernels (the roofline kernel), read back lazily;
    # the per-kernel breakdown comes from the untimed replay below (a full set of events costs
    # about 5 % of the step)
    ctx.set_profiling(0 if args.no_events else 3)
    nn_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        nn_ms.append(ctx.stats()["ms_nn"])  # the step already ended with a stream sync (pose read-back)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- untimed replay of the same pose chain: per-kernel time, pairs, touched points --------
    state = {"pose": d["T_init"].copy(), "s": 0}
    for _ in range(args.warmup):
        one_step()
    touched, cand, passes, pair_counts, maxcand, maxpass = [], [], [], [], [], []
    nn_tile_ms, nn_single_ms, cp_ms, gn_ms, nn_ms_replay, nn_lane_ms = [], [], [], [], [], []
    lane_stats = []
    ctx.set_profiling(1)
    for _ in range(args.steps):
        one_step()
        st = ctx.stats()
        nn_ms_replay.append(st["ms_nn"])
        nn_tile_ms.append(st["ms_nn_tile"])
        nn_lane_ms.append(st["ms_nn_lane"])
        nn_single_ms.append(st["ms_nn_single"])
        cp_ms.append(st["ms_compact"])
        gn_ms.append(st["ms_gn"])
        pair_counts.append(pairs.counts()[0])
    ctx.set_profiling(0)
    log(f"[bench r{rank}] per-step kernel ms (replay; chain position = (warmup + i) % {CYCLE}): lane="
        f"{[round(v, 3) for v in nn_lane_ms]} tile="
        f"{[round(v, 3) for v in nn_tile_ms]} single={[round(v, 3) for v in nn_single_ms]} "
        f"gn={[round(v, 3) for v in gn_ms]}; search in the timed loop: {[round(v, 3) for v in nn_ms]}")
    state = {"pose": d["T_init"].copy(), "s": 0}
    ctx.set_profiling(2)
    for _ in range(min(CYCLE, args.warmup + args.steps)):
        if state["s"] % CYCLE == 0:
            state["pose"] = d["T_init"].copy()
        reg.match(state["pose"])
        st = ctx.stats()
        touched.append(st["nn_points_staged"])
        cand.append(st["nn_candidates_tested"])
        passes.append(st["nn_passes"] / max(1, st["nn_tiles"]))
        maxcand.append(st["nn_max_candidates_one_tile"])
        maxpass.append(st["nn_max_passes_one_tile"])
        lane_stats.append((st["nn_lane_searched"], st["nn_lane_pending"], st["nn_lane_skipped"],
                           st["nn_lane_candidates"], st["nn_lane_voxels"]))
        _e = amd.se3.log(amd.se3.inverse_compose(state["pose"], d["T_gt"]))
        log(f"[bench r{rank}] chain step {state['s']}: err=({np.linalg.norm(_e[:3]):.3f} m, "
            f"{np.degrees(np.linalg.norm(_e[3:])):.2f} deg) pairs={pairs.counts()[0]} "
            f"lane: searched={st['nn_lane_searched']} skipped={st['nn_lane_skipped']} pending={st['nn_lane_pending']} "
            f"cand/q={st['nn_lane_candidates'] / max(1, st['nn_lane_searched']):.1f} "
            f"vox/q={st['nn_lane_voxels'] / max(1, st['nn_lane_searched']):.1f}; "
            f"deferred={st['nn_single_queries']} single_cand/q="
            f"{st['nn_single_candidates'] / max(1, st['nn_single_queries']):.0f} "
            f"tile_cand/tile={st['nn_candidates_tested'] / max(1, st['nn_tiles']):.0f} "
            f"passes/tile={st['nn_passes'] / max(1, st['nn_tiles']):.2f}")
        state["pose"], _ = reg.solve(state["pose"])
        state["s"] += 1
    ctx.set_profiling(0)
    final_err = amd.se3.log(amd.se3.inverse_compose(state["pose"], d["T_gt"]))

    pairs_total = torch.tensor([float(np.sum(pair_counts))], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(pairs_total, op=dist.ReduceOp.SUM)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    # Weak scaling: every rank brings its own n_local-point slice of the local layer, so one step
    # registers world x n_local points jointly (one pose, one unique-global filter, one 6x6
    # system).  The unit of `value` is the metric's configuration -- one outer ICP iteration over
    # n_local local points -- hence the whole-job aggregate is world / t_step; the plain step rate
    # is reported next to it.
    iters_per_s = world * args.steps / elapsed
    nn_ms_avg = max(float(np.mean(nn_ms)), 1e-9)  # 0 only with --no-events (overhead probe)
    # algorithmic bytes of the search kernel per launch:
    #   12 B/query read + 12 B per distinct global point in a visited voxel + 8 B/query written
    alg_bytes = 12.0 * n_l + 12.0 * float(np.mean(touched)) + 8.0 * n_l
    achieved = alg_bytes / (nn_ms_avg * 1e-3) / 1e9
    out = {
        "metric": "icp_iterations_per_sec",
        "value": iters_per_s,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 search / f64 normal equations",
        "data": "synthetic (seeded KITTI-shape street scene: ray-cast scan vs surface-sampled map)",
        "config": {
            "workload": f"{n_l * world}-pt local ({world} x {n_l}) vs {g.shape[0]}-pt global, "
                        "Matcher_Points_DistanceThreshold + Solver_GaussNewton",
            "threshold_m": args.threshold, "thresholdAngularDeg": 0.0,
            "gn_inner_iterations": args.gn_iters, "robust_kernel": "GemanMcClure(0.15)",
            "unique_global_filter": True,
            "pose_chain": f"real ICP chain, restart from perturbed guess every {CYCLE} steps",
            "parallelism": f"local layer sharded x{world}, map replicated",
        },
        "steps_per_sec_wall": args.steps / elapsed,
        "unit_of_work": f"one outer ICP iteration (match + Gauss-Newton solve) over {n_l} local points vs "
                        f"{g.shape[0]} global points; a step does {world} of them jointly",
        "matched_pairs_per_sec": float(pairs_total.item()) / elapsed,
        "queries_per_sec": n_l * world * args.steps / elapsed,
        "pairs_per_step": float(pairs_total.item()) / args.steps,
        "kernel_ms": {"note": "nn_search: hipEvents in the timed loop; the others: same chain replayed "
                              "with an event around every stage",
                      "nn_search": nn_ms_avg, "nn_search_replay": float(np.mean(nn_ms_replay)),
                      "nn_lane_kernel": float(np.mean(nn_lane_ms)),
                      "nn_tile_kernel": float(np.mean(nn_tile_ms)),
                      "nn_single_kernel": float(np.mean(nn_single_ms)),
                      "compact": float(np.mean(cp_ms)),
                      "gn_solve_all_inner": float(np.mean(gn_ms))},
        "nn_stats": {"lane_kernel_searched_frac": float(np.mean([a[0] for a in lane_stats])) / n_l,
                     "lane_kernel_pending_frac": float(np.mean([a[1] for a in lane_stats])) / n_l,
                     "lane_kernel_finished_without_search_frac": float(np.mean([a[2] for a in lane_stats])) / n_l,
                     "avg_passes_per_tile": float(np.mean(passes)),
                     "candidates_tested_per_query": float(np.mean(cand)) / n_l,
                     "global_points_touched": float(np.mean(touched)),
                     "max_candidates_one_tile": int(np.max(maxcand)),
                     "max_passes_one_tile": int(np.max(maxpass)),
                     "voxel_m": info["cell_size"]},
        "index_build_ms": info["build_ms"],
        "final_pose_error": {"trans_m": float(np.linalg.norm(final_err[:3])),
                             "rot_rad": float(np.linalg.norm(final_err[3:]))},
        "roofline": {
            "bound": "hbm",
            "kernel": "nn_lane_kernel + nn_tile_kernel + nn_single_kernel (K1+K3: transform + exact NN search)",
            "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": nn_ms_avg,
            "traffic": None,
        },
    }
    # HBM traffic of the search kernels from the committed rocprofv3 PMC passes of this same
    # command (FETCH_SIZE x2 correction for gfx950 + WRITE_SIZE, MI355X_MICROARCH.md "HBM")
    try:
        import csv
        t = 0.0
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_bench_hbm_pmc.csv"))):
            if r["kernel"].strip('"').startswith(("void mp2p::nn_tile_kernel<32, false>",
                                                   "void mp2p::nn_single_kernel<false>")):
                t += float(r["fetch_bytes_avg_corrected_x2"]) + float(r["write_bytes_avg"])
        if t > 0 and args.n_local == 1_000_000 and args.n_global == 10_000_000 and world == 1:
            out["roofline"]["traffic"] = t
            out["roofline"]["traffic_source"] = "profiles/r01_bench_hbm_pmc.csv (rocprofv3 --pmc)"
    except Exception:
        pass
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
        cores = os.cpu_count() or 1
        t0 = time.time()
        out["cpu_baseline"] = cpu_baseline(d, args.threshold, args.gn_iters, 0.15,
                                           args.cpu_sample, cores)
        out["cpu_baseline"]["wall_s"] = time.time() - t0
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
