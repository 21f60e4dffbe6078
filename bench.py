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

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene a|b] [--config c2|c3|c5]
(--config lines: the other BASELINE configurations, each with its own roofline and cpu_baseline; c3 also runs sharded
 under --gpus N: mp2p_hip_step_sharded_pt2pl)

The default line (no --config) carries, next to the contract's fields:
  roofline       the search kernels (K1+K3) against the HBM roofline
  cpu_baseline   the oracle (CPU port of the reference's algorithm class) on all host threads, plus its
                 single-thread figure
  step_ms        min / median / max of the K timed steps; `stability` = 200 more chained steps
  host_boundary  the same step through HOST containers, i.e. through the reference-side adapter's code
                 (adapter/mp2p_hip_host.hpp): fresh MatchState, pairs into a host vector, marks, solver
  scene_a        the same metric on round 1's scene (scan vs a map sampled uniformly on the surfaces).  `value`,
                 `roofline` and `cpu_baseline` are on the scene SURVEY.md 8d specifies (--scene b: the map is the
                 voxel-thinned union of consecutive scans)
  converging     a third pose chain on the headline scene that reaches millimetre steps (point-to-plane matcher +
                 Gauss-Newton): the regime a converged registration spends most iterations in
N > 1 is launched by torch.distributed.run (one rank per GPU); the local layer is sharded
(weak scaling: every rank holds its own 1 M-point slice of an N x 1 M-point local layer; the block
`strong_scaling` times ONE 1 M-point scan split N ways), the 10 M-point global layer is replicated,
and the exchange steps of mp2p_icp_amd/distributed.py run over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CYCLE = 10  # steps per ICP run before the pose restarts from the initial guess


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------
def build_inputs(n_local, n_global, seed, rank, world, scene="a"):
    """scene a: ray-cast scan vs a map sampled uniformly on the surfaces (round 1's scene);
    scene b: SURVEY.md 8d -- the map is the voxel-thinned union of consecutive scans"""
    from mp2p_icp_amd import synthetic, se3
    if scene == "b":
        return synthetic.make_scan_union_pair(n_local, n_global, seed + 100 * rank,
                                              map_scan_points=max(n_local, 120_000))
    cache = f"/tmp/mp2p_bench_{n_local}_{n_global}_{seed}_{rank}_{world}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        return dict(local=z["local"], glob=z["glob"], T_gt=z["T_gt"], T_init=z["T_init"])
    sc = synthetic.Scene(seed)
    n_rings, n_az = synthetic.rings_for(int(n_local * 1.25))
    sensor = (sc.length * 0.5, 0.0, 0.0)
    yaw = 0.05
    # rank r scans with its own noise stream: its slice of the N x 1M-point local layer
    loc = sc.scan(sensor, yaw, n_rings, n_az, seed + 1 + 1000 * rank)
    if loc.shape[0] > n_local:
        loc = loc[np.linspace(0, loc.shape[0] - 1, n_local).astype(np.int64)]
    glob = sc.sample_map(n_global, seed + 2)  # identical on every rank
    T_gt = se3.from_xyzypr(sensor[0], sensor[1], sensor[2], yaw, 0.0, 0.0)
    T_init = se3.compose(T_gt, se3.from_xyzypr(*synthetic.perturbation(seed + 3)))
    d = dict(local=np.ascontiguousarray(loc), glob=glob, T_gt=T_gt, T_init=T_init)
    try:
        np.savez(cache, **d)
    except Exception:
        pass
    return d


def physical_cores():
    """distinct (socket, core) pairs of /proc/cpuinfo (SMT siblings counted once); os.cpu_count() when it cannot be read"""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(d, threshold, gn_iters, kernel_param, sample, cores):
    """The oracle (CPU port of the reference's algorithm class: exact KD-tree + GN), timed on this
    host's cores on a bounded sample of the same workload; plus the single-thread figure (the
    reference's deterministic, sequential semantics) on a smaller sample."""
    import oracle as orc
    g, l = d["glob"], d["local"]
    t0 = time.time()
    tree = orc.KDTree(g[:, 0], g[:, 1], g[:, 2])
    t_build = time.time() - t0

    gate = []  # (pose, oracle pair list, oracle Gauss-Newton pose) of the whole layer: what the parity gate compares with

    def run(n_s, threads, keep=False):
        ls = l[np.linspace(0, l.shape[0] - 1, n_s).astype(np.int64)]
        tm = ts = npairs = 0.0
        for pose in (d["T_init"], d["T_gt"]):
            t0 = time.time()
            pairs, _ = orc.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], ls[:, 0], ls[:, 1], ls[:, 2], pose,
                                       threshold, 0.0, tree=tree, threads=threads)
            tm += time.time() - t0
            prm = orc.make_gn_params(gn_iters, kernel=orc.KERNEL_GEMANMCCLURE, kernelParam=kernel_param)
            t0 = time.time()
            T_new, *_ = orc.optimal_tf_gauss_newton(pairs, None, None, pose, prm, threads=threads)
            ts += time.time() - t0
            npairs += len(pairs)
            if keep and n_s == l.shape[0]:
                gate.append((np.array(pose), pairs, np.array(T_new)))
        scale = l.shape[0] / n_s
        t_iter = (tm + ts) / 2 * scale  # mean of the hard (initial) and easy (converged) pose
        return 1.0 / t_iter, npairs / 2 * scale / t_iter, tm / 2 * scale, ts / 2 * scale

    n_s = min(sample, l.shape[0])
    run(min(n_s, 50_000), cores)  # (untimed: starts the thread pool, touches the tree)
    v, pps, tm, ts = run(n_s, cores, keep=True)
    n_1 = min(max(2000, sample // 10), l.shape[0])
    v1, pps1, tm1, ts1 = run(n_1, 0)  # threads=0: the sequential loop
    quota = None  # the container's CPU bandwidth limit (cgroup v2 cpu.max: "<quota> <period>" or "max"): what `cores` threads can really use
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    phys = physical_cores()
    return {"value": v, "unit": "iterations/s", "cores": cores, "kind": "port",
            "_gate": gate, "cpu_quota_cores": quota,
            # what the host's cores could give at best: the measured 1-thread figure x the PHYSICAL cores (no quota, perfect scaling)
            "physical_cores": phys, "ideal_scaling_of_single_thread": {"value": v1 * phys, "unit": "iterations/s"},
            "cpu_quota_note": "threads = os.cpu_count(); the cgroup of this container caps their total CPU time at cpu_quota_cores "
                              "cores' worth (cpu.max), which bounds multi_thread_over_single_thread whatever the port does",

            "multi_thread_over_single_thread": v / v1,
            "sample": f"{n_s} of {l.shape[0]} local points (uniform subsample) vs the full "
                      f"{g.shape[0]}-point map; mean of initial-guess and converged pose; "
                      f"KD-tree build {t_build:.1f}s excluded (amortised per map)",
            "pairs_per_s": pps, "matcher_s_per_iteration": tm, "solver_s_per_iteration": ts,
            "single_thread": {"value": v1, "unit": "iterations/s", "cores": 1, "pairs_per_s": pps1,
                              "sample": f"{n_1} of {l.shape[0]} local points, sequential loop",
                              "matcher_s_per_iteration": tm1, "solver_s_per_iteration": ts1}}


# ---------------------------------------------------------------------------------------------------
class Rig:
    """the device-resident pipeline of one scene on one rank"""

    def __init__(self, args, d, rank, world, dist, local_rank, stream, n_offset):
        import mp2p_icp_amd as amd
        from mp2p_icp_amd import _lib, core
        from mp2p_icp_amd.distributed import HipBackend, ShardedRegistration
        self.amd, self.d = amd, d
        self.ctx = amd.Context(local_rank, stream=stream)
        g, l = d["glob"], d["local"]
        t0 = time.time()
        self.gmap = core.GlobalMap(self.ctx, g[:, 0], g[:, 1], g[:, 2], cell_size=args.cell,
                                   target_per_cell=args.target_per_cell, no_occupancy_bitmap=args.no_bitmap)
        self.info = self.gmap.info()
        self.t_index = time.time() - t0
        self.cloud = core.LocalCloud(self.ctx, l[:, 0], l[:, 1], l[:, 2])
        self.n_l = l.shape[0]
        self.prm = _lib.Pt2PtParams(args.threshold, 0.0, 1, 0, 0, 0.20, n_offset, args.r0, args.q, args.grp,
                                    args.budget, args.defer, int(args.cold), args.bricks, 0)
        gnp = _lib.GNParams()
        gnp.maxInnerLoopIterations = args.gn_iters
        gnp.minDelta, gnp.maxCost = 1e-7, 0.0
        gnp.kernel, gnp.kernelParam = _lib.KERNEL_GEMANMCCLURE, 0.15
        gnp.w_pt2pt = gnp.w_pt2pl = 1.0
        self.gnp = gnp
        self.pairs = core.DevicePairs(self.ctx, self.n_l, 0)
        self.reg = ShardedRegistration(HipBackend(self.ctx, self.gmap, self.cloud, self.prm, gnp, self.pairs), dist)
        self.state = {"pose": d["T_init"].copy(), "s": 0}

    def restart(self):
        self.state = {"pose": self.d["T_init"].copy(), "s": 0}

    def one_step(self):
        st = self.state
        if st["s"] % CYCLE == 0:
            st["pose"] = self.d["T_init"].copy()
        st["pose"], _ = self.reg.step(st["pose"])  # ends with the pose read-back (sync)
        st["s"] += 1


def timed_chain(rig, steps, warmup, barrier, events=True):
    """W untimed steps, then exactly K steps between two barrier+synchronize brackets"""
    rig.restart()
    for _ in range(warmup):
        rig.one_step()
    rig.ctx.set_profiling(3 if events else 0)
    nn_ms, step_s = [], []
    import gc
    gc.collect()
    gc.disable()  # as timeit does: a collection inside a 0.4 ms step is a 1 ms outlier
    barrier()
    t0 = time.perf_counter()
    tp = t0
    for _ in range(steps):
        rig.one_step()
        if events:
            nn_ms.append(rig.ctx.stats()["ms_nn"])  # the step already ended with a stream sync (pose read-back)
        tn = time.perf_counter()
        step_s.append(tn - tp)
        tp = tn
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    rig.ctx.set_profiling(0)
    return elapsed, nn_ms, step_s


def replay(rig, steps, warmup):
    """untimed replay of the same pose chain: per-kernel time and pair counts"""
    rig.restart()
    for _ in range(warmup):
        rig.one_step()
    out = {k: [] for k in ("lane", "tile", "single", "compact", "gn", "nn", "pairs")}
    rig.ctx.set_profiling(1)
    for _ in range(steps):
        rig.one_step()
        st = rig.ctx.stats()
        out["nn"].append(st["ms_nn"]), out["lane"].append(st["ms_nn_lane"]), out["tile"].append(st["ms_nn_tile"])
        out["single"].append(st["ms_nn_single"]), out["compact"].append(st["ms_compact"]), out["gn"].append(st["ms_gn"])
        out["pairs"].append(rig.pairs.counts()[0])
    rig.ctx.set_profiling(0)
    return out


def instrumented(rig, n_steps, rank, tag):
    """the device counters of one chain cycle: distinct global points touched, candidates, passes"""
    amd, d = rig.amd, rig.d
    rig.restart()
    st_ = rig.state
    rig.ctx.set_profiling(2)
    rows = []
    for _ in range(n_steps):
        if st_["s"] % CYCLE == 0:
            st_["pose"] = d["T_init"].copy()
        rig.reg.match(st_["pose"])
        st = rig.ctx.stats()
        _e = amd.se3.log(amd.se3.inverse_compose(st_["pose"], d["T_gt"]))
        rows.append(dict(touched=st["nn_points_staged"], cand=st["nn_candidates_tested"],
                         passes=st["nn_passes"] / max(1, st["nn_tiles"]), maxcand=st["nn_max_candidates_one_tile"],
                         maxpass=st["nn_max_passes_one_tile"], pending=st["nn_lane_pending"],
                         skipped=st["nn_lane_skipped"], deferred=st["nn_single_queries"],
                         err_t=float(np.linalg.norm(_e[:3])), err_r=float(np.linalg.norm(_e[3:]))))
        log(f"[bench r{rank}] {tag} chain step {st_['s']}: err=({rows[-1]['err_t']:.3f} m, "
            f"{np.degrees(rows[-1]['err_r']):.2f} deg) pairs={rig.pairs.counts()[0]} "
            f"pending={st['nn_lane_pending']} finished-without-search={st['nn_lane_skipped']} "
            f"deferred={st['nn_single_queries']} single_cand/q="
            f"{st['nn_single_candidates'] / max(1, st['nn_single_queries']):.0f} "
            f"tile_cand/tile={st['nn_candidates_tested'] / max(1, st['nn_tiles']):.0f} "
            f"passes/tile={st['nn_passes'] / max(1, st['nn_tiles']):.2f}")
        st_["pose"], _ = rig.reg.solve(st_["pose"])
        st_["s"] += 1
    rig.ctx.set_profiling(0)
    final_err = amd.se3.log(amd.se3.inverse_compose(st_["pose"], d["T_gt"]))
    return rows, final_err


def converging_block(rig, args):
    """A chain that CONVERGES: point-to-plane matcher (knn 5, 0.4 m) + Gauss-Newton from the perturbed
    guess until the step is below a millimetre, then the headline pair (pt2pt matcher + GN) for 10 more steps from
    there -- the millimetre-step regime a converged registration spends most of its iterations in."""
    import torch
    from mp2p_icp_amd import _lib, core
    amd, d, ctx = rig.amd, rig.d, rig.ctx
    pl = _lib.Pt2PlParams()
    pl.distanceThreshold, pl.searchRadius, pl.knn, pl.minimumPlanePoints, pl.planeEigenThreshold = 0.4, 0.4, 5, 5, 0.05
    pl.bounding_box_intersection_check_epsilon = 0.20
    pairs_pl = core.DevicePairs(ctx, 0, rig.n_l)
    pose, errs, ms_pl = d["T_init"].copy(), [], []

    def err(p):
        e = amd.se3.log(amd.se3.inverse_compose(p, d["T_gt"]))
        return float(np.linalg.norm(e[:3])), float(np.linalg.norm(e[3:]))

    for it in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pairs_pl.clear()
        core.match_pt2pl(ctx, rig.gmap, rig.cloud, pose, pl, None, pairs_pl)
        new = np.array(core.gn_solve(ctx, pairs_pl, pose, rig.gnp).pose)
        ms_pl.append((time.perf_counter() - t0) * 1e3)
        step = amd.se3.log(amd.se3.inverse_compose(pose, new))
        pose = new
        errs.append(err(pose))
        if np.linalg.norm(step[:3]) < 1e-3 and np.linalg.norm(step[3:]) < 1e-4:
            break
    n_pl = len(errs)
    # the headline pair from the converged pose: 3 untimed steps (the warm start settles), 10 timed
    rig.state = {"pose": pose.copy(), "s": 1}  # s != 0 (mod CYCLE): no restart
    for _ in range(3):
        rig.one_step(); rig.state["s"] = 1
    rig.ctx.set_profiling(3)
    ts, nn = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        rig.one_step(); rig.state["s"] = 1
        ts.append((time.perf_counter() - t0) * 1e3)
        nn.append(rig.ctx.stats()["ms_nn"])
    rig.ctx.set_profiling(2)
    rig.reg.match(rig.state["pose"])
    st = rig.ctx.stats()
    rig.ctx.set_profiling(0)
    e_end = err(rig.state["pose"])
    return {"plane_chain": {"iterations": n_pl, "trans_err_m": [round(e[0], 5) for e in errs], "rot_err_rad": [round(e[1], 6) for e in errs],
                            "ms_per_iteration_median": float(np.median(ms_pl[1:])) if n_pl > 1 else float(ms_pl[0])},
            "final_pose_error": {"trans_m": errs[-1][0], "rot_rad": errs[-1][1]},
            "pt2pt_at_the_converged_pose": {"ms_per_step_median": float(np.median(ts)), "iterations_per_s": 1e3 / float(np.median(ts)),
                                            "nn_search_ms": float(np.median(nn)), "pose_error_after": {"trans_m": e_end[0], "rot_rad": e_end[1]},
                                            "finished_without_search_frac": st["nn_lane_skipped"] / rig.n_l,
                                            "deferred_to_one_query_kernel_frac": st["nn_single_queries"] / rig.n_l,
                                            "staged_points_per_tile": st["nn_candidates_tested"] / max(1, st["nn_tiles"])},
            "note": "point-to-plane ICP converges where the point-to-point chain of `value` slides along the street; at the "
                    "converged pose the point matcher's steps are sub-millimetre, its tile kernel tracks a lower bound of every "
                    "query's SECOND-nearest distance (round 4: out of the matrix-pipe prefilter's values minus their proven error "
                    "bound) and a query whose previous neighbour is provably still the nearest skips its search"}


def parity_gate(rig, gate):
    """no timing is reported unless the GPU path reproduces the oracle on THIS workload.  The headline
    layer (all queries, scene and size of `value`) is matched on the GPU at the initial guess and at the ground-truth pose --
    cold, then once more from the warm start the first call left -- and solved; the pair lists must equal the oracle's
    (localIdx, globalIdx, bits of errSq) and the Gauss-Newton pose must agree to 1e-5 m / 1e-5 rad."""
    import oracle as orc
    from mp2p_icp_amd import core
    out = {"poses": len(gate), "pairs_equal": True, "pairs": [], "pose_err": {"trans_m": 0.0, "rot_rad": 0.0}, "matcher_calls": 0}
    for pose, want, T_want in gate:
        for rep in range(2):  # cold (another pose's warm start: far away), then warm from itself
            rig.pairs.clear()
            core.match_pt2pt(rig.ctx, rig.gmap, rig.cloud, pose, rig.prm, None, rig.pairs)
            got = rig.pairs.download_pt2pt()
            out["matcher_calls"] += 1
            same = (len(got) == len(want) and np.array_equal(got["localIdx"], want["localIdx"]) and
                    np.array_equal(got["globalIdx"], want["globalIdx"]) and
                    np.array_equal(got["errorSquareAfterTransformation"].view(np.uint32), want["errSq"].view(np.uint32)))
            out["pairs_equal"] = bool(out["pairs_equal"] and same)
        out["pairs"].append(int(len(want)))
        T_got = np.array(core.gn_solve(rig.ctx, rig.pairs, pose, rig.gnp).pose)
        et, er = orc.pose_err_split(T_got, T_want)
        out["pose_err"]["trans_m"] = max(out["pose_err"]["trans_m"], float(et))
        out["pose_err"]["rot_rad"] = max(out["pose_err"]["rot_rad"], float(er))
    out["passed"] = bool(out["pairs_equal"] and out["pose_err"]["trans_m"] <= 1e-5 and out["pose_err"]["rot_rad"] <= 1e-5)
    return out


def stats_ms(xs):
    a = np.asarray(xs, dtype=np.float64) * 1e3
    return {"min": float(a.min()), "median": float(np.median(a)), "max": float(a.max())}


def csrc_sha16():
    """fingerprint of the kernel sources (what a committed counter capture under profiles/ was taken with)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mp2p_icp_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "include", "mp2p_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PROFILE_ROUND = "r06"


def roofline_block(n_l, touched_mean, nn_ms_avg, tag):
    # algorithmic bytes of the search kernels per launch:
    #   12 B/query read + 12 B per distinct global point in a visited voxel + 8 B/query written
    alg_bytes = 12.0 * n_l + 12.0 * touched_mean + 8.0 * n_l
    achieved = alg_bytes / (max(nn_ms_avg, 1e-9) * 1e-3) / 1e9
    out = {"bound": "hbm",
           "kernel": "nn_lane_kernel + nn_seltile_kernel + nn_single_kernel (K1+K3: transform + exact NN search)",
           "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
           "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": nn_ms_avg,
           "traffic": None, "traffic_from_profiles": None}
    # HBM traffic of the search kernels from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 correction for gfx950 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"): a cited constant
    # from profiles/, not a measurement of this run
    try:
        import csv
        f = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_bench_{tag}_hbm_pmc.csv")
        # the capture names the sources it was taken with: counters of OTHER kernels than the ones timed here are refused,
        # loudly, instead of being quoted next to this run's time
        meta = json.load(open(f.replace(".csv", ".meta.json")))
        if meta.get("csrc_sha16") != csrc_sha16():
            out["traffic_note"] = (f"STALE: {os.path.relpath(f, ROOT)} was captured with kernel sources {meta.get('csrc_sha16')}, "
                                   f"this build is {csrc_sha16()}: re-run tools/gpu_prof.sh + tools/summarize_prof.py")
            log("[bench] WARNING " + out["traffic_note"])
            return out
        t = 0.0
        for r in csv.DictReader(open(f)):
            # the three search launches of the timed path (INSTR = false variants; the instrumented ones
            # only run in the counting replay)
            if re.search(r"mp2p::nn_(lane_kernel<false>|tile_kernel<\d+, false|seltile_kernel<false|single_kernel<false)", r["kernel"]):
                t += float(r["fetch_bytes_avg_corrected_x2"]) + float(r["write_bytes_avg"])
        if t > 0:
            out["traffic_from_profiles"] = t
            out["traffic_source"] = os.path.relpath(f, ROOT) + " (rocprofv3 --pmc; committed, not measured in this run)"
    except Exception:
        pass
    return out


def host_boundary(args, d, dev_ms):
    """one ICP iteration through HOST containers (adapter/mp2p_hip_host.hpp via hostpath_capi.cpp):
    fresh MatchState, pairs into a host vector, marks from the pair list, solver handed the host list"""
    from mp2p_icp_amd import _lib, hostpath
    s = hostpath.Session(d["glob"], d["local"])
    prm = _lib.Pt2PtParams()
    prm.threshold, prm.thresholdAngularDeg, prm.pairingsPerPoint = args.threshold, 0.0, 1
    prm.bounding_box_intersection_check_epsilon = 0.20
    gnp = _lib.GNParams()
    gnp.maxInnerLoopIterations, gnp.minDelta, gnp.maxCost = args.gn_iters, 1e-7, 0.0
    gnp.kernel, gnp.kernelParam, gnp.w_pt2pt, gnp.w_pt2pl = _lib.KERNEL_GEMANMCCLURE, 0.15, 1.0, 1.0
    c0 = hostpath.counters()
    ts, stages, npairs = [], [], []
    pose = d["T_init"].copy()
    n = args.warmup + args.steps
    for it in range(n):
        k = it % CYCLE
        if k == 0:
            pose = d["T_init"].copy()
        t0 = time.perf_counter()
        s.begin_iteration()
        # chain position 0 = ICP iteration 0 of a new ICP::align: both layers are verified in full there
        npairs.append(s.match_pt2pt(pose, prm, icp_iteration=k))
        pose, _ = s.solve_gn(pose, gnp)
        ts.append(time.perf_counter() - t0)
        stages.append(hostpath.stage_ms())
    c1 = hostpath.counters()
    keep = [i for i in range(args.warmup, n) if i % CYCLE != 0]
    first = [i for i in range(args.warmup, n) if i % CYCLE == 0]
    ms = float(np.mean([ts[i] for i in keep])) * 1e3
    out = {"ms_per_step": ms, "iterations_per_s": 1e3 / ms,
           "vs_device_resident_step": ms / dev_ms,
           "ms_per_step_at_icp_iteration_0": float(np.mean([ts[i] for i in first])) * 1e3 if first else None,
           # ICP iteration 0 of a new align: the very first one uploads both layers and builds the index; a later one
           # finds them cached and re-verifies every 61st point (state_in), then searches from a stale warm start
           "icp_iteration_0": {"first_seen_ms": ts[0] * 1e3, "first_seen_stage_ms": stages[0],
                               "re_seen_ms": float(np.mean([ts[i] for i in first])) * 1e3 if first else None,
                               "re_seen_stage_ms": ({k: float(np.mean([stages[i][k] for i in first])) for k in stages[0]} if first else None),
                               "layer_cache": hostpath.cache()},
           "stage_ms": {k: float(np.mean([stages[i][k] for i in keep])) for k in stages[0]},
           "pairs_per_step": float(np.mean([npairs[i] for i in keep])),
           "transfers": {k: c1[k] - c0[k] for k in c0},
           "note": "through adapter/mp2p_hip_host.hpp (the plugin's MRPT-free host layer): packed MatchState "
                   "bit-fields in (none uploaded: the fields are clear), 36 B per emitted pair out, marks set "
                   "from the index arrays while the records are on the link, the solver recognising the device-resident "
                   "list by its fingerprint (length, last record, every 64th record); "
                   "ms_per_step excludes chain position 0, where both layers are fingerprinted in full "
                   "(120 MB) like at ICP iteration 0 of any new ICP::align"}
    s.close()
    return out


def cpu_baseline_config(d, which, cores):
    """cpu_baseline of a --config line: the oracle's matcher(s) + solver of that configuration on this host's cores (and
    the sequential loop on a tenth of the sample), mean of the initial-guess and the converged pose, on a bounded sample of
    the local layer against the full map; the KD-tree build is excluded (amortised per map)."""
    import oracle as orc
    g, l = d["glob"], d["local"]
    t0 = time.time()
    tree = orc.KDTree(g[:, 0], g[:, 1], g[:, 2])
    t_build = time.time() - t0
    gxyz = (g[:, 0], g[:, 1], g[:, 2])

    def run(n_s, threads):
        ls = l[np.linspace(0, l.shape[0] - 1, n_s).astype(np.int64)]
        lxyz = (ls[:, 0], ls[:, 1], ls[:, 2])
        tm = ts = 0.0
        for pose in (d["T_init"], d["T_gt"]):
            t0 = time.time()
            pt = pl = None
            if which == "c2":
                pt, _ = orc.match_pt2pt(*gxyz, *lxyz, pose, 2.0, 0.0, tree=tree, threads=threads)
            elif which == "c3":
                pl, _, _ = orc.match_pt2pl(*gxyz, *lxyz, pose, 0.4, 0.4, 5, 5, 0.05, tree=tree, threads=threads)
            else:  # c5: the plane matcher, then the point matcher on what it left (sequential semantics: one thread)
                lt, gt = np.zeros(ls.shape[0], np.uint8), np.zeros(g.shape[0], np.uint8)
                pl, _, _ = orc.match_pt2pl(*gxyz, *lxyz, pose, 0.25, 0.4, 5, 5, 0.05, tree=tree, local_taken=lt)
                pt, _ = orc.match_pt2pt(*gxyz, *lxyz, pose, 1.0, 0.0, tree=tree, local_taken=lt, global_taken=gt)
            tm += time.time() - t0
            t0 = time.time()
            if which == "c2":
                orc.optimal_tf_horn_wp(pt, None)
            else:
                prm = orc.make_gn_params(3, kernel=orc.KERNEL_CAUCHY if which == "c5" else orc.KERNEL_GEMANMCCLURE, kernelParam=0.15)
                orc.optimal_tf_gauss_newton(pt, pl, None, pose, prm, threads=threads)
            ts += time.time() - t0
        scale = l.shape[0] / n_s
        return 1.0 / ((tm + ts) / 2 * scale), tm / 2 * scale, ts / 2 * scale

    if which == "c5":  # the coupled matchers only exist as the sequential loop in the oracle
        n_1 = min(20_000, l.shape[0])
        v1, tm1, ts1 = run(n_1, 0)
        return {"value": v1, "unit": "iterations/s", "cores": 1, "kind": "port",
                "sample": f"{n_1} of {l.shape[0]} local points (uniform subsample) vs the full {g.shape[0]}-point map, sequential loop; "
                          f"mean of initial-guess and converged pose; KD-tree build {t_build:.1f}s excluded",
                "matcher_s_per_iteration": tm1, "solver_s_per_iteration": ts1}
    n_s = min(200_000, l.shape[0])
    v, tm, ts = run(n_s, cores)
    n_1 = min(max(2000, n_s // 10), l.shape[0])
    v1, tm1, ts1 = run(n_1, 0)
    return {"value": v, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{n_s} of {l.shape[0]} local points vs the full {g.shape[0]}-point map; mean of initial-guess and "
                      f"converged pose; KD-tree build {t_build:.1f}s excluded (amortised per map)",
            "matcher_s_per_iteration": tm, "solver_s_per_iteration": ts,
            "single_thread": {"value": v1, "unit": "iterations/s", "cores": 1, "sample": f"{n_1} of {l.shape[0]} local points, sequential loop",
                              "matcher_s_per_iteration": tm1, "solver_s_per_iteration": ts1}}


# ---------------------------------------------------------------------------------------------------
def bench_config(args, which, local_rank, stream, rank=0, world=1, dist=None):
    """--config c2|c3|c5: the other BASELINE configs as bench lines of their own.  c3 also runs sharded (--gpus N under
    torchrun): every rank holds one scan-sized shard of an N-scan local layer (weak scaling) and the step is
    mp2p_hip_step_sharded_pt2pl -- bounding-box all-reduce + 48 sums per inner iteration over RCCL, inside libmp2p_hip."""
    import mp2p_icp_amd as amd
    from mp2p_icp_amd import _lib, core, synthetic
    ctx = amd.Context(local_rank, stream=stream)
    if which == "c3" and (world > 1 or dist is not None):
        # (--config-scale: a code-path run of the tests at a fraction of the size -- 8 ranks building a 10 M-point scene each on a
        #  one-GPU box's few host cores take minutes; a scaled line says so in its workload and is no measurement of the configuration)
        sc = float(getattr(args, "config_scale", 1.0) or 1.0)
        d = synthetic.make_scan_union_pair(int(120_000 * sc) * world, int(10_000_000 * sc), 3001, map_scan_points=int(1_000_000 * sc))
        n_shard = d["local"].shape[0] // world
        d["local"] = np.ascontiguousarray(d["local"][rank * n_shard:(rank + 1) * n_shard])
        label = (f"{world} x KITTI-shape scan shard (~{n_shard} pts each) vs {10 * sc:g} M-pt map (replicated), Matcher_Point2Plane "
                 "(knn 5, r 0.4) + Solver_GaussNewton, sharded step" + ("" if sc == 1.0 else f" -- SCALED x{sc:g}: a test run, not the configuration"))
    elif which == "c2":
        d = synthetic.make_scan_union_pair(120_000, 2_000_000, 2001, map_scan_points=120_000)
        label = "KITTI-shape scan (~120 k pts) vs 2 M-pt map, Matcher_Points_DistanceThreshold + Solver_Horn"
    elif which == "c3":
        d = synthetic.make_scan_union_pair(120_000, 10_000_000, 3001, map_scan_points=1_000_000)
        label = "KITTI-shape scan (~120 k pts) vs 10 M-pt map, Matcher_Point2Plane (knn 5, r 0.4) + Solver_GaussNewton"
    else:
        d = synthetic.make_scan_union_pair(5_000_000, 5_000_000, 5001, map_scan_points=1_000_000, outlier_frac=0.30)
        label = ("5 M-pt scan (30 % uniform outliers) vs 5 M-pt map, Matcher_Point2Plane + "
                 "Matcher_Points_DistanceThreshold into one Solver_GaussNewton (Cauchy 0.15)")
    g, l = d["glob"], d["local"]
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    n_l = l.shape[0]
    pairs = core.DevicePairs(ctx, n_l, n_l if which != "c2" else 0)
    ms_dev = core.DeviceMatchState(ctx, g.shape[0], n_l) if which == "c5" else None
    pt = _lib.Pt2PtParams()
    pt.threshold, pt.thresholdAngularDeg, pt.pairingsPerPoint = (2.0 if which == "c2" else 1.0), 0.0, 1
    pt.bounding_box_intersection_check_epsilon = 0.20
    pl = _lib.Pt2PlParams()
    pl.distanceThreshold = 0.4 if which == "c3" else 0.25
    pl.searchRadius, pl.knn, pl.minimumPlanePoints, pl.planeEigenThreshold = 0.4, 5, 5, 0.05
    pl.bounding_box_intersection_check_epsilon = 0.20
    gnp = _lib.GNParams()
    gnp.maxInnerLoopIterations, gnp.minDelta, gnp.maxCost = 3, 1e-7, 0.0
    gnp.kernel = _lib.KERNEL_CAUCHY if which == "c5" else _lib.KERNEL_GEMANMCCLURE
    gnp.kernelParam, gnp.w_pt2pt, gnp.w_pt2pl = 0.15, 1.0, 1.0

    sharded = None
    if which == "c3" and (world > 1 or dist is not None):
        from mp2p_icp_amd.distributed import HipPlaneBackend, ShardedRegistration
        sharded = ShardedRegistration(HipPlaneBackend(ctx, gmap, cloud, pl, gnp, pairs, local_index_offset=rank * n_l), dist)
        if dist is not None and world == 1:
            sharded.b.init_native_comm(dist)  # MP2P_BENCH_FORCE_DIST: a one-rank RCCL communicator
        if world > 1 and not getattr(sharded.b, "native", False):
            raise SystemExit("--config c3 --gpus N needs the communicator inside libmp2p_hip (RCCL, or the hooks over gloo of the test boxes)")

    def step(pose, tap=None):
        """one ICP iteration; tap(matcher name) is called behind every matcher call (the replay reads its events there)"""
        if sharded is not None:
            return np.asarray(sharded.step(pose)[0])
        pairs.clear()
        if which == "c2":
            core.match_pt2pt(ctx, gmap, cloud, pose, pt, None, pairs)
            tap and tap("pt2pt")
            T, ok = core.horn_solve(ctx, pairs)
            return np.asarray(T)
        if which == "c3":
            core.match_pt2pl(ctx, gmap, cloud, pose, pl, None, pairs)
            tap and tap("pt2pl")
        else:
            ms_dev.reset()
            core.match_pt2pl(ctx, gmap, cloud, pose, pl, ms_dev, pairs)
            tap and tap("pt2pl")
            core.match_pt2pt(ctx, gmap, cloud, pose, pt, ms_dev, pairs)
            tap and tap("pt2pt")
        return np.array(core.gn_solve(ctx, pairs, pose, gnp).pose)

    import torch
    pose, k = d["T_init"].copy(), 0

    per_matcher = {}  # matcher -> search ms of every replayed call (hipEvents around its search kernels, inside the library)

    def tap(name):
        per_matcher.setdefault(name, []).append(ctx.stats()["ms_nn"])

    def one(tap_=None):
        nonlocal pose, k
        if k % CYCLE == 0:
            pose = d["T_init"].copy()
        pose = step(pose, tap_)
        k += 1

    elapsed, ts = timed_steps(one, args.steps, args.warmup, torch.cuda.synchronize, dist, world, torch.device("cuda"))
    # kernel breakdown + pairs from a replay with events
    pose, k = d["T_init"].copy(), 0
    for _ in range(args.warmup):
        one()
    ctx.set_profiling(1)
    nn, cp, gn, npt, npl = [], [], [], [], []
    for _ in range(min(args.steps, CYCLE)):
        one(tap)
        st = ctx.stats()
        nn.append(st["ms_nn"]), cp.append(st["ms_compact"]), gn.append(st["ms_gn"])
        a, b, _ = pairs.counts()
        npt.append(a), npl.append(b)
    ctx.set_profiling(0)
    err = amd.se3.log(amd.se3.inverse_compose(pose, d["T_gt"]))
    spread = None
    if dist is not None and world > 1:
        # every rank solved the same all-reduced 6x6 systems: one pose
        t_ = torch.tensor(np.asarray(pose, dtype=np.float64).ravel(), dtype=torch.float64,
                          device=torch.device("cuda") if dist.get_backend() == "nccl" else torch.device("cpu"))
        lo, hi = t_.clone(), t_.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN), dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).abs().max().item())
    nn_ms = float(np.mean(nn))
    knn = 5
    # lower bound of the distinct global points touched (N_g,touched >= N_pairs): the
    # pairs' own neighbours; the fraction below is therefore a LOWER bound of the roofline fraction
    touched_lb = float(np.mean(npt)) + min(float(g.shape[0]), knn * float(np.mean(npl)))
    touched_exact = None
    touched_by = {}  # matcher -> exact N_g,touched of its instrumented calls
    if sharded is None:
        # ... and exactly: the instrumented call of every matcher of the step marks each map point its staging loop fetches
        # (profiling level 2; the marks are cleared per call, a point both matchers of C5 fetch counts twice: it is read twice),
        # counted over three mid-chain steps
        cnt = []
        for _ in range(3):
            t = 0
            ctx.set_profiling(2)
            pairs.clear()
            if which == "c2":
                core.match_pt2pt(ctx, gmap, cloud, pose, pt, None, pairs)
                t += ctx.stats()["nn_points_staged"]
                touched_by.setdefault("pt2pt", []).append(t)
            elif which == "c3":
                core.match_pt2pl(ctx, gmap, cloud, pose, pl, None, pairs)
                t += ctx.stats()["nn_points_staged"]
                touched_by.setdefault("pt2pl", []).append(t)
            else:
                ms_dev.reset()
                core.match_pt2pl(ctx, gmap, cloud, pose, pl, ms_dev, pairs)
                t += ctx.stats()["nn_points_staged"]
                touched_by.setdefault("pt2pl", []).append(t)
                core.match_pt2pt(ctx, gmap, cloud, pose, pt, ms_dev, pairs)
                t1_ = ctx.stats()["nn_points_staged"]
                touched_by.setdefault("pt2pt", []).append(t1_)
                t += t1_
            ctx.set_profiling(0)
            cnt.append(t)
            pose = np.asarray(core.horn_solve(ctx, pairs)[0]) if which == "c2" else np.array(core.gn_solve(ctx, pairs, pose, gnp).pose)
        touched_exact = float(np.mean(cnt))
        touched_lb = touched_exact
    out_bytes = 8.0 * n_l if which == "c2" else 72.0 * float(np.mean(npl)) + (8.0 * n_l if which == "c5" else 0.0)
    alg = 12.0 * n_l * (2 if which == "c5" else 1) + 12.0 * touched_lb + out_bytes
    ach = alg / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0  # (no per-stage events: the sharded step over the test boxes' hook communicator)
    # ---- one roofline per matcher of the step (C5's dominant kernel is the plane search, not the last matcher's):
    #      algorithmic bytes = 12 N_l (the layer read) + 12 N_g,touched (exact, counted above) + the matcher's output
    #      (72 B per plane pairing; 8 B per local point for the point matcher's records); duration = hipEvents around the
    #      matcher's search kernels in the replay
    pl_kernel = "pt2pl_seltile_kernel" if n_l > 524288 else "pt2pl_tile_kernel"  # (launch_match_pt2pl's choice, Tune::pl_select = -1)
    names = {"pt2pl": f"pt2pl_cert_kernel + {pl_kernel} (K5: exact k-NN search; the plane fit is timed with the compaction)",
             "pt2pt": "nn_lane_kernel + nn_seltile_kernel + nn_single_kernel (K1+K3: transform + exact NN search)"}
    rooflines = []
    for name, ms_list in per_matcher.items():
        ms_m = float(np.mean(ms_list))
        tch = float(np.mean(touched_by[name])) if name in touched_by else None
        if tch is None or ms_m <= 0:
            continue
        out_b = 72.0 * float(np.mean(npl)) if name == "pt2pl" else 8.0 * n_l
        alg_m = 12.0 * n_l + 12.0 * tch + out_b
        ach_m = alg_m / (ms_m * 1e-3) / 1e9
        rooflines.append({"matcher": "Matcher_Point2Plane" if name == "pt2pl" else "Matcher_Points_DistanceThreshold",
                          "bound": "hbm", "kernel": names[name], "achieved": ach_m, "peak": 8000.0, "unit": "GB/s",
                          "frac": ach_m / 8000.0, "algorithmic_bytes_per_launch": alg_m, "avg_launch_ms": ms_m, "traffic": None,
                          "n_g_touched": tch, "n_g_touched_exact": True})
    rooflines.sort(key=lambda r_: -r_["avg_launch_ms"])
    cpu = None
    if not args.no_cpu_baseline and world == 1 and sharded is None:
        cpu = cpu_baseline_config(d, which, min(256, os.cpu_count() or 1))
    return {
        # N > 1: one step registers `world` scan-sized shards jointly (one pose, one 6x6 system); as on the default
        # line the unit is one iteration over ONE scan-sized layer, hence world / t_step
        "metric": "icp_iterations_per_sec", "value": world * args.steps / elapsed, "unit": "iterations/s", "n_gpus": world,
        "steps_per_sec_wall": args.steps / elapsed,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 search / f64 plane fit and normal equations",
        "data": "synthetic (seeded KITTI-shape street scene; map = voxel-thinned union of consecutive scans, SURVEY.md 8d)",
        "config": {"workload": label, "n_local": int(n_l), "n_global": int(g.shape[0]), "baseline_config": which.upper(),
                   "pose_chain": f"real ICP chain, restart from perturbed guess every {CYCLE} steps"},
        "step_ms": stats_ms(ts),
        "kernel_ms": {**{"search_" + k_: float(np.mean(v_)) for k_, v_ in per_matcher.items()},
                      "search_last_matcher": nn_ms, "compact_last_matcher": float(np.mean(cp)), "solver": float(np.mean(gn)),
                      "note": "hipEvents of the LAST matcher call of the step (c5 runs two matchers)"},
        "pairs_per_step": {"pt2pt": float(np.mean(npt)), "pt2pl": float(np.mean(npl))},
        "matched_pairs_per_sec": (float(np.mean(npt)) + float(np.mean(npl))) * args.steps / elapsed,
        "final_pose_error": {"trans_m": float(np.linalg.norm(err[:3])), "rot_rad": float(np.linalg.norm(err[3:]))},
        # the DOMINANT matcher's search (largest launch duration of the step); every matcher's: "rooflines"
        **({"rooflines": rooflines} if rooflines else {}),
        "roofline": rooflines[0] if rooflines else
                    {"bound": "hbm", "kernel": "search kernel(s) of the last matcher of the step",
                     "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_ms": nn_ms, "traffic": None,
                     "n_g_touched": touched_lb, "n_g_touched_exact": touched_exact is not None,
                     "note": ("N_g,touched counted by the instrumented search (a byte per map point fetched)" if touched_exact is not None else
                              "N_g,touched replaced by its lower bound (the pairs' own neighbours): frac is a lower bound")},
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        **({"final_pose_max_abs_diff_over_ranks": spread} if spread is not None else {}),
    }


# ---------------------------------------------------------------------------------------------------
def timed_steps(one, steps, warmup, sync, dist, world, dev):
    """W untimed calls of one(), then exactly K between two barrier + synchronize brackets; the MAX over the ranks of the
    elapsed time -> (elapsed_s, [step_s]).  The multi-rank core of the --config lines (bench_config); also driven over gloo
    with an oracle-backed step in tests/test_distributed_gloo.py (the dry run of `--gpus N --config c3`)."""
    import torch

    def barrier():
        sync()
        if dist is not None and world > 1:
            dist.barrier()
        sync()

    for _ in range(warmup):
        one()
    barrier()
    t0 = time.perf_counter()
    ts = []
    for _ in range(steps):
        t1 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t1)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, ts


# ---------------------------------------------------------------------------------------------------
def sharded_default_line(args, rank, world, dist, make_rig, dev, sync, build=None):
    """Everything of the default line that involves MORE THAN ONE RANK: the weak-scaling timing (W untimed steps, K steps
    between two barrier + synchronize brackets, MAX over ranks) and the strong-scaling block (ONE scan split N ways).
    make_rig(d, n_offset) -> a rig (restart / one_step / ctx.set_profiling / ctx.stats / info); dev: where the reduction
    tensors live; sync(): drains this rank's device.  main() passes the HIP rig, cuda and torch.cuda.synchronize;
    tests/test_distributed_gloo.py passes a rig whose compute is the CPU oracle, "cpu" and a no-op over gloo -- a dry run
    of everything that is not a kernel, so that the first run on an 8-GPU node is a measurement, not a debug session
    (VERDICT r3 #10).  The product never takes that route: bench.py itself only ever builds the HIP rig."""
    import torch
    build = build or build_inputs

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    t0 = time.time()
    d = build(args.n_local, args.n_global, args.seed, rank, world, args.scene)
    log(f"[bench r{rank}] inputs ready in {time.time() - t0:.1f}s: local {d['local'].shape}, global {d['glob'].shape}")
    n_l = d["local"].shape[0]
    rig = make_rig(d, rank * n_l)
    info = rig.info
    log(f"[bench r{rank}] index: cell {info['cell_size']:.3f} m, {info['n_levels']} levels, "
        f"{info['n_cells_level0']} voxels, {info['device_bytes'] / 1e6:.0f} MB, build "
        f"{info['build_ms']:.1f} ms (upload+build {rig.t_index * 1e3:.0f} ms)")

    # ---- timed region: exactly K steps between two barrier+synchronize brackets --------------
    # two hipEvents per step around the search kernels (the roofline kernels), read back lazily;
    # the per-kernel breakdown comes from the untimed replay below
    elapsed, nn_ms, step_s = timed_chain(rig, args.steps, args.warmup, barrier, events=not args.no_events)
    log("timed steps [ms]:", " ".join("%.3f" % (1e3 * t) for t in step_s))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    def pose_spread(pose):
        """largest |difference| of a pose's 12 numbers over the ranks (every rank solves the same all-reduced 6x6 system)"""
        t_ = torch.tensor(np.asarray(pose, dtype=np.float64), dtype=torch.float64, device=dev)
        lo, hi = t_.clone(), t_.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN), dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return float((hi - lo).abs().max().item())

    # ---- N > 1: strong scaling of ONE scan (every rank a contiguous 1/N of rank 0's scan) -------
    strong = None
    weak_pose_spread = None
    if world > 1:
        weak_pose_spread = pose_spread(rig.state["pose"])
        # a rank that fails locally (out of memory while building its rig, ...) must not leave its peers inside a collective:
        # the set-up is attempted on every rank, its success is all-reduced, and the block is skipped everywhere unless all
        # ranks got through (the first 8-GPU run would have hung instead of reporting an error)
        rig_s, err = None, None
        try:
            d0 = build(args.n_local, args.n_global, args.seed, 0, world, args.scene)
            from mp2p_icp_amd.distributed import shard_range
            b, e = shard_range(d0["local"].shape[0], rank, world)
            ds = dict(d0, local=np.ascontiguousarray(d0["local"][b:e]))
            rig_s = make_rig(ds, b)
        except Exception as ex:
            err = repr(ex)
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() < 0.5:
            strong = {"error": err or "another rank failed to set up its shard"}
        else:
            el_s, _, _ = timed_chain(rig_s, args.steps, args.warmup, barrier, events=False)
            ts_ = torch.tensor([el_s], dtype=torch.float64, device=dev)
            dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
            strong = {"scaling": "strong", "workload": f"ONE {d0['local'].shape[0]}-pt scan split x{world} vs the replicated map",
                      "value": args.steps / float(ts_.item()), "unit": "iterations/s",
                      "ms_per_step": float(ts_.item()) / args.steps * 1e3,
                      "final_pose": [float(v) for v in rig_s.state["pose"]],
                      "final_pose_max_abs_diff_over_ranks": pose_spread(rig_s.state["pose"])}
    if strong is not None and weak_pose_spread is not None:
        strong["weak_chain_final_pose_max_abs_diff_over_ranks"] = weak_pose_spread
    return {"d": d, "rig": rig, "barrier": barrier, "elapsed": elapsed, "nn_ms": nn_ms, "step_s": step_s, "strong": strong}


# ---------------------------------------------------------------------------------------------------
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
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default=None,
                    help="another BASELINE config as a bench line of its own (1 GPU)")
    ap.add_argument("--config-scale", type=float, default=1.0, help="tests only: --gpus N --config c3 at this fraction of its sizes")
    ap.add_argument("--scene", choices=["a", "b"], default="b", help="scene of `value` (see the module docstring)")
    ap.add_argument("--q", type=int, default=0, help="queries per wave (tuning)")
    ap.add_argument("--r0", type=float, default=0.0, help="initial radius in cells (tuning)")
    ap.add_argument("--grp", type=float, default=0.0, help="group radius factor (tuning)")
    ap.add_argument("--bricks", type=int, default=0, help="brick budget of the deferred-query kernel (0 = default)")
    ap.add_argument("--no-bitmap", action="store_true", help="build the map without occupancy bitmaps")
    ap.add_argument("--no-events", action="store_true", help="timed loop without hipEvents (overhead probe)")
    ap.add_argument("--cold", action="store_true", help="disable the warm start from the previous iteration")
    ap.add_argument("--defer", type=float, default=0.0, help="defer radius in cells (tuning)")
    ap.add_argument("--budget", type=int, default=0, help="voxel budget per search box (tuning)")
    ap.add_argument("--cell", type=float, default=0.0, help="voxel edge [m] (0 = automatic)")
    ap.add_argument("--target-per-cell", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip scene_b / host_boundary / stability (tuning runs)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="local points of the CPU baseline's sample (0 = the whole layer)")
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

    # one HIP stream shared with torch so that RCCL collectives are ordered with our kernels: a
    # dedicated torch stream made current (torch's default stream is the null stream, raw value 0)
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    if args.config:
        if world != 1 and args.config != "c3":
            raise SystemExit("--config c2 / c5 lines are single-GPU (c3 shards: mp2p_hip_step_sharded_pt2pl)")
        if world == 1 and os.environ.get("MP2P_BENCH_FORCE_DIST") == "1":
            # test hook: the sharded c3 code path with a one-rank RCCL communicator on a one-GPU box
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        line = bench_config(args, args.config, local_rank, stream, rank, world, dist)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    core_ = sharded_default_line(args, rank, world, dist,
                                 lambda d_, off: Rig(args, d_, rank, world, dist, local_rank, stream, off),
                                 torch.device("cuda"), torch.cuda.synchronize)
    d, rig, barrier = core_["d"], core_["rig"], core_["barrier"]
    elapsed, nn_ms, step_s, strong = core_["elapsed"], core_["nn_ms"], core_["step_s"], core_["strong"]
    n_l, info = d["local"].shape[0], rig.info

    rp = replay(rig, args.steps, args.warmup)
    log(f"[bench r{rank}] per-step kernel ms (replay; chain position = (warmup + i) % {CYCLE}): lane="
        f"{[round(v, 3) for v in rp['lane']]} tile={[round(v, 3) for v in rp['tile']]} "
        f"single={[round(v, 3) for v in rp['single']]} gn={[round(v, 3) for v in rp['gn']]}; "
        f"search in the timed loop: {[round(v, 3) for v in nn_ms]}")
    rows, final_err = instrumented(rig, min(CYCLE, args.warmup + args.steps), rank, "scene " + args.scene)

    pairs_total = torch.tensor([float(np.sum(rp["pairs"]))], dtype=torch.float64, device="cuda")
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
    nn_ms_avg = max(float(np.mean(nn_ms)) if nn_ms else 0.0, 1e-9)  # 0 only with --no-events (overhead probe)
    g = d["glob"]
    scene_txt = {"a": "ray-cast scan vs surface-sampled map",
                 "b": "ray-cast scan vs voxel-thinned union of consecutive scans (SURVEY.md 8d)"}
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
        "data": f"synthetic (seeded KITTI-shape street scene: {scene_txt[args.scene]})",
        "config": {
            "workload": f"{n_l * world}-pt local ({world} x {n_l}) vs {g.shape[0]}-pt global, "
                        "Matcher_Points_DistanceThreshold + Solver_GaussNewton",
            "scene": args.scene,
            "threshold_m": args.threshold, "thresholdAngularDeg": 0.0,
            "gn_inner_iterations": args.gn_iters, "robust_kernel": "GemanMcClure(0.15)",
            "unique_global_filter": True,
            "pose_chain": f"real ICP chain, restart from perturbed guess every {CYCLE} steps",
            "parallelism": f"local layer sharded x{world}, map replicated",
        },
        "step_ms": stats_ms(step_s),
        "steps_per_sec_wall": args.steps / elapsed,
        "unit_of_work": f"one outer ICP iteration (match + Gauss-Newton solve) over {n_l} local points vs "
                        f"{g.shape[0]} global points; a step does {world} of them jointly",
        "matched_pairs_per_sec": float(pairs_total.item()) / elapsed,
        "queries_per_sec": n_l * world * args.steps / elapsed,
        "pairs_per_step": float(pairs_total.item()) / args.steps,
        "kernel_ms": {"note": "nn_search: hipEvents in the timed loop; the others: same chain replayed "
                              "with an event around every stage",
                      "nn_search": nn_ms_avg, "nn_search_replay": float(np.mean(rp["nn"])),
                      "nn_lane_kernel": float(np.mean(rp["lane"])),
                      "nn_tile_kernel": float(np.mean(rp["tile"])),
                      "nn_single_kernel": float(np.mean(rp["single"])),
                      "compact": float(np.mean(rp["compact"])),
                      "gn_solve_all_inner": float(np.mean(rp["gn"])),
                      # launch of the first kernel + dependent-dispatch gaps + pose read-back + host wake-up; a
                      # captured hipGraph does not shorten it (profiles/r03_graph_probe.txt: 12 dependent kernels
                      # 30 us on a stream, 34 us as a graph, 15 us for ONE kernel)
                      # (means on both sides: the search time is the mean over the timed steps, cold restart steps included)
                      "step_minus_kernels": float(np.mean(step_s)) * 1e3 - (nn_ms_avg + float(np.mean(rp["compact"])) + float(np.mean(rp["gn"])))},
        "nn_stats": {"pending_after_prologue_frac": float(np.mean([r["pending"] for r in rows])) / n_l,
                     "finished_without_search_frac": float(np.mean([r["skipped"] for r in rows])) / n_l,
                     "deferred_to_one_query_kernel_frac": float(np.mean([r["deferred"] for r in rows])) / n_l,
                     "avg_passes_per_tile": float(np.mean([r["passes"] for r in rows])),
                     "candidates_tested_per_query": float(np.mean([r["cand"] for r in rows])) / n_l,
                     "global_points_touched": float(np.mean([r["touched"] for r in rows])),
                     "max_candidates_one_tile": int(np.max([r["maxcand"] for r in rows])),
                     "max_passes_one_tile": int(np.max([r["maxpass"] for r in rows])),
                     "voxel_m": info["cell_size"]},
        "index_build_ms": info["build_ms"],
        "final_pose_error": {"trans_m": float(np.linalg.norm(final_err[:3])),
                             "rot_rad": float(np.linalg.norm(final_err[3:]))},
        "roofline": roofline_block(n_l, float(np.mean([r["touched"] for r in rows])), nn_ms_avg, "scene_" + args.scene),
    }
    if strong is not None:
        # `value` at N > 1 is WEAK scaling (each rank its own scan, one joint pose); what BASELINE.json's north star means by
        # "near-linear scaling of the 6x6 all-reduce" is the strong-scaling figure: promoted beside `value`
        out["strong_scaling"] = strong
        out["value_strong_scaling"] = strong.get("value")
    if world == 1 and not args.no_extras:
        # ---- 200 more chained steps: run-to-run / step-to-step spread ------------------------------
        el2, _, st2 = timed_chain(rig, 200, args.warmup, barrier, events=False)
        a = np.asarray(st2) * 1e3
        out["stability"] = {"steps": 200, "iterations_per_s": 200 / el2, "ms_per_step_mean": float(a.mean()),
                            "p5": float(np.percentile(a, 5)), "p50": float(np.percentile(a, 50)),
                            "p95": float(np.percentile(a, 95)), "max": float(a.max()),
                            "note": "no hipEvents in this loop; chain position 0 (the restart from the perturbed guess, "
                                    "stale warm start) is the slow step of every cycle"}
        # ---- a chain that converges -------------------------------------------------------------------------
        try:
            out["converging"] = converging_block(rig, args)
        except Exception as ex:
            out["converging"] = {"error": repr(ex)}
        # ---- the boundary through host containers ------------------------------------------------------
        try:
            out["host_boundary"] = host_boundary(args, d, ms_per_step)
        except Exception as ex:
            out["host_boundary"] = {"error": repr(ex)}
        # ---- the other scene ---------------------------------------------------------------------------
        other = "b" if args.scene == "a" else "a"
        try:
            t0 = time.time()
            d2 = build_inputs(args.n_local, args.n_global, args.seed, 0, 1, other)
            t_gen = time.time() - t0
            rig2 = Rig(args, d2, 0, 1, None, local_rank, stream, 0)
            el, nn2, st2 = timed_chain(rig2, args.steps, args.warmup, barrier)
            rp2 = replay(rig2, args.steps, args.warmup)
            rows2, ferr2 = instrumented(rig2, min(CYCLE, args.warmup + args.steps), 0, "scene " + other)
            n2 = d2["local"].shape[0]
            out["scene_" + other] = {
                "data": scene_txt[other], "n_local": int(n2), "n_global": int(d2["glob"].shape[0]),
                "value": args.steps / el, "unit": "iterations/s", "ms_per_step": el / args.steps * 1e3,
                "step_ms": stats_ms(st2), "pairs_per_step": float(np.mean(rp2["pairs"])),
                "matched_pairs_per_sec": float(np.sum(rp2["pairs"])) / el,
                "kernel_ms": {"nn_search": float(np.mean(nn2)), "nn_lane_kernel": float(np.mean(rp2["lane"])),
                              "nn_tile_kernel": float(np.mean(rp2["tile"])), "nn_single_kernel": float(np.mean(rp2["single"])),
                              "compact": float(np.mean(rp2["compact"])), "gn_solve_all_inner": float(np.mean(rp2["gn"]))},
                "nn_stats": {"pending_after_prologue_frac": float(np.mean([r["pending"] for r in rows2])) / n2,
                             "finished_without_search_frac": float(np.mean([r["skipped"] for r in rows2])) / n2,
                             "global_points_touched": float(np.mean([r["touched"] for r in rows2]))},
                "pose_error_along_the_chain_m": [round(r["err_t"], 4) for r in rows2],
                "final_pose_error": {"trans_m": float(np.linalg.norm(ferr2[:3])), "rot_rad": float(np.linalg.norm(ferr2[3:]))},
                "roofline": roofline_block(n2, float(np.mean([r["touched"] for r in rows2])), float(np.mean(nn2)), "scene_" + other),
                "input_generation_s": t_gen,
            }
        except Exception as ex:
            out["scene_" + other] = {"error": repr(ex)}
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
        cores = os.cpu_count() or 1
        t0 = time.time()
        out["cpu_baseline"] = cpu_baseline(d, args.threshold, args.gn_iters, 0.15, args.cpu_sample or d["local"].shape[0], cores)
        out["cpu_baseline"]["wall_s"] = time.time() - t0
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # (the measured multi-thread figure sits under this container's CPU quota -- cpu_quota_cores; the 50 x target of
        #  BASELINE.json is to be read against BOTH: the quota-bound measurement and the quota-free ideal)
        out["speedup_vs_ideal_scaling_of_single_thread"] = out["value"] / out["cpu_baseline"]["ideal_scaling_of_single_thread"]["value"]
        # ---- the parity gate: the oracle's lists of the WHOLE headline layer were just computed ------------
        gate = out["cpu_baseline"].pop("_gate")
        if gate:
            try:
                out["parity_gate"] = parity_gate(rig, gate)
            except Exception as ex:
                out["parity_gate"] = {"passed": False, "error": repr(ex)}
            if not out["parity_gate"]["passed"]:
                log("[bench] PARITY GATE FAILED on the headline workload: no value is reported")
                out["value_withheld"] = out.pop("value")
                out["value"] = None
        else:
            out["parity_gate"] = {"skipped": "--cpu-sample below the layer's size: the oracle did not see every query"}
    else:
        out["parity_gate"] = {"skipped": "no CPU oracle in this run (--no-cpu-baseline, or a rank of a multi-GPU job: the gate runs at N = 1)"}
    if isinstance(out.get("host_boundary"), dict) and "iterations_per_s" in out["host_boundary"]:
        out["value_through_host_containers"] = out["host_boundary"]["iterations_per_s"]
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
