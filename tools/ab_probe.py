#!/usr/bin/env python3
"""A/B of search variants on the bench workload in ONE process (the scene and the index are built once; knobs are
switched at run time through mp2p_hip_set_tune).  usage:
    python tools/ab_probe.py out.json "name:knob=v,knob=v" "name2:..." [--scene b] [--steps 20] [--tpc 6,3] [--sol]
Each variant: the bench chain (restart from T_init every 10 steps), 10 warm-up + K timed steps with the wall clock,
then a replay with per-stage hipEvents, then one instrumented cycle (device counters).  --sol adds the speed-of-light
decomposition of the tile kernel (tile_sol = 1, 2 against the full kernel, same poses, same warm start)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--scene", default="b")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tpc", default="0", help="comma list of target_per_cell values (0 = the default): one map each")
    ap.add_argument("--cell", default="0", help="comma list of voxel edges [m] (0 = automatic)")
    ap.add_argument("--sol", action="store_true")
    ap.add_argument("--n-local", type=int, default=1_000_000)
    ap.add_argument("--n-global", type=int, default=10_000_000)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ts = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(ts)
    d = bench.build_inputs(a.n_local, a.n_global, 1, 0, 1, a.scene)
    res = []
    maps = [(float(t), float(c)) for t in a.tpc.split(",") for c in a.cell.split(",")]
    for tpc, cell in maps:
        args = argparse.Namespace(threshold=2.0, gn_iters=3, cell=cell, target_per_cell=tpc, no_bitmap=False, r0=0.0, q=0, grp=0.0,
                                  budget=0, defer=0.0, cold=False, bricks=0)
        rig = bench.Rig(args, d, 0, 1, None, 0, ts.cuda_stream, 0)
        info = {k: rig.info[k] for k in rig.info if isinstance(rig.info[k], (int, float))}
        print(f"[ab] map tpc={tpc} cell={cell}: {info}", file=sys.stderr, flush=True)
        for v in a.variants or ["default:"]:
            name, _, knobs = v.partition(":")
            rig.ctx.set_tune("tile_sol=0")
            if knobs:
                rig.ctx.set_tune(knobs)
            el, nn_ms, step_s = bench.timed_chain(rig, a.steps, 10, lambda: torch.cuda.synchronize())
            rp = bench.replay(rig, a.steps, 10)
            rows, _ = bench.instrumented(rig, bench.CYCLE, 0, name)
            row = dict(name=name, knobs=knobs, tpc=tpc, cell=cell, voxel=info.get("cell_size", 0.0), it_s=a.steps / el, ms_step=el / a.steps * 1e3,
                       nn_ms=float(np.mean(nn_ms)), nn_ms_median=float(np.median(nn_ms)),
                       lane=float(np.mean(rp["lane"])), tile=float(np.mean(rp["tile"])), single=float(np.mean(rp["single"])),
                       compact=float(np.mean(rp["compact"])), gn=float(np.mean(rp["gn"])), pairs=float(np.mean(rp["pairs"])),
                       touched=float(np.mean([r["touched"] for r in rows])), cand=float(np.mean([r["cand"] for r in rows])),
                       deferred=float(np.mean([r["deferred"] for r in rows])), maxcand=float(np.max([r["maxcand"] for r in rows])),
                       passes=float(np.mean([r["passes"] for r in rows])))
            st = rig.ctx.stats()
            row["listed_last"], row["needed_last"], row["tiles_last"] = st["nn_sel_voxels_listed"], st["nn_sel_voxels_needed"], st["nn_tiles"]
            res.append(row)
            print("[ab] " + json.dumps(row), file=sys.stderr, flush=True)
            if a.sol and "tile_select=0" not in knobs:
                rig.ctx.set_tune("nn_direct=1")  # (the timing-only variants exist for the fused-prologue build)
                # same poses, same warm start: full step k-1 (sets the warm start), then SOL launches and the full launch at pose k
                from mp2p_icp_amd import core
                rig.restart()
                sol = {0: [], 1: [], 2: [], 3: [], 4: [], 5: []}
                poses = []
                st_ = rig.state
                for _ in range(12):
                    poses.append(st_["pose"].copy())
                    rig.one_step()
                rig.ctx.set_profiling(1)
                rig.restart()
                for k in range(1, 10):
                    rig.ctx.set_tune("tile_sol=0")
                    rig.reg.match(poses[k - 1])
                    for m in (3, 4, 5, 1, 2, 0):
                        rig.ctx.set_tune(f"tile_sol={m}")
                        rig.reg.match(poses[k])
                        s = rig.ctx.stats()
                        sol[m].append(s["ms_nn_tile"])
                rig.ctx.set_profiling(0)
                rig.ctx.set_tune("tile_sol=0")
                row["sol_ms"] = {"prologue": float(np.mean(sol[3])), "plus_list": float(np.mean(sol[4])), "plus_select_resolve": float(np.mean(sol[5])),
                                 "plus_stage": float(np.mean(sol[1])), "plus_prefilter": float(np.mean(sol[2])), "full": float(np.mean(sol[0]))}
                rig.ctx.set_tune("nn_direct=0")
                if knobs:
                    rig.ctx.set_tune(knobs)
                print("[ab] SOL " + json.dumps(row["sol_ms"]), file=sys.stderr, flush=True)
        del rig
    json.dump(res, open(a.out, "w"), indent=1)
    for r in res:
        print(f"{r['name']:24s} tpc={r['tpc']:<4} it/s={r['it_s']:7.0f} step={r['ms_step']:.3f} nn={r['nn_ms']:.3f} (lane {r['lane']:.3f} tile {r['tile']:.3f} single {r['single']:.3f}) "
              f"cand/tile={r['cand'] / max(1, r['tiles_last']):.0f} deferred={r['deferred']:.0f} touched={r['touched']:.0f} pairs={r['pairs']:.0f}"
              + (f" SOL {r['sol_ms']}" if "sol_ms" in r else ""))


if __name__ == "__main__":
    main()
