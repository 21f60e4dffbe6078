#!/usr/bin/env python3
"""The tile kernel's speed-of-light variants on bench-chain poses, for rocprofv3 (tools/gpu_pmc.sh): per chain pose k the full
kernel at pose k-1 (sets the warm start), then tile_sol = 1 (list + select + resolve + stage), 2 (+ matrix-pipe prefilter) and
0 (full) at pose k -- the three are different template instances, so the counters come out per variant."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import argparse
import bench
knobs = sys.argv[1] if len(sys.argv) > 1 else "nn_direct=1"
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=0); torch.cuda.set_stream(ts)
d = bench.build_inputs(1_000_000, 10_000_000, 1, 0, 1, "b")
args = argparse.Namespace(threshold=2.0, gn_iters=3, cell=0.0, target_per_cell=0.0, no_bitmap=False, r0=0.0, q=0, grp=0.0, budget=0, defer=0.0, cold=False, bricks=0)
rig = bench.Rig(args, d, 0, 1, None, 0, ts.cuda_stream, 0)
rig.ctx.set_tune(knobs)
rig.restart()
poses = []
for _ in range(10):
    poses.append(rig.state["pose"].copy()); rig.one_step()
rig.ctx.set_profiling(1)
out = {0: [], 1: [], 2: [], 3: [], 4: [], 5: []}
for k in range(1, 10):
    rig.ctx.set_tune("tile_sol=0"); rig.reg.match(poses[k - 1])
    for m in (3, 4, 5, 1, 2, 0):
        rig.ctx.set_tune(f"tile_sol={m}"); rig.reg.match(poses[k]); out[m].append(rig.ctx.stats()["ms_nn_tile"])
rig.ctx.set_tune("tile_sol=0")
print(json.dumps({f"sol{m}_ms": float(np.mean(v)) for m, v in out.items()}))
