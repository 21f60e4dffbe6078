import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
from mp2p_icp_amd.distributed import HipBackend, _DevArray
ctx = amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
d = synthetic.make_pair(2000, 8000, 5)
g, l = d["glob"], d["local"]
gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
prm = _lib.Pt2PtParams(1.0, 0.0, 1, 0, 0, 0.20, 0, 0.0, 0, 0.0, 0, 0.0, 0)
gnp = _lib.GNParams(); gnp.maxInnerLoopIterations = 3; gnp.minDelta = 1e-7; gnp.kernel = 1; gnp.kernelParam = 0.2; gnp.w_pt2pt = gnp.w_pt2pl = 1.0
pairs = core.DevicePairs(ctx, l.shape[0], 0)
be = HipBackend(ctx, gmap, cloud, prm, gnp, pairs)
be.phase1(d["T_init"]); be.phase2()
be.gn_begin(d["T_init"]); be.gn_accumulate()
torch.cuda.synchronize()
s = be.sums
print("ptr", hex(ctx.gn_sums_ptr()), "tensor ptr", hex(s.data_ptr()), "alias:", ctx.gn_sums_ptr() == s.data_ptr())
print(s[:4].cpu().numpy())
t = torch.as_tensor(_DevArray(ctx.gn_sums_ptr(), 48, "<f8"), device=torch.device("cuda", 0))
print("second view ptr", hex(t.data_ptr()))
