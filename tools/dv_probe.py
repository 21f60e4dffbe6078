#!/usr/bin/env python3
"""Time FilterDecimateVoxels on the GPU (device arrays in and out) vs the CPU oracle."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core
import oracle
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = bench.build_inputs(n, 1000, 77, 0, 1)
p = d["local"]
ctx = amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
t = torch.from_numpy(np.ascontiguousarray(p.T)).cuda()
out = torch.empty_like(t); src = torch.empty(n, dtype=torch.int32, device="cuda")
for res, method in ((2.0, 0), (0.5, 0), (0.1, 1), (0.5, 2)):
    prm = _lib.DecimateParams(res, method, 0, 0.0)
    m = C.c_size_t()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(ctx._L.mp2p_hip_filter_decimate_voxels_device(ctx.handle, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n,
                   C.byref(prm), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), src.data_ptr(), C.byref(m)), ctx.handle)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); o, _ = oracle.filter_decimate_voxels(p[:, 0], p[:, 1], p[:, 2], res, method); tc = time.perf_counter() - t0
    print(f"n={n} res={res} method={method}: voxels={m.value} gpu {np.median(ts)*1e3:.3f} ms, cpu oracle {tc*1e3:.1f} ms, equal={np.array_equal(out[:, :m.value].T.cpu().numpy(), o)}", flush=True)
