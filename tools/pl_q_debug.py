#!/usr/bin/env python3
"""debug: Matcher_Point2Plane with 8-query vs 32-query search tiles on the C3 scene; prints the local
points whose pairing differs and their oracle neighbour lists"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mp2p_icp_amd as amd
from mp2p_icp_amd import _lib, core, synthetic
import oracle as orc

d = synthetic.make_scan_union_pair(120_000, 10_000_000, 3001, map_scan_points=1_000_000)
g, l = d["glob"], d["local"]
prm = _lib.Pt2PlParams(0.4, 0.4, 5, 5, 0.05, 0, 0.20, 0.0, 0)
res = {}
for q in (8, 32):
    os.environ["MP2P_HIP_TUNE"] = f"pl_q={q}"
    ctx = amd.Context(0)
    gmap = core.GlobalMap(ctx, g[:, 0], g[:, 1], g[:, 2])
    cloud = core.LocalCloud(ctx, l[:, 0], l[:, 1], l[:, 2])
    pairs = core.DevicePairs(ctx, 1, l.shape[0])
    core.match_pt2pl(ctx, gmap, cloud, d["T_init"], prm, None, pairs)
    rec, idx = pairs.download_pt2pl()
    res[q] = (rec, idx)
    print(q, len(idx))
a, b = set(res[8][1].tolist()), set(res[32][1].tolist())
diff = sorted(a ^ b)
print("differ:", diff[:20])
tree = orc.KDTree(g[:, 0], g[:, 1], g[:, 2])
tx, ty, tz, _, _ = orc.transform_local_to_global(l[:, 0], l[:, 1], l[:, 2], d["T_init"])
for i in diff[:5]:
    idx, dd = tree.knn((tx[i], ty[i], tz[i]), 8)
    print("query", i, (tx[i], ty[i], tz[i]), "in8" if i in a else "in32")
    print("  oracle knn idx", idx.tolist(), "d2", [float(v) for v in dd], "radSq", np.float32(0.4 * 0.4))
# same neighbours but different planes?
common = sorted(a & b)
m8 = dict(zip(res[8][1].tolist(), range(len(res[8][1]))))
m32 = dict(zip(res[32][1].tolist(), range(len(res[32][1]))))
bad = [i for i in common if not np.allclose(res[8][0]["plane"][m8[i]], res[32][0]["plane"][m32[i]], atol=1e-9)]
print("planes differing among common:", len(bad), bad[:10])
