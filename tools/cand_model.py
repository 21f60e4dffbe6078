"""CPU model of what the pt2pt search has to stage (round 5 design study; no GPU needed).

For real chain poses of a scene-B pair (numpy + scipy, the oracle for the ICP steps) it counts, per tile of Q
Morton-consecutive queries warm-started from the previous pose's neighbours:
  bbox    the rule of rounds 1-4: every occupied voxel within max r of the BOX of the tile's queries
  sphere  a voxel is needed iff some query has |centre - q| <= r_q + rho (rho = half diagonal): the matrix-pipe selection
  exact   a voxel is needed iff some query's ball really reaches the voxel's box
and the points those voxels hold.  usage: python tools/cand_model.py <scene npz> [voxel edge] [n_tiles sampled]
"""
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, ".")
from mp2p_icp_amd import se3  # noqa: E402


def morton_order(p, n_bits=20):
    mn = p.min(0)
    ext = float((p.max(0) - mn).max())
    c = np.minimum(((p - mn) / max(ext / 1048000.0, 1e-9)).astype(np.uint64), (1 << n_bits) - 1)

    def spread(v):
        x = v & np.uint64(0xFFFFF)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
        return x
    key = spread(c[:, 0]) | (spread(c[:, 1]) << np.uint64(1)) | (spread(c[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def main():
    path = sys.argv[1]
    h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.168
    n_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    z = np.load(path)
    loc, glob = z["local"].astype(np.float64), z["glob"].astype(np.float64)
    T = z["T_init"].copy()
    print(f"local {loc.shape[0]}, global {glob.shape[0]}, voxel {h}")
    order = morton_order(z["local"])
    loc = loc[order]
    t0 = time.time()
    tree = cKDTree(glob)
    print(f"tree {time.time() - t0:.1f}s")
    o = glob.min(0)
    vox = np.floor((glob - o) / h).astype(np.int64)
    vkey = (vox[:, 2] << 42) | (vox[:, 1] << 21) | vox[:, 0]
    uk, inv, cnt = np.unique(vkey, return_inverse=True, return_counts=True)
    print(f"occupied voxels {uk.shape[0]}, points per voxel {glob.shape[0] / uk.shape[0]:.1f}")
    occ_xyz = np.stack([uk & 0x1FFFFF, (uk >> 21) & 0x1FFFFF, uk >> 42], 1).astype(np.float64)
    occ_c = o + (occ_xyz + 0.5) * h
    vtree = cKDTree(occ_c)
    rho = h * np.sqrt(3) / 2
    rng = np.random.default_rng(0)

    prev_nn = None
    thr = 2.0
    for step in range(6):
        R, t = se3.Rt(T)
        q = loc @ R.T + t
        d, nn = tree.query(q, k=1, distance_upper_bound=thr, workers=8)
        ok = np.isfinite(d)
        if prev_nn is not None:
            have = prev_nn < glob.shape[0]
            r_ub = np.where(have, np.linalg.norm(q - glob[np.minimum(prev_nn, glob.shape[0] - 1)], axis=1) * (1 + 1 / 512) + 1e-4, thr)
            r_ub = np.minimum(r_ub, thr)
            print(f"step {step}: pairs {ok.sum()}, r_ub percentiles 10/50/90/99: "
                  f"{np.percentile(r_ub, [10, 50, 90, 99]).round(3)}; same NN as before: {(nn == prev_nn)[ok].mean():.3f}; "
                  f"d_nn median {np.median(d[ok]):.3f}")
            for Q in (32, 64, 128, 256):
                tiles = rng.choice(loc.shape[0] // Q, min(n_tiles, loc.shape[0] // Q), replace=False)
                res = {k: [] for k in ("bbox_v", "bbox_p", "sph_v", "sph_p", "ex_v", "ex_p", "ext", "e1_p", "sph1_p")}
                for tl in tiles:
                    sl = slice(tl * Q, tl * Q + Q)
                    qq, rr = q[sl], r_ub[sl]
                    lo, hi = qq.min(0), qq.max(0)
                    rmax = rr.max()
                    # candidates: occupied voxels whose centre is within rmax + rho of the box's centre region
                    ctr, half = (lo + hi) / 2, (hi - lo) / 2
                    ids = np.array(vtree.query_ball_point(ctr, np.linalg.norm(half) + rmax + rho), dtype=np.int64)
                    if ids.size == 0:
                        for k in res:
                            res[k].append(0)
                        continue
                    c = occ_c[ids]
                    vlo, vhi = c - h / 2, c + h / 2
                    # bbox rule: dist(voxel box, query bbox) <= rmax
                    dd = np.maximum(0, np.maximum(vlo - hi, lo - vhi))
                    m_b = (dd ** 2).sum(1) <= rmax ** 2
                    # per query tests
                    dc = np.linalg.norm(c[:, None, :] - qq[None, :, :], axis=2)  # [V, Q]
                    m_s = (dc <= (rr + rho)[None, :]).any(1)
                    db = np.maximum(0, np.maximum(vlo[:, None, :] - qq[None], qq[None] - vhi[:, None, :]))
                    de = np.sqrt((db ** 2).sum(2))
                    m_e = (de <= rr[None, :]).any(1)
                    pc = cnt[ids]
                    res["bbox_v"].append(m_b.sum()), res["bbox_p"].append(pc[m_b].sum())
                    res["sph_v"].append(m_s.sum()), res["sph_p"].append(pc[m_s].sum())
                    res["ex_v"].append(m_e.sum()), res["ex_p"].append(pc[m_e].sum())
                    res["ext"].append(float((hi - lo).max()))
                    res["e1_p"].append(float((pc[:, None] * (de <= rr[None, :])).sum(0).mean()))
                    res["sph1_p"].append(float((pc[:, None] * (dc <= (rr + rho)[None, :])).sum(0).mean()))
                s = {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}
                print(f"   Q={Q:3d}: tile extent med {np.median(s['ext']):.2f} m | points staged per tile  "
                      f"bbox {s['bbox_p'].mean():7.0f} (p95 {np.percentile(s['bbox_p'], 95):6.0f}, max {s['bbox_p'].max():6.0f})  "
                      f"sphere {s['sph_p'].mean():7.0f} (p95 {np.percentile(s['sph_p'], 95):6.0f}, max {s['sph_p'].max():6.0f})  "
                      f"exact {s['ex_p'].mean():7.0f} (p95 {np.percentile(s['ex_p'], 95):6.0f}) | per query alone: exact {s['e1_p'].mean():.0f}, sphere {s['sph1_p'].mean():.0f} "
                      f"| voxels bbox/sphere/exact {s['bbox_v'].mean():.0f}/{s['sph_v'].mean():.0f}/{s['ex_v'].mean():.0f}")
        prev_nn = np.where(ok, nn, glob.shape[0])
        # one ICP step (point-to-point, unique-global filter ignored, plain least squares: the chain only has to be plausible)
        P, G = loc[ok], glob[nn[ok]]
        w = 0.15 ** 2 / (d[ok] ** 2 + 0.15) ** 2
        w = w / w.sum()
        for _ in range(3):
            Rk, tk = se3.Rt(T)
            X = P @ Rk.T + tk
            mu_x, mu_g = (w[:, None] * X).sum(0), (w[:, None] * G).sum(0)
            H = ((X - mu_x) * w[:, None]).T @ (G - mu_g)
            U, _, Vt = np.linalg.svd(H)
            D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
            dR = Vt.T @ D @ U.T
            dt = mu_g - dR @ mu_x
            T = se3.from_Rt(dR @ Rk, dR @ tk + dt)
        e = se3.log(se3.inverse_compose(T, z["T_gt"]))
        print(f"   after step: pose error {np.linalg.norm(e[:3]):.3f} m {np.degrees(np.linalg.norm(e[3:])):.2f} deg")


if __name__ == "__main__":
    main()
