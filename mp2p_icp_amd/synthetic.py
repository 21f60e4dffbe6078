"""Seeded synthetic inputs of KITTI shape: a street scene (ground plane +
building / car boxes), a spinning-LiDAR scan ray-cast in it, and a map sampled on the same
surfaces.  Pure numpy, deterministic for a seed; identical bytes feed the oracle and the GPU.
"""
import math

import numpy as np

GROUND_Z = -1.73
MAX_RANGE = 80.0


class Scene:
    def __init__(self, seed, length=200.0, half_width=40.0, n_buildings=60, n_cars=40):
        rng = np.random.default_rng(seed)
        self.length, self.half_width = float(length), float(half_width)
        boxes = []
        for _ in range(n_buildings):
            sx, sy, h = rng.uniform(6, 22), rng.uniform(6, 18), rng.uniform(3, 15)
            cx = rng.uniform(-20, length + 20)
            side = 1.0 if rng.random() < 0.5 else -1.0
            cy = side * rng.uniform(7 + sy / 2, half_width - sy / 2)
            boxes.append((cx - sx / 2, cy - sy / 2, GROUND_Z, cx + sx / 2, cy + sy / 2, GROUND_Z + h))
        for _ in range(n_cars):
            cx = rng.uniform(-10, length + 10)
            side = 1.0 if rng.random() < 0.5 else -1.0
            cy = side * rng.uniform(2.5, 5.5)
            boxes.append((cx - 2.1, cy - 0.9, GROUND_Z, cx + 2.1, cy + 0.9, GROUND_Z + 1.5))
        self.boxes = np.array(boxes, dtype=np.float64)

    # ---- LiDAR scan (sensor frame) -------------------------------------------------------
    def scan(self, sensor_xyz, yaw, n_rings, n_az, seed, sigma=0.02, chunk=1 << 18):
        rng = np.random.default_rng(seed)
        el = np.deg2rad(np.linspace(-24.8, 2.0, n_rings))
        az = np.linspace(0.0, 2 * math.pi, n_az, endpoint=False)
        EL, AZ = np.meshgrid(el, az, indexing="ij")
        d_s = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], -1).reshape(-1, 3)
        c, s = math.cos(yaw), math.sin(yaw)
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        o = np.asarray(sensor_xyz, dtype=np.float64)
        out = []
        for b in range(0, d_s.shape[0], chunk):
            ds = d_s[b:b + chunk]
            dw = ds @ Rz.T
            t = np.full(ds.shape[0], np.inf)
            with np.errstate(divide="ignore", invalid="ignore"):
                tg = (GROUND_Z - o[2]) / dw[:, 2]
            ok = (dw[:, 2] < -1e-9) & (tg > 0)
            t = np.where(ok, tg, t)
            inv = 1.0 / np.where(np.abs(dw) < 1e-12, 1e-12, dw)
            for bx in self.boxes:
                t0 = (bx[:3] - o) * inv
                t1 = (bx[3:] - o) * inv
                tn = np.minimum(t0, t1).max(axis=1)
                tf = np.maximum(t0, t1).min(axis=1)
                hit = (tn <= tf) & (tf > 0) & (tn > 0.5)
                t = np.where(hit & (tn < t), tn, t)
            keep = t < MAX_RANGE
            r = t[keep] + rng.normal(0.0, sigma, int(keep.sum()))
            out.append((ds[keep] * r[:, None]).astype(np.float32))
        return np.concatenate(out, axis=0)

    # ---- map: points sampled on the surfaces (world frame) -----------------------------------
    def sample_map(self, n, seed, sigma=0.01):
        rng = np.random.default_rng(seed)
        L, W = self.length + 60.0, 2 * self.half_width
        areas = [L * W]
        faces = []  # (origin, u, v) parallelograms
        faces.append((np.array([-30.0, -self.half_width, GROUND_Z]), np.array([L, 0, 0.0]), np.array([0, W, 0.0])))
        for bx in self.boxes:
            x0, y0, z0, x1, y1, z1 = bx
            dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
            for org, u, v in (
                ((x0, y0, z0), (dx, 0, 0), (0, 0, dz)), ((x0, y1, z0), (dx, 0, 0), (0, 0, dz)),
                ((x0, y0, z0), (0, dy, 0), (0, 0, dz)), ((x1, y0, z0), (0, dy, 0), (0, 0, dz)),
                ((x0, y0, z1), (dx, 0, 0), (0, dy, 0))):
                u, v = np.array(u, float), np.array(v, float)
                faces.append((np.array(org, float), u, v))
                areas.append(float(np.linalg.norm(np.cross(u, v))))
        areas = np.array(areas)
        counts = rng.multinomial(n, areas / areas.sum())
        pts = np.empty((n, 3), dtype=np.float32)
        k = 0
        for (org, u, v), c in zip(faces, counts):
            if c == 0:
                continue
            a, b = rng.random(c), rng.random(c)
            p = org[None, :] + a[:, None] * u[None, :] + b[:, None] * v[None, :]
            p += rng.normal(0.0, sigma, p.shape)
            pts[k:k + c] = p.astype(np.float32)
            k += c
        rng.shuffle(pts, axis=0)
        return pts


def rings_for(n_points):
    """(n_rings, n_az) giving about n_points rays with KITTI-like aspect (64 x 1875 = 120k)."""
    if n_points <= 150_000:
        return 64, max(8, int(round(n_points / 64)))
    n_rings = 128
    return n_rings, int(round(n_points / n_rings))


def perturbation(seed, max_t=0.5, max_r_deg=3.0):
    rng = np.random.default_rng(seed)
    t = rng.uniform(-max_t, max_t, 3)
    r = np.deg2rad(rng.uniform(-max_r_deg, max_r_deg, 3))
    return np.concatenate([t, r])  # x y z yaw pitch roll


def make_pair(n_local, n_global, seed, max_t=0.5, max_r_deg=3.0):
    """Returns dict(local[N,3] f32 in the sensor frame, global[M,3] f32 world, T_gt, T_init)."""
    from . import se3
    scene = Scene(seed)
    n_rings, n_az = rings_for(int(n_local * 1.25))  # some rays miss (sky / range)
    sensor = (scene.length * 0.5, 0.0, 0.0)
    yaw = 0.05
    loc = scene.scan(sensor, yaw, n_rings, n_az, seed + 1)
    if loc.shape[0] > n_local:
        # keep ring/azimuth order, drop uniformly
        idx = np.linspace(0, loc.shape[0] - 1, n_local).astype(np.int64)
        loc = loc[idx]
    glob = scene.sample_map(n_global, seed + 2)
    T_gt = se3.from_xyzypr(sensor[0], sensor[1], sensor[2], yaw, 0.0, 0.0)
    T_init = se3.compose(T_gt, se3.from_xyzypr(*perturbation(seed + 3, max_t, max_r_deg)))
    return dict(local=np.ascontiguousarray(loc), glob=glob, T_gt=T_gt, T_init=T_init)


def random_cloud_pair(n_local, n_global, seed, extent=20.0, noise=0.02, outlier_frac=0.0):
    """Small generic clouds for parity tests: global = random surface-ish points, local = a
    subset moved by the inverse of a random pose (+noise, + optional uniform outliers)."""
    from . import se3
    rng = np.random.default_rng(seed)
    g = rng.uniform(-extent, extent, (n_global, 3))
    g[:, 2] = 0.2 * np.sin(g[:, 0] * 0.7) + 0.1 * g[:, 1] + rng.normal(0, 0.3, n_global)
    g = g.astype(np.float32)
    T_gt = se3.from_xyzypr(*(rng.uniform(-1, 1, 3)), *(np.deg2rad(rng.uniform(-10, 10, 3))))
    sel = rng.choice(n_global, size=min(n_local, n_global), replace=n_local > n_global)
    R, t = se3.Rt(T_gt)
    l = (g[sel].astype(np.float64) - t) @ R  # R^T (g - t)
    l += rng.normal(0, noise, l.shape)
    n_out = int(outlier_frac * l.shape[0])
    if n_out:
        l[rng.choice(l.shape[0], n_out, replace=False)] = rng.uniform(-extent, extent, (n_out, 3))
    T_init = se3.compose(T_gt, se3.from_xyzypr(*(rng.uniform(-0.2, 0.2, 3)),
                                               *(np.deg2rad(rng.uniform(-2, 2, 3)))))
    return dict(local=l.astype(np.float32), glob=g, T_gt=T_gt, T_init=T_init)


# ---- SURVEY.md section 8d scenes: the map is a UNION OF CONSECUTIVE SCANS, voxel-thinned ----------
# (scan and map densities then match, unlike sample_map's uniform surface sampling).  The ray
# casting runs in torch -- on the GPU when there is one (input generation only; the product never
# sees torch tensors) -- because a map needs tens of scans.
def scan_torch(scene, sensor_xyz, yaw, n_rings, n_az, seed, sigma=0.02, device=None, chunk=1 << 18):
    """Scene.scan with torch tensors: points of one spinning-LiDAR scan in the SENSOR frame (float32
    numpy [N,3]).  Deterministic for a seed (noise from a CPU generator)."""
    import torch
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    f64 = torch.float64
    el = torch.deg2rad(torch.linspace(-24.8, 2.0, n_rings, dtype=f64))
    az = torch.arange(n_az, dtype=f64) * (2 * math.pi / n_az)
    EL, AZ = torch.meshgrid(el, az, indexing="ij")
    d_s = torch.stack([torch.cos(EL) * torch.cos(AZ), torch.cos(EL) * torch.sin(AZ), torch.sin(EL)], -1).reshape(-1, 3)
    c, s = math.cos(yaw), math.sin(yaw)
    Rz = torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1.0]], dtype=f64)
    o = torch.tensor(np.asarray(sensor_xyz, dtype=np.float64))
    boxes = torch.tensor(scene.boxes, dtype=f64, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    noise = torch.randn(d_s.shape[0], generator=gen, dtype=f64) * sigma
    out = []
    od = o.to(dev)
    for b in range(0, d_s.shape[0], chunk):
        ds = d_s[b:b + chunk].to(dev)
        dw = ds @ Rz.T.to(dev)
        tg = (GROUND_Z - od[2]) / dw[:, 2]
        t = torch.where((dw[:, 2] < -1e-9) & (tg > 0), tg, torch.full_like(tg, float("inf")))
        inv = 1.0 / torch.where(dw.abs() < 1e-12, torch.full_like(dw, 1e-12), dw)
        t0 = (boxes[None, :, :3] - od) * inv[:, None, :]
        t1 = (boxes[None, :, 3:] - od) * inv[:, None, :]
        tn = torch.minimum(t0, t1).amax(dim=2)
        tf = torch.maximum(t0, t1).amin(dim=2)
        hit = (tn <= tf) & (tf > 0) & (tn > 0.5)
        tb = torch.where(hit, tn, torch.full_like(tn, float("inf"))).amin(dim=1)
        t = torch.minimum(t, tb)
        keep = t < MAX_RANGE
        r = t[keep] + noise[b:b + chunk].to(dev)[keep]
        out.append((ds[keep] * r[:, None]).to(torch.float32).cpu())
    return torch.cat(out, 0).numpy()


def _voxel_thin(pts, vox):
    """one point (the first in input order) per voxel of edge `vox`; returns the kept points"""
    import torch
    p = torch.from_numpy(pts)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    q = torch.floor(p.to(dev).to(torch.float64) / vox).to(torch.int64) + (1 << 20)
    key = (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]
    ks, order = torch.sort(key, stable=True)
    first = torch.ones_like(ks, dtype=torch.bool)
    first[1:] = ks[1:] != ks[:-1]
    keep = torch.sort(order[first]).values.cpu()
    return pts[keep.numpy()]


def make_scan_union_pair(n_local, n_global, seed, map_scan_points=None, spacing=1.0, max_t=0.5,
                         max_r_deg=3.0, outlier_frac=0.0, max_scans=400):
    """local = one scan of ~n_local points (sensor frame); global = the union of
    consecutive scans taken every `spacing` metres along a gently curving trajectory through the
    street scene, voxel-thinned and cut to exactly n_global points (world frame).  The local scan is
    taken in the middle of the mapped stretch with its own noise stream.  outlier_frac of the local
    points are replaced by uniform points of the scan's bounding box (BASELINE config C5).
    Returns dict(local, glob, T_gt, T_init, n_scans, voxel)."""
    from . import se3
    cache = f"/tmp/mp2p_scene_b_{n_local}_{n_global}_{seed}_{map_scan_points}_{spacing}_{outlier_frac}.npz"
    try:
        import os
        if os.path.exists(cache):
            z = np.load(cache)
            return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    except Exception:
        pass
    rng = np.random.default_rng(seed)
    msp = int(map_scan_points or max(n_local, 120_000))
    # scene long enough for the stretch that n_global needs (estimated, grown below if too short)
    scene = Scene(seed, length=200.0)
    rings_m, az_m = rings_for(int(msp * 1.25))

    def pose_at(k):  # sensor pose of scan k: 1 m steps, slow sinusoidal sway and heading
        x = 20.0 + k * spacing
        y = 1.5 * math.sin(x / 35.0)
        yaw = 0.05 * math.cos(x / 35.0)
        return (x, y, 0.0), yaw

    # scans are added until the thinned union holds enough points; the union is thinned incrementally
    # ("first in input order" per voxel, so accumulating a thinned prefix changes nothing)
    vox = 0.05
    raw, n_scans, k = [], 0, 0
    acc = np.zeros((0, 3), np.float32)
    while acc.shape[0] < n_global * 1.1 and k < max_scans:
        (sx, sy, sz), yaw = pose_at(k)
        if sx > scene.length + 20.0:
            break
        p = scan_torch(scene, (sx, sy, sz), yaw, rings_m, az_m, seed * 7919 + 11 + k)
        c, s = math.cos(yaw), math.sin(yaw)
        w = np.empty_like(p)
        w[:, 0] = c * p[:, 0] - s * p[:, 1] + sx
        w[:, 1] = s * p[:, 0] + c * p[:, 1] + sy
        w[:, 2] = p[:, 2] + sz
        raw.append(w)
        k += 1
        n_scans = k
        if k % 4 == 0 or k < 4:
            acc = _voxel_thin(np.concatenate([acc] + raw[-(4 if k % 4 == 0 else 1):], 0) if k >= 4
                              else np.concatenate(raw, 0), vox)
    thin = _voxel_thin(np.concatenate(raw, 0), vox) if acc.shape[0] < n_global or k % 4 else acc
    if thin.shape[0] < n_global:
        allp = np.concatenate(raw, 0)
        while thin.shape[0] < n_global and vox > 0.004:  # not enough voxels: thin less
            vox *= 0.7
            thin = _voxel_thin(allp, vox)
    if thin.shape[0] < n_global:
        raise RuntimeError(f"scene too small for a {n_global}-point map ({thin.shape[0]} after {n_scans} scans)")
    sel = np.sort(rng.choice(thin.shape[0], n_global, replace=False))
    glob = np.ascontiguousarray(thin[sel])
    rng.shuffle(glob, axis=0)  # a map has no scan order
    # the local scan: middle of the stretch, its own noise
    (sx, sy, sz), yaw = pose_at(n_scans // 2)
    rings_l, az_l = rings_for(int(n_local * 1.25))
    loc = scan_torch(scene, (sx + 0.37 * spacing, sy, sz), yaw, rings_l, az_l, seed * 7919 + 5)
    if loc.shape[0] > n_local:
        loc = loc[np.linspace(0, loc.shape[0] - 1, n_local).astype(np.int64)]
    loc = np.ascontiguousarray(loc)
    n_out = int(outlier_frac * loc.shape[0])
    if n_out:
        lo, hi = loc.min(0), loc.max(0)
        idx = rng.choice(loc.shape[0], n_out, replace=False)
        loc[idx] = rng.uniform(lo, hi, (n_out, 3)).astype(np.float32)
    T_gt = se3.from_xyzypr(sx + 0.37 * spacing, sy, sz, yaw, 0.0, 0.0)
    T_init = se3.compose(T_gt, se3.from_xyzypr(*perturbation(seed + 3, max_t, max_r_deg)))
    d = dict(local=loc, glob=glob, T_gt=T_gt, T_init=T_init, n_scans=n_scans, voxel=vox)
    try:
        np.savez(cache, **d)
    except Exception:
        pass
    return d
