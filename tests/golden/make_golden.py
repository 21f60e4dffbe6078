#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ with THIS repo's CPU oracle
(oracle/mp2p_oracle.c).  The reference itself cannot be built in this image, so
these are oracle outputs -- labelled as such -- pinned indirectly through the reference's
known-answer tests that tests/test_oracle_kat.py restates.  bunny_decim.xyz.gz is a data file
of the reference's own tests (demos/, used by tests/test-mp2p_icp_algos.cpp).

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as orc  # noqa: E402
from mp2p_icp_amd import synthetic  # noqa: E402


def main():
    out = {}
    # ---- pt2pt matcher, two poses, with and without the angular term -------------------------
    d = synthetic.random_cloud_pair(1500, 6000, 101, outlier_frac=0.1)
    g, l = d["glob"], d["local"]
    out["pt2pt_glob"], out["pt2pt_local"] = g, l
    out["pt2pt_T_gt"], out["pt2pt_T_init"] = d["T_gt"], d["T_init"]
    for tag, T, thr, ang in (("a", d["T_init"], 0.6, 0.0), ("b", d["T_gt"], 0.25, 0.4)):
        pairs, pot = orc.match_pt2pt(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, thr, ang)
        out[f"pt2pt_{tag}_params"] = np.array([thr, ang])
        out[f"pt2pt_{tag}_pairs"] = pairs
        out[f"pt2pt_{tag}_potential"] = np.array([pot])
    # ---- Gauss-Newton on the pairs of case a ---------------------------------------------------
    prm = orc.make_gn_params(3, kernel=orc.KERNEL_GEMANMCCLURE, kernelParam=0.15)
    T, it, H, gg = orc.optimal_tf_gauss_newton(out["pt2pt_a_pairs"], None, None, d["T_init"], prm)
    out["gn_a_pose"], out["gn_a_iters"], out["gn_a_H"], out["gn_a_g"] = T, np.array([it]), H, gg
    # ---- pt2pl matcher on a piecewise-planar scene ----------------------------------------------
    rng = np.random.default_rng(202)
    n = 4000
    a = rng.uniform(-4, 4, (n, 2))
    planes = [np.c_[a[:, 0], a[:, 1], 0.02 * rng.normal(size=n)],
              np.c_[a[:, 0], 4 + 0.02 * rng.normal(size=n), a[:, 1] + 4],
              np.c_[4 + 0.02 * rng.normal(size=n), a[:, 0], a[:, 1] + 4]]
    clutter = rng.uniform(-4, 4, (1500, 3)) + np.array([0, 0, 4.0])
    g2 = np.concatenate(planes + [clutter]).astype(np.float32)
    sel = rng.choice(g2.shape[0], 1200, replace=False)
    Tp = orc.pose_from_xyzypr(0.05, -0.03, 0.04, 0.01, -0.008, 0.012)
    R = Tp[:9].reshape(3, 3)
    l2 = ((g2[sel].astype(np.float64) - Tp[9:]) @ R + rng.normal(0, 0.01, (1200, 3))).astype(np.float32)
    prm_pl = dict(distanceThreshold=0.15, searchRadius=0.35, knn=6, minimumPlanePoints=5,
                  planeEigenThreshold=0.05)
    pl, idx, pot = orc.match_pt2pl(g2[:, 0], g2[:, 1], g2[:, 2], l2[:, 0], l2[:, 1], l2[:, 2],
                                   orc.pose_identity(), **prm_pl)
    out["pt2pl_glob"], out["pt2pl_local"] = g2, l2
    out["pt2pl_params"] = np.array([prm_pl[k] for k in ("distanceThreshold", "searchRadius", "knn",
                                                        "minimumPlanePoints", "planeEigenThreshold")])
    out["pt2pl_pairs"], out["pt2pl_local_idx"], out["pt2pl_potential"] = pl, idx, np.array([pot])
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote golden_v1.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})
    return out


def main_v2(v1):
    """golden_v2.npz: the components of SURVEY.md section 8f on the scenes of v1"""
    out = {}
    g, l, T = v1["pt2pl_glob"], v1["pt2pl_local"], orc.pose_from_xyzypr(0.02, 0.01, -0.01, 0.004, 0.0, -0.002)
    out["pose"] = T
    # ---- Matcher_Adaptive (plane detection on, two point pairings per local point) -----------------
    A = dict(confidenceInterval=0.8, firstToSecondDistanceMax=1.5, absoluteMaxSearchDistance=1.0,
             minimumCorrDist=0.05, enableDetectPlanes=True, maxPt2PtCorrespondences=2, planeSearchPoints=8,
             planeMinimumFoundPoints=4, planeMinimumDistance=0.10, planeEigenThreshold=0.01)
    r = orc.match_adaptive(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, **A)
    out["adaptive_params"] = np.array([A[k] for k in sorted(A)], np.float64)
    out["adaptive_pt2pt"], out["adaptive_pt2pl"], out["adaptive_pl_idx"] = r["pt2pt"], r["pt2pl"], r["pl_local_idx"]
    out["adaptive_ci_high"] = np.array([r["ci_high"]])
    out["adaptive_hist"] = np.concatenate([[r["hist"]["minSq"], r["hist"]["maxSq"], r["hist"]["count"]],
                                           r["hist"]["bins"]]).astype(np.float64)
    # ---- Matcher_Points_InlierRatio -------------------------------------------------------------------
    ir, pot = orc.match_inlier_ratio(g[:, 0], g[:, 1], g[:, 2], l[:, 0], l[:, 1], l[:, 2], T, 0.6)
    out["inlier_ratio_pairs"] = ir
    # ---- Solver_Horn: WeightParameters on the adaptive matcher's point pairings (+ 25 gross outliers),
    #      and the pt2ln_pl_to_pt2pt conversion of its plane pairings -----------------------------------
    pt = r["pt2pt"].copy()
    rng = np.random.default_rng(7)
    bad = rng.choice(len(pt), 25, replace=False)
    pt["gx"][bad] += rng.uniform(2, 4, 25).astype(np.float32)
    Th, rc, fl = orc.optimal_tf_horn_wp(pt, None, use_scale_outlier_detector=True, scale_outlier_threshold=1.15,
                                        point_weights=[(300, 2.0), (len(pt), 0.5)])
    assert rc == 1
    out["horn_pairs"], out["horn_pose"], out["horn_outliers"] = pt, Th, np.flatnonzero(fl).astype(np.uint32)
    out["converted_pairs"] = orc.pt2ln_pl_to_pt2pt(r["pt2pl"], None, T)
    # ---- covariance() of point + plane pairings ----------------------------------------------------------
    cov, H, ok = orc.covariance(r["pt2pt"], r["pt2pl"], None, None, T)
    assert ok
    out["cov"], out["cov_H"] = cov, H
    # ---- FilterDecimateVoxels -------------------------------------------------------------------------------
    for name, method in (("first", 0), ("closest", 1), ("average", 2)):
        xyz, src = orc.filter_decimate_voxels(g[:, 0], g[:, 1], g[:, 2], 0.5, method)
        out[f"decimate_{name}_xyz"], out[f"decimate_{name}_src"] = xyz, src
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main_v2(main())
