"""The error bound behind the tile kernel's matrix-pipe prefilter (mp2p_icp_amd/csrc/nn_query.hip, `mtol`).

The kernel evaluates S = -2 c'.q' + |c'|^2 + |q'|^2 on box-centred fp32 coordinates with fp32 MFMAs and keeps
for exact recomputation every candidate with S <= best + tol, tol = (hx^2 + hy^2 + hz^2) / 32768 where h = half
extent of the search box grown by one voxel.  That is only exact if |S - d2| <= tol for every query inside the
box and every candidate inside the grown box, whatever the order in which the hardware adds the six terms and
whether it rounds the products or not.  This test evaluates S in fp32 in all those variants on adversarial
inputs (boxes from 5 cm to 100 m, coordinates up to 8 km from the origin, points on the box faces) and checks
the bound against the kernel's exact fp32 sequence d2 = ((dx dx) + (dy dy)) + (dz dz)."""
import itertools

import numpy as np

F = np.float32


def _exact_d2(q, c):
    d = (q - c).astype(F)
    return ((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F) + (d[:, 2] * d[:, 2]).astype(F)


def test_prefilter_tolerance_covers_every_summation_order():
    rng = np.random.default_rng(7)
    worst = 0.0
    for trial in range(400):
        half = (10.0 ** rng.uniform(-1.6, 2.0, 3)).astype(F)              # 2.5 cm .. 100 m half extents
        hs = F(10.0 ** rng.uniform(-1.3, 0.5))                             # voxel edge 5 cm .. 3 m
        centre = (rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3.9)).astype(F)   # up to 8 km away
        lo, hi = (centre - half).astype(F), (centre + half).astype(F)
        o = (F(0.5) * (lo + hi)).astype(F)                                 # the kernel's centre
        h = (F(0.5) * (hi - lo) + hs).astype(F)
        tol = float((h * h).sum(dtype=F)) / 32768.0
        n = 256
        q = (lo + (hi - lo) * rng.random((n, 3)).astype(F)).astype(F)
        c = ((lo - hs) + (hi - lo + 2 * hs) * rng.random((n, 3)).astype(F)).astype(F)
        # points on the faces / corners of the two boxes: the largest terms
        k = n // 4
        q[:k] = np.where(rng.random((k, 3)) < 0.5, lo, hi)
        c[:k] = np.where(rng.random((k, 3)) < 0.5, lo - hs, hi + hs)
        q, c = np.clip(q, lo, hi).astype(F), np.clip(c, lo - hs, hi + hs).astype(F)
        cq, cc = (q - o).astype(F), (c - o).astype(F)
        nq = ((cq[:, 0] * cq[:, 0] + cq[:, 1] * cq[:, 1]).astype(F) + cq[:, 2] * cq[:, 2]).astype(F)
        nc = ((cc[:, 0] * cc[:, 0] + cc[:, 1] * cc[:, 1]).astype(F) + cc[:, 2] * cc[:, 2]).astype(F)
        d2 = _exact_d2(q, c).astype(np.float64)
        for rounded_products in (True, False):
            terms64 = [(-2.0 * cq[:, a].astype(np.float64)) * cc[:, a].astype(np.float64) for a in range(3)]
            terms = [t.astype(F).astype(np.float64) if rounded_products else t for t in terms64]
            terms += [nc.astype(np.float64), nq.astype(np.float64)]
            for order in itertools.permutations(range(5)):
                s = np.zeros(n, F)
                for i in order:                       # fp32 accumulation, one rounding per addition
                    s = (s.astype(np.float64) + terms[i]).astype(F)
                err = float(np.max(np.abs(s.astype(np.float64) - d2)))
                worst = max(worst, err / tol)
                assert err <= tol, (trial, order, rounded_products, err, tol)
    assert worst < 0.1     # measured 0.04: the bound has more than a factor 10 to spare on these inputs
