"""CPU model of the point-to-plane search's certificate (nn_pt2pl.hip, PlArgs::lb_io / pt2pl_cert_kernel): the rule that
lets a query skip its k-NN search is checked against brute force on random clouds and random small moves.

Model of what the kernel keeps per query after a search that covered the ball of radius `cov` completely:
  list S = the (up to knn) nearest points within the search radius `rad`,
  lb     = min(distance of the nearest point NOT in S that the search tested, cov)
and of the rule at the next call, after the query moved by `disp`:
  lbn = lb - disp - margin;  S re-measured from the new position, members beyond `rad` dropped (m2 of m left);
  full list (m == knn):   certified iff m2 == m and max re-measured distance < lbn
  short list (m < knn):   certified iff m2 == m and lbn > rad
A certified query must have exactly the brute-force answer at the new position.  The second test shows why `m2 == m`
is part of the short-list rule (round 3: a member that drops out is an outsider NEARER than lb; without the condition
it is missed when it comes back into reach)."""
import numpy as np


def _knn(pts, q, knn, rad):
    d = np.linalg.norm(pts - q, axis=1)
    order = np.lexsort((np.arange(len(d)), d))
    sel = [i for i in order[:knn] if d[i] <= rad]
    return sel, d


def _search(pts, q, knn, rad, cov_extra):
    """what a completed search leaves: list, lb (cov = the radius it covered: the k-th distance or rad, plus a margin)"""
    sel, d = _knn(pts, q, knn, rad)
    cov = (d[sel[-1]] if len(sel) == knn else rad) + cov_extra
    outsiders = np.setdiff1d(np.arange(len(pts)), sel)
    tested = outsiders[d[outsiders] <= cov * 1.5]          # the staged region is larger than the covered ball
    nearest_rejected = d[tested].min() if len(tested) else np.inf
    return sel, min(nearest_rejected, cov)


def _certify(pts, S, lb, q_new, disp, knn, rad, margin, require_all_members=True):
    dn = np.linalg.norm(pts[S] - q_new, axis=1) if len(S) else np.zeros(0)
    keep = dn <= rad
    m, m2 = len(S), int(keep.sum())
    lbn = lb - disp - margin
    if m == knn:
        ok = m2 == m and lbn > 0 and dn.max() < lbn
    else:
        ok = (m2 == m or not require_all_members) and lbn > rad
    S2 = [S[i] for i in np.lexsort((np.asarray(S), dn)) if keep[i]]
    return ok, S2, lbn


def test_certified_lists_equal_brute_force():
    rng = np.random.default_rng(7)
    knn, rad, margin = 5, 0.4, 1e-6
    n_cert_full = n_cert_short = 0
    for trial in range(300):
        pts = rng.uniform(-1, 1, (int(rng.integers(20, 400)), 3))
        q = rng.uniform(-0.3, 0.3, 3)
        S, lb = _search(pts, q, knn, rad, cov_extra=float(rng.uniform(0.0, 0.05)))
        for step in range(6):                                   # a chain of small moves, the bound decaying each time
            move = rng.normal(0, 1, 3)
            move *= float(rng.uniform(0, 0.01)) / np.linalg.norm(move)
            q_new = q + move
            disp = float(np.linalg.norm(move))
            ok, S2, lbn = _certify(pts, S, lb, q_new, disp, knn, rad, margin)
            want, _ = _knn(pts, q_new, knn, rad)
            if ok:
                assert S2 == want, (trial, step, S2, want)
                n_cert_full += len(S) == knn
                n_cert_short += len(S) < knn
                S, lb = S2, lbn
            else:
                S, lb = _search(pts, q_new, knn, rad, cov_extra=float(rng.uniform(0.0, 0.05)))
            q = q_new
    assert n_cert_full > 100 and n_cert_short > 100, (n_cert_full, n_cert_short)   # both branches exercised


def test_short_list_needs_every_member_to_stay():
    """the hole the GPU test found: a member leaves the radius (certified with one member less), the bound is kept, the
    member returns -- and the list misses it.  With m2 == m in the rule the query is searched instead."""
    knn, rad, margin = 5, 0.4, 1e-6
    pts = np.array([[0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0], [0.3995, 0, 0], [2.0, 0, 0]])
    q0 = np.zeros(3)
    S, lb = _search(pts, q0, knn, rad, cov_extra=0.05)
    assert len(S) == 4 and lb > 0.44
    q1 = np.array([-0.001, 0, 0])                              # the 4th member is now 0.4005 away: out
    ok_bad, S_bad, lb_bad = _certify(pts, S, lb, q1, 0.001, knn, rad, margin, require_all_members=False)
    assert ok_bad and len(S_bad) == 3
    q2 = np.zeros(3)                                            # ... and back in
    ok_bad2, S_bad2, _ = _certify(pts, S_bad, lb_bad, q2, 0.001, knn, rad, margin, require_all_members=False)
    assert ok_bad2 and S_bad2 != _knn(pts, q2, knn, rad)[0]     # the flawed rule certifies a wrong list
    ok, _, _ = _certify(pts, S, lb, q1, 0.001, knn, rad, margin)
    assert not ok                                               # the rule as built sends the query to the search


# ---- the point-to-point search's certificate (round 4; nn_query.hip, NNArgs::lb2nd) ----------------------------------
def _top2_model(vals, lim, rng):
    """what the tracking build of the tile kernel inserts for one query out of the prefilter values `vals` of a pass (in
    blocks of 16 rows = 4 groups of 4), given the limit of the moment: a block without a value within the limit is stood for
    by its minimum; in a block with one, a group outside the limit by its minimum, a group within it by its four values.
    Returns the second smallest inserted value (the kernel's t2)."""
    ins = []
    vals = list(vals)
    rng.shuffle(vals)
    while len(vals) % 16:
        vals.append(1e36)  # the padding slots
    for b in range(0, len(vals), 16):
        blk = vals[b:b + 16]
        if min(blk) > lim:
            ins.append(min(blk))
            continue
        for k in range(4):
            grp = blk[4 * k:4 * k + 4]
            if min(grp) > lim:
                ins.append(min(grp))
            else:
                ins.extend(grp)
    ins.sort()
    return ins[1] if len(ins) > 1 else np.inf


def test_pt2pt_certificate_rule_against_brute_force():
    """(1) the second smallest INSERTED prefilter value is a lower bound of the second smallest of ALL values, whatever the
    order of the candidates and whatever the limit (as long as the smallest value is within it: the nearest neighbour's
    block is always entered); (2) the rule built on it -- previous neighbour re-measured < (bound - displacement - margin)
    -- only ever certifies the true, unique nearest neighbour; chains of small moves with the bound decaying."""
    rng = np.random.default_rng(11)
    n_cert = 0
    for trial in range(400):
        pts = rng.uniform(-1, 1, (int(rng.integers(3, 300)), 3))
        q = rng.uniform(-0.3, 0.3, 3)
        d = np.linalg.norm(pts - q, axis=1)
        tol = 1e-5
        S = d * d + rng.uniform(-tol, tol, len(d))              # the prefilter's values: exact d2 within its error bound
        cover = float(rng.uniform(0.3, 2.0))                    # the radius the final pass covered
        staged = d <= cover * float(rng.uniform(1.0, 1.5))      # everything within the covered ball is staged (and more)
        if not staged.any():
            continue
        best = int(np.argmin(np.where(staged, d, np.inf)))
        lim = S[best] + float(rng.uniform(0, 0.2))              # any limit the nearest neighbour's value is within
        t2 = _top2_model(S[staged], lim, rng)
        true_s2 = np.sort(S[staged])[1] if staged.sum() > 1 else np.inf
        assert t2 <= true_s2 + 1e-15
        lb = min(np.sqrt(max(t2 - tol, 0.0)), cover)            # the kernel's bound on every point but `best`
        others = np.delete(d, best)
        assert (others >= lb - 1e-12).all()
        margin = 1e-6
        for step in range(8):
            move = rng.normal(0, 1, 3)
            move *= float(rng.uniform(0, 0.004)) / np.linalg.norm(move)
            q = q + move
            disp = float(np.linalg.norm(move))
            room = lb - disp - margin
            dn = np.linalg.norm(pts - q, axis=1)
            if lb > 0 and dn[best] + margin < room:
                assert int(np.argmin(dn)) == best and (np.delete(dn, best) > dn[best]).all(), (trial, step)
                n_cert += 1
                lb = room
            else:
                break
    assert n_cert > 300, n_cert
