"""MRPT-free driver reproducing mp2p_icp::ICP::align (ICP.cpp:36-382): outer loop, formula
parameters, termination criteria, quality (QualityEvaluator_PairedRatio, checkpoints) and final
covariance (ICP.cpp:316-337).  It is the *caller* of the hot path, kept minimal: the other
quality evaluators and log records are out of scope."""
import numpy as np

from . import core, se3
from .matcher import (MatchContext, Matcher_Points_DistanceThreshold, MatchState, Pairings,
                      run_matchers)
from .parameterizable import ParameterSource
from .solver import OptimalTF_Result, SolverContext, run_solvers


class IterTermReason:  # IterTermReason.h:26-35
    Undefined, NoPairings, SolverError, MaxIterations, Stalled, QualityCheckpointFailed, \
        HookRequest = range(7)


class Parameters:  # Parameters.h:42-100
    def __init__(self, maxIterations=40, minAbsStep_trans=5e-4, minAbsStep_rot=1e-4,
                 debugPrintIterationProgress=False, quality_checkpoints=None):
        self.maxIterations = maxIterations
        self.minAbsStep_trans = minAbsStep_trans
        self.minAbsStep_rot = minAbsStep_rot
        self.debugPrintIterationProgress = debugPrintIterationProgress
        # iteration -> minimum quality (Parameters.h: {50: 0.05, 100: 0.10})
        self.quality_checkpoints = {50: 0.05, 100: 0.10} if quality_checkpoints is None else dict(quality_checkpoints)


class QualityEvaluator_PairedRatio:
    """QualityEvaluator_PairedRatio.cpp:27-73: matched / potential pairings, either of the ICP's last
    pairings (reuse_icp_pairings, the default) or of a fresh Matcher_Points_DistanceThreshold run
    that may pair a global point several times."""

    def __init__(self):
        self.matcher_ = Matcher_Points_DistanceThreshold()
        self.reuse_icp_pairings = True
        self.absolute_minimum_pairing_ratio = 0.20

    def initialize(self, params):
        params = dict(params or {})
        self.reuse_icp_pairings = bool(params.get("reuse_icp_pairings", True))
        self.absolute_minimum_pairing_ratio = float(params.get("absolute_minimum_pairing_ratio", 0.20))
        if not self.reuse_icp_pairings:
            params.setdefault("allowMatchAlreadyMatchedGlobalPoints", True)
            for k in ("reuse_icp_pairings", "absolute_minimum_pairing_ratio"):
                params.pop(k, None)
            self.matcher_.initialize(params)

    def evaluate(self, pcGlobal, pcLocal, localPose, pairingsFromICP):
        """-> (quality, hard_discard)"""
        if self.reuse_icp_pairings:
            pairings = pairingsFromICP
        else:
            pairings = Pairings()
            self.matcher_.match(pcGlobal, pcLocal, localPose, MatchContext(), MatchState(pcGlobal, pcLocal),
                                pairings)
        n = pairings.potential_pairings if pairings is not None else 0
        q = (pairings.size() / float(n)) if n else 0.0
        return q, q < self.absolute_minimum_pairing_ratio


def evaluate_quality(evaluators, pcGlobal, pcLocal, localPose, finalPairings):  # ICP.cpp:608-634
    assert evaluators
    sw = se = 0.0
    for obj, w in evaluators:
        assert w > 0
        q, hard = obj.evaluate(pcGlobal, pcLocal, localPose, finalPairings)
        if hard:
            return 0.0
        se += w * q
        sw += w
    return se / sw


class Results:  # Results.h
    def __init__(self):
        self.optimal_tf = se3.identity()
        self.nIterations = 0
        self.terminationReason = IterTermReason.Undefined
        self.finalPairings = None
        self.quality = 0.0
        self.optimal_tf_cov = None  # CPose3DPDFGaussian::cov of Results::optimal_tf (6x6)


def covariance(pairings, finalAlignSolution, finDif_xyz=1e-7, finDif_angles=1e-7, ctx=None):
    """mp2p_icp::covariance (covariance.cpp:29-141) on the pairings' device lists."""
    ctx = pairings.ctx or ctx or core.default_context()
    dev = pairings.device
    if dev is None:
        dev = pairings._ensure_dev(ctx, 1, 0)
    if len(pairings.paired_ln2ln):
        raise NotImplementedError("paired_ln2ln is not supported")
    n_ln, n_pp = len(pairings.paired_pt2ln), len(pairings.paired_pl2pl)
    if n_ln or n_pp or getattr(dev, "_has_lines_planes", False):
        dev.upload_lines_planes(pairings.paired_pt2ln, pairings.paired_pl2pl)
        dev._has_lines_planes = bool(n_ln or n_pp)
    cov, _, _ = core.covariance(ctx, dev, finalAlignSolution, finDif_xyz, finDif_angles)
    return cov


class ICP:
    def __init__(self, ctx=None):
        self.ctx = ctx or core.default_context()
        self.matchers_ = []
        self.solvers_ = []
        self.ownParamSource_ = ParameterSource()
        self.iteration_hook_ = None
        self.quality_evaluators_ = [(QualityEvaluator_PairedRatio(), 1.0)]  # ICP.h: the default list

    def set_quality_evaluators(self, lst):
        """[(evaluator, relativeWeight)]"""
        self.quality_evaluators_ = list(lst)

    def set_matchers(self, m):
        self.matchers_ = list(m)

    def set_solvers(self, s):
        self.solvers_ = list(s)

    def align(self, pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, prior=None):
        assert self.matchers_ and self.solvers_
        assert not pcGlobal.empty() and not pcLocal.empty()
        result = Results()
        sources = set()

        def add_own(obj):
            ps = obj.attachedSource()
            if ps is None:
                obj.attachToParameterSource(self.ownParamSource_)
                ps = self.ownParamSource_
            ps.updateVariable("ICP_ITERATION", result.nIterations)
            sources.add(ps)

        ms_pose = np.asarray(initialGuessLocalWrtGlobal, dtype=np.float64)
        initGuess = ms_pose if ms_pose.size == 12 else se3.from_xyzypr(*ms_pose)
        cur = OptimalTF_Result()
        cur.optimalPose = initGuess.copy()
        prev_solution = cur.optimalPose.copy()
        prev2_solution = None
        lastCorrection = None
        sc = SolverContext()
        sc.prior = prior
        pairings = Pairings(self.ctx)

        it = 0
        while it < p.maxIterations:  # ICP.cpp:123
            result.nIterations = it
            for o in self.matchers_ + self.solvers_:
                add_own(o)
            for ps in sources:
                ps.realize()
            mc = MatchContext(it)
            ms = MatchState(pcGlobal, pcLocal, self.ctx)
            pairings = run_matchers(self.matchers_, pcGlobal, pcLocal, cur.optimalPose, mc, ms,
                                    out=pairings)  # :143-144
            if pairings.empty():
                result.terminationReason = IterTermReason.NoPairings
                break
            sc.icpIteration = it
            sc.guessRelativePose = cur.optimalPose.copy()
            sc.currentCorrectionFromInitialGuess = se3.inverse_compose(cur.optimalPose, initGuess)
            sc.lastIcpStepIncrement = lastCorrection
            if not run_solvers(self.solvers_, pairings, cur, sc):  # :170-171
                result.terminationReason = IterTermReason.SolverError
                break

            def incrs(d):
                xi = se3.log(d)
                return float(np.linalg.norm(xi[:3])), float(np.linalg.norm(xi[3:]))

            deltaSol = se3.inverse_compose(cur.optimalPose, prev_solution)
            lastCorrection = deltaSol
            dxyz, drot = incrs(deltaSol)
            if prev2_solution is not None:
                d2, r2 = incrs(se3.inverse_compose(cur.optimalPose, prev2_solution))
                dxyz, drot = min(dxyz, d2), min(drot, r2)
            if p.debugPrintIterationProgress:
                print(f"[ICP] Iter={it:3d} dt={dxyz:9.2e}, dR={np.degrees(drot):6.3f} deg, "
                      f"pairs={pairings.contents_summary()}")
            if abs(dxyz) < p.minAbsStep_trans and abs(drot) < p.minAbsStep_rot:  # :228-229
                result.terminationReason = IterTermReason.Stalled
                break
            if it in p.quality_checkpoints:  # ICP.cpp:258-284
                q = evaluate_quality(self.quality_evaluators_, pcGlobal, pcLocal, cur.optimalPose, pairings)
                if q < p.quality_checkpoints[it]:
                    result.terminationReason = IterTermReason.QualityCheckpointFailed
                    break
            if self.iteration_hook_ is not None and self.iteration_hook_(it, pairings, cur):
                result.terminationReason = IterTermReason.HookRequest
                break
            prev2_solution = prev_solution
            prev_solution = cur.optimalPose.copy()
            it += 1
            result.nIterations = it
        if result.nIterations >= p.maxIterations:
            result.terminationReason = IterTermReason.MaxIterations
        result.optimal_tf = cur.optimalPose
        result.finalPairings = pairings
        if pairings is not None:
            result.quality = evaluate_quality(self.quality_evaluators_, pcGlobal, pcLocal, result.optimal_tf,
                                              pairings)  # ICP.cpp:316-325
            result.optimal_tf_cov = covariance(pairings, result.optimal_tf, ctx=self.ctx)  # ICP.cpp:334-337
        return result
