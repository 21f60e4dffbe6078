"""MRPT-free driver reproducing mp2p_icp::ICP::align (ICP.cpp:36-382): outer loop, formula
parameters, termination criteria, final covariance (ICP.cpp:334-337).  It is the *caller* of the
hot path, kept minimal: quality evaluators / checkpoints and log records are out of scope
(SURVEY.md section 2 rows 14-16)."""
import numpy as np

from . import core, se3
from .matcher import MatchContext, MatchState, Pairings, run_matchers
from .parameterizable import ParameterSource
from .solver import OptimalTF_Result, SolverContext, run_solvers


class IterTermReason:  # IterTermReason.h:26-35
    Undefined, NoPairings, SolverError, MaxIterations, Stalled, QualityCheckpointFailed, \
        HookRequest = range(7)


class Parameters:  # Parameters.h:42-100
    def __init__(self, maxIterations=40, minAbsStep_trans=5e-4, minAbsStep_rot=1e-4,
                 debugPrintIterationProgress=False):
        self.maxIterations = maxIterations
        self.minAbsStep_trans = minAbsStep_trans
        self.minAbsStep_rot = minAbsStep_rot
        self.debugPrintIterationProgress = debugPrintIterationProgress


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
            result.optimal_tf_cov = covariance(pairings, result.optimal_tf, ctx=self.ctx)  # ICP.cpp:334-337
        return result
