"""Host-side mirror of the reference's Solver plugin interface for the hot path.

  Solver / SolverContext   mp2p_icp/src/Solver.cpp:28-64, Solver.h:43-102
  Solver_GaussNewton       Solver_GaussNewton.cpp:29-67 -> optimal_tf_gauss_newton
  Solver_Horn              Solver_Horn.cpp:33-61 -> pt2ln_pl_to_pt2pt, optimal_tf_horn (WeightParameters)
  run_solvers              ICP.cpp:469-479
"""
import ctypes as C

import numpy as np

from . import _lib, core, se3
from .parameterizable import Parameterizable

ROBUST_KERNELS = {"RobustKernel::None": _lib.KERNEL_NONE, "None": _lib.KERNEL_NONE,
                  "RobustKernel::GemanMcClure": _lib.KERNEL_GEMANMCCLURE,
                  "GemanMcClure": _lib.KERNEL_GEMANMCCLURE,
                  "RobustKernel::Cauchy": _lib.KERNEL_CAUCHY, "Cauchy": _lib.KERNEL_CAUCHY}


class OptimalTF_Result:  # OptimalTF_Result.h:32-39
    def __init__(self):
        self.optimalPose = se3.identity()
        self.optimalScale = 1.0
        self.outliers = []
        self.gn = None  # extra: last normal equations / iteration count of the GN solver


class PosePrior:
    """SolverContext::prior (CPose3DPDFGaussianInf): mean pose + 6x6 information matrix"""

    def __init__(self, mean, cov_inv):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.cov_inv = np.asarray(cov_inv, dtype=np.float64).reshape(6, 6)


class SolverContext:  # Solver.h:43-62
    def __init__(self):
        self.guessRelativePose = None
        self.currentCorrectionFromInitialGuess = None
        self.lastIcpStepIncrement = None
        self.icpIteration = None
        self.prior = None
        self.perSolverPersistentData = {}


class PairWeights:  # PairWeights.h:34-52
    def __init__(self):
        self.pt2pt = self.pt2ln = self.pt2pl = self.ln2ln = self.pl2pl = 1.0

    def load_from(self, p):  # PairWeights.cpp:26-34: all five required
        for k in ("pt2pt", "pt2pl", "pt2ln", "ln2ln", "pl2pl"):
            if k not in p:
                raise KeyError(f"Required parameter `{k}` not an existing key in dictionary.")
            setattr(self, k, float(p[k]))


class Solver(Parameterizable):
    def __init__(self):
        super().__init__()
        self.runFromIteration = 0
        self.runUpToIteration = 0
        self.enabled = True
        self.runUntilTranslationCorrectionSmallerThan = 0.0

    def initialize(self, params):
        params = params or {}
        self.runFromIteration = int(params.get("runFromIteration", 0))
        self.runUpToIteration = int(params.get("runUpToIteration", 0))
        self.enabled = bool(params.get("enabled", True))
        self.runUntilTranslationCorrectionSmallerThan = float(
            params.get("runUntilTranslationCorrectionSmallerThan", 0.0))

    def optimal_pose(self, pairings, out, sc):  # Solver.cpp:36-64
        if not self.enabled:
            return False
        it = sc.icpIteration
        if it is not None and it < self.runFromIteration:
            return False
        if it is not None and self.runUpToIteration > 0 and it > self.runUpToIteration:
            return False
        if self.runUntilTranslationCorrectionSmallerThan > 0:
            my = sc.perSolverPersistentData.setdefault(id(self), {})
            if "finished" in my:
                return False
            if sc.lastIcpStepIncrement is not None and np.linalg.norm(
                    sc.lastIcpStepIncrement[9:12]) < self.runUntilTranslationCorrectionSmallerThan:
                my["finished"] = True
                return False
        return self.impl_optimal_pose(pairings, out, sc)


class Solver_GaussNewton(Solver):
    def __init__(self):
        super().__init__()
        self.maxIterations = 5
        self.innerLoopVerbose = False
        self.robustKernel = _lib.KERNEL_NONE
        self.robustKernelParam = 1.0
        self.pairWeights = PairWeights()

    def initialize(self, params):  # Solver_GaussNewton.cpp:29-40
        super().initialize(params)
        if params is None or "maxIterations" not in params:
            raise KeyError("Required parameter `maxIterations` not an existing key in dictionary.")
        self.maxIterations = int(params["maxIterations"])
        self.innerLoopVerbose = bool(params.get("innerLoopVerbose", False))
        rk = params.get("robustKernel", "RobustKernel::None")
        if rk not in ROBUST_KERNELS:
            raise ValueError(f"Unknown kernel type: {rk}")
        self.robustKernel = ROBUST_KERNELS[rk]
        self.declare_parameter_opt(params, "robustKernelParam")
        if "pair_weights" in params:
            self.pairWeights.load_from(params["pair_weights"])

    def gn_params(self, sc, point_weights=None):
        p = _lib.GNParams()
        p.maxInnerLoopIterations = self.maxIterations
        p.minDelta, p.maxCost = 1e-7, 0.0  # optimal_tf_gauss_newton.h:46-58
        p.kernel, p.kernelParam = self.robustKernel, float(self.robustKernelParam)
        p.w_pt2pt, p.w_pt2pl = self.pairWeights.pt2pt, self.pairWeights.pt2pl
        p.w_pt2ln, p.w_pl2pl = self.pairWeights.pt2ln, self.pairWeights.pl2pl
        p.has_prior = 0
        if sc.prior is not None:
            p.has_prior = 1
            p.prior_mean = (C.c_double * 12)(*sc.prior.mean)
            p.prior_cov_inv = (C.c_double * 36)(*sc.prior.cov_inv.ravel())
        p.n_weight_blocks = 0
        if point_weights:
            if len(point_weights) > _lib.MAX_WEIGHT_BLOCKS:
                raise ValueError("at most 32 point_weights blocks are supported")
            p.n_weight_blocks = len(point_weights)
            for i, (cnt, w) in enumerate(point_weights):
                p.weight_block_count[i] = int(cnt)
                p.weight_block_w[i] = float(w)
        return p

    def impl_optimal_pose(self, pairings, out, sc):  # Solver_GaussNewton.cpp:42-67
        self.checkAllParametersAreRealized()
        if sc.guessRelativePose is None:
            raise RuntimeError("ASSERT_(sc.guessRelativePose.has_value())")
        ctx = pairings.ctx or core.default_context()
        dev = pairings.device
        if dev is None:
            dev = pairings._ensure_dev(ctx, 1, 0)
        if len(pairings.paired_ln2ln):
            raise NotImplementedError("paired_ln2ln (optimal_tf_gauss_newton.cpp:204-225) is not supported")
        n_ln, n_pp = len(pairings.paired_pt2ln), len(pairings.paired_pl2pl)
        if n_ln or n_pp or getattr(dev, "_has_lines_planes", False):
            dev.upload_lines_planes(pairings.paired_pt2ln, pairings.paired_pl2pl)
            dev._has_lines_planes = bool(n_ln or n_pp)
        res = core.gn_solve(ctx, dev, sc.guessRelativePose, self.gn_params(sc, pairings.point_weights))
        out.__init__()
        out.optimalPose = np.array(res.pose)
        out.gn = core.gn_result_to_dict(res)
        return True  # optimal_tf_gauss_newton always returns true (:369)


class WeightParameters:
    """WeightParameters.h:34-72; load_from = WeightParameters.cpp:47-55"""

    def __init__(self):
        self.use_scale_outlier_detector = False
        self.scale_outlier_threshold = 1.20
        self.pair_weights = PairWeights()
        self.robust_kernel = _lib.KERNEL_NONE
        self.currentEstimateForRobust = None
        self.robust_kernel_param = 1.0

    def load_from(self, p):
        for k in ("use_scale_outlier_detector", "robust_kernel"):  # MCP_LOAD_REQ
            if k not in p:
                raise KeyError(f"Required parameter `{k}` not an existing key in dictionary.")
        self.use_scale_outlier_detector = bool(p["use_scale_outlier_detector"])
        self.scale_outlier_threshold = float(p.get("scale_outlier_threshold", self.scale_outlier_threshold))
        rk = p["robust_kernel"]
        if rk not in ROBUST_KERNELS:
            raise ValueError(f"Unknown kernel type: {rk}")
        self.robust_kernel = ROBUST_KERNELS[rk]
        self.robust_kernel_param = float(p.get("robust_kernel_param", self.robust_kernel_param))
        if "pair_weights" in p:
            self.pair_weights.load_from(p["pair_weights"])

    def to_lib(self):
        w = _lib.HornParams()
        w.use_scale_outlier_detector = int(self.use_scale_outlier_detector)
        w.scale_outlier_threshold = float(self.scale_outlier_threshold)
        w.w_pt2pt, w.w_ln2ln, w.w_pl2pl = self.pair_weights.pt2pt, self.pair_weights.ln2ln, self.pair_weights.pl2pl
        w.robust_kernel, w.robust_kernel_param = int(self.robust_kernel), float(self.robust_kernel_param)
        if self.currentEstimateForRobust is not None:
            w.has_current_estimate = 1
            w.current_estimate[:] = [float(v) for v in self.currentEstimateForRobust]
        return w


def optimal_tf_horn(pairings, wp, result):
    """optimal_tf_horn.cpp:201-252 on a Pairings (point pairings in HBM, paired_pl2pl uploaded);
    result.outliers = indices of the point pairings the scale detector discarded"""
    ctx = pairings.ctx or core.default_context()
    result.__init__()
    if len(pairings.paired_ln2ln):
        raise NotImplementedError("paired_ln2ln is not supported")
    dev = pairings.device
    if dev is None:
        dev = pairings._ensure_dev(ctx, 1, 0)
    n_pt, n_pl, _ = dev.counts()
    if n_pl or len(pairings.paired_pt2ln):  # visit_correspondences.h:55-56
        raise RuntimeError("This solver cannot handle point-to-plane / point-to-line pairings yet.")
    n_pp = len(pairings.paired_pl2pl)
    if n_pp or getattr(dev, "_has_lines_planes", False):
        dev.upload_lines_planes(None, pairings.paired_pl2pl)
        dev._has_lines_planes = bool(n_pp)
    T, ok, n_out = core.horn_solve_wp(ctx, dev, wp.to_lib(), pairings.point_weights)
    if ok:
        result.optimalPose = T
        if n_out:
            result.outliers = np.flatnonzero(core.horn_outlier_flags(ctx, n_pt)).tolist()
    return ok


def pt2ln_pl_to_pt2pt(pairings, sc, reuse=None):
    """pt2ln_pl_to_pt2pt.cpp:47-113 -> a Pairings holding only the converted point pairings (`reuse`: the
    container of an earlier conversion on the same context, so that an ICP loop allocates it once)"""
    from .matcher import Pairings
    if sc.guessRelativePose is None:
        raise RuntimeError("ASSERT_(sc.guessRelativePose.has_value())")
    ctx = pairings.ctx or core.default_context()
    dev = pairings.device
    if dev is None:
        dev = pairings._ensure_dev(ctx, 1, 0)
    n_ln = len(pairings.paired_pt2ln)
    if n_ln or getattr(dev, "_has_lines_planes", False):
        dev.upload_lines_planes(pairings.paired_pt2ln, pairings.paired_pl2pl)
        dev._has_lines_planes = bool(n_ln or len(pairings.paired_pl2pl))
    n_pl = dev.counts()[1]
    out = reuse if (reuse is not None and reuse.ctx is ctx and reuse is not pairings) else Pairings(ctx, max(1, n_pl + n_ln), 0)
    odev = out._ensure_dev(ctx, max(1, n_pl + n_ln), 0)
    if out is reuse:
        odev.clear()  # the conversion appends to an empty list (pt2ln_pl_to_pt2pt.cpp:49)
    core.pairs_pt2ln_pl_to_pt2pt(ctx, dev, sc.guessRelativePose, odev)
    out._ub = [n_pl + n_ln, 0]
    return out


class Solver_Horn(Solver):
    """Solver_Horn.cpp:33-61: WeightParameters from `pairingsWeightParameters`; point-to-plane /
    point-to-line pairings go through pt2ln_pl_to_pt2pt first."""

    def __init__(self):
        super().__init__()
        self.pairingsWeightParameters = WeightParameters()

    @property
    def pairWeights(self):
        return self.pairingsWeightParameters.pair_weights

    def initialize(self, params):
        super().initialize(params)
        params = params or {}
        if "pairingsWeightParameters" in params:
            self.pairingsWeightParameters.load_from(params["pairingsWeightParameters"])

    def impl_optimal_pose(self, pairings, out, sc):
        if pairings.device is None and not len(pairings.paired_pl2pl) and not len(pairings.paired_pt2ln):
            out.__init__()
            return False
        eff = pairings
        n_pl = pairings.device.counts()[1] if pairings.device is not None else 0
        if n_pl or len(pairings.paired_pt2ln):  # :51-55
            eff = self._converted = pt2ln_pl_to_pt2pt(pairings, sc, reuse=getattr(self, "_converted", None))
        return optimal_tf_horn(eff, self.pairingsWeightParameters, out)


def run_solvers(solvers, pairings, out, sc):  # ICP.cpp:469-479
    for s in solvers:
        assert s is not None
        if s.optimal_pose(pairings, out, sc):
            return True
    return False
