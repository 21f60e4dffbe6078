"""mp2p_icp_amd -- MI355X (gfx950) implementation of the mp2p_icp per-iteration hot path.

The compute lives in libmp2p_hip.so (hand-written HIP, C ABI in include/mp2p_hip.h).  This
package is the host-side mirror of the reference's Matcher / Solver plugin interface for
that path; it has no CPU fallback (importing works without a GPU, computing does not).
"""
from . import _build, _lib, core, se3  # noqa: F401
from ._lib import Mp2pHipError  # noqa: F401
from .core import Context, DevicePairs, GlobalMap, LocalCloud, default_context  # noqa: F401
from .filters import FilterDecimateVoxels  # noqa: F401
from .icp import (ICP, IterTermReason, Parameters, QualityEvaluator_PairedRatio, Results,  # noqa: F401
                  covariance, evaluate_quality)
from .matcher import (MatchContext, Matcher, Matcher_Adaptive, Matcher_Point2Plane,  # noqa: F401
                      Matcher_Points_InlierRatio, Matcher_Points_DistanceThreshold, MatchState, Pairings,
                      run_matchers)
from .metric_map import PT_LAYER_RAW, PointLayer, metric_map_t  # noqa: F401
from .parameterizable import ParameterSource  # noqa: F401
from .solver import (OptimalTF_Result, PosePrior, Solver, Solver_GaussNewton,  # noqa: F401
                     Solver_Horn, SolverContext, run_solvers)

__version__ = "0.1.0"
