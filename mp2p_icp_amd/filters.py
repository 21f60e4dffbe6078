"""Host-side mirror of mp2p_icp_filters::FilterDecimateVoxels (FilterDecimateVoxels.cpp:40-381) over
the HIP voxel decimation (csrc/filter_decimate.hip).  Same YAML keys, same error behaviour."""
import numpy as np

from . import _lib, core
from .metric_map import PointLayer
from .parameterizable import Parameterizable

DECIMATE_METHODS = {
    "DecimateMethod::FirstPoint": _lib.DECIMATE_FIRST_POINT,
    "DecimateMethod::ClosestToAverage": _lib.DECIMATE_CLOSEST_TO_AVERAGE,
    "DecimateMethod::VoxelAverage": _lib.DECIMATE_VOXEL_AVERAGE,
    "DecimateMethod::RandomPoint": None,
}


class FilterDecimateVoxels(Parameterizable):
    def __init__(self):
        super().__init__()
        self.input_pointcloud_layer = ["raw"]
        self.error_on_missing_input_layer = True
        self.output_pointcloud_layer = ""
        self.voxel_filter_resolution = 1.0
        self.use_tsl_robin_map = True  # accepted; the output follows the std::map order either way
        self.decimate_method = _lib.DECIMATE_FIRST_POINT
        self.flatten_to = None
        self.minimum_input_points_to_filter = 0

    def initialize(self, c):  # Parameters::load_from_yaml, FilterDecimateVoxels.cpp:40-84
        if c is None or "input_pointcloud_layer" not in c:
            raise KeyError("YAML configuration must have an entry `input_pointcloud_layer` with a "
                           "scalar or sequence.")
        cfg = c["input_pointcloud_layer"]
        self.input_pointcloud_layer = [cfg] if isinstance(cfg, str) else list(cfg)
        assert self.input_pointcloud_layer
        self.error_on_missing_input_layer = bool(c.get("error_on_missing_input_layer", True))
        if "decimate_method" not in c:
            raise KeyError("Required parameter `decimate_method` not an existing key in dictionary.")
        name = c["decimate_method"]
        if name not in DECIMATE_METHODS:
            raise ValueError(f"Unknown DecimateMethod: {name}")
        if DECIMATE_METHODS[name] is None:
            raise NotImplementedError("DecimateMethod::RandomPoint draws from mrpt::random: not offered")
        self.decimate_method = DECIMATE_METHODS[name]
        if "output_pointcloud_layer" not in c:
            raise KeyError("Required parameter `output_pointcloud_layer` not an existing key in dictionary.")
        self.output_pointcloud_layer = c["output_pointcloud_layer"]
        self.minimum_input_points_to_filter = int(c.get("minimum_input_points_to_filter", 0))
        self.declare_parameter_req(c, "voxel_filter_resolution")
        self.use_tsl_robin_map = bool(c.get("use_tsl_robin_map", True))
        self.flatten_to = float(c["flatten_to"]) if "flatten_to" in c else None

    def _use_single_grid(self):  # FilterDecimateVoxels.h:129
        return self.decimate_method == _lib.DECIMATE_FIRST_POINT

    def filter(self, inOut, ctx=None):  # :107-381
        self.checkAllParametersAreRealized()
        ctx = ctx or core.default_context()
        pcs = []
        for name in self.input_pointcloud_layer:
            if name in inOut.layers:
                pcs.append(inOut.layers[name])
            elif self.error_on_missing_input_layer:
                raise RuntimeError(f"Input layer '{name}' not found on input map.")
        assert pcs, "ASSERT_(!pcPtrs.empty())"
        assert self.output_pointcloud_layer
        out = [inOut.layers[self.output_pointcloud_layer].xyz()] if self.output_pointcloud_layer in inOut.layers else []
        # layers at or below minimum_input_points_to_filter are appended unfiltered (:157-190)
        if self.minimum_input_points_to_filter > 0:
            keep = []
            for pc in pcs:
                if pc.size() > self.minimum_input_points_to_filter:
                    keep.append(pc)
                    continue
                p = pc.xyz().copy()
                if self.flatten_to is not None:
                    p[:, 2] = self.flatten_to
                out.append(p)
            pcs = keep
        if pcs:
            if not self._use_single_grid() and len(pcs) != 1:
                raise RuntimeError("Only one input layer allowed when requiring the non-single decimating grid")
            # FirstPoint over several layers: one grid fed layer after layer (:208-215) = the
            # concatenation in layer order
            pts = np.concatenate([pc.xyz() for pc in pcs]) if len(pcs) > 1 else pcs[0].xyz()
            dec, _ = core.filter_decimate_voxels(ctx, pts[:, 0], pts[:, 1], pts[:, 2],
                                                 float(self.voxel_filter_resolution), self.decimate_method,
                                                 self.flatten_to)
            out.append(dec)
        res = np.concatenate(out) if out else np.zeros((0, 3), np.float32)
        inOut.layers[self.output_pointcloud_layer] = PointLayer(res)  # mark_as_modified: a new layer object
