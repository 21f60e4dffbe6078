"""Minimal stand-in for mp2p_icp::metric_map_t (metricmap.h:64-258): named point layers.

Only what the hot path consumes: SoA float32 coordinates per layer (what
CPointsMap::getPointsBufferRef_{x,y,z} exposes) and a modification counter that invalidates
the device-side index (the role of CPointsMap::mark_as_modified())."""
import itertools

import numpy as np

from . import core

PT_LAYER_RAW = "raw"  # metric_map_t::PT_LAYER_RAW

_uid = itertools.count(1)


class PointLayer:
    """A point-cloud layer.  Device objects (NN index as a global layer, sorted copy as a local
    layer) are created lazily and cached until the layer is modified."""

    def __init__(self, x, y=None, z=None, cell_size=0.0, target_per_cell=0.0, no_occupancy_bitmap=False):
        if y is None:
            pts = np.asarray(x, dtype=np.float32)
            x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        self.x = np.ascontiguousarray(x, dtype=np.float32)
        self.y = np.ascontiguousarray(y, dtype=np.float32)
        self.z = np.ascontiguousarray(z, dtype=np.float32)
        assert self.x.size == self.y.size == self.z.size
        self.cell_size = cell_size
        self.target_per_cell = target_per_cell
        self.no_occupancy_bitmap = no_occupancy_bitmap
        self.uid = next(_uid)
        self._version = 0
        self._cache = {}

    def size(self):
        return self.x.size

    def xyz(self):
        return np.stack([self.x, self.y, self.z], 1)

    def empty(self):
        return self.x.size == 0

    def mark_as_modified(self):
        self._version += 1
        self._cache.clear()

    def as_global(self, ctx):
        key = ("g", id(ctx))
        if key not in self._cache:
            self._cache[key] = core.GlobalMap(ctx, self.x, self.y, self.z, self.cell_size,
                                              self.target_per_cell,
                                              no_occupancy_bitmap=self.no_occupancy_bitmap)
        return self._cache[key]

    def as_local(self, ctx):
        key = ("l", id(ctx))
        if key not in self._cache:
            self._cache[key] = core.LocalCloud(ctx, self.x, self.y, self.z)
        return self._cache[key]


class metric_map_t:
    PT_LAYER_RAW = PT_LAYER_RAW

    def __init__(self, layers=None):
        self.layers = dict(layers or {})

    def empty(self):
        return all(l.empty() for l in self.layers.values()) if self.layers else True
