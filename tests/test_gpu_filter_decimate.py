"""GPU parity of FilterDecimateVoxels (csrc/filter_decimate.hip through the C ABI) against the CPU
oracle: the same points, bit for bit, in the same (std::map) order, for the three deterministic
DecimateMethods, flatten_to, negative coordinates (truncation toward zero makes the voxels that
touch 0 twice as wide), and the demo configurations."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
METHODS = ["DecimateMethod::FirstPoint", "DecimateMethod::ClosestToAverage", "DecimateMethod::VoxelAverage"]


@pytest.fixture(scope="module")
def amd():
    import mp2p_icp_amd
    return mp2p_icp_amd


def _run(amd, pts, params, extra_layers=None):
    mm = amd.metric_map_t({"raw": amd.PointLayer(pts)})
    for k, v in (extra_layers or {}).items():
        mm.layers[k] = amd.PointLayer(v)
    f = amd.FilterDecimateVoxels()
    f.initialize(params)
    f.filter(mm)
    return mm.layers[params["output_pointcloud_layer"]].xyz()


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("res", [0.05, 0.5, 2.0])
def test_random_cloud_vs_oracle(amd, oracle, method, res):
    rng = np.random.default_rng(10 * method + int(res * 100))
    pts = rng.normal(0, 3.0, (60_000, 3)).astype(np.float32)     # both signs: cells around 0
    pts[100:200] = pts[0:100]                                        # duplicated points
    want, wsrc = oracle.filter_decimate_voxels(pts[:, 0], pts[:, 1], pts[:, 2], res, method)
    got = _run(amd, pts, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "decimated",
                          "decimate_method": METHODS[method], "voxel_filter_resolution": res})
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and through the raw entry point: source indices
    from mp2p_icp_amd import core
    xyz, src = core.filter_decimate_voxels(amd.default_context(), pts[:, 0], pts[:, 1], pts[:, 2], res, method)
    assert np.array_equal(src, wsrc)
    if method != 2:
        assert np.array_equal(xyz, pts[src])


def test_definition_small(amd, oracle):
    """against a dictionary restatement of the reference loop (PointCloudToVoxelGrid.cpp:57-92)"""
    rng = np.random.default_rng(4)
    pts = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    res = np.float32(0.25)
    vox = {}
    for i, p in enumerate(pts):
        key = tuple(int(v) for v in (p / res).astype(np.float32).astype(np.int32))  # trunc toward zero
        vox.setdefault(key, []).append(i)
    first = np.array([pts[vox[k][0]] for k in sorted(vox)])
    got = _run(amd, pts, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "d",
                          "decimate_method": METHODS[0], "voxel_filter_resolution": float(res)})
    assert np.array_equal(got, first)
    mean = []
    for k in sorted(vox):
        m = np.zeros(3, np.float32)
        for i in vox[k]:
            m = (m + pts[i]).astype(np.float32)
        mean.append((m * np.float32(np.float32(1.0) / np.float32(len(vox[k])))).astype(np.float32))
    got = _run(amd, pts, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "d",
                          "decimate_method": METHODS[2], "voxel_filter_resolution": float(res)})
    assert np.array_equal(got, np.array(mean))


def test_flatten_minimum_points_and_layers(amd, oracle):
    rng = np.random.default_rng(9)
    pts = rng.uniform(-5, 5, (20_000, 3)).astype(np.float32)
    want, _ = oracle.filter_decimate_voxels(pts[:, 0], pts[:, 1], pts[:, 2], 1.0, 1, flatten_to=0.5)
    got = _run(amd, pts, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "flat",
                          "decimate_method": METHODS[1], "voxel_filter_resolution": 1.0, "flatten_to": 0.5})
    assert np.array_equal(got, want) and np.all(got[:, 2] == np.float32(0.5))
    assert len(got) == len({(int(a), int(b)) for a, b in (pts[:, :2] / np.float32(1.0)).astype(np.int32)})
    # a small layer is passed through unfiltered, a missing one raises (or is skipped on request)
    small = pts[:50]
    out = _run(amd, small, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "d",
                            "decimate_method": METHODS[0], "voxel_filter_resolution": 5.0,
                            "minimum_input_points_to_filter": 100})
    assert np.array_equal(out, small)
    with pytest.raises(RuntimeError):
        _run(amd, pts, {"input_pointcloud_layer": ["raw", "nope"], "output_pointcloud_layer": "d",
                        "decimate_method": METHODS[0], "voxel_filter_resolution": 1.0})
    # FirstPoint over two layers = one grid fed layer after layer
    a, b = pts[:9000], pts[9000:]
    want, _ = oracle.filter_decimate_voxels(pts[:, 0], pts[:, 1], pts[:, 2], 1.0, 0)
    got = _run(amd, a, {"input_pointcloud_layer": ["raw", "second"], "output_pointcloud_layer": "d",
                        "decimate_method": METHODS[0], "voxel_filter_resolution": 1.0}, {"second": b})
    assert np.array_equal(got, want)
    with pytest.raises(NotImplementedError):
        f = amd.FilterDecimateVoxels()
        f.initialize({"input_pointcloud_layer": "raw", "output_pointcloud_layer": "d",
                      "decimate_method": "DecimateMethod::RandomPoint", "voxel_filter_resolution": 1.0})


def test_demo_configurations(amd, oracle):
    """demos/icp-settings-kitti.yaml:78-83 (2.0 m, FirstPoint) on a KITTI-shape scan and
    demos/icp-settings-example1.yaml:60-72 (0.01, ClosestToAverage) on the bunny"""
    from mp2p_icp_amd import synthetic
    d = synthetic.make_pair(120_000, 200_000, 2001)
    scan = d["local"]
    want, _ = oracle.filter_decimate_voxels(scan[:, 0], scan[:, 1], scan[:, 2], 2.0, 0)
    got = _run(amd, scan, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "decimated",
                           "decimate_method": METHODS[0], "voxel_filter_resolution": 2.0})
    assert np.array_equal(got, want) and 100 < len(got) < len(scan) / 10
    with gzip.open(os.path.join(HERE, "golden", "bunny_decim.xyz.gz"), "rt") as f:
        bunny = np.loadtxt(f, dtype=np.float32)[:, :3]
    want, _ = oracle.filter_decimate_voxels(bunny[:, 0], bunny[:, 1], bunny[:, 2], 0.01, 1)
    got = _run(amd, bunny, {"input_pointcloud_layer": "raw", "output_pointcloud_layer": "decimated",
                            "decimate_method": METHODS[1], "voxel_filter_resolution": 0.01})
    assert np.array_equal(got, want)
    # the decimated layer is what the matcher then runs on (same pipeline shape as the demos)
    mmG = amd.metric_map_t({"decimated": amd.PointLayer(d["glob"])})
    mmL = amd.metric_map_t({"decimated": amd.PointLayer(_run(amd, scan, {
        "input_pointcloud_layer": "raw", "output_pointcloud_layer": "decimated",
        "decimate_method": METHODS[0], "voxel_filter_resolution": 0.5}))})
    m = amd.Matcher_Points_DistanceThreshold()
    m.initialize({"threshold": 2.0, "thresholdAngularDeg": 0.0})
    pairs = amd.Pairings()
    assert m.match(mmG, mmL, d["T_gt"], amd.MatchContext(), amd.MatchState(mmG, mmL), pairs)
    assert len(pairs.paired_pt2pt) > 1000
