"""adapter/mp2p_hip_plugin.cpp cannot be BUILT in this image (MRPT and mp2p_icp are absent), but it can
be syntax- and type-checked: g++ -fsyntax-only against declaration-only stand-ins of the MRPT /
mp2p_icp interfaces it touches (tests/adapter_stubs/).  Where /root/reference exists, the headers of
the reference that are self-contained given those MRPT stand-ins -- pointcloud_bitfield.h (whose
PRIVATE member the plugin reaches), NearestPlaneCapable.h, point_plane_pair_t.h, plane_patch.h,
layer_name_t.h, robust_kernels.h -- replace their stand-ins, so the plugin is checked against the real
declarations of exactly the types it converts."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "adapter_stubs")
PLUGIN = os.path.join(ROOT, "adapter", "mp2p_hip_plugin.cpp")
REF = "/root/reference"
REAL = {  # header -> directory of the reference that holds it
    "pointcloud_bitfield.h": "mp2p_icp_map/include/mp2p_icp",
    "NearestPlaneCapable.h": "mp2p_icp_map/include/mp2p_icp",
    "point_plane_pair_t.h": "mp2p_icp_map/include/mp2p_icp",
    "plane_patch.h": "mp2p_icp_map/include/mp2p_icp",
    "layer_name_t.h": "mp2p_icp_map/include/mp2p_icp",
    "robust_kernels.h": "mp2p_icp/include/mp2p_icp",
}


def _gxx():
    return shutil.which("g++") or pytest.skip("g++ not available")


def _check(src, extra_inc=()):
    cmd = [_gxx(), "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=return-type"]
    for d in extra_inc:
        cmd += ["-I", d]
    cmd += ["-I", STUBS, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"), src]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_plugin_compiles_against_the_stand_ins():
    r = _check(PLUGIN)
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_check_is_live():
    """a misspelt member of a reference type must be caught (guards against a stand-in that accepts anything)"""
    src = open(PLUGIN).read()
    with tempfile.TemporaryDirectory() as d:
        for old, new in (("out.paired_pt2pt)", "out.paired_pt2pt_typo)"),
                         ("&BitField::dense_>", "&BitField::dense>"),
                         ("sc.prior->cov_inv(i, j)", "sc.prior->cov_inverse(i, j)"),
                         ("outPc->insertPointFrom(*todo[li], i)", "outPc->insertPointFromLayer(*todo[li], i)"),
                         ("mrpt::math::confidenceIntervalsFromHistogram(xs,", "mrpt::math::confidenceIntervalFromHistogram(xs,"),
                         ("class FilterDecimateVoxels : public mp2p_icp_filters::FilterBase", "class FilterDecimateVoxels : public mp2p_icp_filters::FilterBaze"),
                         ("n_pairs = pairingsFromICP.size(), potential", "n_pairs = pairingsFromICP.sizes(), potential"),
                         ("class QualityEvaluator_PairedRatio : public mp2p_icp::QualityEvaluator", "class QualityEvaluator_PairedRatio : public mp2p_icp::QualityEvaluatr"),
                         ("mp2p_icp::MatchState ms(pcGlobal, pcLocal);  // :59", "mp2p_icp::MatchState ms(pcGlobal);  // :59")):
            assert old in src
            p = os.path.join(d, "bad.cpp")
            open(p, "w").write(src.replace(old, new, 1))
            r = _check(p)
            assert r.returncode != 0, (old, new)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on this machine")
def test_plugin_compiles_against_the_reference_headers_where_self_contained():
    with tempfile.TemporaryDirectory() as d:
        inc = os.path.join(d, "mp2p_icp")
        os.makedirs(inc)
        for h, sub in REAL.items():
            real = os.path.join(REF, sub, h)
            assert os.path.exists(real), real
            os.symlink(real, os.path.join(inc, h))  # a link for the compiler, never a copy in the repo
        r = _check(PLUGIN, extra_inc=(d,))
        assert r.returncode == 0, r.stderr[-4000:]
        # and the link really took precedence: the real pointcloud_bitfield.h defines its methods inline
        probe = os.path.join(d, "probe.cpp")
        open(probe, "w").write('#include <mp2p_icp/pointcloud_bitfield.h>\n'
                               'int main(){ mp2p_icp::pointcloud_bitfield_t b; mp2p_icp::metric_map_t m; '
                               'b.initialize_from(m); return 0; }\n')
        assert _check(probe, extra_inc=(d,)).returncode == 0
        assert _check(probe).returncode != 0  # the stand-in has no initialize_from


def test_host_layer_compiles_standalone():
    """adapter/mp2p_hip_host.hpp needs nothing but the C ABI header"""
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.cpp")
        open(p, "w").write('#include "mp2p_hip_host.hpp"\nint main(){return 0;}\n')
        r = subprocess.run([_gxx(), "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I",
                            os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"), p],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
