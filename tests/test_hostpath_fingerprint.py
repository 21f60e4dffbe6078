"""The Pairings fingerprint of the adapter's host layer (adapter/mp2p_hip_host.hpp, ListPrint): feeding a list
in chunks equals feeding it at once (two matchers appending to one Pairings), it is the stated definition
(length, last record, first 32 and every 64th record), and a swap at a sampled position changes it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_list_fingerprint(tmp_path):
    sys.path.insert(0, ROOT)
    from mp2p_icp_amd import _build
    lib_dir = os.path.join(ROOT, "mp2p_icp_amd")
    if not os.path.exists(os.path.join(lib_dir, "libmp2p_hip.so")):
        _build.build()
    exe = str(tmp_path / "fingerprint_check")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adapter"),
           os.path.join(ROOT, "tests", "cpp", "fingerprint_check.cpp"), "-o", exe, "-L" + lib_dir, "-lmp2p_hip",
           "-Wl,-rpath," + lib_dir, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_layer_cache_against_a_mock_library(tmp_path):
    """Runtime's device-object cache (LRU, byte budget, re-seen / edited layers, release hooks) driven on the CPU
    against a mock of the C-ABI calls it makes: tests/cpp/layer_cache_check.cpp"""
    exe = str(tmp_path / "layer_cache_check")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adapter"),
           os.path.join(ROOT, "tests", "cpp", "layer_cache_check.cpp"), "-o", exe, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-3000:] + r.stderr[-1000:]
