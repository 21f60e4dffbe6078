"""Wall-time bounds, kept out of the parity suite (`-m gpu`): run with `python -m pytest tests -m perf` on a quiet GPU box.
They re-run two GPU tests with their timing assertions switched on (MP2P_PERF_ASSERTS=1)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.perf
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_filter_decimate_wall_time(monkeypatch):
    monkeypatch.setenv("MP2P_PERF_ASSERTS", "1")
    m = _load("test_gpu_scratch")
    import mp2p_icp_amd as amd
    m.test_decimation_of_one_million_points_device_time(amd)


def test_host_path_wall_time(monkeypatch, oracle):
    monkeypatch.setenv("MP2P_PERF_ASSERTS", "1")
    m = _load("test_gpu_boundary_hostpath")
    m.test_host_path_cost_at_full_size(oracle)
