"""pytest configuration.

Markers:
  gpu -- needs a real MI355X (run by the driver with `-m gpu` on a GPU box).  Everything
         else must pass on a CPU-only container (`-m "not gpu"`).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real AMD GPU (MI355X)")
    config.addinivalue_line("markers", "perf: wall-time bounds (needs a GPU and a quiet box); NOT part of `-m gpu`: a busy box must not "
                                       "turn the parity suite red -- run with `-m perf`")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords or "perf" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc
