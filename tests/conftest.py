import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import qpc_pkg  # noqa: E402

qpc_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: a LONG gpu test (a larger sample, more seeds, a second variant of something the fast set covers): "
                                       "left out of `-m gpu`, run with `-m gpu_slow` (or `-m \"gpu or gpu_slow\"`); always marked gpu as well")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` is what the driver runs inside a time limit: the tests that are also marked gpu_slow are deselected unless the marker
    expression names gpu_slow itself.  (The fast set keeps every SURVEY 8 row and every BASELINE config: tests/README_GPU_SUITE.md.)"""
    if "gpu_slow" in (config.option.markexpr or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_slow") is not None else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def pkg():
    return qpc_pkg.load()
