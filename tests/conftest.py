import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import qpc_pkg  # noqa: E402

qpc_pkg.load()


_SESSION_JIT_DIR = None


def pytest_configure(config):
    # One code-object cache for the SESSION (qpc_jit.cpp, QPC_JIT_CACHE_DIR): a fresh, empty directory, removed at the end.  Every distinct
    # controller is still compiled by hiprtc once per run; the checks that start child processes (tests/gpu_torch_checks.py, the bench
    # ranks) and repeat a controller the session has built read it back instead of spending 5 s each on the same compilation.
    global _SESSION_JIT_DIR
    if "QPC_JIT_CACHE_DIR" not in os.environ:
        import tempfile
        try:
            _SESSION_JIT_DIR = tempfile.mkdtemp(prefix="qpc_jit_session_")
            os.environ["QPC_JIT_CACHE_DIR"] = _SESSION_JIT_DIR
        except OSError:
            _SESSION_JIT_DIR = None  # (no writable temporary directory: every process compiles for itself, as before)
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: a LONG gpu test (a larger sample, more seeds, a second variant of something the fast set covers): "
                                       "left out of `-m gpu`, run with `-m gpu_slow` (or `-m \"gpu or gpu_slow\"`); always marked gpu as well")


def pytest_unconfigure(config):
    global _SESSION_JIT_DIR
    if _SESSION_JIT_DIR:
        import shutil
        shutil.rmtree(_SESSION_JIT_DIR, ignore_errors=True)
        os.environ.pop("QPC_JIT_CACHE_DIR", None)
        _SESSION_JIT_DIR = None


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
