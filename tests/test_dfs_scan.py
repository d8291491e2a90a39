"""Opt-in QPC_DFS_SCAN=1 (qpc_tick.h, inverse dynamics): the wrench sum over a joint's subtree as the difference of two wave prefix sums
instead of a loop over the subtree, for models whose bodies are numbered depth-first.  Same torques up to the cancellation of the
two prefix sums (which is why it is not the default: DESIGN.md 7b), statuses and accelerations untouched."""
import os

import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import workloads as W


def _run(factory, B, config, scan):
    old = os.environ.pop("QPC_DFS_SCAN", None)
    if scan:
        os.environ["QPC_DFS_SCAN"] = "1"
    try:
        c, wl = W.atlas_workload(factory, B, config)
    finally:
        os.environ.pop("QPC_DFS_SCAN", None)
        if old is not None:
            os.environ["QPC_DFS_SCAN"] = old
    return c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)


def _same(a, b):
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"])
    assert np.array_equal(a["vdot"], b["vdot"])  # the solve is the same; only the last phase of the inverse dynamics differs
    ok = a["status"] == 0
    scale = np.maximum(1.0, np.abs(b["tau"][ok]).max(axis=1, keepdims=True))
    err = float(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max())
    assert 0.0 < err <= 1e-12, err  # (0 would mean the switch did nothing)


@pytest.mark.parametrize("config", [3, 5])
def test_prefix_scan_equals_subtree_loop_emulated(config):
    emu.build()
    emu.race_report()
    _same(_run(emu.factory, 24, config, True), _run(emu.factory, 24, config, False))
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.gpu_slow  # (5 s; an internal property of the kinematics sweep that every parity scenario exercises; its emulated twin runs on the CPU)
def test_prefix_scan_equals_subtree_loop_gpu():
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech, binding.EXPERIMENTS_LIB), N, fl)  # noqa: E731
    _same(_run(f, 2048, 3, True), _run(f, 2048, 3, False))
