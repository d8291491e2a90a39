"""K blocks in closed form (one friction coefficient, Solver::k_row_closed) and from the static eigen-decomposition table
(qpc_tick.h Solver::kr_blocks / ky_phase, D.k_tab) against the general
Gauss-Jordan form of the same matrices ((H_FF + eps I)^-1 per contact): QPC_NO_KTAB=1 at finalize selects the latter.
Both are the same matrix to rounding, so the working-set search takes the same path: statuses and change counts are
equal, change counts equal up to ties decided by rounding, and tau agrees to 1e-8 of its scale (median 1e-12).  Controllers the table does not cover (N != 4, per-QP surfaces, different mu per
contact) must not select it."""
import os

import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import workloads as W


def _run(factory, B, config, no_table, no_closed=False):
    """no_table: Gauss-Jordan form; no_closed: the eigen-decomposition table also where the closed form (one friction
    coefficient, Solver::k_row_closed) would be taken"""
    want = {"QPC_NO_KTAB": no_table, "QPC_K_CLOSED": not no_closed and not no_table}
    old = {k: os.environ.pop(k, None) for k in want}
    for k, on in want.items():
        if on:
            os.environ[k] = "1"
    try:
        c, wl = W.atlas_workload(factory, B, config)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    r = c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    r["k_tables"] = c.L.qpc_debug_k_tables(c.h)
    return r


def _same(a, b):
    assert a["k_tables"] >= 1 and b["k_tables"] == 0
    # the same QPs are solved; HOW an infeasible one fails (INFEASIBLE / ITERATION_LIMIT on the cycle guard) may depend on rounding
    assert np.array_equal(a["status"] == 0, b["status"] == 0), (np.where((a["status"] == 0) != (b["status"] == 0))[0], a["status"][a["status"] != b["status"]], b["status"][a["status"] != b["status"]])
    assert np.mean(a["status"] != b["status"]) <= 0.01
    ok = a["status"] == 0
    assert ok.sum() >= len(ok) // 2
    # the same sequence of working sets (a tie decided differently by rounding is allowed for a few QPs)
    assert np.mean(a["iters"][ok] % 1000 != b["iters"][ok] % 1000) <= 0.05
    scale = np.maximum(1.0, np.abs(b["tau"][ok]).max(axis=1, keepdims=True))
    # relative to the QP's torque scale: config 5 holds single-contact QPs with |tau| ~ 3e4 N m that are degenerate at the
    # optimum; two roundings of the same K then differ by up to 1.5e-9 of that scale (measured on 4 096 states; typical 1e-14)
    err = float(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max())
    assert err <= 1e-8, err
    assert float(np.median(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max(axis=1))) <= 1e-12


@pytest.mark.parametrize("no_closed", [False, True])
@pytest.mark.parametrize("config", [3, 4, 5])
def test_table_equals_gauss_jordan_emulated(config, no_closed):
    """no_closed = False: the closed-form rows of K (one friction coefficient, opt-in: QPC_K_CLOSED=1); True: the
    eigen-decomposition table (the default)"""
    emu.build()
    emu.race_report()
    _same(_run(emu.factory, 48, config, False, no_closed), _run(emu.factory, 48, config, True))
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def _mixed_mu(factory):
    """controllers whose contacts alternate between two friction coefficients (one table per distinct mu)"""
    def make(mech, N, fl):
        c = factory(mech, N, fl)
        real, count = c.add_contact, [0]

        def add_contact(body, position, normal, mu, *a, **kw):
            count[0] += 1
            return real(body, position, normal, mu * (0.5 if count[0] % 2 else 1.0), *a, **kw)
        c.add_contact = add_contact
        return c
    return make


def test_two_friction_coefficients_emulated():
    emu.build()
    emu.race_report()
    f = _mixed_mu(emu.factory)
    a, b = _run(f, 48, 5, False), _run(f, 48, 5, True)
    assert a["k_tables"] == 2
    _same(a, b)
    plain = _run(emu.factory, 48, 5, False)
    assert not np.allclose(a["tau"][a["status"] == 0], plain["tau"][a["status"] == 0])  # the second mu really is in effect
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.gpu_slow  # (5 s of hiprtc; the emulated twin runs on the CPU, the fast set keeps test_table_equals_gauss_jordan_gpu)
def test_two_friction_coefficients_gpu():
    f = _mixed_mu(lambda mech, N, fl: binding.Controller(binding.Model(mech, binding.EXPERIMENTS_LIB), N, fl))
    _same(_run(f, 2048, 5, False), _run(f, 2048, 5, True))


@pytest.mark.gpu
@pytest.mark.parametrize("no_closed", [False, True])
@pytest.mark.parametrize("config", [3, 5])
def test_table_equals_gauss_jordan_gpu(config, no_closed):
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech, binding.EXPERIMENTS_LIB), N, fl)  # noqa: E731
    _same(_run(f, 4096, config, False, no_closed), _run(f, 4096, config, True))
