"""The product's device code (qpcontrol.jl_amd/csrc/qpc_tick.h) compiled as a lane-serial CPU
emulation (tests/emul/) against the oracle on every scenario, with the emulator's LDS race
detector armed: no same-phase cross-lane RAW/WAR/WAW, no uninitialised LDS reads, no writes outside
a phase.  Runs without a GPU; the `-m gpu` parity tests run the same source on the MI355X."""
import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


@pytest.fixture(scope="module")
def emu_cases():
    return {c.name: c for c in SC.all_cases(emu.factory, small=True)}


@pytest.fixture(scope="module")
def ora_cases():
    return {c.name: c for c in SC.all_cases(ora_factory, small=True)}


NAMES = ["fixed_base_hard", "fixed_base_soft", "acrobot_config1", "acrobot_point", "free_fall", "momentum_rate",
         "spatial_constraint", "spatial_scalar", "spatial_scalar_param", "spatial_matrix", "spatial_matrix_param",
         "mixed_tasks", "mixed_tasks_fixed", "atlas_config2", "atlas_config3", "atlas_config5"]


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernel_matches_oracle(emu_cases, ora_cases, name):
    g, o = emu_cases[name], ora_cases[name]
    emu.race_report()
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
    parity.compare(g, rg, o, ro)


def test_mixed_contact_sets_32():
    """BASELINE config 5 shape at B = 32 (includes a single-active-contact QP on which plain OSQP
    stalls at max_iter; the oracle's exact mode and the kernel agree on it)."""
    g, o = SC.atlas_config(emu.factory, 5, 32), SC.atlas_config(ora_factory, 5, 32)
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf)
    parity.compare(g, rg, o, ro)
    assert (ro["status"] == 2).sum() >= 1  # infeasible elements are reported as such, not resampled
    assert np.array_equal(rg["status"] == 2, ro["status"] == 2)


def test_race_detector_detects():
    """The emulator's detector must flag a deliberately racy access pattern (self-test)."""
    import ctypes
    L = ctypes.CDLL(emu.EMU_LIB)
    if not hasattr(L, "emu_selftest_race"):
        pytest.skip("self-test hook not built")
    assert L.emu_selftest_race() > 0
