"""Warm-started ticks (qpc_enable_warm_start, qpc_control_batch_warm): the counterpart of the reference's
persistent OSQP workspace -- momentum.jl:58 calls solve! on the SAME Parametron model every tick, so OSQP
starts from the previous tick's iterate.  Here the carried state is the working set (two 64-bit masks per QP).

Contract under test: the answer never depends on the hint -- tau from the exact previous set, from a
neighbouring state's set and from garbage masks equals the cold tau to 1e-9, statuses are
equal, and from the exact set the search needs (almost) no working-set changes."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import workloads as W


def _make(factory, B, config):
    real = binding.Controller.finalize

    def fin(self, device_id=0):
        self.enable_warm_start()
        return real(self, device_id)
    binding.Controller.finalize = fin
    try:
        return W.atlas_workload(factory, B, config)
    finally:
        binding.Controller.finalize = real


def _check(factory, B, config):
    c, wl = _make(factory, B, config)
    args = (wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    cold = c.control_batch(*args)
    ok = cold["status"] == 0
    assert ok.sum() >= B // 2

    def same_tau(r):
        # 1e-9 relative to the size of each QP's torques: config 5 holds single-contact QPs with |tau| ~ 3e4 N m and
        # 2e5 N contact forces, degenerate at the optimum (two optimal working sets), where 1e-9 ABSOLUTE is 3e-14 relative
        scale = np.maximum(1.0, np.abs(cold["tau"][ok]).max(axis=1, keepdims=True))
        assert np.all(np.abs(r["tau"][ok] - cold["tau"][ok]) <= 1e-9 * scale), np.abs((r["tau"][ok] - cold["tau"][ok]) / scale).max()
    # (a) cold start through the warm entry (zero masks) -> returns the final working sets
    ws = np.zeros((B, 2), dtype=np.uint64)
    r0 = c.control_batch(*args, working_set=ws)
    assert np.array_equal(r0["status"], cold["status"])
    same_tau(r0)
    assert np.all(ws[~ok] == 0)  # a failed solve hands back "start cold"
    if config == 5:
        assert (ws[ok, 0] != 0).any()  # disabled contacts sit at their lower bound
    # (b) from the exact set: same answer, (almost) no working-set changes
    ws_exact = ws.copy()
    r1 = c.control_batch(*args, working_set=ws)
    assert np.array_equal(r1["status"], cold["status"])
    same_tau(r1)
    assert np.array_equal(ws[ok], ws_exact[ok])  # a fixed point
    it_cold, it_warm = (cold["iters"][ok] % 1000).mean(), (r1["iters"][ok] % 1000).mean()
    assert it_warm <= 0.25 * it_cold + 0.25, (it_cold, it_warm)
    # (c) from the NEIGHBOUR's set (a different state)
    ws_n = np.roll(ws_exact, 1, axis=0).copy()
    r2 = c.control_batch(*args, working_set=ws_n)
    assert np.array_equal(r2["status"], cold["status"])
    same_tau(r2)
    # (d) from garbage masks (both bits set for some rays, bits beyond nrho, upper bounds active)
    rng = np.random.default_rng(config)
    ws_g = rng.integers(0, 2 ** 63, size=(B, 2), dtype=np.uint64)
    r3 = c.control_batch(*args, working_set=ws_g)
    assert np.array_equal(r3["status"], cold["status"])
    same_tau(r3)
    return it_cold, it_warm


@pytest.mark.parametrize("config", [3, 5])
def test_warm_start_emulated(config):
    emu.build()
    emu.race_report()
    _check(emu.factory, 12, config)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_warm_entry_needs_the_flag():
    emu.build()
    c, wl = W.atlas_workload(emu.factory, 2, 3)
    with pytest.raises(binding.QpcError, match="qpc_enable_warm_start"):
        c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce,
                        working_set=np.zeros((2, 2), dtype=np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("config", [pytest.param(3, marks=pytest.mark.gpu_slow), 5])
def test_warm_start_gpu(config):
    it_cold, it_warm = _check(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 2048, config)
    print(f"config {config}: working-set changes per QP cold {it_cold:.2f} -> warm {it_warm:.2f}")


@pytest.mark.gpu
def test_cold_kernel_unaffected_by_the_warm_variant():
    """the cold kernel of a warm-enabled controller is the same code as that of a plain controller: bitwise-equal output"""
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
    cw, wl = _make(f, 512, 3)
    cc, _ = W.atlas_workload(f, 512, 3)
    a = cw.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    b = cc.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    assert np.array_equal(a["tau"], b["tau"]) and np.array_equal(a["iters"], b["iters"])
    assert cw.L.qpc_debug_warm_active(cw.h) == 1 and cc.L.qpc_debug_warm_active(cc.h) == 0
