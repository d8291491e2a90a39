"""Poisoned records in a batch (NaN / Inf / absurd magnitudes in q, v, desired, contact weights and caps): the reference
would throw QPSolveFailure for THAT state (momentum.jl:83-91); the batched kernel must report a non-OPTIMAL status for it,
terminate (a persistent kernel whose search never ends would hang the device), and leave every other QP of the batch
bitwise unchanged.  Degenerate-but-legal inputs (zero contact weight, zero cap = disable!, contacts.jl:72) must still solve."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import workloads as W


def _run(factory, B):
    c, wl = W.atlas_workload(factory, B, 3)
    args = [wl.q.copy(), wl.v.copy(), wl.desired.copy(), wl.contact_weight.copy(), wl.contact_maxnormalforce.copy()]
    clean = c.control_batch(*args)
    assert (clean["status"] == 0).all()
    q, v, d, cw, cm = args
    bad = {}
    q[0, 5] = np.nan; bad[0] = "nan in q"
    v[1, 3] = np.inf; bad[1] = "inf in v"
    d[2, 0] = np.nan; bad[2] = "nan in desired"
    cw[3, 2] = np.nan; bad[3] = "nan contact weight"
    cm[4, 1] = np.nan; bad[4] = "nan contact cap"
    q[5, :4] = 0.0; bad[5] = "zero quaternion"
    v[6, :] = 1e200; bad[6] = "absurd velocity"
    d[7, :] = -1e300; bad[7] = "absurd desired"
    cw[8, :] = -1.0; bad[8] = "negative contact weights"
    cm[11, 3] = -5.0; bad[11] = "negative contact cap"
    legal = {}
    cw[9, :] = 0.0; legal[9] = "zero contact weights"
    cm[10, :4] = 0.0; legal[10] = "half of the contacts disabled"
    r = c.control_batch(q, v, d, cw, cm)
    for b, what in bad.items():
        # (a negative weight makes the QP non-convex: anything but a silent OPTIMAL with a NaN inside is acceptable)
        if r["status"][b] in (0, 1):
            assert np.isfinite(r["tau"][b]).all(), what
            assert what in ("negative contact weights", "zero quaternion", "absurd velocity", "absurd desired"), (what, r["status"][b])
        assert r["status"][b] in (0, 1, 2, 3, 4, 5), what
    for b, what in legal.items():
        assert r["status"][b] in (0, 1), (what, r["status"][b])
        assert np.isfinite(r["tau"][b]).all(), what
    rest = [b for b in range(B) if b not in bad and b not in legal]
    assert np.array_equal(r["tau"][rest], clean["tau"][rest])
    assert np.array_equal(r["status"][rest], clean["status"][rest])
    assert r["status"][11] == 2  # 0 <= rho <= rho_max < 0: INFEASIBLE, as the reference's solver would say
    nan_rows = [b for b, what in bad.items() if "nan" in what or "inf" in what]
    assert all(r["status"][b] not in (0, 1) for b in nan_rows), r["status"][nan_rows]
    return r


def test_bad_inputs_emulated():
    emu.build()
    _run(emu.factory, 16)


@pytest.mark.gpu
def test_bad_inputs_gpu():
    r = _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 4096)
    print("statuses of the poisoned records:", r["status"][:12])
