"""Solver-independent physical checks of a whole tick, in the spirit of the reference's own tests
(test/controller.jl:92-96 inverse o forward dynamics; :276-281 achieved task accelerations):

* forward dynamics (oracle, independent of the kernel's RNEA) driven by the returned tau and the
  returned contact wrenches reproduces the returned v̇ -- Newton-Euler consistency of QP solution,
  wrench balance, extraction and inverse dynamics together;
* every contact force lies in its friction cone and pushes (f_n >= 0);
* the hard tasks are met: J v̇ + J̇v = desired for the foot / pelvis tasks."""
import numpy as np
import pytest

import emu
from oracle import oracle as O
from qpcontrol_jl_amd import workloads as W


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def check(factory, B, config, fused_laws=False):
    c, wl = W.atlas_workload(factory, B, config, fused_laws=fused_laws)
    r = c.control_batch(wl.q, wl.v, None if fused_laws else wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    oc, _ = W.atlas_workload(ora_factory, 1, config)
    om, mech, st = oc.model, wl.mech, wl.setup
    ok = np.where(r["status"] == 0)[0]
    assert len(ok) == B if config != 5 else len(ok) >= B // 2  # configs 2 and 3: every QP solves
    for b in ok:
        om.set_state(wl.q[b], wl.v[b])
        wext = np.zeros((mech.nbodies, 6))
        for ci, cp in enumerate(mech.contact_points if config != 2 else []):
            wext[cp.body] += r["cwrench"][b, ci]
            # friction cone (contacts.jl:16-23: the basis rays lie on the cone, their hull is inside it)
            _, _, R, p = oc.contact_eval(ci, cp.normal, cp.mu, wl.contact_maxnormalforce[b, ci])
            f = R.T @ r["cwrench"][b, ci, 3:]
            assert f[2] >= -1e-7 and np.hypot(f[0], f[1]) <= cp.mu * f[2] + 1e-6, (b, ci, f)
            np.testing.assert_allclose(r["cwrench"][b, ci, :3], np.cross(p, r["cwrench"][b, ci, 3:]), atol=1e-9)
        vd_fd = om.forward_dynamics(r["tau"][b], wext)
        np.testing.assert_allclose(vd_fd, r["vdot"][b], atol=5e-10, rtol=0)  # achieved: ~3e-11 (reference ID o FD bar: 1e-10 on a 3-joint arm)
        for tid in list(st.foot_tasks) + [st.pelvis_task]:  # hard tasks
            J, bias = oc.task_eval(tid)
            o = st.offsets[tid]
            np.testing.assert_allclose(J @ r["vdot"][b] + bias, wl.desired[b, o:o + len(bias)], atol=1e-10, rtol=0)  # achieved: ~3e-15
    return len(ok)


@pytest.mark.parametrize("config", [2, 3, 5])
def test_physical_consistency_emulated(config):
    emu.build()
    check(emu.factory, 6, config)


def test_physical_consistency_emulated_fused_laws():
    emu.build()
    check(emu.factory, 4, 3, fused_laws=True)


@pytest.mark.gpu
@pytest.mark.parametrize("config", [2, 3, 5])
def test_physical_consistency_gpu(config):
    from qpcontrol_jl_amd import binding
    check(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 96, config)
