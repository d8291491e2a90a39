"""L1 (qpc_assemble_solve_batch): assembly + solve from blocks the ORACLE evaluates on the host
(task Jacobians/biases, momentum matrix, contact frames) against the oracle's own tick.  It is the
dense one-lane-per-QP path, independent of the fused kernel; the last test cross-checks the two
device paths against each other."""
import numpy as np
import pytest

import emu
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import binding


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def oracle_blocks(o):
    """what task_error / ContactPoint / add_wrench_balance_constraint! evaluate from the state"""
    oc, om, mech = o.ctrl, o.ctrl.model, o.mech
    lay = oc.layout
    nt, nc, nv, nd = lay.n_tasks, lay.n_contacts, lay.nv, lay.n_desired
    J, bias = np.zeros((o.B, nd, nv)), np.zeros((o.B, nd))
    Am, mrhs = np.zeros((o.B, 6, nv)), np.zeros((o.B, 6))
    cR, cp = np.zeros((o.B, max(nc, 1), 9)), np.zeros((o.B, max(nc, 1), 3))
    for b in range(o.B):
        om.set_state(o.q[b], o.v[b])
        r = 0
        for t in range(nt):
            Jt, bt = oc.task_eval(t)
            d = len(bt)
            J[b, r:r + d], bias[b, r:r + d] = Jt, bt
            r += d
        A, Adv = om.momentum_matrix()
        fg = mech.total_mass() * np.array(mech.gravity)
        Wg = np.concatenate([np.cross(om.center_of_mass(), fg), fg])
        Am[b], mrhs[b] = A, Wg - Adv
        k = 0
        for ci, (normal, mu, per_qp) in enumerate(getattr(oc, "contact_specs", [])):
            if per_qp:
                normal, mu = o.surf[b, 4 * k:4 * k + 3], o.surf[b, 4 * k + 3]
                k += 1
            _, _, R, p = oc.contact_eval(ci, normal, mu, o.cmaxf[b, ci])
            cR[b, ci], cp[b, ci] = R.reshape(-1), p
    return J, bias, Am, mrhs, cR[:, :nc], cp[:, :nc]


def run_l1(case, o):
    J, bias, Am, mrhs, cR, cp = oracle_blocks(o)
    fl = case.floating >= 0
    return binding.assemble_solve_batch(case.ctrl, J, bias, case.desired, Am if fl else None, mrhs if fl else None,
                                        cR, cp, case.cweight, case.cmaxf, case.tw, case.surf)


NAMES = ["fixed_base_hard", "fixed_base_soft", "acrobot_point", "free_fall", "momentum_rate", "spatial_matrix_param",
         "mixed_tasks", "mixed_tasks_fixed", "atlas_config2", "atlas_config3", "atlas_config5"]


def _check(factory, name):
    case = next(iter(SC.all_cases(factory, small=True, only=name)))
    o = next(iter(SC.all_cases(ora_factory, small=True, only=name)))
    r = run_l1(case, o)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
    concl = (ro["status"] == 0) | (ro["status"] == 2)
    assert np.array_equal((r["status"] == 0)[concl], (ro["status"] == 0)[concl]), (r["status"], ro["status"])
    ok = ro["status"] == 0
    assert ok.any()
    if np.isfinite(case.tol_tau):  # v̇ unique (regularised)
        np.testing.assert_allclose(r["vdot"][ok], ro["vdot"][ok], atol=max(case.tol_tau, 1e-8), rtol=0)
        np.testing.assert_allclose(r["cwrench"][ok], ro["cwrench"][ok], atol=1e-6, rtol=0)
    return case, r


@pytest.mark.parametrize("name", NAMES)
def test_l1_emulated_matches_oracle(name):
    emu.build()
    _check(emu.factory, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acrobot_point", "momentum_rate", "mixed_tasks", "atlas_config3", "atlas_config5"])
def test_l1_gpu_matches_oracle_and_fused_kernel(name):
    def gpu_factory(mech, n_basis, floating):
        return binding.Controller(binding.Model(mech), n_basis, floating)
    case, r = _check(gpu_factory, name)
    rf = case.ctrl.control_batch(case.q, case.v, case.desired, case.cweight, case.cmaxf, case.tw, case.surf)
    both = (r["status"] == 0) & (rf["status"] == 0)
    assert np.array_equal(r["status"] == 0, rf["status"] == 0)
    if np.isfinite(case.tol_tau):
        np.testing.assert_allclose(r["vdot"][both], rf["vdot"][both], atol=1e-8, rtol=0)
        np.testing.assert_allclose(r["cwrench"][both], rf["cwrench"][both], atol=1e-6, rtol=0)
