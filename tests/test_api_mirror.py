"""Host mirror of the reference API (qpcontrol.jl_amd/api.py) driven end-to-end on CPU through the
emulated device code: same names / call sequence as the reference notebook
(notebooks/Standing controller.ipynb cells 5-8) and its error behaviour (QPSolveFailure,
additions after the first tick are rejected)."""
import numpy as np
import pytest

import emu
from oracle import oracle as O
from qpcontrol_jl_amd import api, binding, models as M, workloads as W


@pytest.fixture(scope="module")
def controller():
    emu.build()
    mech = M.atlas()
    c = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world", lib_path=emu.EMU_LIB)
    contacts = []
    for cp in mech.contact_points:  # notebook cell 5
        pt = c.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce = 1e6
        pt.weight = 1e-3
        contacts.append(pt)
    for j in range(mech.nbodies):
        c.regularize(j, 0.05)
    feet = [api.SpatialAccelerationTask(mech, "world", f) for f in ("l_foot", "r_foot")]
    for t in feet:
        c.addtask(t)
    linmom = api.LinearMomentumRateTask(mech)
    c.addtask(linmom, 1.0)
    pelvis = api.AngularAccelerationTask(mech, "world", "pelvis")
    c.addtask(pelvis)
    on_path = set(mech.path_joints(M.WORLD, mech.findbody("l_foot"))) | set(mech.path_joints(M.WORLD, mech.findbody("r_foot")))
    joints = {}
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE and j not in on_path:
            joints[j] = api.JointAccelerationTask(mech, j)
            c.addtask(joints[j])
    return mech, c, contacts, feet, linmom, pelvis, joints


def test_standing_tick_matches_oracle(controller):
    mech, c, contacts, feet, linmom, pelvis, joints = controller
    B = 5
    q, v = W.standing_states(mech, B, 11)

    def ofactory(mech_, N, fl):
        return O.OracleController(O.OracleModel(mech_), N, fl)
    oc, _ = W.atlas_workload(ofactory, 1, 3)
    st = W.standing_setup(_Null(), mech, M.atlas_nominal_configuration(mech))
    des = W.standing_desired(st, q, v)
    api.setdesired(linmom, des[:, st.offsets[st.linmom_task]:st.offsets[st.linmom_task] + 3])
    api.setdesired(pelvis, des[:, st.offsets[st.pelvis_task]:st.offsets[st.pelvis_task] + 3])
    for j, t in joints.items():
        api.setdesired(t, des[:, st.offsets[st.joint_tasks[j]]:st.offsets[st.joint_tasks[j]] + 1])
    tau = np.zeros((B, mech.nv))
    out = c(tau, 0.0, (q, v))
    assert out is tau and c.result_vdot.shape == (B, mech.nv) and c.pointwrenches.shape == (B, 8, 6)
    ro = oc.tick_batch(q, v, des, np.full((B, 8), 1e-3), np.full((B, 8), 1e6))
    ok = ro["status"] == 0
    np.testing.assert_allclose(tau[ok], ro["tau"][ok], atol=1e-8, rtol=0)
    # contactwrenches[BodyID(body)] (momentum.jl:64-72): one wrench per foot, the sum over the foot's four contact points -- against
    # the oracle's per-contact wrenches summed by body
    feet_ids = [mech.findbody("l_foot"), mech.findbody("r_foot")]
    assert sorted(c.contactwrenches) == sorted(feet_ids) and all(w.shape == (B, 6) for w in c.contactwrenches.values())
    body_of = [cp.body for cp in mech.contact_points]
    for f in feet_ids:
        want = sum(ro["cwrench"][:, k] for k in range(8) if body_of[k] == f)
        np.testing.assert_allclose(c.contactwrenches[f][ok], want[ok], atol=1e-8 * max(1.0, np.abs(want[ok]).max()), rtol=0)
    # ... and Newton-Euler: the bodies' wrenches together balance gravity and the momentum rate -- the forces' sum has the robot's weight's size
    total = sum(c.contactwrenches.values())
    assert np.all(total[ok][:, 5] > 0.5 * mech.total_mass() * 9.81)


def test_disable_and_failure(controller):
    mech, c, contacts, feet, linmom, pelvis, joints = controller
    q, v = W.standing_states(mech, 2, 12)
    for t in [linmom, pelvis] + list(joints.values()):
        api.setdesired(t, np.zeros(t.dim))  # one value for every state
    for pt in contacts:
        api.disable(pt)
    assert not api.isenabled(contacts[0])
    with pytest.raises(api.QPSolveFailure):  # floating base + hard foot tasks + no contact force: infeasible
        c(None, 0.0, (q, v))
    tau = c(None, 0.0, (q, v), throw_on_failure=False)
    assert np.all(c.status == 2) and tau.shape == (2, mech.nv)
    for pt in contacts:
        pt.maxnormalforce = 1e6


def test_additions_after_first_tick_are_rejected(controller):
    mech, c, *_ = controller
    with pytest.raises(binding.QpcError):
        c.addtask(api.MomentumRateTask(mech))
    with pytest.raises(binding.QpcError):
        c.regularize(1, 1.0)


class _Null:
    """Controller stand-in that only hands out ids (to reuse workloads.standing_setup for offsets)."""

    def __init__(self):
        self.n = 0

    def add_task(self, *a, **k):
        self.n += 1
        return self.n - 1

    def add_contact(self, *a, **k):
        return 0

    def regularize(self, *a, **k):
        pass


def test_standing_controller_mirror_matches_oracle():
    """api.StandingController == the reference's highlevel/standing.jl object: same constructor
    arguments, tick = (tau, t, state) -> tau with the PD laws evaluated on the device."""
    emu.build()
    mech = M.atlas()
    low = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world", lib_path=emu.EMU_LIB)
    for cp in mech.contact_points:  # notebook cell 5
        pt = low.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce, pt.weight = 1e6, 1e-3
    ctrl = api.StandingController(low, ["l_foot", "r_foot"], "pelvis", M.atlas_nominal_configuration(mech))
    B = 4
    q, v = W.standing_states(mech, B, 12)
    tau = ctrl(np.zeros((B, mech.nv)), 0.0, (q, v))
    assert low.layout.n_desired_inputs == 0

    def ofactory(mech_, N, fl):
        return O.OracleController(O.OracleModel(mech_), N, fl)
    oc, _ = W.atlas_workload(ofactory, 1, 3)
    st = W.standing_setup(_Null(), mech, M.atlas_nominal_configuration(mech))
    ro = oc.tick_batch(q, v, W.standing_desired(st, q, v), np.full((B, 8), 1e-3), np.full((B, 8), 1e6))
    ok = ro["status"] == 0
    assert ok.any()
    np.testing.assert_allclose(tau[ok], ro["tau"][ok], atol=1e-8, rtol=0)
