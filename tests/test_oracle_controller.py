"""Restates reference test/controller.jl against the oracle (reference-layout QP + OSQP
restatement): the invariants that pin the oracle in place of golden vectors."""
import numpy as np
import pytest

from oracle import oracle as O
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W
from qpcontrol_jl_amd.workloads import (HARD, JOINT, MATRIX, MATRIX_PER_QP, MOMENTUM_RATE, SCALAR,
                                        SCALAR_PER_QP, SPATIAL)


def humanoid(seed=0):
    """Valkyrie stand-in: the synthetic Atlas (ValkyrieRobot is not available)."""
    return M.atlas()


def test_reference_layout_sizes():
    """143 variables, 114 equality + 64 inequality rows for the Atlas notebook."""
    def factory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech), N, fl)
    oc, _ = W.atlas_workload(factory, 1, 3)
    L = oc.layout
    assert (L.ref_n_var, L.ref_n_eq, L.ref_n_ineq, L.n_desired) == (143, 114, 64, 36)
    oc2, _ = W.atlas_workload(factory, 1, 2)
    assert (oc2.layout.ref_n_var, oc2.layout.ref_n_eq, oc2.layout.ref_n_ineq) == (39, 36, 0)


def test_qp_structure_matches_appendix_a():
    def factory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech), N, fl)
    oc, wl = W.atlas_workload(factory, 1, 3)
    P, q, A, l, u = oc.build_qp(wl.q[0], wl.v[0], wl.desired[0], wl.contact_weight[0], wl.contact_maxnormalforce[0])
    d = np.diag(P)
    assert np.allclose(d[:36], 0.1) and np.allclose(d[-3:], 2.0) and np.count_nonzero(P - np.diag(d)) == 0
    assert np.count_nonzero(d) == 63 and np.all(q == 0)
    # maxrho quirk: maxnormalforce / (N sqrt(mu^2+1))
    rho_rows = [r for r in range(len(l)) if u[r] < 1e29 and l[r] < -1e29]
    assert len(rho_rows) == 32 and np.allclose(u[rho_rows], 1e6 / (4 * np.sqrt(0.8**2 + 1)))


@pytest.mark.parametrize("constrained", [True, False])
def test_fixed_base_joint_space_control(constrained):  # test/controller.jl:68-97
    rng = np.random.default_rng(42)
    mech = M.rand_tree([M.PRISMATIC, M.REVOLUTE, M.REVOLUTE], seed=42)
    om = O.OracleModel(mech)
    oc = O.OracleController(om, 4, -1)
    for j in range(mech.nbodies):
        oc.add_task(JOINT, joint=j, wkind=HARD if constrained else SCALAR, weight=None if constrained else [1.0])
    oc.finalize()
    desired = rng.uniform(size=mech.nv)
    q, v = M.rand_state(mech, rng)
    r = oc.tick(q, v, desired)
    assert r["status"] == 0
    om.set_state(q, v)
    np.testing.assert_allclose(om.forward_dynamics(r["tau"]), desired, atol=1e-10)


def test_zero_velocity_free_fall():  # test/controller.jl:128-165
    mech = humanoid()
    om = O.OracleModel(mech)
    fl = mech.findbody("pelvis")
    oc = O.OracleController(om, 4, fl)
    for cp in mech.contact_points:
        oc.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        if j != fl:
            oc.regularize(j, 1.0)
    oc.finalize()
    rng = np.random.default_rng(5354)
    q, _ = M.rand_state(mech, rng)
    nc = len(mech.contact_points)
    r = oc.tick(q, np.zeros(mech.nv), (), np.zeros(nc), np.zeros(nc))  # defaults weight = maxnormalforce = 0
    assert r["status"] == 0
    vd = r["vdot"]
    np.testing.assert_allclose(vd[6:], 0, atol=1e-4)
    om.set_state(q, np.zeros(mech.nv))
    acc = om.spatial_accelerations(vd)[fl]
    R, p = om.transform(fl)
    np.testing.assert_allclose(R.T @ acc[:3], 0, atol=1e-4)
    # linear acceleration of the pelvis origin (v = 0): a_lin_world - p x alpha ... alpha = 0
    np.testing.assert_allclose(acc[3:] + np.cross(acc[:3], p), mech.gravity, atol=1e-4)
    np.testing.assert_allclose(r["cwrench"], 0, atol=1e-6)


def test_achievable_momentum_rate():  # test/controller.jl:169-230 (mixed active contact sets)
    mech = humanoid()
    om = O.OracleModel(mech)
    fl = mech.findbody("pelvis")
    rng = np.random.default_rng(533454)
    oc = O.OracleController(om, 4, fl)
    nc = len(mech.contact_points)
    for cp in mech.contact_points:
        oc.add_contact(cp.body, cp.position, cp.normal, cp.mu, per_qp_surface=1)
    oc.add_task(MOMENTUM_RATE, wkind=HARD)
    for j in range(mech.nbodies):
        oc.regularize(j, 1e-6)
    oc.finalize()
    normals = rng.normal(size=(nc, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    mus = rng.uniform(size=nc)
    surf = np.concatenate([normals, mus[:, None]], axis=1).reshape(-1)
    for p_active in np.linspace(0, 1, 5):
        q, v = M.rand_state(mech, rng)
        om.set_state(q, v)
        com = om.center_of_mass()
        fg = mech.total_mass() * np.array(mech.gravity)
        hdes = np.concatenate([np.zeros(3), fg])
        cw, cmax = np.zeros(nc), np.zeros(nc)
        for ci, cp in enumerate(mech.contact_points):
            if rng.uniform() < p_active:
                n, mu = normals[ci], mus[ci]
                cw[ci], cmax[ci] = 1e-6, 1e9
                fn = 50.0 * rng.uniform()
                mured = np.sqrt(2) / 2 * mu
                t = rng.normal(size=3)
                ft = mured * fn * rng.uniform() * np.cross(n, t / np.linalg.norm(t))
                f = fn * n + ft
                R, pb = om.transform(cp.body)
                fw = R @ f
                pw = R @ cp.position + pb
                hdes += np.concatenate([np.cross(pw - com, fw), fw])
        r = oc.tick(q, v, hdes, cw, cmax, surf=surf)
        assert r["status"] == 0, (p_active, r["status"])
        A, Adv = om.momentum_matrix()
        h = A @ r["vdot"] + Adv
        hc = np.concatenate([h[:3] - np.cross(com, h[3:]), h[3:]])
        np.testing.assert_allclose(hc, hdes, atol=1e-3)
        assert np.all(r["tau"][:6] == 0)


@pytest.mark.parametrize("mode", ["constraint", "scalar", "scalar_param", "matrix", "matrix_param"])
def test_spatial_acceleration_modes(mode):  # test/controller.jl:232-285
    mech = humanoid()
    om = O.OracleModel(mech)
    fl = mech.findbody("pelvis")
    rng = np.random.default_rng(533)
    q, v = M.rand_state(mech, rng)
    oc = O.OracleController(om, 4, fl)
    body, base = mech.findbody("l_foot"), mech.findbody("r_hand")
    tw = None
    if mode == "constraint":
        oc.add_task(SPATIAL, base=base, target=body, frame=base, wkind=HARD)
        for j in range(mech.nbodies):
            oc.regularize(j, 1.0)
    elif mode == "scalar":
        oc.add_task(SPATIAL, base=base, target=body, frame=base, wkind=SCALAR, weight=[1.0])
    elif mode == "scalar_param":
        oc.add_task(SPATIAL, base=base, target=body, frame=base, wkind=SCALAR_PER_QP)
        tw = [1.0]
    elif mode == "matrix":
        oc.add_task(SPATIAL, base=base, target=body, frame=base, wkind=MATRIX, weight=np.eye(6))
    else:
        oc.add_task(SPATIAL, base=base, target=body, frame=base, wkind=MATRIX_PER_QP)
        tw = np.eye(6).reshape(-1)
    oc.finalize()
    desired = rng.uniform(size=6)
    r = oc.tick(q, v, desired, tw=tw)
    assert r["status"] == 0
    om.set_state(q, v)
    acc = om.spatial_accelerations(r["vdot"])
    rel = om.transform_accel(acc[body] - acc[base], body, base, base)
    np.testing.assert_allclose(rel, desired, atol=1e-8)


def test_parameterized_contact_normal():  # test/controller.jl:22-66
    mech = M.Mechanism()
    mech.add_body("body", M.WORLD, "rx", M.REVOLUTE, (1, 0, 0), mass=10.0, inertia=(1, 1, 1, 0, 0, 0))
    om = O.OracleModel(mech)
    oc = O.OracleController(om, 4, -1)
    oc.add_contact(0, np.zeros(3), np.array([0, 0, 1.0]), 1.0, per_qp_surface=1)
    oc.finalize()
    for th in np.linspace(-np.pi, np.pi, 10):
        om.set_state([th], [0.0])
        R, _ = om.transform(0)
        normal_body = R.T @ np.array([0, 0, 1.0])
        B, maxrho, Rc, pc = oc.contact_eval(0, normal_body, 1.0, 1e3)
        np.testing.assert_allclose(Rc @ np.array([0, 0, 1.0]), [0, 0, 1.0], atol=1e-12)
        assert maxrho > 0
        np.testing.assert_allclose(np.linalg.norm(B, axis=0), 1.0, atol=1e-15)


def test_infeasible_is_reported():
    """Floating base, hard foot/pelvis/joint tasks and no contacts: 39 equality rows on 36 unknowns
    (SURVEY 8d config 2 note) -> the status must not be OPTIMAL (QPSolveFailure in the reference)."""
    def factory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech), N, fl)
    oc, wl = W.atlas_workload(factory, 1, 3)
    r = oc.tick(wl.q[0], wl.v[0], wl.desired[0], wl.contact_weight[0], np.zeros(8))  # all contacts disabled
    assert r["status"] != 0
