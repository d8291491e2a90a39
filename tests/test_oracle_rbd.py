"""Pins the oracle's rigid-body restatement (oracle/ora_rbd.c) by physical invariants: the same
identities RigidBodyDynamics guarantees and the reference's tests rely on (test/tasks.jl,
test/controller.jl:92-96), checked against finite differences and an independent numpy FK."""
import numpy as np
import pytest

from oracle import oracle as O
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W

MECHS = {
    "revolute10": lambda: M.rand_tree([M.REVOLUTE] * 10, seed=1),
    "mixed_floating": lambda: M.rand_tree([M.QUAT_FLOATING] + [M.REVOLUTE] * 8 + [M.PRISMATIC] * 2, seed=3),
    "atlas": M.atlas,
}


@pytest.fixture(params=list(MECHS))
def setup(request):
    mech = MECHS[request.param]()
    rng = np.random.default_rng(7)
    q, v = M.rand_state(mech, rng)
    return mech, O.OracleModel(mech), q, v, rng


def test_numpy_fk_agrees(setup):
    mech, om, q, v, _ = setup
    kin = W.batch_kinematics(mech, q[None], v[None])
    om.set_state(q, v)
    for i in range(mech.nbodies):
        R, p = om.transform(i)
        np.testing.assert_allclose(R, kin["R"][i][0], atol=1e-13)
        np.testing.assert_allclose(p, kin["p"][i][0], atol=1e-13)
        np.testing.assert_allclose(om.twist(i)[:3], kin["w"][i][0], atol=1e-12)
        np.testing.assert_allclose(om.twist(i)[3:], kin["v"][i][0], atol=1e-12)
    np.testing.assert_allclose(om.center_of_mass(), kin["com"][0], atol=1e-13)
    A, _ = om.momentum_matrix()
    np.testing.assert_allclose((A @ v)[3:], kin["linear_momentum"][0], atol=1e-11)


def test_twist_is_jacobian_times_v(setup):
    mech, om, q, v, _ = setup
    om.set_state(q, v)
    for b in range(mech.nbodies):
        np.testing.assert_allclose(om.geometric_jacobian(-1, b, -1) @ v, om.twist(b), atol=1e-11)


def test_bias_zero_at_zero_velocity(setup):
    mech, om, q, v, _ = setup
    om.set_state(q, np.zeros(mech.nv))
    for b in range(mech.nbodies):
        assert np.all(om.bias_acceleration(b) == 0.0)  # exact, as test/tasks.jl:46,51 demands
    assert np.all(om.momentum_matrix()[1] == 0.0)


def test_relative_acceleration_finite_difference(setup):
    """d/dt [relative twist expressed in a moving frame] == J v̇ + transform(bias) -- the identity
    every acceleration task relies on (src/tasks.jl:31-44)."""
    mech, om, q, v, rng = setup
    vd = rng.normal(size=mech.nv)
    dt = 1e-6

    def reltw(qq, vv, body, base, frame):
        om.set_state(qq, vv)
        return om.relative_twist(body, base, frame)

    for _ in range(6):
        base, body, frame = (int(x) for x in rng.integers(-1, mech.nbodies, 3))
        fd = (reltw(om.integrate(q, v, dt), v + vd * dt, body, base, frame)
              - reltw(om.integrate(q, v, -dt), v - vd * dt, body, base, frame)) / (2 * dt)
        om.set_state(q, v)
        J = om.geometric_jacobian(base, body, frame)
        b6 = om.transform_accel(om.bias_acceleration(body) - om.bias_acceleration(base), body, base, frame)
        np.testing.assert_allclose(fd, J @ vd + b6, atol=2e-7 * (1 + np.abs(fd).max()))


def test_momentum_rate_finite_difference(setup):
    mech, om, q, v, rng = setup
    vd = rng.normal(size=mech.nv)
    dt = 1e-6
    om.set_state(q, v)
    A, Adv = om.momentum_matrix()
    h0 = A @ v
    om.set_state(om.integrate(q, v, dt), v + vd * dt)
    hp, cp = om.momentum_matrix()[0] @ (v + vd * dt), om.center_of_mass()
    om.set_state(om.integrate(q, v, -dt), v - vd * dt)
    hm, cm = om.momentum_matrix()[0] @ (v - vd * dt), om.center_of_mass()
    scale = 1 + np.abs(h0).max()
    np.testing.assert_allclose((hp - hm) / (2 * dt), A @ vd + Adv, atol=1e-6 * scale)
    np.testing.assert_allclose(mech.total_mass() * (cp - cm) / (2 * dt), h0[3:], atol=1e-6 * scale)


def test_dynamics_round_trip(setup):
    mech, om, q, v, rng = setup
    vd = rng.normal(size=mech.nv)
    om.set_state(q, v)
    Mm = om.mass_matrix()
    assert np.abs(Mm - Mm.T).max() == 0 and np.linalg.eigvalsh(Mm).min() > 0
    wext = rng.normal(size=(mech.nbodies, 6))
    tau = om.inverse_dynamics(vd, wext)
    np.testing.assert_allclose(om.forward_dynamics(tau, wext), vd, atol=1e-10)
    c = om.inverse_dynamics(np.zeros(mech.nv), wext)
    np.testing.assert_allclose(tau, Mm @ vd + c, atol=1e-9 * (1 + np.abs(tau).max()))
    # external wrench enters as -J^T w
    c0 = om.inverse_dynamics(np.zeros(mech.nv))
    jt = sum(om.geometric_jacobian(-1, b, -1).T @ wext[b] for b in range(mech.nbodies))
    np.testing.assert_allclose(c - c0, -jt, atol=1e-10 * (1 + np.abs(jt).max()))


def test_floating_base_torque_is_wrench_balance_residual():
    """S'(A v̇ + Ȧv - Wg - sum w) is the floating-joint torque of inverse dynamics: the identity
    behind add_wrench_balance_constraint! + zero_floating_joint_torques! (momentum.jl:93-97,162-193)."""
    mech = M.atlas()
    om = O.OracleModel(mech)
    rng = np.random.default_rng(3)
    q, v = M.rand_state(mech, rng)
    vd = rng.normal(size=mech.nv)
    om.set_state(q, v)
    wext = np.zeros((mech.nbodies, 6))
    wext[mech.findbody("l_foot")] = rng.normal(size=6) * 50
    tau = om.inverse_dynamics(vd, wext)
    A, Adv = om.momentum_matrix()
    fg = mech.total_mass() * np.array(mech.gravity)
    Wg = np.concatenate([np.cross(om.center_of_mass(), fg), fg])
    S = om.geometric_jacobian(-1, 0, -1)[:, :6]
    np.testing.assert_allclose(S.T @ (A @ vd + Adv - Wg - wext.sum(0)), tau[:6], atol=1e-9 * (1 + np.abs(tau).max()))
