"""Pins that depend on NEITHER the library NOR the oracle: problems whose answer has a closed form that is written out here.

1. A single floating rigid body on contact points (the smallest MomentumBasedController with contacts: momentum.jl:15-33,
   addcontact! :133-148, ContactPoint / forcebasis contacts.jl:16-69, wrench balance momentum.jl:162-193): with a hard
   SpatialAccelerationTask on the body the accelerations are the desired ones and the contact forces are the MINIMUM-NORM forces
   that balance Newton-Euler,
       min sum_c w |f_c|^2   s.t.   sum_c f_c = m (a - g),   sum_c (p_c - c) x f_c = I omega_dot + omega x I omega,
   i.e. f = A' (A A')^-1 b -- six lines of numpy -- as long as every force stays inside its friction pyramid (checked).  The
   kernel must return these forces as world-frame contact wrenches (value(wrench_world), contacts.jl:82-89), zero torque on the
   floating joint (momentum.jl:93-97) and vdot = desired.
2. (examples/acrobot_tick.c, tests/test_c_example.py: the 2-link manipulator equation.)

Tolerance 1e-9 (the proximal regularisation of the rank-deficient ray blocks is removed by the final polish)."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding, models as M
from qpcontrol_jl_amd.workloads import HARD, SPATIAL


def _box(factory, seed):
    rng = np.random.default_rng(seed)
    mass = float(rng.uniform(5.0, 40.0))
    Ib = np.diag(rng.uniform(0.3, 2.0, size=3))
    mech = M.Mechanism()
    mech.add_body("box", M.WORLD, "floating", M.QUAT_FLOATING, mass=mass, com=(0, 0, 0),
                  inertia=(Ib[0, 0], Ib[1, 1], Ib[2, 2], 0, 0, 0))
    pts = np.array([[0.3, 0.2, -0.1], [0.3, -0.25, -0.1], [-0.35, 0.2, -0.1], [-0.3, -0.2, -0.12]]) + rng.normal(scale=0.01, size=(4, 3))
    c = factory(mech, 4, 0)
    for p in pts:
        c.add_contact(0, p, [0.0, 0.0, 1.0], 1.0)
    c.add_task(SPATIAL, base=M.WORLD, target=0, frame=M.FRAME_DEFAULT, wkind=HARD)
    c.regularize(0, 1e-3)
    c.finalize()
    return mech, c, mass, Ib, pts


def _check(factory, B=6, seed=3):
    mech, c, mass, Ib, pts = _box(factory, seed)
    rng = np.random.default_rng(seed + 1)
    nc = len(pts)
    psi = rng.uniform(-np.pi, np.pi, size=B)                  # yawed about the (vertical) contact normal ...
    q = np.zeros((B, 7)); q[:, 0] = np.cos(psi / 2); q[:, 3] = np.sin(psi / 2)
    q[:, 4:7] = rng.normal(scale=0.5, size=(B, 3))            # ... anywhere in the world
    v = np.zeros((B, 6)); v[:, :3] = rng.normal(scale=0.2, size=(B, 3))  # spinning (omega x I omega enters), not translating
    des = np.zeros((B, 6))
    des[:, :3] = rng.normal(scale=0.3, size=(B, 3))           # angular acceleration (body frame)
    des[:, 3:] = rng.normal(scale=0.5, size=(B, 3)) + [0, 0, 1.0]
    w = rng.uniform(0.5e-3, 2e-3, size=(B, 1)) * np.ones((B, nc))   # one weight per QP (equal over its contacts)
    cap = np.full((B, nc), 1e6)
    r = c.control_batch(q, v, des, w, cap)
    assert (r["status"] == 0).all(), r["status"]
    g = np.array([0.0, 0.0, -9.81])
    for b in range(B):
        com = q[b, 4:7]
        om = v[b, :3]
        cz, sz = np.cos(psi[b]), np.sin(psi[b])
        R = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])  # body -> world
        # v̇ of a QuaternionFloating joint = (omega_dot_body, v_dot_body); with v_lin = 0 the body-origin acceleration is v_dot_body
        np.testing.assert_allclose(r["vdot"][b], des[b], rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["tau"][b], 0.0, rtol=0, atol=0)  # floating joint: exactly zero (momentum.jl:93-97)
        A = np.zeros((6, 3 * nc))
        for k, p in enumerate(pts):
            A[:3, 3 * k:3 * k + 3] = np.eye(3)
            px, py, pz = R @ p  # p_c - c in world axes (c = body origin)
            A[3:, 3 * k:3 * k + 3] = np.array([[0, -pz, py], [pz, 0, -px], [-py, px, 0]])
        # desired spatial acceleration and omega are body-frame quantities (v̇ of the floating joint): to world axes with R
        rhs = np.concatenate([mass * (R @ des[b, 3:] - g), R @ (Ib @ des[b, :3] + np.cross(om, Ib @ om))])
        f = (A.T @ np.linalg.solve(A @ A.T, rhs)).reshape(nc, 3)
        # the 4-ray pyramid of mu = 1 contains the cone |f_xy| <= f_z / sqrt(2) whatever its yaw
        assert np.all(np.hypot(f[:, 0], f[:, 1]) < 0.65 * f[:, 2]), "closed form leaves the friction pyramid: pick another seed"
        wr = r["cwrench"][b].reshape(nc, 6)
        for k, p in enumerate(pts):
            np.testing.assert_allclose(wr[k, 3:], f[k], rtol=0, atol=1e-9 * mass * 10)
            np.testing.assert_allclose(wr[k, :3], np.cross(com + R @ p, f[k]), rtol=0, atol=1e-9 * mass * 10)  # about the WORLD origin


def test_rigid_body_on_contacts_closed_form_emulated():
    emu.build()
    emu.race_report()
    _check(emu.factory)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
def test_rigid_body_on_contacts_closed_form_gpu():
    _check(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), B=64, seed=5)


# ---- the cap of contacts.jl:57: rho <= maxnormalforce / (N sqrt(mu^2 + 1)) ---------------------------------------------------------
def _check_cap(factory, B=5, seed=9):
    """A box on four SYMMETRIC contact points is asked (soft SpatialAccelerationTask) to jump harder than its contacts allow: every
    ray then sits on its upper bound rho_max = maxnormalforce / (N sqrt(mu^2 + 1)) (contacts.jl:57), the horizontal components cancel,
    and each contact pushes with N rho_max / sqrt(mu^2 + 1) = maxnormalforce / (mu^2 + 1) -- NOT maxnormalforce: the reference's
    quirk, reproduced.  Closed form:  vdot_z = 4 maxnormalforce / ((mu^2 + 1) m) - g , everything else zero."""
    from qpcontrol_jl_amd.workloads import SCALAR
    rng = np.random.default_rng(seed)
    mass, mu = 12.0, 0.6
    mech = M.Mechanism()
    mech.add_body("box", M.WORLD, "floating", M.QUAT_FLOATING, mass=mass, com=(0, 0, 0), inertia=(0.5, 0.7, 0.9, 0, 0, 0))
    c = factory(mech, 4, 0)
    for sx in (1, -1):
        for sy in (1, -1):
            c.add_contact(0, [0.3 * sx, 0.2 * sy, -0.1], [0.0, 0.0, 1.0], mu)
    c.add_task(SPATIAL, base=M.WORLD, target=0, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[10.0])
    c.regularize(0, 1e-3)
    c.finalize()
    q = np.zeros((B, 7)); q[:, 0] = 1.0
    q[:, 4:7] = rng.normal(scale=0.3, size=(B, 3))
    v = np.zeros((B, 6))
    des = np.zeros((B, 6)); des[:, 5] = 50.0
    cap = rng.uniform(20.0, 60.0, size=(B, 1)) * np.ones((B, 4))
    w = np.full((B, 4), 1e-3)
    r = c.control_batch(q, v, des, w, cap)
    assert (r["status"] == 0).all(), r["status"]
    az = 4.0 * cap[:, 0] / ((mu * mu + 1.0) * mass) - 9.81
    assert np.all(az < 50.0)
    expect = np.zeros((B, 6)); expect[:, 5] = az
    np.testing.assert_allclose(r["vdot"], expect, rtol=0, atol=1e-8)
    wr = r["cwrench"].reshape(B, 4, 6)
    np.testing.assert_allclose(wr[:, :, 5], cap / (mu * mu + 1.0), rtol=0, atol=1e-8)
    np.testing.assert_allclose(wr[:, :, 3:5], 0.0, rtol=0, atol=1e-8)


def test_contact_cap_closed_form_emulated():
    emu.build()
    _check_cap(emu.factory)


@pytest.mark.gpu
def test_contact_cap_closed_form_gpu():
    _check_cap(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), B=32)
