"""Pin of the rigid-body half of the tick (a2 kinematics + a16 inverse_dynamics!, momentum.jl:57,75) on the full 31-body Atlas tree
that uses neither the library's nor the oracle's dynamics: Lagrange's equations
    tau_j = d/dt (dT/dv_j) - dT/dq_j + dV/dq_j  =  (M vdot)_j + d/ds p_j(q + s qdot)|_0 - dT/dq_j + dV/dq_j ,   p = M(q) v,
with every term from ENERGIES evaluated with the independent numpy forward kinematics (workloads.batch_kinematics):
  * M a by polarisation of the kinetic energy, (M a)_j = T(a + e_j) - T(a) - T(e_j)  -- exact, T is quadratic in v;
  * the derivatives along q by Richardson-extrapolated central differences (O(h^4)); V = m_tot g z_com.
The joints move (random revolute velocities: Coriolis and centrifugal terms of the whole tree), the base is at rest.
The controller pins every acceleration with hard JointAccelerationTasks (fixed base: floatingjoint = nothing), so vdot is the
desired vector and tau is pure inverse dynamics.  Revolute rows only (the base rows would need a derivative on SE(3)).
Tolerance 1e-7 N m on torques of order 10^2 (measured ~1e-9)."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding, models as M, workloads as W
from qpcontrol_jl_amd.workloads import HARD, JOINT


def _kinetic(mech, q, v):
    k = W.batch_kinematics(mech, q, v)
    T = np.zeros(np.atleast_2d(q).shape[0])
    for i in range(mech.nbodies):
        R, p, w, vl = k["R"][i], k["p"][i], k["w"][i], k["v"][i]
        cw = np.einsum("bij,j->bi", R, mech.com[i]) + p
        vc = vl + np.cross(w, cw)
        Ixx, Iyy, Izz, Ixy, Ixz, Iyz = mech.inertia[i]
        Ib = np.array([[Ixx, Ixy, Ixz], [Ixy, Iyy, Iyz], [Ixz, Iyz, Izz]])
        wb = np.einsum("bji,bj->bi", R, w)  # omega in body axes
        T += 0.5 * mech.mass[i] * np.einsum("bi,bi->b", vc, vc) + 0.5 * np.einsum("bi,ij,bj->b", wb, Ib, wb)
    return T


def _potential(mech, q):
    k = W.batch_kinematics(mech, q, np.zeros((np.atleast_2d(q).shape[0], mech.nv)))
    g = np.asarray(mech.gravity)
    return -k["mass"] * (k["com"] @ g)


def _check(factory, B, seed=7):
    mech = M.atlas()
    c = factory(mech, 4, M.WORLD)
    for j in range(mech.nbodies):
        if mech.joint_type[j] != M.FIXED:
            c.add_task(JOINT, joint=j, wkind=HARD)
            c.regularize(j, 0.1)
    c.finalize()
    rng = np.random.default_rng(seed)
    qs, _ = zip(*(M.rand_state(mech, rng, 0.0) for _ in range(B)))
    q = np.array(qs)
    v = np.zeros((B, mech.nv))
    v[:, 6:] = rng.normal(scale=0.8, size=(B, mech.nv - 6))  # the pelvis joint (velocity indices 0..5) stays at rest
    assert mech.joint_type[0] == M.QUAT_FLOATING
    des_v = rng.normal(size=(B, mech.nv))           # desired accelerations in velocity-index order
    # the desired record follows the task order (one JOINT task per joint, added in body order = velocity order here)
    lay = c.layout
    assert lay.n_desired == mech.nv and lay.n_contacts == 0
    r = c.control_batch(q, v, des_v)
    assert (r["status"] == 0).all(), r["status"]
    np.testing.assert_allclose(r["vdot"], des_v, rtol=0, atol=1e-10)
    # revolute velocity indices and their configuration indices
    rows = []
    qo = vo = 0
    for i in range(mech.nbodies):
        t = mech.joint_type[i]
        if t == M.REVOLUTE:
            rows.append((vo, qo))
        qo += M._NQ[t]; vo += M._NV[t]
    Ta = _kinetic(mech, q, des_v)
    err = 0.0
    h = 2e-3
    qdot = np.zeros_like(q)  # configuration rates: revolute joints only (base at rest)
    for vj, qj in rows:
        qdot[:, qj] = v[:, vj]

    def richardson(f):  # derivative of f at 0
        d = lambda hh: (f(hh) - f(-hh)) / (2 * hh)  # noqa: E731
        return (4.0 * d(h / 2) - d(h)) / 3.0
    for vj, qj in rows:
        e = np.zeros((B, mech.nv)); e[:, vj] = 1.0
        Ma = _kinetic(mech, q, des_v + e) - Ta - _kinetic(mech, q, e)

        def shifted(hh):
            qq = q.copy(); qq[:, qj] += hh
            return qq
        grav = richardson(lambda hh: _potential(mech, shifted(hh)))
        dTdq = richardson(lambda hh: _kinetic(mech, shifted(hh), v))
        pdot = richardson(lambda s_: _kinetic(mech, q + s_ * qdot, v + e) - _kinetic(mech, q + s_ * qdot, v) - _kinetic(mech, q + s_ * qdot, e))
        err = max(err, float(np.abs(r["tau"][:, vj] - (Ma + pdot - dTdq + grav)).max()))
    assert np.abs(r["tau"]).max() > 10.0
    assert err < 1e-7, err
    return err


def _check_foot_task(factory, B, seed=11):
    """a3 (SpatialAccelerationTask, src/tasks.jl:3-44) pinned by differentiating the independent forward kinematics in TIME: with the
    base and every other joint pinned, a hard task on the left foot leaves the six leg accelerations as the only unknowns; along
    q(t) = q + v t + vdot t^2 / 2 the foot's world twist (about the world origin) must then accelerate by exactly the desired spatial
    acceleration, re-expressed in the foot frame (source = world, frame = target: the velocity cross term vanishes)."""
    mech = M.atlas()
    foot = mech.findbody("l_foot")
    leg = []
    b = foot
    while mech.joint_type[b] != M.QUAT_FLOATING:
        leg.append(b)
        b = mech.parent[b]
    assert len(leg) == 6
    c = factory(mech, 4, M.WORLD)
    from qpcontrol_jl_amd.workloads import SPATIAL
    t_foot = c.add_task(SPATIAL, base=M.WORLD, target=foot, frame=M.FRAME_DEFAULT, wkind=HARD)
    pinned = []
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.FIXED:
            continue
        c.regularize(j, 0.1)
        if j not in leg:
            pinned.append((j, c.add_task(JOINT, joint=j, wkind=HARD)))
    c.finalize()
    rng = np.random.default_rng(seed)
    qs, _ = zip(*(M.rand_state(mech, rng, 0.0) for _ in range(B)))
    q = np.array(qs)
    v = np.zeros((B, mech.nv)); v[:, 6:] = rng.normal(scale=0.7, size=(B, mech.nv - 6))
    des = np.zeros((B, c.layout.n_desired))
    a_des = rng.normal(size=(B, 6))
    o, n = c.task_desired_range(t_foot)
    des[:, o:o + n] = a_des
    voff = {}
    qo = vo = 0
    for i in range(mech.nbodies):
        voff[i] = (qo, vo)
        qo += M._NQ[mech.joint_type[i]]; vo += M._NV[mech.joint_type[i]]
    acc_pin = rng.normal(scale=0.5, size=(B, mech.nv)); acc_pin[:, :6] = 0.0  # the base stays at rest and unaccelerated
    for j, tid in pinned:
        o, n = c.task_desired_range(tid)
        des[:, o:o + n] = acc_pin[:, voff[j][1]:voff[j][1] + n]
    r = c.control_batch(q, v, des)
    assert (r["status"] == 0).all(), r["status"]
    vd = r["vdot"]
    for j, _ in pinned:
        np.testing.assert_allclose(vd[:, voff[j][1]:voff[j][1] + M._NV[mech.joint_type[j]]], acc_pin[:, voff[j][1]:voff[j][1] + M._NV[mech.joint_type[j]]], atol=1e-10, rtol=0)
    qdot = np.zeros_like(q); qdd = np.zeros_like(q)
    for i in range(mech.nbodies):
        if mech.joint_type[i] == M.REVOLUTE:
            qdot[:, voff[i][0]] = v[:, voff[i][1]]; qdd[:, voff[i][0]] = vd[:, voff[i][1]]

    def twist(t):
        k = W.batch_kinematics(mech, q + t * qdot + 0.5 * t * t * qdd, v + t * vd)
        return np.concatenate([k["w"][foot], k["v"][foot]], axis=1)
    h = 1e-4  # (the leg accelerations that realise a random foot acceleration can be large: 2e-3 leaves 3e-4 of truncation error)
    d = lambda hh: (twist(hh) - twist(-hh)) / (2 * hh)  # noqa: E731
    aw = (4.0 * d(h / 2) - d(h)) / 3.0  # world-frame spatial acceleration of the foot about the world origin
    k0 = W.batch_kinematics(mech, q, v)
    R, p = k0["R"][foot], k0["p"][foot]
    ang = np.einsum("bji,bj->bi", R, aw[:, :3])
    lin = np.einsum("bji,bj->bi", R, aw[:, 3:] - np.cross(p, aw[:, :3]))
    err = float(np.abs(np.concatenate([ang, lin], axis=1) - a_des).max())
    assert err < 1e-7, err
    return err


def test_atlas_foot_task_against_differentiated_kinematics_emulated():
    emu.build()
    err = _check_foot_task(emu.factory, 3)
    print(f"max |d/dt twist(foot) - desired| = {err:.2e}")


@pytest.mark.gpu
def test_atlas_foot_task_against_differentiated_kinematics_gpu():
    err = _check_foot_task(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 16)
    print(f"max |d/dt twist(foot) - desired| = {err:.2e}")


def test_atlas_inverse_dynamics_against_energies_emulated():
    emu.build()
    emu.race_report()
    err = _check(emu.factory, 3)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first
    print(f"max |tau - Lagrange(T, V)| = {err:.2e}")


@pytest.mark.gpu
def test_atlas_inverse_dynamics_against_energies_gpu():
    err = _check(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 16)
    print(f"max |tau - Lagrange(T, V)| = {err:.2e}")
