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


# ---- Newton-Euler of the WHOLE robot on the BASELINE config-3 workload (floating base, 8 contact points) --------------------------
def _momenta(mech, q, v):
    """linear momentum and angular momentum about the world origin from the independent numpy kinematics"""
    k = W.batch_kinematics(mech, q, v)
    Bn = np.atleast_2d(q).shape[0]
    L = np.zeros((Bn, 3)); K = np.zeros((Bn, 3))
    for i in range(mech.nbodies):
        R, p, w, vl = k["R"][i], k["p"][i], k["w"][i], k["v"][i]
        cw = np.einsum("bij,j->bi", R, mech.com[i]) + p
        vc = vl + np.cross(w, cw)
        Ixx, Iyy, Izz, Ixy, Ixz, Iyz = mech.inertia[i]
        Ib = np.array([[Ixx, Ixy, Ixz], [Ixy, Iyy, Iyz], [Ixz, Iyz, Izz]])
        Iw_w = np.einsum("bij,bj->bi", R, np.einsum("ij,bj->bi", Ib, np.einsum("bji,bj->bi", R, w)))
        L += mech.mass[i] * vc
        K += np.cross(cw, mech.mass[i] * vc) + Iw_w
    return L, K, k["com"], k["mass"]


def _advance(mech, q, v, t):
    """q (+) v t: revolute / prismatic coordinates linearly, the floating base on SE(3) (v = body-frame twist)"""
    qn = q.copy()
    qo = vo = 0
    for i in range(mech.nbodies):
        ty = mech.joint_type[i]
        if ty in (M.REVOLUTE, M.PRISMATIC):
            qn[:, qo] = q[:, qo] + t * v[:, vo]
        elif ty == M.QUAT_FLOATING:
            w, x, y, z = q[:, qo], q[:, qo + 1], q[:, qo + 2], q[:, qo + 3]
            R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                          np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                          np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
            om = v[:, vo:vo + 3] * t
            th = np.linalg.norm(om, axis=1)
            half = 0.5 * th
            sc = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1.0), 0.5)
            dq = np.concatenate([np.cos(half)[:, None], sc[:, None] * om], axis=1)  # exp of the body-frame rotation vector
            a, b, c_, d = dq[:, 0], dq[:, 1], dq[:, 2], dq[:, 3]
            qn[:, qo] = w * a - x * b - y * c_ - z * d
            qn[:, qo + 1] = w * b + x * a + y * d - z * c_
            qn[:, qo + 2] = w * c_ - x * d + y * a + z * b
            qn[:, qo + 3] = w * d + x * c_ - y * b + z * a
            qn[:, qo + 4:qo + 7] = q[:, qo + 4:qo + 7] + t * np.einsum("bij,bj->bi", R, v[:, vo + 3:vo + 6])
        qo += M._NQ[ty]; vo += M._NV[ty]
    return qn


def _check_momentum_balance(factory, B, config=3):
    """a8 / a11 / a15 at once: with the kernel's vdot, the robot's momentum (independent kinematics, differentiated in time along
    q (+) v t, v + vdot t) must change at the rate of the wrench its own contact forces and gravity apply:
        dL/dt = m g + sum_c f_c ,     dK/dt = c x m g + sum_c n_c     (moments about the world origin; momentum.jl:162-193)"""
    c, wl = W.atlas_workload(factory, B, config)
    mech = M.atlas()
    r = c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    ok = r["status"] == 0
    assert ok.mean() > 0.8
    q, v, vd = wl.q[ok], wl.v[ok], r["vdot"][ok]
    wr = r["cwrench"][ok].reshape(ok.sum(), -1, 6)
    h = 1e-4

    def mom(t):
        L, K, _, _ = _momenta(mech, _advance(mech, q, v, t), v + t * vd)
        return np.concatenate([L, K], axis=1)
    d = lambda hh: (mom(hh) - mom(-hh)) / (2 * hh)  # noqa: E731
    rate = (4.0 * d(h / 2) - d(h)) / 3.0
    _, _, com, mtot = _momenta(mech, q, v)
    g = np.asarray(mech.gravity)
    force = mtot * g[None, :] + wr[:, :, 3:].sum(axis=1)
    moment = np.cross(com, mtot * g[None, :]) + wr[:, :, :3].sum(axis=1)
    scale = mtot * 9.81
    err = float(np.abs(rate - np.concatenate([force, moment], axis=1)).max() / scale)
    assert err < 1e-7, err
    return err


def test_whole_robot_momentum_balance_emulated():
    emu.build()
    err = _check_momentum_balance(emu.factory, 8)
    print(f"max |d/dt momentum - applied wrench| / (m g) = {err:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 5])
def test_whole_robot_momentum_balance_gpu(config):
    err = _check_momentum_balance(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 256, config)
    print(f"config {config}: max |d/dt momentum - applied wrench| / (m g) = {err:.2e}")


def _check_torques_with_contacts(factory, B, config=3):
    """a16 (inverse_dynamics! with the contact wrenches as external wrenches, momentum.jl:65-75) on the config-3 workload: for every
    revolute joint, Lagrange's equation with the generalised force of the contact forces,
        tau_j = d/dt (M v)_j - dT/dq_j + dV/dq_j - sum_c f_c . dp_c/dq_j ,
    the base moving on SE(3) (its velocity coordinates are quasi-velocities; the Hamel coefficients only couple the base rows, so
    the revolute rows keep the classical form).  Energies, momenta and contact points from the independent numpy kinematics,
    vdot and f_c from the kernel."""
    c, wl = W.atlas_workload(factory, B, config)
    mech = M.atlas()
    r = c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    ok = r["status"] == 0
    assert ok.mean() > 0.8
    q, v, vd, tau = wl.q[ok], wl.v[ok], r["vdot"][ok], r["tau"][ok]
    n = int(ok.sum())
    wr = r["cwrench"][ok].reshape(n, -1, 6)
    assert np.all(tau[:, :6] == 0.0)  # zero_floating_joint_torques! (momentum.jl:93-97)
    h = 1e-4

    def rich(f):
        d = lambda hh: (f(hh) - f(-hh)) / (2 * hh)  # noqa: E731
        return (4.0 * d(h / 2) - d(h)) / 3.0

    def contact_points(qq):
        k = W.batch_kinematics(mech, qq, np.zeros((n, mech.nv)))
        return np.stack([np.einsum("bij,j->bi", k["R"][cp.body], np.asarray(cp.position)) + k["p"][cp.body] for cp in mech.contact_points], axis=1)
    err = 0.0
    qo = vo = 0
    for i in range(mech.nbodies):
        ty = mech.joint_type[i]
        if ty == M.REVOLUTE:
            e = np.zeros((n, mech.nv)); e[:, vo] = 1.0

            def shifted(hh, qo=qo):
                qq = q.copy(); qq[:, qo] += hh
                return qq

            def pj(t, e=e):  # (M v)_j along the motion, by polarisation of the kinetic energy
                qt, vt = _advance(mech, q, v, t), v + t * vd
                return _kinetic(mech, qt, vt + e) - _kinetic(mech, qt, vt) - _kinetic(mech, qt, e)
            lag = rich(pj) - rich(lambda hh: _kinetic(mech, shifted(hh), v)) + rich(lambda hh: _potential(mech, shifted(hh)))
            dp = rich(lambda hh: contact_points(shifted(hh)))       # [n, nc, 3]
            gen = np.einsum("bci,bci->b", dp, wr[:, :, 3:])
            err = max(err, float(np.abs(tau[:, vo] - (lag - gen)).max()))
        qo += M._NQ[ty]; vo += M._NV[ty]
    assert np.abs(tau).max() > 10.0
    assert err < 1e-6, err
    return err


def test_torques_with_contacts_against_lagrange_emulated():
    emu.build()
    err = _check_torques_with_contacts(emu.factory, 4)
    print(f"max |tau - (Lagrange - J_c' f_c)| = {err:.2e}")


@pytest.mark.gpu
def test_torques_with_contacts_against_lagrange_gpu():
    err = _check_torques_with_contacts(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 32)
    print(f"max |tau - (Lagrange - J_c' f_c)| = {err:.2e}")


# ---- a6 PointAccelerationTask and a8 MomentumRateTask against differentiated kinematics ---------------------------------------------
def _check_point_task(factory, B, seed=13):
    """PointAccelerationTask (src/tasks.jl:125-171) with base = world: the CLASSICAL acceleration of a point fixed in the hand, i.e. the
    second time derivative of its world position along q(t) = q + v t + vdot t^2 / 2 (base and every joint off the arm pinned)."""
    from qpcontrol_jl_amd.workloads import POINT
    mech = M.atlas()
    hand = mech.findbody("r_hand")
    arm = []
    b = hand
    while mech.body_names[b] != "utorso" and mech.joint_type[b] != M.QUAT_FLOATING:
        arm.append(b)
        b = mech.parent[b]
    assert 5 <= len(arm) <= 8
    pt = np.array([0.05, -0.1, 0.02])
    c = factory(mech, 4, M.WORLD)
    t_pt = c.add_task(POINT, base=M.WORLD, target=hand, point=pt, wkind=HARD)
    pinned = []
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.FIXED:
            continue
        c.regularize(j, 0.1)
        if j not in arm:
            pinned.append((j, c.add_task(JOINT, joint=j, wkind=HARD)))
    c.finalize()
    rng = np.random.default_rng(seed)
    qs, _ = zip(*(M.rand_state(mech, rng, 0.0) for _ in range(B)))
    q = np.array(qs)
    v = np.zeros((B, mech.nv)); v[:, 6:] = rng.normal(scale=0.5, size=(B, mech.nv - 6))
    des = np.zeros((B, c.layout.n_desired))
    a_des = rng.normal(size=(B, 3))
    o, n = c.task_desired_range(t_pt)
    des[:, o:o + n] = a_des
    voff = {}
    qo = vo = 0
    for i in range(mech.nbodies):
        voff[i] = (qo, vo)
        qo += M._NQ[mech.joint_type[i]]; vo += M._NV[mech.joint_type[i]]
    acc = rng.normal(scale=0.5, size=(B, mech.nv)); acc[:, :6] = 0.0
    for j, tid in pinned:
        o, n = c.task_desired_range(tid)
        des[:, o:o + n] = acc[:, voff[j][1]:voff[j][1] + n]
    r = c.control_batch(q, v, des)
    assert (r["status"] == 0).all(), r["status"]
    vd = r["vdot"]
    qdot = np.zeros_like(q); qdd = np.zeros_like(q)
    for i in range(mech.nbodies):
        if mech.joint_type[i] == M.REVOLUTE:
            qdot[:, voff[i][0]] = v[:, voff[i][1]]; qdd[:, voff[i][0]] = vd[:, voff[i][1]]

    def pos(t):
        k = W.batch_kinematics(mech, q + t * qdot + 0.5 * t * t * qdd, v)
        return np.einsum("bij,j->bi", k["R"][hand], pt) + k["p"][hand]
    h = 2e-3
    d2 = lambda hh: (pos(hh) - 2.0 * pos(0.0) + pos(-hh)) / (hh * hh)  # noqa: E731
    acc_fd = (4.0 * d2(h / 2) - d2(h)) / 3.0
    err = float(np.abs(acc_fd - a_des).max())
    assert err < 1e-6, err  # (second differences: 1e-16 |p| / h^2 of rounding; measured 7e-10)
    return err


def _check_momentum_rate_task(factory, B, seed=17):
    """MomentumRateTask (src/tasks.jl:192-236) as a HARD task on the floating Atlas with its contacts: the rate of change of the momentum
    (independent kinematics, differentiated along the motion), moved to the centre of mass WITHOUT differentiating the transform
    (k_c' = K_O' - c x L', src/tasks.jl:208-219), must be the desired 6-vector [angular; linear]."""
    from qpcontrol_jl_amd.workloads import MOMENTUM_RATE
    mech = M.atlas()
    c = factory(mech, 4, mech.findbody("pelvis"))
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        if mech.joint_type[j] != M.FIXED:
            c.regularize(j, 0.05)
    t_m = c.add_task(MOMENTUM_RATE, wkind=HARD)
    c.finalize()
    q, v = W.standing_states(mech, B, seed, dq=0.1, drot=0.05, vstd=0.2)
    mtot = mech.total_mass()
    des = np.zeros((B, c.layout.n_desired))
    rng = np.random.default_rng(seed)
    o, n = c.task_desired_range(t_m)
    des[:, o:o + 3] = rng.normal(scale=5.0, size=(B, 3))            # angular momentum rate about the centre of mass
    des[:, o + 3:o + 6] = rng.normal(scale=0.05 * mtot, size=(B, 3))  # linear momentum rate (a few % of the weight)
    nc = len(mech.contact_points)
    r = c.control_batch(q, v, des, np.full((B, nc), 1e-3), np.full((B, nc), 1e6))
    ok = r["status"] == 0
    assert ok.mean() > 0.7, r["status"]
    q, v, vd, want = q[ok], v[ok], r["vdot"][ok], des[ok]
    h = 1e-4

    def mom(t):
        L, K, _, _ = _momenta(mech, _advance(mech, q, v, t), v + t * vd)
        return np.concatenate([L, K], axis=1)
    d = lambda hh: (mom(hh) - mom(-hh)) / (2 * hh)  # noqa: E731
    rate = (4.0 * d(h / 2) - d(h)) / 3.0
    _, _, com, _ = _momenta(mech, q, v)
    kc = rate[:, 3:] - np.cross(com, rate[:, :3])
    err = float(max(np.abs(kc - want[:, o:o + 3]).max(), np.abs(rate[:, :3] - want[:, o + 3:o + 6]).max()) / (mtot * 9.81))
    assert err < 1e-7, err
    return err


def test_point_and_momentum_rate_tasks_against_differentiated_kinematics_emulated():
    emu.build()
    e1 = _check_point_task(emu.factory, 3)
    e2 = _check_momentum_rate_task(emu.factory, 6)
    print(f"point task {e1:.2e} m/s^2, momentum-rate task {e2:.2e} of m g")


@pytest.mark.gpu
def test_point_and_momentum_rate_tasks_against_differentiated_kinematics_gpu():
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
    e1 = _check_point_task(f, 16)
    e2 = _check_momentum_rate_task(f, 64)
    print(f"point task {e1:.2e} m/s^2, momentum-rate task {e2:.2e} of m g")
