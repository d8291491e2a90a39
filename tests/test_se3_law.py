"""SE3PDController + SE3Trajectory evaluated ON THE DEVICE (QPC_LAW_SE3_PD; reference
src/lowlevel/se3pdcontroller.jl:13-18, src/trajectories/se3.jl, interpolated.jl:20-83).

The host mirror (qpcontrol.jl_amd/api.SE3PDController over trajectories.py) is pinned by the reference's own known
answers (tests/test_trajectories.py transcribes test/trajectories.jl).  Here: a controller whose hand task carries the
device law and receives the per-QP trajectory RECORD must produce the same tau as a twin whose task receives the desired
spatial acceleration the host mirror computed for the same state and time -- for times before, inside and after
[x0, xf], the identity interpolator and a fit_quintic time warp and a degree-5 Bezier one."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import api, binding
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import trajectories as T
from qpcontrol_jl_amd.workloads import HARD, JOINT, SCALAR, SPATIAL


def _controller(factory, with_law, gains, pelvis_law=None):
    mech = M.atlas()
    fl = mech.findbody("pelvis")
    c = factory(mech, 4, fl)
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        if mech.joint_type[j] != M.FIXED:
            c.regularize(j, 0.05)
    base, body = mech.findbody("pelvis"), mech.findbody("r_hand")
    t_hand = c.add_task(SPATIAL, base=base, target=body, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[10.0])
    t_foot = c.add_task(SPATIAL, base=M.WORLD, target=mech.findbody("l_foot"), frame=M.FRAME_DEFAULT, wkind=HARD)
    t_j = c.add_task(JOINT, joint=mech.findbody("r_hand") - 1, wkind=SCALAR, weight=[0.5])  # a task AFTER the record slice
    t_pel = None
    if pelvis_law is not None:  # a second kind of state-dependent law in the same controller (QpcDims::law_mask carries both)
        from qpcontrol_jl_amd.workloads import ANGULAR
        t_pel = c.add_task(ANGULAR, base=M.WORLD, target=mech.findbody("pelvis"), frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[3.0])
        if with_law:
            c.set_task_law(t_pel, binding.LAW_ORIENTATION_PD, pelvis_law[0], pelvis_law[1])
    if with_law:
        c.set_task_law(t_hand, binding.LAW_SE3_PD, gains.angular.k, gains.angular.d, [gains.linear.k, gains.linear.d])
    c.finalize()
    return mech, c, base, body, (t_hand, t_foot, t_j) + ((t_pel,) if t_pel is not None else ())


def _run(factory, B=10, seed=5, pelvis_law=None):
    rng = np.random.default_rng(seed)
    gains = api.SE3PDGains(api.PDGains(100.0, 20.0), api.PDGains(1000.0, 200.0))  # test/controller.jl:13
    mech, claw, base, body, tids = _controller(factory, True, gains, pelvis_law)
    _, cref, _, _, _ = _controller(factory, False, gains, pelvis_law)
    off_law = [claw.task_desired_range(t) for t in tids]
    off_ref = [cref.task_desired_range(t) for t in tids]
    assert off_law[0][1] == binding.SE3_RECORD and off_ref[0][1] == 6
    assert off_law[1][0] == off_law[0][0] + 29 and off_ref[1][0] == off_ref[0][0] + 6  # later slices shift with the record
    assert claw.layout.n_desired == cref.layout.n_desired + 23
    qs, vs = zip(*(M.rand_state(mech, rng, 0.3) for _ in range(B)))
    q, v = np.array(qs), np.array(vs)
    pel_des = None
    if pelvis_law is not None:  # -kp rotation_vector(R) - kd R' omega of the pelvis (standing.jl:72-77), host evaluation
        from qpcontrol_jl_amd import workloads as W
        kin = W.batch_kinematics(mech, q, v)
        pb = mech.findbody("pelvis")
        Rp = kin["R"][pb]
        pel_des = -pelvis_law[0] * W.rotation_vector(Rp) - pelvis_law[1] * np.einsum("bji,bj->bi", Rp, kin["w"][pb])
    nc = len(mech.contact_points)
    cw, cm = np.full((B, nc), 1e-3), np.full((B, nc), 1e6)
    for warp in (None, "quintic", "antipodal", "bezier"):
        f = None if warp in (None, "antipodal") else T.fit_quintic(x0=0.0, xf=1.0, y0=0.0, yd0=0.0, ydd0=0.0, yf=1.0, ydf=0.0, yddf=0.0)
        if warp == "bezier":  # `const BezierCurve = BernsteinPolynomial` (trajectories.jl:26), degree 5: the device record takes its monomial form
            f = T.BezierCurve([0.0, 0.0, 0.15, 0.7, 1.0, 1.0])
        yf = T.Quat.from_angle_axis(np.pi / 2, [1.0, 0.3, -0.2])
        if warp == "antipodal":  # same rotation, opposite sign: inv(y0) * yf has w < 0 -> the reference takes the long arc (3 pi / 2)
            yf = T.Quat(*(-yf.q))
        ang = T.Interpolated(0.2, 1.7, T.Quat.identity(), yf, f, min_num_derivs=2)
        lin = T.Interpolated(0.2, 1.7, np.array([0.0, 1.0, 2.0]) * 0.2, np.array([2.0, 3.0, 4.0]) * 0.15, f, min_num_derivs=2)
        traj = T.SE3Trajectory(body, base, ang, lin)
        ctl = api.SE3PDController(mech, base, body, traj, np.diag([0, 0, 0, 10.0, 10, 10]), gains)
        times = np.array([-0.5, 0.2, 0.21, 0.6, 0.95, 1.3, 1.69, 1.7, 2.5, 0.8][:B])
        des_law = np.zeros((B, claw.layout.n_desired))
        des_ref = np.zeros((B, cref.layout.n_desired))
        rec = ctl.device_record(times)
        foot_des = rng.normal(size=(B, 6)) * 0.1
        jd = rng.normal(size=(B, 1))
        for b in range(B):
            des_law[b, off_law[0][0]:off_law[0][0] + 29] = rec[b]
            des_ref[b, off_ref[0][0]:off_ref[0][0] + 6] = ctl(times[b], (q[b:b + 1], v[b:b + 1]))[0]
        for dl, offs in ((des_law, off_law), (des_ref, off_ref)):
            dl[:, offs[1][0]:offs[1][0] + 6] = foot_des
            dl[:, offs[2][0]:offs[2][0] + 1] = jd
        if pel_des is not None:
            des_ref[:, off_ref[3][0]:off_ref[3][0] + 3] = pel_des  # the law controller ignores its slice
        rl = claw.control_batch(q, v, des_law, cw, cm)
        rr = cref.control_batch(q, v, des_ref, cw, cm)
        assert np.array_equal(rl["status"], rr["status"]) and (rr["status"] == 0).all(), (rl["status"], rr["status"])
        assert np.abs(rr["tau"]).max() > 1.0
        np.testing.assert_allclose(rl["tau"], rr["tau"], atol=1e-9 * max(1.0, np.abs(rr["tau"]).max()), rtol=0)
        np.testing.assert_allclose(rl["vdot"], rr["vdot"], atol=1e-9 * max(1.0, np.abs(rr["vdot"]).max()), rtol=0)


def test_se3_pd_law_emulated():
    emu.build()
    emu.race_report()
    _run(emu.factory)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_se3_and_orientation_laws_in_one_controller_emulated():
    """two kinds of state-dependent laws (SE(3) PD on the hand, orientation PD on the pelvis) in one controller: the specialised
    kernel carries exactly the kinds in use (QpcDims::law_mask)"""
    emu.build()
    emu.race_report()
    _run(emu.factory, pelvis_law=(20.0, 9.0))
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_se3_law_needs_a_body_frame_spatial_task():
    emu.build()
    mech = M.atlas()
    c = emu.factory(mech, 4, mech.findbody("pelvis"))
    t = c.add_task(SPATIAL, base=M.WORLD, target=3, frame=M.WORLD, wkind=HARD)
    with pytest.raises(binding.QpcError, match="target body's frame"):
        c.set_task_law(t, binding.LAW_SE3_PD, 1.0, 1.0, [1.0, 1.0])


@pytest.mark.gpu
def test_se3_pd_law_gpu():
    _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), B=10)


@pytest.mark.gpu
@pytest.mark.gpu_slow  # (6 s of hiprtc; each law keeps a fast GPU test of its own, the combination its emulated twin)
def test_se3_and_orientation_laws_in_one_controller_gpu():
    _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), B=10, pelvis_law=(20.0, 9.0))


# ---------------------------------------------------------------------------------------------------------------------
# Piecewise trajectories on the device (/root/reference/src/trajectories/piecewise.jl:1-30): the record carries the breaks
# and every piece; the piece is chosen per QP on the device.
def _run_piecewise(factory, B=12, seed=8, n_pieces=3):
    rng = np.random.default_rng(seed)
    gains = api.SE3PDGains(api.PDGains(100.0, 20.0), api.PDGains(1000.0, 200.0))
    mech = M.atlas()
    fl = mech.findbody("pelvis")
    base, body = mech.findbody("pelvis"), mech.findbody("r_hand")

    def build(with_law):
        c = factory(mech, 4, fl)
        for cp in mech.contact_points:
            c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
        for j in range(mech.nbodies):
            c.regularize(j, 0.05)
        th = c.add_task(SPATIAL, base=base, target=body, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[10.0])
        tf = c.add_task(SPATIAL, base=M.WORLD, target=mech.findbody("l_foot"), frame=M.FRAME_DEFAULT, wkind=HARD)
        if with_law:
            c.set_task_law(th, binding.LAW_SE3_PD, gains.angular.k, gains.angular.d, [gains.linear.k, gains.linear.d, float(n_pieces)])
        c.finalize()
        return c, th, tf
    claw, th, tf = build(True)
    cref, th2, tf2 = build(False)
    assert claw.task_desired_range(th)[1] == binding.se3_pw_record(n_pieces) and cref.task_desired_range(th2)[1] == 6
    # a trajectory of n_pieces Interpolated pieces through random way points; every piece on its own local clock [0, duration]
    breaks = np.concatenate([[0.3], 0.3 + np.cumsum(rng.uniform(0.3, 0.8, size=n_pieces))])
    quats = [T.Quat.identity()] + [T.Quat.from_angle_axis(rng.uniform(0.2, 1.2), rng.normal(size=3)) for _ in range(n_pieces)]
    pts = [rng.normal(size=3) * 0.3 for _ in range(n_pieces + 1)]
    f = T.fit_quintic(x0=0.0, xf=1.0, y0=0.0, yd0=0.0, ydd0=0.0, yf=1.0, ydf=0.0, yddf=0.0)
    angs = [T.Interpolated(0.0, breaks[i + 1] - breaks[i], quats[i], quats[i + 1], f if i % 2 == 0 else None, min_num_derivs=2) for i in range(n_pieces)]
    lins = [T.Interpolated(0.0, breaks[i + 1] - breaks[i], pts[i], pts[i + 1], f if i % 2 == 0 else None, min_num_derivs=2) for i in range(n_pieces)]
    traj = T.SE3Trajectory(body, base, T.Piecewise(angs, breaks), T.Piecewise(lins, breaks))
    ctl = api.SE3PDController(mech, base, body, traj, np.eye(6), gains)
    assert ctl.n_pieces() == n_pieces
    qs, vs = zip(*(M.rand_state(mech, rng, 0.3) for _ in range(B)))
    q, v = np.array(qs), np.array(vs)
    # times before the first break, on every break, inside every piece, after the last break
    times = np.concatenate([[breaks[0] - 0.4], breaks, 0.5 * (breaks[:-1] + breaks[1:]), [breaks[-1] + 1.0]])
    times = np.resize(times, B)
    rec = ctl.device_record(times)
    ol, orf = claw.task_desired_range(th)[0], cref.task_desired_range(th2)[0]
    des_law, des_ref = np.zeros((B, claw.layout.n_desired)), np.zeros((B, cref.layout.n_desired))
    foot = rng.normal(size=(B, 6)) * 0.1
    for b in range(B):
        des_law[b, ol:ol + rec.shape[1]] = rec[b]
        des_ref[b, orf:orf + 6] = ctl(times[b], (q[b:b + 1], v[b:b + 1]))[0]
    des_law[:, claw.task_desired_range(tf)[0]:claw.task_desired_range(tf)[0] + 6] = foot
    des_ref[:, cref.task_desired_range(tf2)[0]:cref.task_desired_range(tf2)[0] + 6] = foot
    nc = len(mech.contact_points)
    cw, cm = np.full((B, nc), 1e-3), np.full((B, nc), 1e6)
    rl, rr = claw.control_batch(q, v, des_law, cw, cm), cref.control_batch(q, v, des_ref, cw, cm)
    assert np.array_equal(rl["status"], rr["status"]) and (rr["status"] == 0).all()
    np.testing.assert_allclose(rl["tau"], rr["tau"], atol=1e-9 * max(1.0, np.abs(rr["tau"]).max()), rtol=0)
    # the pieces differ: the same state at two times in different pieces gives different torques
    assert np.abs(rr["tau"]).max() > 1.0


@pytest.mark.parametrize("n_pieces", [2, 4])
def test_se3_piecewise_law_emulated(n_pieces):
    emu.build()
    emu.race_report()
    _run_piecewise(emu.factory, n_pieces=n_pieces)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_se3_piecewise_law_validation():
    emu.build()
    mech = M.atlas()
    c = emu.factory(mech, 4, mech.findbody("pelvis"))
    t = c.add_task(SPATIAL, base=mech.findbody("pelvis"), target=mech.findbody("r_hand"), frame=M.FRAME_DEFAULT, wkind=HARD)
    for bad in (1.0, 5.0, 2.5):
        with pytest.raises(binding.QpcError, match="number of pieces"):
            c.set_task_law(t, binding.LAW_SE3_PD, 1.0, 1.0, [1.0, 1.0, bad])


@pytest.mark.gpu
def test_se3_piecewise_law_gpu():
    _run_piecewise(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), B=12, n_pieces=3)
