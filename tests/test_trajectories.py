"""Known-answer and property tests of the Trajectories host mirror, transcribed case by case from the
reference's own test file (test/trajectories.jl: fit_cubic :18-41, fit_quintic :43-70, Interpolated
identity :72-108, Interpolated polynomial :110-136, Piecewise :138-181, Point/FreeVector :183-217,
SE(3) :219-232) -- the only neighbour of the hot path for which the reference tree holds golden values."""
import numpy as np
import pytest

from qpcontrol_jl_amd import trajectories as T


def test_fit_cubic():
    rng = np.random.default_rng(15)
    for _ in range(10):
        x0, xf, y0, yd0, yf, ydf = rng.uniform(size=6)
        p = T.fit_cubic(x0=x0, xf=xf, y0=y0, yd0=yd0, yf=yf, ydf=ydf)
        pd = p.derivative()
        assert abs(p(x0) - y0) < 1e-6 and abs(pd(x0) - yd0) < 1e-6
        assert abs(p(xf) - yf) < 1e-6 and abs(pd(xf) - ydf) < 1e-6
        assert len(p.coeffs) == 4


def test_fit_quintic():
    rng = np.random.default_rng(15)
    for _ in range(10):
        x0, xf, y0, yd0, ydd0, yf, ydf, yddf = rng.uniform(size=8)
        p = T.fit_quintic(x0=x0, xf=xf, y0=y0, yd0=yd0, ydd0=ydd0, yf=yf, ydf=ydf, yddf=yddf)
        pd = p.derivative()
        pdd = pd.derivative()
        for got, want in ((p(x0), y0), (pd(x0), yd0), (pdd(x0), ydd0), (p(xf), yf), (pd(xf), ydf), (pdd(xf), yddf)):
            assert abs(got - want) < 1e-6
        assert len(p.coeffs) == 6


def test_interpolated_identity():
    traj = T.Interpolated(0.0, 1.0, 1, 2)
    assert traj(0.5) == 1.5
    assert traj(0.5, 2) == (1.5, 1.0, 0.0)
    traj = T.Interpolated(0.0, 1.0, np.array([1.0, 2, 3]), np.array([0.0, 2, 4]))
    y, yd, ydd = traj(0.5, 2)
    np.testing.assert_allclose(y, [0.5, 2.0, 3.5], atol=1e-15)
    np.testing.assert_allclose(yd, [-1.0, 0.0, 1.0], atol=1e-15)
    np.testing.assert_allclose(ydd, [0.0, 0.0, 0.0], atol=1e-15)
    angle, axis = np.pi / 2, np.array([1.0, 0.0, 0.0])
    y0, yf = T.Quat.identity(), T.Quat.from_angle_axis(angle, axis)
    traj = T.Interpolated(0.0, 1.0, y0, yf)
    assert traj(0.0).isapprox(y0, 1e-15) and traj(1.0).isapprox(yf, 1e-15)
    y, yd, ydd = traj(0.5, 2)
    assert y.isapprox(T.Quat.from_angle_axis(0.5 * angle, axis), 1e-15)
    np.testing.assert_allclose(yd / np.linalg.norm(yd), axis, atol=1e-15)
    assert abs(np.linalg.norm(yd) - abs(angle)) < 1e-15
    np.testing.assert_allclose(ydd, [0.0, 0.0, 0.0], atol=1e-15)


def test_interpolated_polynomial():
    interpolator = T.fit_quintic(x0=0.0, xf=1.0, y0=0.0, yd0=0.0, ydd0=0.0, yf=1.0, ydf=0.0, yddf=0.0)
    x0, xf, y0, yf = -1.0, 2.0, 2.0, 3.0
    traj = T.Interpolated(x0, xf, y0, yf, interpolator, min_num_derivs=2)
    y, yd, ydd = traj(x0, 2)
    assert abs(y - y0) < 1e-10 and abs(yd) < 1e-10 and abs(ydd) < 1e-10
    y, yd, ydd = traj(xf, 2)
    assert abs(y - yf) < 1e-10 and abs(yd) < 1e-10 and abs(ydd) < 1e-10
    h = 1e-5
    for x in np.linspace(x0 + 0.05, xf - 0.05, 10):  # the reference differentiates with ForwardDiff
        _, yd, ydd = traj(x, 2)
        assert abs(yd - (traj(x + h) - traj(x - h)) / (2 * h)) < 1e-4
        assert abs(ydd - (traj(x + h) - 2 * traj(x) + traj(x - h)) / h ** 2) < 1e-4


def _test_piecewise(traj):
    n = len(traj.subfunctions)
    for i, t in enumerate(traj.breaks):
        sub = traj.subfunctions[min(i, n - 1)]
        assert traj(t) == sub(t - traj.breaks[min(i, n - 1)])
        if i < n:
            tmid = (t + traj.breaks[i + 1]) / 2
            assert traj(tmid) == sub(tmid - t)
            assert traj(tmid, 2) == sub(tmid - t, 2)
    t0, tf = traj.breaks[0], traj.breaks[-1]
    if traj.clamp:
        assert traj(t0 - 1) == traj(t0) and traj(tf + 1) == traj(tf)
    else:
        with pytest.raises(T.TrajectoryDomainError):
            traj(t0 - 1)
        with pytest.raises(T.TrajectoryDomainError):
            traj(tf + 1)


def test_piecewise_constant():
    n = 5
    subs = [T.Constant(i) for i in range(1, n + 1)]
    breaks = [i ** 2 - 1 for i in range(1, n + 2)]
    _test_piecewise(T.Piecewise(subs, breaks, clamp=True))
    _test_piecewise(T.Piecewise(subs, breaks, clamp=False))


def test_piecewise_interpolated():
    rng = np.random.default_rng(1)
    n = 5
    breaks = [i ** 2 - 1 for i in range(1, n + 2)]
    subs = [T.Interpolated(0.0, float(dt), rng.uniform(), rng.uniform()) for dt in np.diff(breaks)]
    _test_piecewise(T.Piecewise(subs, breaks, clamp=True))
    _test_piecewise(T.Piecewise(subs, breaks, clamp=False))


def test_point_and_free_vector_trajectory():
    rng = np.random.default_rng(2)
    interpolator = T.fit_cubic(x0=0.0, xf=1.0, y0=rng.uniform(), yd0=rng.uniform(), yf=rng.uniform(), ydf=rng.uniform())
    interpolated = T.Interpolated(-1.0, 2.0, rng.uniform(size=3), rng.uniform(size=3), interpolator, min_num_derivs=2)
    for cls in (T.PointTrajectory, T.FreeVectorTrajectory):
        traj = cls("frame", interpolated)
        f, y = traj(1.0)
        assert f == "frame" and np.array_equal(y, interpolated(1.0))
        for (f, got), want in zip(traj(1.0, 2), interpolated(1.0, 2)):
            assert f == "frame" and np.array_equal(got, want)


def test_se3():
    angle, axis = np.pi / 2, np.array([1.0, 0.0, 0.0])
    angular = T.Interpolated(0.0, 1.0, T.Quat.identity(), T.Quat.from_angle_axis(angle, axis))
    linear = T.Interpolated(0.0, 1.0, np.array([0.0, 1.0, 2.0]), np.array([2.0, 3.0, 4.0]))
    traj = T.SE3Trajectory("body", "base", angular, linear)
    H, Tw, Td = traj(0.5, 2)
    assert H["from_frame"] == "body" and H["to_frame"] == "base"
    for X in (Tw, Td):
        assert X["body"] == "body" and X["base"] == "base" and X["frame"] == "body"
    # values (not asserted by the reference, which only checks frames): constant-rate motion
    R = T.Quat.from_angle_axis(angle / 2, axis).matrix()
    np.testing.assert_allclose(H["R"], R, atol=1e-15)
    np.testing.assert_allclose(H["p"], [1.0, 2.0, 3.0], atol=1e-15)
    np.testing.assert_allclose(Tw["v"][:3], R.T @ (axis * angle), atol=1e-15)
    np.testing.assert_allclose(Tw["v"][3:], R.T @ np.array([2.0, 2.0, 2.0]), atol=1e-15)
    np.testing.assert_allclose(Td["v"][:3], 0.0, atol=1e-15)
    np.testing.assert_allclose(Td["v"][3:], np.cross(Tw["v"][:3], Tw["v"][3:]), atol=1e-15)  # nu_d = R' pdd + w x nu, pdd = 0


def test_se3pd_controller():
    """src/lowlevel/se3pdcontroller.jl; the reference only smoke-tests it (test/controller.jl:1-20, "FIXME:
    proper tests"): same construction here, plus two properties of the law -- on the reference trajectory
    the feedback vanishes (output = reference acceleration), off it the linear term pulls back."""
    from qpcontrol_jl_amd import api, models as M
    mech = M.Mechanism()  # one floating body whose joint frame coincides with the world frame
    mech.add_body("body", M.WORLD, "floating", M.QUAT_FLOATING, mass=3.0, com=(0.1, 0.0, 0.0), inertia=(1.0, 1.0, 1.0, 0.0, 0.0, 0.0))
    angle, axis = np.pi / 2, np.array([1.0, 0.0, 0.0])
    angular = T.Interpolated(0.0, 1.0, T.Quat.identity(), T.Quat.from_angle_axis(angle, axis))
    linear = T.Interpolated(0.0, 1.0, np.array([0.0, 1.0, 2.0]), np.array([2.0, 3.0, 4.0]))
    traj = T.SE3Trajectory("body", "base", angular, linear)
    gains = api.SE3PDGains(api.PDGains(100.0, 20.0), api.PDGains(1000.0, 200.0))
    ctrl = api.SE3PDController(mech, M.WORLD, 0, traj, np.diag([0, 0, 0, 10.0, 10.0, 10.0]), gains)
    rng = np.random.default_rng(1)
    q, v = M.rand_state(mech, rng)
    out = ctrl(0.5, (q, v))  # the reference's smoke call
    assert out.shape == (1, 6) and np.all(np.isfinite(out))
    # put the floating body exactly on the trajectory at t = 0.5
    H, Tw, Td = traj(0.5, 2)
    qq = q.copy()
    qq[:4] = T.Quat.from_angle_axis(angle / 2, axis).q
    qq[4:7] = H["p"]
    vv = v.copy()
    vv[:6] = Tw["v"]  # floating joint velocity = body-frame twist
    np.testing.assert_allclose(ctrl(0.5, (qq, vv))[0], Td["v"], atol=1e-9)
    # displaced along +x of the base frame: the linear feedback points back (-x), rotated into the body frame
    qd = qq.copy()
    qd[4] += 0.01
    fb = ctrl(0.5, (qd, vv))[0] - Td["v"]
    np.testing.assert_allclose(H["R"] @ fb[3:], [-1000.0 * 0.01, 0.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(fb[:3], 0.0, atol=1e-9)


def test_quat_interpolation_takes_the_long_arc_for_antipodal_signs():
    """interpolated.jl:76-83 with Rotations 0.13.0's AngleAxis(::Quat) (theta = 2 atan(|v|, w), no hemisphere flip): when
    inv(y0) * yf has w < 0 the angle is 2 pi - a about -n; same end point, different path and Lie derivative"""
    a, n = np.pi / 2, np.array([1.0, 0.0, 0.0])
    y0 = T.Quat.identity()
    yf_short = T.Quat.from_angle_axis(a, n)
    yf_long = T.Quat(*(-yf_short.q))
    for yf, angle, axis in ((yf_short, a, n), (yf_long, 2 * np.pi - a, -n)):
        ang, ax = (y0.inv() * yf).angle_axis()
        assert abs(ang - angle) < 1e-15 and np.allclose(ax, axis, atol=1e-15)
        traj = T.Interpolated(0.0, 1.0, y0, yf, min_num_derivs=2)
        y, yd, ydd = traj(0.5, 2)
        assert y.isapprox(T.Quat.from_angle_axis(0.5 * angle, axis), atol=1e-15)
        assert np.allclose(yd, axis * angle, atol=1e-14) and np.allclose(ydd, 0.0, atol=1e-14)
        y1 = traj(1.0, 2)[0]
        assert np.allclose(y1.matrix(), yf_short.matrix(), atol=1e-15)  # both arcs end on the same rotation


def test_bezier_curve_matches_the_power_basis_and_drives_an_interpolated_trajectory():
    """BezierCurve = BernsteinPolynomial (trajectories.jl:26): the Bernstein form of the smoothstep 3 t^2 - 2 t^3 is
    (0, 0, 1, 1); value and derivatives agree with the power basis, and as the interpolator of an Interpolated trajectory it
    gives the same samples as the fit_cubic time warp (interpolated.jl:20-57)."""
    b = T.BezierCurve([0.0, 0.0, 1.0, 1.0])
    p = T.Polynomial([0.0, 0.0, 3.0, -2.0])
    for t in np.linspace(0.0, 1.0, 11):
        assert abs(b(t) - p(t)) < 1e-15
        assert abs(b.derivative()(t) - p.derivative()(t)) < 1e-14
        assert abs(b.derivative().derivative()(t) - p.derivative().derivative()(t)) < 1e-13
    assert T.BezierCurve([2.5]).derivative()(0.3) == 0.0
    f = T.fit_cubic(x0=0.0, xf=1.0, y0=0.0, yd0=0.0, yf=1.0, ydf=0.0)
    y0, yf = np.array([1.0, 2.0, 3.0]), np.array([-1.0, 0.5, 4.0])
    tb = T.Interpolated(2.0, 5.0, y0, yf, b, min_num_derivs=2)
    tp = T.Interpolated(2.0, 5.0, y0, yf, f, min_num_derivs=2)
    for x in (1.0, 2.0, 2.7, 4.1, 5.0, 6.0):
        for u, w in zip(tb(x, 2), tp(x, 2)):
            np.testing.assert_allclose(u, w, rtol=0, atol=1e-13)
