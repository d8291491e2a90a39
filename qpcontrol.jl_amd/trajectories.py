"""Host mirror of QPControl.Trajectories (reference src/trajectories/*.jl; exports
src/trajectories/trajectories.jl:3-13): the reference signals a high-level controller samples before
it calls setdesired!.  Pure scalar / small-vector arithmetic, evaluated per batch
element on the host with numpy -- it is a caller of the hot path, nothing here touches the device.

Conventions kept from the reference: a trajectory is called as `traj(x)` for the value or
`traj(x, n)` for `(value, 1st derivative, ..., n-th derivative)` (Julia: `traj(x, Val(n))`);
rotations are unit quaternions [w, x, y, z] and their derivatives are 3-vectors (angular velocity
in the interpolation's own frame, "Lie derivative", interpolated.jl:82-89); `clamp` holds the end
values outside [x0, xf], otherwise a `DomainError` (here: ValueError) is raised (trajectories.jl:29-31).
"""
from __future__ import annotations

import numpy as np


class TrajectoryDomainError(ValueError):
    """trajectories.jl:29-31 throw_trajectory_domain_error"""


# ------------------------------------------------------------------------------------------
# polynomials (StaticUnivariatePolynomials.Polynomial: coefficients in ascending order)
# ------------------------------------------------------------------------------------------
class Polynomial:
    """p(x) = sum_j c_j s^j with s = (x - shift) / scale.  The reference's Polynomial is the plain monomial
    form (shift 0, scale 1); the fitted polynomials keep the interval-normalised variable because their
    monomial coefficients in x are huge and cancel catastrophically when x0 and xf are close."""

    def __init__(self, coeffs, shift=0.0, scale=1.0):
        self.c = tuple(float(c) for c in coeffs)
        self.shift, self.scale = float(shift), float(scale)

    def __call__(self, x):
        sx = (x - self.shift) / self.scale
        acc = 0.0
        for c in reversed(self.c):  # Horner
            acc = acc * sx + c
        return acc

    def derivative(self):
        return Polynomial([k * c / self.scale for k, c in enumerate(self.c)][1:] or [0.0], self.shift, self.scale)

    @property
    def coeffs(self):
        """monomial coefficients in x, ascending (the reference's representation)"""
        out = np.zeros(len(self.c))
        for j, cj in enumerate(self.c):
            term = np.array([1.0])
            for _ in range(j):
                term = np.convolve(term, np.array([-self.shift / self.scale, 1.0 / self.scale]))
            out[:j + 1] += cj * term
        return tuple(out)


class BezierCurve:
    """`const BezierCurve = BernsteinPolynomial` (trajectories.jl:26; the type comes from the un-vendored
    StaticUnivariatePolynomials): b(t) = sum_i c_i C(n, i) t^i (1 - t)^(n - i) on [0, 1], evaluated by de Casteljau's
    recurrence; `derivative()` is the Bernstein polynomial of degree n - 1 with coefficients n (c_{i+1} - c_i).  Usable
    as the interpolator of an `Interpolated` trajectory (it maps [0, 1] -> [0, 1] when c_0 = 0 and c_n = 1)."""

    def __init__(self, coeffs):
        self.c = tuple(float(c) for c in coeffs)
        if not self.c:
            raise ValueError("a Bezier curve needs at least one control value")

    def __call__(self, t):
        b = list(self.c)
        for k in range(len(b) - 1, 0, -1):
            b = [(1.0 - t) * b[i] + t * b[i + 1] for i in range(k)]
        return b[0]

    def derivative(self):
        n = len(self.c) - 1
        return BezierCurve([n * (self.c[i + 1] - self.c[i]) for i in range(n)] or [0.0])


def _coefficient_gradient(x, num_coeffs, order):
    """d^order/dx^order of [1, x, x^2, ...] (SUP.coefficient_gradient)"""
    row = []
    for k in range(num_coeffs):
        if k < order:
            row.append(0.0)
        else:
            f = 1.0
            for j in range(order):
                f *= (k - j)
            row.append(f * x ** (k - order))
    return row


def _fit_hermite(x0, xf, left, right):
    """Polynomial of degree 2k - 1 with prescribed value and first k - 1 derivatives at x0 and xf.
    The reference solves the Vandermonde-type system directly in x (fit_polynomial.jl), which is
    ill-conditioned when x0 and xf are close; here the same interpolation problem is solved in
    s = (x - x0) / (xf - x0) and kept in that variable (same polynomial, better rounding)."""
    k = len(left)
    n = 2 * k
    h = float(xf) - float(x0)
    A = np.array([_coefficient_gradient(0.0, n, d) for d in range(k)] + [_coefficient_gradient(1.0, n, d) for d in range(k)])
    rhs = np.array([left[d] * h ** d for d in range(k)] + [right[d] * h ** d for d in range(k)], dtype=float)
    return Polynomial(np.linalg.solve(A, rhs), shift=x0, scale=h)  # p(x) = sum_j c_j ((x - x0) / h)^j


def fit_cubic(*, x0, xf, y0, yd0, yf, ydf):
    """fit_polynomial.jl:1-11"""
    return _fit_hermite(x0, xf, (y0, yd0), (yf, ydf))


def fit_quintic(*, x0, xf, y0, yd0, ydd0, yf, ydf, yddf):
    """fit_polynomial.jl:13-25"""
    return _fit_hermite(x0, xf, (y0, yd0, ydd0), (yf, ydf, yddf))


class PolyDerivEvaluator:
    """poly_deriv_evaluator.jl: the derivative polynomials are formed once"""

    def __init__(self, p, num_derivs):  # p: Polynomial or BezierCurve (anything with __call__ and derivative())
        self.f = p
        self.derivs = []
        q = p
        for _ in range(num_derivs):
            q = q.derivative()
            self.derivs.append(q)

    def __call__(self, theta, num_derivs):
        if num_derivs < 0:
            raise ValueError("num_derivs must be nonnegative")
        while len(self.derivs) < num_derivs:
            self.derivs.append((self.derivs[-1] if self.derivs else self.f).derivative())
        return (self.f(theta),) + tuple(d(theta) for d in self.derivs[:num_derivs])


# ------------------------------------------------------------------------------------------
# rotations
# ------------------------------------------------------------------------------------------
class Quat:
    """unit quaternion [w, x, y, z] (Rotations.Quat)"""

    def __init__(self, w, x, y, z):
        self.q = np.array([w, x, y, z], dtype=float)

    @staticmethod
    def identity():
        return Quat(1.0, 0.0, 0.0, 0.0)

    @staticmethod
    def from_angle_axis(angle, axis):
        axis = np.asarray(axis, dtype=float)
        n = np.linalg.norm(axis)
        a = axis / n if n > 0 else axis
        return Quat(np.cos(angle / 2), *(np.sin(angle / 2) * a))

    def __mul__(self, o):
        w1, x1, y1, z1 = self.q
        w2, x2, y2, z2 = o.q
        return Quat(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                    w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2)

    def inv(self):
        w, x, y, z = self.q
        return Quat(w, -x, -y, -z)

    def angle_axis(self):
        # AngleAxis(::Quat) of Rotations 0.13.0 (the version pinned in the reference's Manifest.toml:136-140):
        # theta = 2 atan(|v|, w), axis = v / |v| -- NO flip to the w >= 0 hemisphere, so w < 0 yields an angle in
        # (pi, 2 pi] and interpolate(::Quat, ::Quat, alpha) (interpolated.jl:76-83) then takes the long arc
        w, v = self.q[0], self.q[1:]
        s = np.linalg.norm(v)
        angle = 2.0 * np.arctan2(s, w)
        axis = v / s if s > 1e-300 else np.array([1.0, 0.0, 0.0])
        return angle, axis

    def matrix(self):
        w, x, y, z = self.q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def isapprox(self, o, atol=1e-12):
        return min(np.abs(self.q - o.q).max(), np.abs(self.q + o.q).max()) <= atol


def _zero_tangent(y):
    """trajectories.jl:33-35"""
    if isinstance(y, Quat):
        return np.zeros(3)
    return np.zeros_like(np.asarray(y, dtype=float)) if np.ndim(y) else 0.0


def _interpolate(y0, yf, alpha):
    """interpolated.jl:68-90: value and dy/dalpha"""
    if isinstance(y0, Quat):
        angle, axis = (y0.inv() * yf).angle_axis()
        return y0 * Quat.from_angle_axis(alpha * angle, axis), axis * angle
    y0a, yfa = np.asarray(y0, dtype=float), np.asarray(yf, dtype=float)
    dy = yfa - y0a
    y = y0a + alpha * dy
    return (y if np.ndim(y) else float(y)), (dy if np.ndim(dy) else float(dy))


# ------------------------------------------------------------------------------------------
# trajectories
# ------------------------------------------------------------------------------------------
class Constant:
    """constant.jl"""

    def __init__(self, value):
        self.value = value

    def __call__(self, x, num_derivs=None):
        if num_derivs is None:
            return self.value
        return (self.value,) + tuple(_zero_tangent(self.value) for _ in range(num_derivs))


class Interpolated:
    """interpolated.jl:1-66: y(x) = interpolate(y0, yf, alpha(theta(x))), theta = clamp((x - x0) / (xf - x0), 0, 1),
    alpha = interpolator (identity or a Polynomial warping [0, 1] -> [0, 1])."""

    def __init__(self, x0, xf, y0, yf, f=None, min_num_derivs=0, clamp=True):
        self.x0, self.xf, self.y0, self.yf, self.clamp = float(x0), float(xf), y0, yf, clamp
        self.interpolator = PolyDerivEvaluator(f, min_num_derivs) if isinstance(f, (Polynomial, BezierCurve)) else f

    def _alpha(self, theta, n):
        if self.interpolator is None:  # identity: nthderiv (interpolated.jl:99)
            return (theta,) + tuple(1.0 if k == 1 else 0.0 for k in range(1, n + 1))
        return self.interpolator(theta, n)

    def __call__(self, x, num_derivs=None):
        n = 0 if num_derivs is None else num_derivs
        if not self.clamp and (x < self.x0 or x > self.xf):
            raise TrajectoryDomainError(f"Trajectory evaluated outside of range [{self.x0}, {self.xf}]")
        dx = self.xf - self.x0
        theta = (x - self.x0) / dx
        if theta <= 0.0:
            theta, dtheta = 0.0, 0.0
        elif theta >= 1.0:
            theta, dtheta = 1.0, 0.0
        else:
            dtheta = 1.0 / dx
        vals = self._alpha(theta, n)
        alpha = vals[0]
        alpha_x = [vals[i] * dtheta ** i for i in range(1, n + 1)]  # higher derivatives of theta vanish
        y, dyda = _interpolate(self.y0, self.yf, alpha)
        if num_derivs is None:
            return y
        return (y,) + tuple(dyda * a for a in alpha_x)


class Piecewise:
    """piecewise.jl: subfunctions[i] lives on [breaks[i], breaks[i+1]] and is evaluated at x - breaks[i]"""

    def __init__(self, subfunctions, breaks, clamp=True):
        breaks = list(breaks)
        assert all(breaks[i] <= breaks[i + 1] for i in range(len(breaks) - 1))
        assert len(breaks) == len(subfunctions) + 1
        self.subfunctions, self.breaks, self.clamp = list(subfunctions), breaks, clamp

    def __call__(self, x, *args):
        x0, xf = self.breaks[0], self.breaks[-1]
        if self.clamp:
            x = min(max(x, x0), xf)
        elif x < x0 or x > xf:
            raise TrajectoryDomainError(f"Trajectory evaluated outside of range [{x0}, {xf}]")
        index = min(int(np.searchsorted(self.breaks, x, side="right")) - 1, len(self.subfunctions) - 1)
        return self.subfunctions[index](x - self.breaks[index], *args)


class PointTrajectory:
    """point_vector.jl:1-12: (Point3D, FreeVector3D...) = (frame, array) pairs"""

    def __init__(self, frame, trajectory):
        self.frame, self.trajectory = frame, trajectory

    def __call__(self, x, num_derivs=None):
        if num_derivs is None:
            return (self.frame, self.trajectory(x))
        return tuple((self.frame, v) for v in self.trajectory(x, num_derivs))


class FreeVectorTrajectory(PointTrajectory):
    """point_vector.jl:15-26"""


class SE3Trajectory:
    """se3.jl: angular = rotation body -> base (derivatives in base), linear = translation body -> base
    (derivatives in base).  Returns H = (R, p), twist and spatial acceleration of body w.r.t. base
    expressed in BODY frame as 6-vectors [angular; linear]."""

    def __init__(self, body, base, angular, linear):
        self.body, self.base, self.angular, self.linear = body, base, angular, linear

    def __call__(self, x, num_derivs=2):
        if num_derivs != 2:
            raise ValueError("SE3Trajectory: num_derivs must be 2 (se3.jl:9)")
        Rq, w_base, wd_base = self.angular(x, 2)
        p, pd, pdd = self.linear(x, 2)
        R = Rq.matrix() if isinstance(Rq, Quat) else np.asarray(Rq, dtype=float)
        Rinv = R.T
        w_body, nu_body = Rinv @ w_base, Rinv @ pd
        wd_body = Rinv @ wd_base
        nud_body = Rinv @ pdd + np.cross(w_body, nu_body)
        H = dict(from_frame=self.body, to_frame=self.base, R=R, p=np.asarray(p, dtype=float))
        T = dict(body=self.body, base=self.base, frame=self.body, v=np.concatenate([w_body, nu_body]))
        Td = dict(body=self.body, base=self.base, frame=self.body, v=np.concatenate([wd_body, nud_body]))
        return H, T, Td
