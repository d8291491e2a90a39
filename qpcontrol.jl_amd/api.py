"""Host mirror of the reference's operator interface for the hot path, batched.

Same names and argument meaning as `src/QPControl.jl:5-33` (Julia `!` dropped): tasks are
constructed from a mechanism and body/joint names, `addtask` has the hard / scalar-weight /
matrix-weight forms of `momentum.jl:99-117`, `addcontact` returns a `ContactPoint` whose `weight`
and `maxnormalforce` cells (contacts.jl:35-36, default 0 as in contacts.jl:50) may be scalars or
per-state arrays, `setdesired` stores what the next tick uses, and the controller is callable as
`controller(tau, t, x)` with `x = (q[B, nq], v[B, nv])`.  After a call `controller.result_vdot` and
`controller.contactwrenches` hold what `controller.result.v̇` / `controller.contactwrenches` hold
in the reference (momentum.jl:62-72).  Everything numeric happens in libqpcontrol_hip.so.
"""
from __future__ import annotations

import numpy as np

from . import binding
from .models import FRAME_DEFAULT, WORLD, Mechanism
from .workloads import (ANGULAR, HARD, JOINT, LINEAR, LINEAR_MOMENTUM_RATE, MATRIX, MATRIX_PER_QP,
                        MOMENTUM_RATE, POINT, SCALAR, SCALAR_PER_QP, SPATIAL)


class QPSolveFailure(RuntimeError):
    """exceptions.jl:1-11 -- raised by checkstatus (momentum.jl:83-91) for the first bad element."""

    def __init__(self, index, status):
        self.index, self.status = index, status
        # the reference's three fields (terminationstatus, primalstatus, dualstatus), from qpc_status_triple
        self.terminationstatus, self.primalstatus, self.dualstatus = binding.status_triple(status)
        super().__init__(f"QP solve unsuccessful (batch element {index}). Termination status: {self.terminationstatus}. "
                         f"Primal status: {self.primalstatus}. Dual status: {self.dualstatus}.")


def _body(mech: Mechanism, b):
    return b if isinstance(b, (int, np.integer)) else mech.findbody(b)


def _joint(mech: Mechanism, j):
    return j if isinstance(j, (int, np.integer)) else mech.findjoint(j)


class AbstractMotionTask:
    kind = None
    dim = 0

    def __init__(self):
        self.desired = None
        self._id = None

    def _args(self):
        return {}


class _PathTask(AbstractMotionTask):
    def __init__(self, mechanism, base, target, frame=FRAME_DEFAULT):
        super().__init__()
        self.base, self.target = _body(mechanism, base), _body(mechanism, target)
        self.frame = frame if frame == FRAME_DEFAULT else _body(mechanism, frame)
        self.desired = np.zeros(self.dim)

    def _args(self):
        return dict(base=self.base, target=self.target, frame=self.frame)


class SpatialAccelerationTask(_PathTask):  # src/tasks.jl:3-44
    kind, dim = SPATIAL, 6


class AngularAccelerationTask(_PathTask):  # src/tasks.jl:47-84
    kind, dim = ANGULAR, 3


class LinearAccelerationTask(_PathTask):  # src/tasks.jl:86-123
    kind, dim = LINEAR, 3


class PointAccelerationTask(AbstractMotionTask):  # src/tasks.jl:125-171
    kind, dim = POINT, 3

    def __init__(self, mechanism, base, target, point):
        super().__init__()
        self.base, self.target = _body(mechanism, base), _body(mechanism, target)
        self.point = np.asarray(point, dtype=np.float64)
        self.desired = np.zeros(3)

    def _args(self):
        return dict(base=self.base, target=self.target, point=self.point)


class JointAccelerationTask(AbstractMotionTask):  # src/tasks.jl:173-189
    kind = JOINT

    def __init__(self, mechanism, joint):
        super().__init__()
        self.joint = _joint(mechanism, joint)
        self.dim = len(mechanism.velocity_range(self.joint))
        self.desired = np.zeros(self.dim)

    def _args(self):
        return dict(joint=self.joint)


class MomentumRateTask(AbstractMotionTask):  # src/tasks.jl:192-236
    kind, dim = MOMENTUM_RATE, 6

    def __init__(self, mechanism=None, centroidalframe=None):
        super().__init__()
        self.desired = np.zeros(6)


class LinearMomentumRateTask(AbstractMotionTask):  # src/tasks.jl:239-262
    kind, dim = LINEAR_MOMENTUM_RATE, 3

    def __init__(self, mechanism=None, centroidalframe=None):
        super().__init__()
        self.desired = np.zeros(3)


def setdesired(task: AbstractMotionTask, desired):
    """setdesired!(task, desired): a (dim,) value for every state or a (B, dim) array."""
    d = np.asarray(desired, dtype=np.float64)
    if d.ndim == 0:
        d = d.reshape(1)
    if d.shape[-1] != task.dim:
        raise ValueError(f"desired has trailing dimension {d.shape[-1]}, task dimension is {task.dim}")
    task.desired = d


class ContactPoint:
    """contacts.jl:27-36: `weight` and `maxnormalforce` default to 0 (a fresh contact is disabled)."""

    def __init__(self, cid, body, position, normal, mu, parametric):
        self._id, self.body = cid, body
        self.position, self.normal, self.mu = np.asarray(position, float), np.asarray(normal, float), mu
        self.weight = 0.0
        self.maxnormalforce = 0.0
        self.parametric = parametric


def disable(point: ContactPoint):
    point.maxnormalforce = 0.0


def isenabled(point: ContactPoint):
    return bool(np.any(np.asarray(point.maxnormalforce) > 0))


class PDGains:
    """RigidBodyDynamics.PDControl.PDGains(k, d) as used by standing.jl:26-28"""

    def __init__(self, k, d):
        self.k, self.d = float(k), float(d)


class Parameter:
    """Marker for a per-state ("Parameter" in Parametron) weight: `value` is (B,) or (B, dim, dim)."""

    def __init__(self, value):
        self.value = np.asarray(value, dtype=np.float64)


class MomentumBasedController:
    """MomentumBasedController{N}(mechanism, optimizer; floatingjoint) -- momentum.jl:1-81 -- without an
    optimizer argument: the QP is solved by the HIP kernels."""

    def __init__(self, mechanism: Mechanism, n_basis=4, floatingjoint=None, lib_path=None):
        self.mechanism = mechanism
        self.floating = WORLD if floatingjoint is None else _joint(mechanism, floatingjoint)
        self._c = binding.Controller(binding.Model(mechanism, lib_path), n_basis, self.floating)
        self.tasks, self.contacts, self._weights = [], [], []
        self.initialized = False
        self.result_vdot = None
        self.contactwrenches = None  # {body: [B, 6]} after a tick: the reference's per-body sums
        self.pointwrenches = None    # [B, nc, 6]: per contact point
        self.status = None

    # -- construction ------------------------------------------------------------------------
    def addtask(self, task: AbstractMotionTask, weight=None):
        if weight is None:
            wk, w = HARD, None
        elif isinstance(weight, Parameter):
            wk, w = (SCALAR_PER_QP if weight.value.ndim <= 1 else MATRIX_PER_QP), None
            self._weights.append((task, weight))
        elif np.ndim(weight) == 0:
            wk, w = SCALAR, [float(weight)]
        else:
            wk, w = MATRIX, np.asarray(weight, dtype=np.float64)
        task._id = self._c.add_task(task.kind, wkind=wk, weight=w, **task._args())
        self.tasks.append(task)
        return task

    def regularize(self, joint, weight):
        self._c.regularize(_joint(self.mechanism, joint), weight)

    def setlaw(self, task: AbstractMotionTask, law, gains: PDGains = None, ref=()):
        """Replace the per-tick `setdesired!(task, pd(...))` of a high-level controller
        (standing.jl:58-89) by a law the device evaluates from each state (qpc_task_set_law)."""
        g = gains or PDGains(0.0, 0.0)
        self._c.set_task_law(task._id, law, g.k, g.d, ref)
        task.desired = None

    def addcontact(self, body, position, normal, mu, parametric=False):
        b = _body(self.mechanism, body)
        cid = self._c.add_contact(b, position, normal, mu, per_qp_surface=int(parametric))
        cp = ContactPoint(cid, b, position, normal, mu, parametric)
        self.contacts.append(cp)
        return cp

    def finalize(self, device_id=0):
        """initialize!(controller) (momentum.jl:150-156); implicit on the first call."""
        self.layout = self._c.finalize(device_id)
        self.initialized = True
        return self.layout

    # -- the tick ----------------------------------------------------------------------------
    def _inputs(self, B):
        """per-QP input arrays from the mutable cells of the reference objects (task.desired,
        point.weight, point.maxnormalforce, Parameter weights), broadcast over the batch"""
        lay = self.layout

        def bc(val, n):
            return np.ascontiguousarray(np.broadcast_to(np.asarray(val, dtype=np.float64).reshape(-1, n) if np.ndim(val) > 1
                                                        else np.asarray(val, dtype=np.float64), (B, n)))

        des = None
        if self.tasks and lay.n_desired_inputs:
            des = np.concatenate([bc(tk.desired if tk.desired is not None else np.zeros(tk.dim), tk.dim)
                                  for tk in self.tasks], axis=1)
        cw = np.stack([np.broadcast_to(np.asarray(c.weight, float), (B,)) for c in self.contacts], 1) if self.contacts else None
        cm = np.stack([np.broadcast_to(np.asarray(c.maxnormalforce, float), (B,)) for c in self.contacts], 1) if self.contacts else None
        tw = None
        if self._weights:
            tw = np.concatenate([np.broadcast_to(w.value.reshape(w.value.shape[0] if w.value.ndim else 1, -1)
                                                 if w.value.ndim else w.value.reshape(1, 1),
                                                 (B, 1 if w.value.ndim <= 1 else tk.dim * tk.dim))
                                 for tk, w in self._weights], axis=1)
        return des, cw, cm, tw

    def __call__(self, tau, t, x, surfaces=None, throw_on_failure=True):
        if not self.initialized:
            self.finalize()
        q, v = x
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        v = np.atleast_2d(np.asarray(v, dtype=np.float64))
        des, cw, cm, tw = self._inputs(q.shape[0])
        r = self._c.control_batch(q, v, des, cw, cm, tw, surfaces)
        self.result_vdot, self.pointwrenches, self.status = r["vdot"], r["cwrench"], r["status"]
        # contactwrenches[BodyID(body)] (momentum.jl:64-72): per body that carries contact points, the sum of its points' wrenches
        if self.contacts:
            bodies, _ = self._c.contact_bodies()
            sums = self._c.contact_wrench_per_body(r["cwrench"])
            self.contactwrenches = {b: sums[:, i] for i, b in enumerate(bodies)}
        else:
            self.contactwrenches = {}
        self.iters = r["iters"]
        if throw_on_failure:
            bad = np.where(r["status"] > 1)[0]
            if len(bad):
                raise QPSolveFailure(int(bad[0]), int(r["status"][bad[0]]))
        if tau is not None:
            np.copyto(tau, r["tau"])
            return tau
        return r["tau"]

    # -- closed loop -------------------------------------------------------------------------
    def simulate(self, x, n_steps, ground: "SoftGround", dt=1e-3, surfaces=None):
        """Batched counterpart of the notebook's acceptance run (Standing controller.ipynb cells 9-11:
        `Dynamics(mechanism, controller)` + soft contacts + `simulate`): n_steps x (tick -> forward-dynamics
        step on compliant ground at this controller's contact points), all on the device
        (qpc_rollout_batch).  Returns dict(q, v, tau, status, bad) with bad[b] = failed QPs of robot b."""
        if not self.initialized:
            self.finalize()
        q, v = x
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        v = np.atleast_2d(np.asarray(v, dtype=np.float64))
        des, cw, cm, tw = self._inputs(q.shape[0])
        r = self._c.rollout(q, v, n_steps, ground.params(dt), des, cw, cm, tw, surfaces)
        self.status = r["status"]
        return r


class SoftGround:
    """Compliant half space z <= z under the controller's contact points: normal spring/damper,
    viscous tangential friction capped by Coulomb's law (qpc_sim_params)."""

    def __init__(self, z, stiffness=6.0e4, damping=400.0, tangential_damping=300.0):
        self.z, self.stiffness, self.damping, self.tangential_damping = z, stiffness, damping, tangential_damping

    def params(self, dt):
        return binding.SimParams(dt, self.z, self.stiffness, self.damping, self.tangential_damping)


class StandingController:
    """StandingController(lowlevel, feet, pelvis, nominalstate; ...) -- src/highlevel/standing.jl:1-89.

    Construction registers exactly the reference's task set on `lowlevel` (standing.jl:35-49).  The
    tick of the reference evaluates three PD laws on the CPU and calls `setdesired!` (standing.jl:
    58-86) before `lowlevel(tau, t, state)`; here the laws are handed to the device at construction,
    so the call is a pure (q, v) -> tau map for a batch of states."""

    def __init__(self, lowlevel: MomentumBasedController, feet, pelvis, nominal_q, joint_regularization=0.05,
                 linear_momentum_weight=1.0, comgains=None, pelvisgains=None, jointgains=None, comref=None,
                 jointrefs=None):
        from .models import FIXED, REVOLUTE
        from .workloads import batch_kinematics
        mech = lowlevel.mechanism
        self.lowlevel = lowlevel
        self.robotmass = mech.total_mass()
        self.comgains = comgains or PDGains(10.0, 2 * np.sqrt(10.0))      # standing.jl:26
        self.pelvisgains = pelvisgains or PDGains(20.0, 2 * np.sqrt(20.0))  # standing.jl:27
        default_joint = PDGains(100.0, 20.0)                              # standing.jl:28
        nominal_q = np.asarray(nominal_q, dtype=np.float64)
        if comref is None:  # standing.jl:29
            comref = batch_kinematics(mech, nominal_q, np.zeros(mech.nv))["com"][0] - np.array([0.0, 0.0, 0.05])
        self.comref = np.asarray(comref, dtype=np.float64)
        for j in range(mech.nbodies):  # standing.jl:35
            if mech.joint_type[j] != FIXED:
                lowlevel.regularize(j, joint_regularization)
        feet = [_body(mech, f) for f in feet]
        self.foottasks = {f: lowlevel.addtask(SpatialAccelerationTask(mech, WORLD, f)) for f in feet}  # :37-38
        self.linmomtask = lowlevel.addtask(LinearMomentumRateTask(mech), linear_momentum_weight)      # :40-41
        self.pelvistask = lowlevel.addtask(AngularAccelerationTask(mech, WORLD, _body(mech, pelvis)))  # :43-44
        on_foot_path = set()
        for f in feet:
            on_foot_path.update(mech.path_joints(WORLD, f))
        self.jointtasks, self.jointrefs, self.jointgains = {}, {}, {}
        for j in range(mech.nbodies):  # standing.jl:46-49
            if mech.joint_type[j] == REVOLUTE and j not in on_foot_path:
                self.jointtasks[j] = lowlevel.addtask(JointAccelerationTask(mech, j))
                self.jointrefs[j] = float(nominal_q[mech.configuration_range(j)[0]]) if jointrefs is None else jointrefs[j]
                self.jointgains[j] = default_joint if jointgains is None else jointgains[j]
        # the laws of standing.jl:58-86, evaluated per state on the device
        for t in self.foottasks.values():
            lowlevel.setlaw(t, binding.LAW_CONSTANT, ref=np.zeros(6))
        lowlevel.setlaw(self.linmomtask, binding.LAW_COM_PD, self.comgains, self.comref)
        lowlevel.setlaw(self.pelvistask, binding.LAW_ORIENTATION_PD, self.pelvisgains)
        for j, t in self.jointtasks.items():
            lowlevel.setlaw(t, binding.LAW_JOINT_PD, self.jointgains[j], [self.jointrefs[j]])

    def __call__(self, tau, t, x, **kw):
        return self.lowlevel(tau, t, x, **kw)


class SE3PDGains:
    """RigidBodyDynamics.PDControl.SE3PDGains(FramePDGains(frame, PDGains(k, d)) x 2) with scalar gains,
    both expressed in the controlled body's frame (test/controller.jl:13)."""

    def __init__(self, angular: PDGains, linear: PDGains):
        self.angular, self.linear = angular, linear


class SE3PDController:
    """SE3PDController(base, body, trajectory, weight, gains) -- src/lowlevel/se3pdcontroller.jl:1-18.

    `controller(t, (q, v))` returns, per state, the desired spatial acceleration of `body` w.r.t.
    `base` expressed in the body frame: `Tdref + pd(gains, H, Href, T, Tref)` (se3pdcontroller.jl:13-18),
    ready for `setdesired(SpatialAccelerationTask(mechanism, base, body), ...)`; `weight` is kept for
    the caller's `addtask(task, weight)` as in the reference struct.  The SE(3) PD law itself lives in
    the un-vendored RigidBodyDynamics.PDControl; it is restated here from its documented form (double
    geodesic law, Bullo & Murray 1995, Thm. 12): with e = inv(Href) H (actual body frame -> desired
    frame) and edot = T - Ad(inv(e)) Tref,  ang = -k_w log(R_e) - d_w edot_ang,
    lin = -k_v R_e' p_e - d_v edot_lin.  [recalled semantics: the reference tree only smoke-tests this
    controller (test/controller.jl:1-20), so its numerics are unpinned.]"""

    def __init__(self, mechanism: Mechanism, base, body, trajectory, weight, gains: SE3PDGains):
        self.mechanism = mechanism
        self.base, self.body = _body(mechanism, base), _body(mechanism, body)
        self.trajectory, self.weight, self.gains = trajectory, weight, gains

    @staticmethod
    def _rotvec(R):
        from .trajectories import Quat  # log map through the quaternion
        tr = np.trace(R)
        w = 0.5 * np.sqrt(max(1.0 + tr, 0.0))
        if w > 1e-8:
            v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4.0 * w)
        else:  # rotation by pi
            d = np.sqrt(np.maximum((np.diag(R) + 1.0) / 2.0, 0.0))
            v = d
        angle, axis = Quat(w, *v).angle_axis()
        return angle * axis

    def device_record(self, t):
        """The trajectory record of the device-side law QPC_LAW_SE3_PD (include/qpc.h) for time(s) t: an array
        [..., 29] = [6 unused | t x0 xf | c0..c5 of the interpolator | quaternion y0 | quaternion yf | p0 | pf].  With it
        in the task's slice of `desired` (binding.Controller.task_desired_range) and
        `set_task_law(task, LAW_SE3_PD, k_ang, d_ang, [k_lin, d_lin])`, the tick evaluates this controller per QP on the
        device instead of the host calling it and `setdesired!` before every tick."""
        from math import comb
        from .trajectories import BezierCurve, Interpolated, Piecewise, Polynomial, PolyDerivEvaluator, Quat, SE3Trajectory
        tr = self.trajectory
        if isinstance(tr, SE3Trajectory) and isinstance(tr.angular, Piecewise) and isinstance(tr.linear, Piecewise):
            return self._piecewise_record(t)
        if not (isinstance(tr, SE3Trajectory) and isinstance(tr.angular, Interpolated) and isinstance(tr.linear, Interpolated)):
            raise TypeError("device_record: an SE3Trajectory of two Interpolated (or two Piecewise) trajectories is needed")
        a, l = tr.angular, tr.linear
        if (a.x0, a.xf) != (l.x0, l.xf) or not isinstance(a.y0, Quat) or not isinstance(a.yf, Quat):
            raise ValueError("device_record: angular and linear parts must share [x0, xf]; angular end points must be Quat")

        def coeffs(f):
            if f is None:
                return (0.0, 1.0, 0.0, 0.0, 0.0, 0.0)
            p = f.f if isinstance(f, PolyDerivEvaluator) else f
            if isinstance(p, BezierCurve):
                # `const BezierCurve = BernsteinPolynomial` (trajectories.jl:26): b(t) = sum_i c_i C(n, i) t^i (1 - t)^(n - i); the device
                # record holds monomial coefficients: a_j = C(n, j) sum_{i <= j} (-1)^(j - i) C(j, i) c_i (exact for the small
                # integers involved: degree <= 5)
                n = len(p.c) - 1
                if n > 5:
                    raise ValueError("device_record: a Bezier interpolator of degree <= 5 is needed")
                mono = [comb(n, j) * sum((-1) ** (j - i) * comb(j, i) * p.c[i] for i in range(j + 1)) for j in range(n + 1)]
                return tuple(mono) + (0.0,) * (5 - n)
            if not isinstance(p, Polynomial) or p.shift != 0.0 or p.scale != 1.0 or len(p.c) > 6:
                raise ValueError("device_record: the interpolator must be a monomial Polynomial or a BezierCurve of degree <= 5 on [0, 1]")
            return tuple(p.c) + (0.0,) * (6 - len(p.c))
        ca = coeffs(a.interpolator)
        if ca != coeffs(l.interpolator):
            raise ValueError("device_record: angular and linear parts must share the interpolator")
        t = np.asarray(t, dtype=np.float64)
        rec = np.zeros(t.shape + (29,))
        rec[..., 6] = t
        rec[..., 7], rec[..., 8] = a.x0, a.xf
        rec[..., 9:15] = ca
        rec[..., 15:19], rec[..., 19:23] = a.y0.q, a.yf.q
        rec[..., 23:26], rec[..., 26:29] = np.asarray(l.y0, dtype=float), np.asarray(l.yf, dtype=float)
        return rec

    def n_pieces(self):
        """pieces of a Piecewise trajectory (the third reference value of `set_task_law(task, LAW_SE3_PD, ...)`), 1 otherwise"""
        from .trajectories import Piecewise
        return len(self.trajectory.angular.subfunctions) if isinstance(self.trajectory.angular, Piecewise) else 1

    def _piecewise_record(self, t):
        """record of a Piecewise SE(3) trajectory (include/qpc.h QPC_SE3_PW_RECORD): [6 unused | t | np + 1 breaks | np x 22]; the
        piece is chosen per QP on the device (piecewise.jl:14-27)"""
        from .trajectories import Interpolated, SE3Trajectory
        tr = self.trajectory
        a, l = tr.angular, tr.linear
        npc = len(a.subfunctions)
        if list(a.breaks) != list(l.breaks) or len(l.subfunctions) != npc or not (2 <= npc <= binding.SE3_MAX_PIECES) or not (a.clamp and l.clamp):
            raise ValueError("device_record: clamped angular and linear Piecewise parts with the same breaks, 2 .. SE3_MAX_PIECES pieces")
        t = np.asarray(t, dtype=np.float64)
        rec = np.zeros(t.shape + (binding.se3_pw_record(npc),))
        rec[..., 6] = t
        rec[..., 7:8 + npc] = np.asarray(a.breaks, dtype=float)
        for i in range(npc):
            if not (isinstance(a.subfunctions[i], Interpolated) and isinstance(l.subfunctions[i], Interpolated)):
                raise TypeError("device_record: the pieces must be Interpolated trajectories")
            one = SE3PDController(self.mechanism, self.base, self.body, SE3Trajectory(tr.body, tr.base, a.subfunctions[i], l.subfunctions[i]),
                                  self.weight, self.gains).device_record(0.0)
            rec[..., 8 + npc + 22 * i:8 + npc + 22 * (i + 1)] = one[7:29]
        return rec

    def __call__(self, t, x):
        from .workloads import batch_kinematics
        q, v = x
        kin = batch_kinematics(self.mechanism, q, v)
        B = np.atleast_2d(q).shape[0]
        Href, Tref, Tdref = self.trajectory(t, 2)
        out = np.zeros((B, 6))

        def pose(b, i):
            return (np.eye(3), np.zeros(3)) if i < 0 else (kin["R"][i][b], kin["p"][i][b])

        def twist(b, i):  # world-frame twist about the world origin
            return (np.zeros(3), np.zeros(3)) if i < 0 else (kin["w"][i][b], kin["v"][i][b])
        for b in range(B):
            Rb, pb = pose(b, self.body)
            Ra, pa = pose(b, self.base)
            R = Ra.T @ Rb                      # H = relative_transform(body -> base)
            p = Ra.T @ (pb - pa)
            wb, vb = twist(b, self.body)
            wa, va = twist(b, self.base)
            w_rel = wb - wa                    # relative twist, world coordinates about the world origin
            v_rel = vb - va
            T_ang = Rb.T @ w_rel               # ... re-expressed in the body frame (origin at the body frame)
            T_lin = Rb.T @ (v_rel + np.cross(w_rel, pb))
            Re = Href["R"].T @ R               # e = inv(Href) * H
            pe = Href["R"].T @ (p - Href["p"])
            # Tref (given in the DESIRED body frame) seen from the actual body frame: Ad(inv(e))
            Tr_ang = Re.T @ Tref["v"][:3]
            Tr_lin = Re.T @ (Tref["v"][3:] + np.cross(Tref["v"][:3], pe))
            ed_ang, ed_lin = T_ang - Tr_ang, T_lin - Tr_lin
            ang = -self.gains.angular.k * self._rotvec(Re) - self.gains.angular.d * ed_ang
            lin = -self.gains.linear.k * (Re.T @ pe) - self.gains.linear.d * ed_lin
            out[b] = Tdref["v"] + np.concatenate([ang, lin])
        return out
