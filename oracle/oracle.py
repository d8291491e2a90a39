"""ctypes front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See oracle/oracle.h for the parity status
("parity unpinned": the Julia reference cannot run in this image).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


class Layout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "nq", "nv", "n_tasks", "n_contacts", "n_basis", "n_desired", "n_task_weights", "n_surface",
        "ref_n_var", "ref_n_eq", "ref_n_ineq", "n_free", "n_pinned", "n_slack", "n_rho",
        "n_eq_rows", "lds_bytes_per_qp", "n_desired_inputs")]


class OsqpSettings(ctypes.Structure):
    _fields_ = [("rho", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double),
                ("eps_prim_inf", ctypes.c_double), ("eps_dual_inf", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("scaling", ctypes.c_int), ("adaptive_rho", ctypes.c_int),
                ("adaptive_rho_interval", ctypes.c_int), ("check_termination", ctypes.c_int),
                ("warm_start", ctypes.c_int), ("adaptive_rho_tolerance", ctypes.c_double),
                ("polish", ctypes.c_int)]


class TickInfo(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("iters", ctypes.c_int), ("polish_ok", ctypes.c_int),
                ("admm_restarts", ctypes.c_int), ("prim_res", ctypes.c_double),
                ("dual_res", ctypes.c_double), ("kkt_stat", ctypes.c_double),
                ("kkt_prim", ctypes.c_double), ("kkt_comp", ctypes.c_double), ("obj", ctypes.c_double)]


def build(native=False):
    """Compile the oracle with the committed Makefile (gcc, seconds)."""
    target = "_build/liboracle_native.so" if native else "_build/liboracle.so"
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".build.lock"), "w") as lk:  # one builder at a time (pytest-xdist workers)
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


def lib(native=False):
    global _LIB
    key = "native" if native else "std"
    if _LIB is None:
        _LIB = {}
    if key not in _LIB:
        path = build(native)  # `make`: a no-op when the library is newer than its sources, a rebuild when they changed
        L = ctypes.CDLL(path)
        L.ora_model_create.restype = ctypes.c_void_p
        L.ora_state_create.restype = ctypes.c_void_p
        L.ora_ctrl_create.restype = ctypes.c_void_p
        for f in ("ora_model_destroy", "ora_state_destroy", "ora_ctrl_destroy"):
            getattr(L, f).restype = None
        _LIB[key] = L
    return _LIB[key]


def _p(a):
    return a.ctypes.data_as(c_double_p)


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


class OracleModel:
    def __init__(self, mech, native=False):
        self.mech = mech
        self.L = lib(native)
        d, self._keep = mech.desc()
        self.h = ctypes.c_void_p(self.L.ora_model_create(ctypes.byref(d)))
        assert self.h, "ora_model_create failed"
        self.nq, self.nv, self.nb = mech.nq, mech.nv, mech.nbodies
        self.s = ctypes.c_void_p(self.L.ora_state_create())

    def __del__(self):
        try:
            self.L.ora_state_destroy(self.s)
            self.L.ora_model_destroy(self.h)
        except Exception:
            pass

    # -- state -------------------------------------------------------------------------------
    def set_state(self, q, v):
        q, v = _arr(q, self.nq), _arr(v, self.nv)
        self.L.ora_state_update(self.h, _p(q), _p(v), self.s)
        return self

    def transform(self, body):
        R, p = np.zeros(9), np.zeros(3)
        self.L.ora_get_transform(self.s, int(body), _p(R), _p(p))
        return R.reshape(3, 3), p

    def twist(self, body):
        T = np.zeros(6)
        self.L.ora_get_twist(self.s, int(body), _p(T))
        return T

    def bias_acceleration(self, body):
        A = np.zeros(6)
        self.L.ora_get_bias(self.s, int(body), _p(A))
        return A

    def center_of_mass(self):
        c = np.zeros(3)
        self.L.ora_get_com(self.s, _p(c))
        return c

    # PD laws of the StandingController tick (standing.jl:58-89) on the current state
    def standing_com_law(self, v, kp, kd, comref):
        out = np.zeros(3)
        self.L.ora_standing_com_law(self.h, self.s, _p(_arr(v, self.nv)), ctypes.c_double(kp), ctypes.c_double(kd),
                                    _p(_arr(comref, 3)), _p(out))
        return out

    def standing_orientation_law(self, body, kp, kd):
        out = np.zeros(3)
        self.L.ora_standing_orientation_law(self.s, int(body), ctypes.c_double(kp), ctypes.c_double(kd), _p(out))
        return out

    def standing_joint_law(self, kp, kd, q, qref, v):
        f = self.L.ora_standing_joint_law
        f.restype = ctypes.c_double
        return f(ctypes.c_double(kp), ctypes.c_double(kd), ctypes.c_double(q), ctypes.c_double(qref), ctypes.c_double(v))

    def momentum_matrix(self):
        A, Adv = np.zeros((6, self.nv)), np.zeros(6)
        self.L.ora_get_momentum_matrix(self.h, self.s, _p(A), _p(Adv))
        return A, Adv

    def geometric_jacobian(self, base, target, frame=-1):
        J = np.zeros((6, self.nv))
        self.L.ora_geometric_jacobian(self.h, self.s, int(base), int(target), int(frame), _p(J))
        return J

    def mass_matrix(self):
        M = np.zeros((self.nv, self.nv))
        self.L.ora_mass_matrix(self.h, self.s, _p(M))
        return M

    def inverse_dynamics(self, vd, wext=None):
        vd = _arr(vd, self.nv)
        tau = np.zeros(self.nv)
        w = None if wext is None else _arr(wext, 6 * self.nb)
        self.L.ora_inverse_dynamics(self.h, self.s, _p(vd), _p(w) if w is not None else None, _p(tau))
        return tau

    def forward_dynamics(self, tau, wext=None):
        tau = _arr(tau, self.nv)
        vd = np.zeros(self.nv)
        w = None if wext is None else _arr(wext, 6 * self.nb)
        rc = self.L.ora_forward_dynamics(self.h, self.s, _p(tau), _p(w) if w is not None else None, _p(vd))
        assert rc == 0
        return vd

    def spatial_accelerations(self, vd):
        vd = _arr(vd, self.nv)
        acc = np.zeros((self.nb, 6))
        self.L.ora_spatial_accelerations(self.h, self.s, _p(vd), _p(acc))
        return acc

    def integrate(self, q, v, dt):
        q = _arr(q, self.nq).copy()
        v = _arr(v, self.nv)
        self.L.ora_integrate(self.h, _p(q), _p(v), ctypes.c_double(dt))
        return q

    def relative_twist(self, body, base, frame):
        T = np.zeros(6)
        self.L.ora_relative_twist(self.h, self.s, int(body), int(base), int(frame), _p(T))
        return T

    def transform_accel(self, Aw, body, base, frame):
        out = np.zeros(6)
        Aw = _arr(Aw, 6)
        self.L.ora_transform_accel(self.h, self.s, _p(Aw), int(body), int(base), int(frame), _p(out))
        return out


def test_settings(polish=2):
    s = OsqpSettings()
    lib().ora_osqp_test_settings(ctypes.byref(s))
    s.polish = polish
    return s


def notebook_settings(polish=0):
    s = OsqpSettings()
    lib().ora_osqp_notebook_settings(ctypes.byref(s))
    s.polish = polish
    return s


class OracleController:
    """Reference-layout controller: same construction calls as the product's C ABI."""

    def __init__(self, model: OracleModel, n_basis=4, floating_body=-1):
        self.model = model
        self.L = model.L
        self.h = ctypes.c_void_p(self.L.ora_ctrl_create(model.h, int(n_basis), int(floating_body)))
        assert self.h
        self.N = n_basis
        self.layout = None
        self.task_dims = []
        self.nc = 0
        self.var_blocks = []  # reference-layout variable blocks after v̇, in creation order

    def __del__(self):
        try:
            self.L.ora_ctrl_destroy(self.h)
        except Exception:
            pass

    def add_task(self, kind, base=-1, target=-1, frame=-2, point=None, joint=-1, wkind=0, weight=None):
        pt = _arr(point if point is not None else np.zeros(3), 3)
        w = None if weight is None else _arr(np.asarray(weight).reshape(-1))
        tid = self.L.ora_add_task(self.h, kind, base, target, frame, _p(pt), joint, wkind,
                                  _p(w) if w is not None else None)
        assert tid >= 0, "ora_add_task failed"
        self.task_dims.append(self.L.ora_task_dim(self.h, tid))
        if wkind != 0:
            self.var_blocks.append(("slack", tid, self.task_dims[-1]))
        return tid

    def regularize(self, joint, w):
        assert self.L.ora_regularize(self.h, int(joint), ctypes.c_double(w)) == 0

    def regularize_per_qp(self, joint):
        """regularize!(c, joint, weight::Parameter): -> index of the weight in the per-QP task-weight record"""
        off = self.L.ora_regularize_per_qp(self.h, int(joint))
        assert off >= 0
        return off

    def add_contact(self, body, pos, normal, mu, per_qp_surface=0):
        cid = self.L.ora_add_contact(self.h, int(body), _p(_arr(pos, 3)), _p(_arr(normal, 3)),
                                     ctypes.c_double(mu), int(per_qp_surface))
        assert cid >= 0
        self.nc += 1
        self.var_blocks.append(("contact", cid, self.N + 9))
        self.contact_specs = getattr(self, "contact_specs", []) + [(np.asarray(normal, float), float(mu), int(per_qp_surface))]
        return cid

    def finalize(self):
        L = Layout()
        assert self.L.ora_finalize(self.h, ctypes.byref(L)) == 0
        self.layout = L
        return L

    def set_settings(self, s: OsqpSettings):
        self.L.ora_set_osqp_settings(self.h, ctypes.byref(s))

    def task_eval(self, task):
        d = self.task_dims[task]
        J, b = np.zeros((d, self.model.nv)), np.zeros(d)
        assert self.L.ora_task_eval(self.h, self.model.s, task, _p(J), _p(b)) == 0
        return J, b

    def contact_eval(self, ci, normal, mu, maxf):
        B, mr, R, p = np.zeros((3, self.N)), ctypes.c_double(0), np.zeros(9), np.zeros(3)
        n = _arr(normal, 3).copy()
        self.L.ora_contact_eval(self.h, self.model.s, ci, _p(n), ctypes.c_double(mu),
                                ctypes.c_double(maxf), _p(B), ctypes.byref(mr), _p(R), _p(p))
        return B, mr.value, R.reshape(3, 3), p

    def _inputs(self, q, v, desired, cweight, cmaxf, tw, surf, B=1):
        L = self.layout
        q, v = _arr(q, B * L.nq), _arr(v, B * L.nv)
        desired = _arr(desired if L.n_desired else np.zeros(1))
        cweight = _arr(cweight if L.n_contacts else np.zeros(1))
        cmaxf = _arr(cmaxf if L.n_contacts else np.zeros(1))
        tw = _arr(tw) if L.n_task_weights else None
        surf = _arr(surf) if L.n_surface else None
        return q, v, desired, cweight, cmaxf, tw, surf

    def build_qp(self, q, v, desired=(), cweight=(), cmaxf=(), tw=None, surf=None):
        L = self.layout
        n, m = L.ref_n_var, L.ref_n_eq + L.ref_n_ineq
        q, v, desired, cweight, cmaxf, tw, surf = self._inputs(q, v, desired, cweight, cmaxf, tw, surf)
        P, qv, A, l, u = np.zeros((n, n)), np.zeros(n), np.zeros((m, n)), np.zeros(m), np.zeros(m)
        rc = self.L.ora_build_qp(self.h, _p(q), _p(v), _p(desired), _p(cweight), _p(cmaxf),
                                 _p(tw) if tw is not None else None,
                                 _p(surf) if surf is not None else None, _p(P), _p(qv), _p(A), _p(l), _p(u))
        assert rc == 0
        return P, qv, A, l, u

    def tick(self, q, v, desired=(), cweight=(), cmaxf=(), tw=None, surf=None, warm=False):
        L = self.layout
        q, v, desired, cweight, cmaxf, tw, surf = self._inputs(q, v, desired, cweight, cmaxf, tw, surf)
        tau, vd = np.zeros(L.nv), np.zeros(L.nv)
        cw = np.zeros((max(L.n_contacts, 1), 6))
        x = np.zeros(L.ref_n_var)
        info = TickInfo()
        st = self.L.ora_tick(self.h, _p(q), _p(v), _p(desired), _p(cweight), _p(cmaxf),
                             _p(tw) if tw is not None else None,
                             _p(surf) if surf is not None else None, int(warm), _p(tau), _p(vd),
                             _p(cw), _p(x), ctypes.byref(info))
        return dict(status=st, tau=tau, vdot=vd, cwrench=cw[:L.n_contacts], x=x, info=info)

    def tick_batch(self, q, v, desired=None, cweight=None, cmaxf=None, tw=None, surf=None, warm=False):
        L = self.layout
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        q, v, desired, cweight, cmaxf, tw, surf = self._inputs(
            q, v, desired if desired is not None else (), cweight if cweight is not None else (),
            cmaxf if cmaxf is not None else (), tw, surf, B=B)
        tau, vd = np.zeros((B, L.nv)), np.zeros((B, L.nv))
        cw = np.zeros((B, max(L.n_contacts, 1) * 6))
        status, iters = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        bad = self.L.ora_tick_batch(self.h, ctypes.c_int64(B), _p(q), _p(v), _p(desired), _p(cweight),
                                    _p(cmaxf), _p(tw) if tw is not None else None,
                                    _p(surf) if surf is not None else None, int(warm), _p(tau), _p(vd),
                                    _p(cw) if L.n_contacts else None,
                                    status.ctypes.data_as(c_int32_p), iters.ctypes.data_as(c_int32_p))
        return dict(tau=tau, vdot=vd, cwrench=cw[:, :6 * L.n_contacts].reshape(B, L.n_contacts, 6),
                    status=status, iters=iters, n_bad=bad)


def qp_solve(P, q, A, l, u, settings=None):
    """Generic dense QP through the OSQP restatement (+exact polish by default)."""
    L = lib()
    n, m = len(q), len(l)
    P, q, A, l, u = _arr(P, n * n), _arr(q), _arr(A, m * n), _arr(np.clip(l, -1e30, 1e30)), _arr(np.clip(u, -1e30, 1e30))
    st = settings or test_settings(2)
    x, y = np.zeros(n), np.zeros(max(m, 1))
    info = TickInfo()
    rc = L.ora_qp_solve(n, m, _p(P), _p(q), _p(A), _p(l), _p(u), ctypes.byref(st), _p(x), _p(y),
                        ctypes.byref(info))
    return dict(status=rc, x=x, y=y[:m], info=info)


def qp_time(P, q, A, l, u, settings, reps=5, warm=False):
    """Microseconds per solve of the OSQP restatement on k dense problems of one pattern: setup once, then per problem `reps` x
    (update of all coefficients + solve) -- what `Parametron.solve!` costs the reference's optimizer every tick (momentum.jl:58).
    warm: from the previous solution of the same problem; warm = 2: the k problems are consecutive ticks of one closed loop, problem
    b > 0 is solved ONCE from the solution of problem b - 1.  -> dict(us[k], iters[k], status[k])"""
    L = lib()
    P, A = np.ascontiguousarray(P, dtype=np.float64), np.ascontiguousarray(A, dtype=np.float64)
    k, n, m = P.shape[0], P.shape[1], A.shape[1]
    q = np.ascontiguousarray(q, dtype=np.float64); l = np.ascontiguousarray(np.clip(l, -1e30, 1e30)); u = np.ascontiguousarray(np.clip(u, -1e30, 1e30))
    us = np.zeros(k); iters = np.zeros(k, dtype=np.int32); status = np.zeros(k, dtype=np.int32)
    L.ora_qp_time.restype = ctypes.c_int
    rc = L.ora_qp_time(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(k), _p(P), _p(q), _p(A), _p(l), _p(u), ctypes.byref(settings), ctypes.c_int(reps),
                       ctypes.c_int(int(warm)), _p(us), iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return dict(us=us, iters=iters, status=status)
