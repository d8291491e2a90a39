"""Solver-independent KKT certificate of a tick result in the REFERENCE LAYOUT (SURVEY.md 4 item 6,
8c "what replaces golden vectors (2)").  Given v̇ and the contact wrenches returned by any backend,
the full reference-layout point x = [v̇; per contact (rho, f_local, w_ang, w_lin); slacks] is
rebuilt, primal feasibility is checked against the oracle-built (P, q, A, l, u), and a dual
certificate y with the right signs is searched by bounded least squares (scipy).  The objective is
strictly convex in (v̇, f) when regularised, so a certified point IS the unique optimum -- no
reference solver run is needed."""
import numpy as np
from scipy.optimize import lsq_linear


def lift(oc, q, v, desired, cweight, cmaxf, vdot, cwrench, tw=None, surf=None):
    L = oc.layout
    om = oc.model
    om.set_state(q, v)
    x = np.zeros(L.ref_n_var)
    x[:L.nv] = vdot
    off = L.nv
    desired = np.asarray(desired, float).reshape(-1)
    des_off = np.concatenate([[0], np.cumsum(oc.task_dims)])
    nsurf = 0
    for kind, idx, n in oc.var_blocks:
        if kind == "contact":
            normal, mu, per_qp = oc.contact_specs[idx]
            if per_qp:
                normal, mu = surf[4 * nsurf:4 * nsurf + 3], surf[4 * nsurf + 3]
                nsurf += 1
            B, maxrho, R, p = oc.contact_eval(idx, normal, mu, cmaxf[idx])
            wl = cwrench[idx][3:]
            f = R.T @ wl
            N = oc.N
            if maxrho > 0:
                # any rho in [0, maxrho] with B rho = f; prefer an interior one (few active bounds)
                res = lsq_linear(np.vstack([B, 1e-6 * np.eye(N)]), np.concatenate([f, 1e-6 * np.full(N, f[2] / max(B[2].sum(), 1e-300))]),
                                 bounds=(0.0, maxrho), lsq_solver="exact")
                rho = res.x
            else:
                rho = np.zeros(N)
            x[off:off + N] = rho
            x[off + N:off + N + 3] = B @ rho
            x[off + N + 3:off + N + 6] = cwrench[idx][:3]
            x[off + N + 6:off + N + 9] = wl
        else:
            J, bias = oc.task_eval(idx)
            x[off:off + n] = J @ vdot + bias - desired[des_off[idx]:des_off[idx] + n]
        off += n
    return x


def certify(oc, q, v, desired, cweight, cmaxf, vdot, cwrench, tw=None, surf=None, tol=1e-7):
    """Returns dict(primal, stationarity (relative), ok)."""
    P, qv, A, l, u = oc.build_qp(q, v, desired, cweight, cmaxf, tw, surf)
    x = lift(oc, q, v, desired, cweight, cmaxf, vdot, cwrench, tw, surf)
    Ax = A @ x
    scale = 1.0 + np.abs(Ax).max()
    primal = max(np.max(l - Ax), np.max(Ax - u), 0.0) / scale
    eq = (u - l) < 1e-12 * scale
    at_l = (~eq) & (Ax - l < tol * scale)
    at_u = (~eq) & (u - Ax < tol * scale)
    lo = np.where(eq | at_l, -np.inf, 0.0)
    hi = np.where(eq | at_u, np.inf, 0.0)
    act = eq | at_l | at_u
    g = P @ x + qv
    res = lsq_linear(A.T[:, act], -g, bounds=(lo[act], hi[act]), lsq_solver="exact")
    y = np.zeros(len(l))
    y[act] = res.x
    stat = np.abs(A.T @ y + g).max() / (1.0 + max(np.abs(g).max(), np.abs(y).max()))
    return dict(primal=primal, stationarity=stat, ok=bool(primal <= tol and stat <= tol), x=x, y=y)
