"""numpy prototype of the REDUCED-FORM algorithm the HIP kernels implement (DESIGN.md "Solver"):
range-space (Schur complement) KKT solves + Goldfarb-Idnani dual active set on the contact
multipliers rho + proximal-method-of-multipliers refinement.  Test helper only: it documents the
kernel's math in executable form and cross-checks it against the oracle (which solves the
143-variable reference-layout problem with a completely different algorithm) on CPU, so that
algorithmic bugs are found without a GPU.  Kinematic inputs come from the oracle."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd.workloads import (HARD, JOINT, MATRIX, MATRIX_PER_QP, SCALAR, SCALAR_PER_QP)


@dataclass
class Settings:
    detect_cycles: bool = False
    multi_drop: bool = False  # experiment: phase B drops every wrong-signed multiplier at once
    max_pdas: int = 4
    max_active_set_iters: int = 200
    max_refine_iters: int = 6
    eps_primal_reg: float = 1e-9
    eps_dual_reg: float = 1e-11  # relative pivot threshold of the infinite-pivot Cholesky
    tol_residual: float = 1e-10
    tol_bound: float = 1e-10


class Recorder:
    """Controller-like object that records the construction calls and forwards them to `inner`."""

    def __init__(self, inner, mech, n_basis, floating):
        self.inner, self.mech, self.N, self.floating = inner, mech, n_basis, floating
        self.tasks, self.contacts, self.reg = [], [], np.zeros(mech.nv)

    def add_task(self, kind, base=-1, target=-1, frame=-2, point=None, joint=-1, wkind=0, weight=None):
        tid = self.inner.add_task(kind, base, target, frame, point, joint, wkind, weight)
        self.tasks.append(dict(kind=kind, joint=joint, wkind=wkind,
                               weight=None if weight is None else np.asarray(weight, dtype=float)))
        return tid

    def regularize(self, joint, w):
        self.inner.regularize(joint, w)
        for k in self.mech.velocity_range(joint):
            self.reg[k] += w

    def add_contact(self, body, pos, normal, mu, per_qp_surface=0):
        cid = self.inner.add_contact(body, pos, normal, mu, per_qp_surface)
        self.contacts.append(dict(body=body, pos=np.asarray(pos, float), normal=np.asarray(normal, float),
                                  mu=mu, per_qp=per_qp_surface))
        return cid

    def finalize(self):
        self.layout = self.inner.finalize()
        return self.layout


@dataclass
class Reduced:
    free: np.ndarray
    pinned: np.ndarray
    vd_pinned: np.ndarray
    Hv: np.ndarray
    gv: np.ndarray
    slack_W: list
    E_v: np.ndarray
    E_e: np.ndarray
    G: np.ndarray  # 6 x nrho (wrench rows are the LAST 6 rows of E when present)
    d: np.ndarray
    Hrho: list  # per contact N x N
    rho_max: np.ndarray
    n_wrench: int
    info: dict = field(default_factory=dict)


def build_reduced(rec: Recorder, q, v, desired, cweight, cmaxf, tw=None, surf=None):
    """Assemble the reduced QP exactly as the kernel does (pins eliminated, slacks kept, f/wrench
    variables of the reference layout eliminated)."""
    oc, mech = rec.inner, rec.mech
    om = oc.model
    om.set_state(q, v)
    nv, N = mech.nv, rec.N
    desired = np.asarray(desired, dtype=float).reshape(-1)
    # --- pins: hard JointAccelerationTask on a not-yet-pinned velocity index
    pinned_val = {}
    off = 0
    task_rows = []
    tw_off = 0
    Hv_full = 2.0 * rec.reg.copy()
    gv_full = np.zeros(nv)
    for ti, t in enumerate(rec.tasks):
        dim = oc.task_dims[ti]
        des = desired[off:off + dim]
        off += dim
        W = None
        if t["wkind"] in (SCALAR, MATRIX):
            W = t["weight"]
        elif t["wkind"] == SCALAR_PER_QP:
            W = np.asarray(tw[tw_off:tw_off + 1]); tw_off += 1
        elif t["wkind"] == MATRIX_PER_QP:
            W = np.asarray(tw[tw_off:tw_off + dim * dim]); tw_off += dim * dim
        if t["kind"] == JOINT:
            vr = list(mech.velocity_range(t["joint"]))
            if t["wkind"] == HARD and all(k not in pinned_val for k in vr):
                for k, dk in zip(vr, des):
                    pinned_val[k] = dk
                continue
            if t["wkind"] in (SCALAR, SCALAR_PER_QP):  # fold: w (v̇ - d)^2
                for k, dk in zip(vr, des):
                    Hv_full[k] += 2.0 * W[0]
                    gv_full[k] += -2.0 * W[0] * dk
                continue
        J, b = oc.task_eval(ti)
        task_rows.append((J, des - b, t["wkind"], W, dim))
    pinned = np.array(sorted(pinned_val), dtype=int)
    free = np.array([k for k in range(nv) if k not in pinned_val], dtype=int)
    vdp = np.array([pinned_val[k] for k in pinned])
    rows_v, rows_e, rhs, slack_W = [], [], [], []
    ne = sum(r[4] for r in task_rows if r[2] != HARD)
    eo = 0
    for J, r, wkind, W, dim in task_rows:
        rr = r - (J[:, pinned] @ vdp if len(pinned) else 0.0)
        Ee = np.zeros((dim, ne))
        if wkind != HARD:
            Ee[:, eo:eo + dim] = -np.eye(dim)
            eo += dim
            if wkind in (SCALAR, SCALAR_PER_QP):
                slack_W.append(2.0 * W[0] * np.eye(dim))
            else:
                Wm = np.asarray(W).reshape(dim, dim)
                slack_W.append(Wm + Wm.T)
        rows_v.append(J[:, free]); rows_e.append(Ee); rhs.append(rr)
    nc = len(rec.contacts)
    nrho = nc * N
    G = np.zeros((6, nrho))
    Hrho, rho_max = [], np.zeros(nrho)
    for ci, c in enumerate(rec.contacts):
        normal, mu = c["normal"], c["mu"]
        if c["per_qp"]:
            k = sum(1 for cc in rec.contacts[:ci] if cc["per_qp"])
            normal, mu = surf[4 * k:4 * k + 3], surf[4 * k + 3]
        B, maxrho, R, p = oc.contact_eval(ci, normal, mu, cmaxf[ci])
        RB = R @ B
        G[3:, ci * N:(ci + 1) * N] = RB
        G[:3, ci * N:(ci + 1) * N] = np.cross(p[None, :], RB.T).T
        Hrho.append(2.0 * cweight[ci] * B.T @ B)
        rho_max[ci * N:(ci + 1) * N] = maxrho
    n_wrench = 0
    if rec.floating >= 0:
        n_wrench = 6
        A, Adv = om.momentum_matrix()
        fg = mech.total_mass() * np.array(mech.gravity)
        Wg = np.concatenate([np.cross(om.center_of_mass(), fg), fg])
        rr = Wg - Adv - (A[:, pinned] @ vdp if len(pinned) else 0.0)
        rows_v.append(A[:, free]); rows_e.append(np.zeros((6, ne))); rhs.append(rr)
    m = sum(len(r) for r in rhs)
    E_v = np.vstack(rows_v) if rows_v else np.zeros((0, len(free)))
    E_e = np.vstack(rows_e) if rows_e else np.zeros((0, ne))
    d = np.concatenate(rhs) if rhs else np.zeros(0)
    assert E_v.shape[0] == m
    return Reduced(free, pinned, vdp, Hv_full[free], gv_full[free], slack_W, E_v, E_e, G, d, Hrho,
                   rho_max, n_wrench)


def chol_inf(S, tau):
    """Cholesky with the 'infinite pivot' rule for (numerically) dependent rows: a pivot below
    tau * (original diagonal) marks the constraint row as redundant -- its multiplier becomes 0 and
    it no longer influences the other rows.  Returns (L, n_dropped)."""
    n = len(S)
    L = np.zeros_like(S)
    dropped = 0
    for k in range(n):
        d = S[k, k] - L[k, :k] @ L[k, :k]
        if not d > tau * max(S[k, k], 1e-300):
            L[k, k] = 1e150
            dropped += 1
            continue  # column k of L stays 0
        L[k, k] = np.sqrt(d)
        L[k + 1:, k] = (S[k + 1:, k] - L[k + 1:, :k] @ L[k, :k]) / L[k, k]
    return L, dropped


def solve_reduced(P: Reduced, st: Settings = Settings(), stats=None, guess=None):
    """x = [v̇_free; e; rho].  Returns dict(vd_free, e, rho, status, iters, refine)."""
    nf, ne, nrho = len(P.Hv), P.E_e.shape[1], len(P.rho_max)
    m, nw = len(P.d), P.n_wrench
    N = nrho // max(len(P.Hrho), 1) if nrho else 0
    use_rho = nw > 0 and nrho > 0
    hdiag = [np.abs(P.Hv).max() if nf else 0.0] + [np.abs(np.diag(W)).max() for W in P.slack_W] + \
            ([max(np.abs(np.diag(H)).max() for H in P.Hrho)] if use_rho else [])
    hscale = max(hdiag) if max(hdiag, default=0.0) > 0 else 1.0
    eps = st.eps_primal_reg * hscale
    Hv_inv = 1.0 / (P.Hv + eps)
    He_inv = np.zeros((ne, ne)); He = np.zeros((ne, ne))
    o = 0
    for W in P.slack_W:
        k = len(W)
        He[o:o + k, o:o + k] = W
        He_inv[o:o + k, o:o + k] = np.linalg.inv(W + eps * np.eye(k))
        o += k
    S0 = (P.E_v * Hv_inv[None, :]) @ P.E_v.T + P.E_e @ He_inv @ P.E_e.T
    lo = np.zeros(nrho)
    hi = P.rho_max.copy()
    enabled = (hi > 0) & use_rho
    ws = np.where(enabled, 0, -1).astype(int)  # 0 free, -1 at lower, +1 at upper
    Hrho_full = np.zeros((nrho, nrho))
    for ci, H in enumerate(P.Hrho):
        Hrho_full[ci * N:(ci + 1) * N, ci * N:(ci + 1) * N] = H
    nsolves = [0]

    class Fact:
        pass

    def factor():
        F = Fact()
        F.fr = np.where(ws == 0)[0]
        S = S0.copy()
        if len(F.fr):
            F.Kr = np.linalg.inv(Hrho_full[np.ix_(F.fr, F.fr)] + eps * np.eye(len(F.fr)))
            Gf = P.G[:, F.fr]
            S[m - nw:, m - nw:] += Gf @ F.Kr @ Gf.T
        else:
            F.Kr = np.zeros((0, 0))
        F.L, F.dropped = chol_inf(S, st.eps_dual_reg)
        if stats is not None:
            stats.setdefault("dropped", []).append(F.dropped)
        return F

    def kkt_solve(F, r_v, r_e, r_r, r_c):
        """[H+eps, E'; E, 0] [x; lam] = [r1; r_c] on the free set (fixed rho: x = 0)."""
        nsolves[0] += 1
        fr = F.fr
        t = P.E_v @ (Hv_inv * r_v) + P.E_e @ (He_inv @ r_e) - r_c
        if len(fr):
            t[m - nw:] += -P.G[:, fr] @ (F.Kr @ r_r[fr])
        lam = np.linalg.solve(F.L.T, np.linalg.solve(F.L, t)) if m else np.zeros(0)
        xv = Hv_inv * (r_v - P.E_v.T @ lam)
        xe = He_inv @ (r_e - P.E_e.T @ lam)
        xr = np.zeros(nrho)
        if len(fr):
            xr[fr] = F.Kr @ (r_r[fr] + P.G[:, fr].T @ lam[m - nw:])
        return xv, xe, xr, lam

    def residuals(xv, xe, xr, lam):
        """true (unregularised) KKT residuals of the equality-constrained problem on the working set"""
        r_v = -(P.Hv * xv + P.gv + P.E_v.T @ lam)
        r_e = -(He @ xe + P.E_e.T @ lam)
        r_c = P.d - (P.E_v @ xv + P.E_e @ xe)
        gl = np.zeros(nrho)
        if use_rho:
            r_c = r_c.copy()
            r_c[m - nw:] += P.G @ xr
            gl = Hrho_full @ xr - P.G.T @ lam[m - nw:]
        r_r = np.where(ws == 0, -gl, 0.0)
        return r_v, r_e, r_r, r_c, gl

    def eqp(F, refine_steps):
        """solve the EQP of the current working set from scratch, then iterative refinement"""
        xr = np.where(ws == 1, hi, 0.0)
        xv, xe, lam = np.zeros(nf), np.zeros(ne), np.zeros(m)
        nref = 0
        for it in range(refine_steps + 1):
            r_v, r_e, r_r, r_c, _ = residuals(xv, xe, xr, lam)
            dv, de, dr, dl = kkt_solve(F, r_v, r_e, r_r, r_c)
            xv, xe, xr, lam = xv + dv, xe + de, xr + dr, lam + dl
            nref = it
            big = max([np.abs(a).max() for a in (xv, xe, xr) if len(a)] + [0.0])
            dbig = max([np.abs(a).max() for a in (dv, de, dr) if len(a)] + [0.0])
            if it > 0 and dbig <= 1e-13 * (1.0 + big):
                break
        return xv, xe, xr, lam, nref

    def multipliers(xr, lam):
        gl = Hrho_full @ xr - (P.G.T @ lam[m - nw:] if use_rho else 0.0)
        return np.where(ws == -1, gl, np.where(ws == 1, -gl, 0.0)), gl

    status, iters, refine = 0, 0, 0
    F = factor()
    xv, xe, xr, lam, _ = eqp(F, 0)
    if use_rho and guess is not None:  # experiment hook: initial working set from the unconstrained solve
        new = guess(P, xr, ws, enabled, N)
        if not np.array_equal(new, ws):
            ws = new
            iters += 1
            F = factor()
            xv, xe, xr, lam, _ = eqp(F, 0)
    if use_rho:
        # ---- phase A: primal-dual active set passes (multi-change), cheap and usually enough
        seen = [tuple(ws)]
        for _pd in range(st.max_pdas):  # more sweeps rarely converge; the finite phase B is cheaper from there
            u, gl = multipliers(xr, lam)
            gscale = 1.0 + np.abs(gl).max()
            tolb = st.tol_bound * (1.0 + np.abs(xr).max())
            new = ws.copy()
            new[(ws != 0) & enabled & (u < -st.tol_bound * gscale)] = 0
            new[(ws == 0) & (xr < lo - tolb)] = -1
            new[(ws == 0) & (xr > hi + tolb)] = 1
            if np.array_equal(new, ws):
                break
            if st.detect_cycles and tuple(new) in seen:
                break  # primal-dual sweeps cycle: hand over to the finite phase B now
            seen.append(tuple(new))
            ws = new
            iters += 1
            F = factor()
            xv, xe, xr, lam, _ = eqp(F, 0)
        # ---- phase B: safe single-change loop (drops until dual feasible, then GI additions)
        while True:
            if iters > st.max_active_set_iters:
                status = 4
                break
            if not (np.all(np.isfinite(xv)) and np.all(np.isfinite(lam)) and np.all(np.isfinite(xr))):
                status = 5
                break
            u, gl = multipliers(xr, lam)
            gscale = 1.0 + np.abs(gl).max()
            tolb = st.tol_bound * (1.0 + np.abs(xr).max())
            act = np.where((ws != 0) & enabled)[0]
            bad = [j for j in act if u[j] < -st.tol_bound * gscale]
            if bad:
                if stats is not None:
                    stats["drops"] = stats.get("drops", 0) + 1
                if st.multi_drop:
                    for j in bad:
                        ws[j] = 0
                else:
                    j = min(bad, key=lambda j: u[j])
                    ws[j] = 0
                F = factor(); xv, xe, xr, lam, _ = eqp(F, 0); iters += 1
                continue
            viol = np.where(ws == 0, np.maximum(lo - xr, xr - hi), 0.0)
            p = int(np.argmax(viol))
            if viol[p] <= tolb:
                break
            side = -1 if (lo[p] - xr[p]) > (xr[p] - hi[p]) else 1
            nsign = 1.0 if side == -1 else -1.0
            fail = False
            while True:
                if stats is not None:
                    stats["adds"] = stats.get("adds", 0) + 1
                iters += 1
                if iters > st.max_active_set_iters:
                    fail = True
                    break
                r_r = np.zeros(nrho); r_r[p] = nsign
                zv, ze, zr, zl = kkt_solve(F, np.zeros(nf), np.zeros(ne), r_r, np.zeros(m))
                hz = Hrho_full @ zr + eps * zr - P.G.T @ zl[m - nw:]
                sj = np.where(ws == -1, 1.0, -1.0)
                r = np.where((ws != 0) & enabled, -sj * hz, 0.0)
                sp = nsign * xr[p] - (lo[p] if side == -1 else -hi[p])
                zn = nsign * zr[p]
                t2 = -sp / zn if zn > 1e-300 else np.inf
                cand = [(u[j] / r[j], j) for j in act if ws[j] != 0 and r[j] > 1e-300]
                t1, jdrop = min(cand) if cand else (np.inf, -1)
                t = min(t1, t2)
                if not np.isfinite(t):
                    status = 2
                    fail = True
                    break
                xv, xe, xr, lam = xv + t * zv, xe + t * ze, xr + t * zr, lam + t * zl
                u = u - t * r
                if t2 <= t1:
                    ws[p] = side
                    F = factor()
                    xv, xe, xr, lam, _ = eqp(F, 0)
                    break
                ws[jdrop] = 0
                u[jdrop] = 0.0
                act = np.where((ws != 0) & enabled)[0]
                F = factor()
            if fail:
                if status == 0:
                    status = 4
                break
    # ---- final polish on the converged working set + acceptance test
    if status == 0:
        xv, xe, xr, lam, refine = eqp(F, st.max_refine_iters)
        r_v, r_e, r_r, r_c, gl = residuals(xv, xe, xr, lam)
        sc_p = 1.0 + (np.abs(P.d).max() if m else 0.0)
        r_prim = (np.abs(r_c).max() if m else 0.0) / sc_p
        sc_s = 1.0 + max(np.abs(P.gv).max() if nf else 0.0, np.abs(P.E_v.T @ lam).max() if nf and m else 0.0)
        r_stat = max([np.abs(a).max() for a in (r_v, r_e, r_r) if len(a)] + [0.0]) / sc_s
        if not np.isfinite(r_prim + r_stat):
            status = 5
        elif r_prim > 1e-7:
            status = 2  # equality rows inconsistent on the feasible set: INFEASIBLE
        elif r_stat > 1e-7:
            status = 1
        if stats is not None:
            stats.setdefault("r_prim", []).append(r_prim)
            stats.setdefault("r_stat", []).append(r_stat)
    if stats is not None:
        stats.setdefault("solves", []).append(nsolves[0])
    return dict(vd_free=xv, e=xe, rho=xr, lam=lam, status=status, iters=iters, refine=refine, ws=ws)


def tick(rec: Recorder, q, v, desired=(), cweight=(), cmaxf=(), tw=None, surf=None, st=Settings()):
    """Full prototype tick: reduce, solve, contact wrenches, inverse dynamics, zero floating tau."""
    mech = rec.mech
    P = build_reduced(rec, q, v, desired, cweight, cmaxf, tw, surf)
    sol = solve_reduced(P, st)
    vd = np.zeros(mech.nv)
    vd[P.free] = sol["vd_free"]
    vd[P.pinned] = P.vd_pinned
    N = rec.N
    nc = len(rec.contacts)
    cwrench = np.zeros((nc, 6))
    wext = np.zeros((mech.nbodies, 6))
    rho = sol["rho"] if P.n_wrench else np.zeros(nc * N)
    for ci, c in enumerate(rec.contacts):
        cwrench[ci] = P.G[:, ci * N:(ci + 1) * N] @ rho[ci * N:(ci + 1) * N]
        if c["body"] >= 0:
            wext[c["body"]] += cwrench[ci]
    om = rec.inner.model
    om.set_state(q, v)
    tau = om.inverse_dynamics(vd, wext)
    if rec.floating >= 0:
        for k in mech.velocity_range(rec.floating):
            tau[k] = 0.0
    return dict(tau=tau, vdot=vd, cwrench=cwrench, status=sol["status"], iters=sol["iters"],
                refine=sol["refine"], rho=rho, reduced=P)
