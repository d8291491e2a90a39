"""Feasibility probe for a full-fp32 tick: condition of the Schur complement S = E H^-1 E' of the reduced QP (with and
without Jacobi scaling) and the error of an fp32 range-space solve of the optimal-working-set EQP vs fp64."""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import qpc_pkg; qpc_pkg.load()
import proto_reduced as PR, scenarios as SC
from oracle import oracle as O
def factory(mech, N, fl): return PR.Recorder(O.OracleController(O.OracleModel(mech), N, fl), mech, N, fl)
case = SC.atlas_config(factory, 3, 24)
rec = case.ctrl
conds, conds_s, errs = [], [], []
for b in range(case.B):
    P = PR.build_reduced(rec, case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
    r = PR.solve_reduced(P)
    if r["status"] != 0: continue
    rho = r["rho"]; free = rho > 1e-9
    nf, ne = len(P.Hv), P.E_e.shape[1]
    N = len(P.Hrho[0]); nrho = len(rho)
    Hr = np.zeros((nrho,nrho))
    for ci,H in enumerate(P.Hrho): Hr[ci*N:(ci+1)*N, ci*N:(ci+1)*N] = H
    We = np.zeros((ne,ne)); o=0
    for W in P.slack_W: k=len(W); We[o:o+k,o:o+k]=W; o+=k
    m = len(P.d)
    Erho = np.zeros((m, nrho)); Erho[-6:, :] = -P.G
    idx = np.where(free)[0]
    eps = 1e-9*max(P.Hv.max(), 2.0)
    def solve(dt, eps):
        Hd = np.concatenate([P.Hv, np.diag(We)]).astype(dt)  # slack W diagonal here
        E = np.hstack([P.E_v, P.E_e, Erho[:, idx]]).astype(dt)
        Hfull = np.zeros((nf+ne+len(idx),)*2, dtype=dt)
        Hfull[:nf,:nf] = np.diag(P.Hv); Hfull[nf:nf+ne,nf:nf+ne]=We; Hfull[nf+ne:,nf+ne:]=Hr[np.ix_(idx,idx)]
        Hfull = Hfull.astype(dt) + dt(eps)*np.eye(len(Hfull),dtype=dt)
        g = np.concatenate([P.gv, np.zeros(ne+len(idx))]).astype(dt)
        Hi = np.linalg.inv(Hfull.astype(np.float64)).astype(dt) if dt==np.float64 else np.linalg.inv(Hfull)
        S = (E @ Hi @ E.T).astype(dt)
        rhs = (E @ (Hi @ (-g)) - P.d.astype(dt)).astype(dt)
        L = np.linalg.cholesky(S + dt(1e-30)*np.eye(m,dtype=dt))
        lam = np.linalg.solve(L.T, np.linalg.solve(L, rhs)).astype(dt)
        x = (Hi @ (-g - E.T @ lam)).astype(dt)
        return x, S
    x64, S = solve(np.float64, eps)
    d = np.sqrt(np.diag(S)); Ss = S/np.outer(d,d)
    conds.append(np.linalg.cond(S)); conds_s.append(np.linalg.cond(Ss))
    try:
        x32, _ = solve(np.float32, 1e-5*max(P.Hv.max(),2.0))
        errs.append(np.abs(x32[:nf]-x64[:nf]).max()/max(np.abs(x64[:nf]).max(),1e-9))
    except np.linalg.LinAlgError as e:
        errs.append(np.inf)
print("cond(S)  median %.2e max %.2e" % (np.median(conds), np.max(conds)))
print("cond(S) Jacobi-scaled median %.2e max %.2e" % (np.median(conds_s), np.max(conds_s)))
print("fp32 EQP relative error of vdot_free: median %.2e max %.2e" % (np.median(errs), np.max(errs)))
