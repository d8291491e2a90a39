#!/usr/bin/env python3
"""Generates tests/golden/l0/atlas_143x178.npz: the 143 x 178 reference-layout QPs (what
Parametron hands to its optimizer for the Atlas standing controller, `momentum.jl:27,58`) of a few seeded config-3
states, as the CPU ORACLE builds them (oracle/ora_ctrl.c), with the oracle's exact solution -- inputs and expected
outputs of the L0 entry `qpc_qp_solve_batch`, so that tools/bench_latency.py --l0 and the GPU tests can exercise the
b-1 plug without importing the oracle.  Regression data of this repository, not reference output (the reference is
Julia).  Stored sparse (the dense matrices are 98 % zeros).

    python tests/golden/l0/make_l0_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import scenarios as SC  # noqa: E402
from oracle import oracle as O  # noqa: E402


def momentum_weight_matrix():
    """the 6 x 6 W of the matrix-weighted fixture: symmetric positive definite, NOT diagonal (seeded)"""
    rng = np.random.default_rng(61)
    Q = rng.normal(size=(6, 6))
    return Q @ Q.T / 6.0 + np.diag([0.5, 0.5, 0.5, 1.0, 1.0, 1.0])


def main(B=8):
    O.build()
    factory = lambda mech, N, fl: O.OracleController(O.OracleModel(mech), N, fl)  # noqa: E731
    write(SC.atlas_config(factory, 3, B), "atlas_143x178.npz")
    # the same controller with `addtask!(controller, MomentumRateTask, W::AbstractMatrix)` (momentum.jl:113-117; the reference's test
    # modes OBJECTIVE_TERM_MATRIX_WEIGHT*, test/controller.jl:232-285) in place of the scalar-weighted linear momentum task:
    # six slack variables, a 6 x 6 block in P
    from qpcontrol_jl_amd import workloads as W
    ctrl, wl = W.atlas_workload(factory, B, 3, momentum_weight_matrix=momentum_weight_matrix())
    write(SC.Case("atlas_matrix_weight", wl.mech, ctrl, wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce,
                  floating=wl.floating_body), "atlas_matrix_weight_146x181.npz")


def closed_loop(K=96, fname="atlas_closed_loop_143x178.npz"):
    """The regime the reference runs its optimizer in (notebooks/Standing controller.ipynb: ONE Atlas, a tick every millisecond, the
    optimizer warm-started from the previous tick): K CONSECUTIVE reference-layout problems of a closed loop -- a perturbed standing
    state, the StandingController laws re-evaluated every tick, the state stepped with the QP's own v̇ (explicit Euler at 1 kHz, as
    tests/test_moi_optimizer.py does), a push of 6 cm/s on the pelvis at tick K/3 and a contact whose force limit drops to zero at tick
    2K/3 (so that the working set changes along the way).  Same file layout as the other fixtures; the problems are in tick order."""
    from qpcontrol_jl_amd import workloads as W
    factory = lambda mech, N, fl: O.OracleController(O.OracleModel(mech), N, fl)  # noqa: E731
    oc, wl = W.atlas_workload(factory, 4, 3)
    om = oc.model
    nv = wl.mech.nv
    q, v = wl.q[0].copy(), wl.v[0].copy()
    cw, cm = wl.contact_weight[0].copy(), wl.contact_maxnormalforce[0].copy()
    Ps, qs, As, ls, us, xs = [], [], [], [], [], []
    for k in range(K):
        if k == K // 3:
            v[3:6] += np.array([0.06, -0.03, 0.0])
        if k == (2 * K) // 3:
            cm[1] = 0.0
        des = W.standing_desired(wl.setup, q[None], v[None])[0]
        P, qv, A, l, u = oc.build_qp(q, v, des, cw, cm)
        r = oc.tick(q, v, des, cw, cm)
        assert r["status"] == 0, (k, r["status"])
        Ps.append(P); qs.append(qv); As.append(A); ls.append(l); us.append(u); xs.append(r["x"])
        v = v + 1e-3 * r["x"][:nv]
        q = om.integrate(q, v, 1e-3)
    save(np.array(Ps), np.array(As), qs, ls, us, xs, nv, fname)


def save(P, A, qs, ls, us, xs, nv, fname):
    pat_P, pat_A = np.any(P != 0.0, axis=0), np.any(A != 0.0, axis=0)
    np.savez_compressed(os.path.join(HERE, fname), n=np.int32(P.shape[1]), m=np.int32(A.shape[1]),
                        P_idx=np.argwhere(pat_P).astype(np.int16), P_val=P[:, pat_P], A_idx=np.argwhere(pat_A).astype(np.int16),
                        A_val=A[:, pat_A], q=np.array(qs), l=np.array(ls), u=np.array(us), x=np.array(xs), nv=np.int32(nv))
    print(f"{fname}: {P.shape[0]} QPs, n = {P.shape[1]}, m = {A.shape[1]}, nnz(P) = {int(pat_P.sum())}, nnz(A) = {int(pat_A.sum())}")


def write(case, fname):
    oc = case.ctrl
    Ps, qs, As, ls, us, xs = [], [], [], [], [], []
    for b in range(case.B):
        P, q, A, l, u = oc.build_qp(case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
        r = oc.tick(case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
        if r["status"] != 0:
            continue
        Ps.append(P); qs.append(q); As.append(A); ls.append(l); us.append(u); xs.append(r["x"])
    save(np.array(Ps), np.array(As), qs, ls, us, xs, case.mech.nv, fname)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "closed_loop":  # (by itself: the eight-state fixtures stay byte for byte what they are)
        O.build()
        closed_loop()
    else:
        main()
