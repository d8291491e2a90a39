"""L0 (qpc_qp_solve_batch, raw dense QPs in OSQP form -- the MOI-optimizer-level plug) against the
oracle's OSQP restatement + exact polish: random QPs, and the full reference-layout problems the
oracle builds for the Acrobot and the Atlas standing controller (143 x 178)."""
import numpy as np
import pytest

import emu
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import binding


def _random_qp(rng, n, m, n_eq):
    M = rng.normal(size=(n, n))
    P = M @ M.T * 0.3 + 0.05 * np.eye(n)
    q = rng.normal(size=n)
    A = rng.normal(size=(m, n))
    x0 = rng.normal(size=n)
    b = A @ x0
    l, u = b - rng.uniform(0.1, 1.0, size=m), b + rng.uniform(0.1, 1.0, size=m)
    l[:n_eq] = u[:n_eq] = b[:n_eq]
    l[n_eq::3] = -1e30  # one-sided rows
    return P, q, A, l, u


def _random_structured_qp(rng, n, m, n_eq, density=0.15):
    """what Parametron builds for this controller: diagonal P, sparse A -- the problems the workgroup form of L0 (csrc/qpc_qp_wg.h)
    takes; bounds chosen so that a good part of the inequality rows ends up active"""
    P = np.diag(rng.uniform(0.05, 2.0, size=n))
    q = rng.normal(size=n)
    A = rng.normal(size=(m, n)) * (rng.uniform(size=(m, n)) < density)
    A[np.arange(m), rng.integers(0, n, size=m)] += rng.uniform(0.5, 1.5, size=m)  # no empty row
    x0 = rng.normal(size=n)
    b = A @ x0
    l, u = b - rng.uniform(0.0, 0.6, size=m), b + rng.uniform(0.0, 0.6, size=m)
    l[:n_eq] = u[:n_eq] = b[:n_eq]
    l[n_eq::3] = -1e30
    return P, q, A, l, u


def _check_structured(backend_lib, device_id=0, expect_wg=None):
    rng = np.random.default_rng(11)
    sizes = ((6, 9, 2), (20, 30, 6), (40, 70, 12), (64, 100, 20), (143, 178, 60))
    n_wg = 0
    for n, m, ne in sizes:
        for _ in range(3):
            P, q, A, l, u = _random_structured_qp(rng, n, m, ne)
            ro = O.qp_solve(P, q, A, l, u)
            if ro["status"] != 0:
                continue
            r = binding.qp_solve_batch(P, q, A, l, u, device_id=device_id, lib_path=backend_lib)
            assert r["status"][0] == 0, (n, m, r["status"])
            np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
            np.testing.assert_allclose(r["y"][0], ro["y"], atol=1e-6)
            n_wg += 1
    assert n_wg >= 10
    if expect_wg is not None:
        assert expect_wg() >= n_wg - 4  # (a QP whose sweeps do not settle goes to the one-lane form: rare)


def _check(backend_lib, device_id=0):
    rng = np.random.default_rng(5)
    probs = [_random_qp(rng, n, m, ne) for n, m, ne in ((5, 7, 2), (12, 20, 4), (30, 45, 10), (8, 0, 0))]
    for P, q, A, l, u in probs:
        r = binding.qp_solve_batch(P, q, A, l, u, device_id=device_id, lib_path=backend_lib)
        assert r["status"][0] == 0
        if len(l):
            ro = O.qp_solve(P, q, A, l, u)
            assert ro["status"] == 0
            np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
            np.testing.assert_allclose(r["y"][0], ro["y"], atol=1e-6)
        else:
            np.testing.assert_allclose(r["x"][0], -np.linalg.solve(P, q), atol=1e-9)


def _check_reference_layout(backend_lib, device_id=0):
    def ofactory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech), N, fl)
    for case in (SC.acrobot_point_task(ofactory, B=2), SC.atlas_config(ofactory, 3, 3)):
        oc = case.ctrl
        Ps, qs, As, ls, us, xs = [], [], [], [], [], []
        for b in range(case.B):
            P, q, A, l, u = oc.build_qp(case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
            r = oc.tick(case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
            assert r["status"] == 0
            Ps.append(P); qs.append(q); As.append(A); ls.append(l); us.append(u); xs.append(r["x"])
        r = binding.qp_solve_batch(np.array(Ps), np.array(qs), np.array(As), np.array(ls), np.array(us),
                                   device_id=device_id, lib_path=backend_lib)
        assert np.all(r["status"] == 0), r["status"]
        nv = case.mech.nv
        for b in range(case.B):  # v̇ is unique (regularised); rho of the N = 4 basis is not
            np.testing.assert_allclose(r["x"][b][:nv], xs[b][:nv], atol=1e-8)


def test_l0_emulated_random():
    emu.build()
    _check(emu.EMU_LIB)


def _emu_wg_count():
    import ctypes
    L = ctypes.CDLL(emu.EMU_LIB)
    L.emu_l0_wg_count.restype = ctypes.c_long
    return L.emu_l0_wg_count()


def test_l0_emulated_reference_layout():
    emu.build()
    _emu_wg_count()
    _check_reference_layout(emu.EMU_LIB)
    assert _emu_wg_count() == 2 + 3  # every reference-layout problem (diagonal P, sparse A) is solved by the workgroup form


def test_l0_emulated_structured():
    emu.build()
    _emu_wg_count()
    _check_structured(emu.EMU_LIB, expect_wg=_emu_wg_count)


def test_l0_workgroup_form_equals_one_lane_form(monkeypatch):
    """the two device forms of L0 on the same problems: same statuses, x to 1e-9"""
    emu.build()
    rng = np.random.default_rng(3)
    probs = [_random_structured_qp(rng, n, m, ne) for n, m, ne in ((12, 20, 4), (40, 70, 12), (90, 120, 30))]
    a = [binding.qp_solve_batch(*p, lib_path=emu.EMU_LIB) for p in probs]
    monkeypatch.setenv("QPC_L0_ONE_LANE", "1")
    b = [binding.qp_solve_batch(*p, lib_path=emu.EMU_LIB) for p in probs]
    for ra, rb in zip(a, b):
        assert ra["status"][0] == rb["status"][0]
        if ra["status"][0] == 0:
            np.testing.assert_allclose(ra["x"][0], rb["x"][0], atol=1e-9)


def test_l0_infeasible_and_status():
    emu.build()
    P, q = np.eye(2), np.zeros(2)
    A = np.array([[1.0, 0.0], [1.0, 0.0]])
    r = binding.qp_solve_batch(P, q, A, np.array([1.0, -1e30]), np.array([1e30, 0.0]), lib_path=emu.EMU_LIB)  # x0 >= 1 and x0 <= 0
    assert r["status"][0] != 0


@pytest.mark.gpu
def test_l0_gpu_random():
    _check(None)


@pytest.mark.gpu
def test_l0_gpu_reference_layout():
    _check_reference_layout(None)


@pytest.mark.gpu
def test_l0_gpu_structured():
    _check_structured(None)


@pytest.mark.gpu
def test_l0_gpu_fixture_batch():
    """the committed 143 x 178 Atlas problems (tests/golden/l0/) as one batch of 64: every QP OPTIMAL, v̇ as stored"""
    import sys
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tools"))
    import bench_latency as BL
    P, q, A, l, u, x, nv = BL.load_l0_fixture()
    idx = np.arange(64) % P.shape[0]
    r = binding.qp_solve_batch(P[idx], q[idx], A[idx], l[idx], u[idx])
    assert (r["status"] == 0).all()
    np.testing.assert_allclose(r["x"][:, :nv], x[idx][:, :nv], atol=1e-8)


def test_one_lane_form_keeps_equality_rows_before_active_bounds():
    """A 231 x 246 block-arrow problem with blocks in P (no plan fits LDS, P is not diagonal: the one-lane form) whose sweeps put a bound into the
    working set that, with the equality rows before it, says what an equality row BEHIND it in the caller's order says.  The infinite-pivot
    factorization drops the later of two dependent rows -- it has to be the bound: in the caller's order the equality row went, the answer
    violated it by 0.17 and the QP was called INFEASIBLE (tests/offline/fuzz_l0_big.py seed 11, trial 29; the oracle and the dense
    workgroup form on the diagonal of P solve it)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "offline"))
    import fuzz_l0_big
    emu.build()
    rng = np.random.default_rng(11)
    for _ in range(30):
        P, q, A, l, u, n, m, ne = fuzz_l0_big.gen(rng)
    assert (n, m, ne) == (231, 246, 160) and np.any(P - np.diag(np.diag(P)) != 0.0)
    ro = O.qp_solve(P, q, A, l, u)
    assert ro["status"] == 0
    pr = binding.QpProblem(n, m, P != 0, A != 0, lib_path=emu.EMU_LIB)
    Pv, Av = pr.values(P, A)
    r = pr.solve(Pv[0], q, Av[0], l, u)
    assert pr.plan_info()["in_use"] == 0
    pr.close()
    assert r["status"][0] == 0, r["status"]
    Ax = A @ r["x"][0]
    assert max(np.maximum(l - Ax, 0).max(), np.maximum(Ax - u, 0).max()) < 1e-10
    np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-7)
