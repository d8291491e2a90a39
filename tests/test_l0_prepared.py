"""L0 with the optimizer's life cycle (qpc_qp_setup / qpc_qp_solve / qpc_qp_free: the pattern once, values every call, warm start) --
what `QPCOptimizer <: MOI.AbstractOptimizer` binds (momentum.jl:15-16,27,58; test/runtests.jl:35-43).  Checked against the dense
entry (qpc_qp_solve_batch), the oracle's OSQP restatement and the committed 143 x 178 Atlas problems; CPU emulation and GPU."""
import os
import sys

import numpy as np
import pytest

import emu
from oracle import oracle as O
from qpcontrol_jl_amd import binding
from test_l0_dense_qp import _emu_wg_count, _random_qp, _random_structured_qp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def _fixture():
    import bench_latency as BL
    return BL.load_l0_fixture()


def _check_fixture(lib):
    """the committed Atlas problems: prepared == dense entry, v̇ as stored; then the same batch warm-started"""
    P, q, A, l, u, x, nv = _fixture()
    n, m = P.shape[1], A.shape[1]
    pr = binding.QpProblem(n, m, (P != 0).any(0), (A != 0).any(0), lib_path=lib)
    assert len(pr.P_rows) == 63 and len(pr.A_rows) == 1004  # what Parametron builds for this controller
    Pv, Av = pr.values(P, A)
    rd = binding.qp_solve_batch(P, q, A, l, u, lib_path=lib)
    rp = pr.solve(Pv, q, Av, l, u)
    assert (rp["status"] == 0).all() and (rd["status"] == 0).all()
    np.testing.assert_array_equal(rp["iters"], rd["iters"])
    np.testing.assert_allclose(rp["x"], rd["x"], atol=1e-9)
    np.testing.assert_allclose(rp["y"], rd["y"], atol=1e-6)
    np.testing.assert_allclose(rp["x"][:, :nv], x[:, :nv], atol=1e-8)
    rw = pr.solve(Pv, q, Av, l, u, warm_start=True)  # from its own solution: nothing (or one weakly active row) to change
    assert (rw["status"] == 0).all() and (rw["iters"] <= 1).all(), rw["iters"]
    np.testing.assert_allclose(rw["x"], rp["x"], atol=1e-9)
    idx = np.roll(np.arange(P.shape[0]), 1)  # from ANOTHER state's solution: still the right answer
    rx = pr.solve(Pv[idx], q[idx], Av[idx], l[idx], u[idx], warm_start=True)
    assert (rx["status"] == 0).all()
    np.testing.assert_allclose(rx["x"][:, :nv], x[idx][:, :nv], atol=1e-8)
    pr.close()
    return rp, rw


def _check_random(lib):
    rng = np.random.default_rng(23)
    # structured (diagonal P): the workgroup solver; explicit zeros in the pattern, values that vanish
    for n, m, ne in ((6, 9, 2), (20, 30, 6), (64, 100, 20)):
        P, q, A, l, u = _random_structured_qp(rng, n, m, ne)
        pat_A = (A != 0) | (rng.uniform(size=A.shape) < 0.05)  # structural zeros
        pr = binding.QpProblem(n, m, P != 0, pat_A, lib_path=lib)
        Pv, Av = pr.values(P, A)
        ro = O.qp_solve(P, q, A, l, u)
        r = pr.solve(Pv, q, Av, l, u)
        if ro["status"] == 0:
            assert r["status"][0] == 0
            np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
            rd = binding.qp_solve_batch(P, q, A, l, u, lib_path=lib)  # (multipliers: against the dense entry -- a degenerate vertex has no unique y)
            np.testing.assert_allclose(r["y"][0], rd["y"][0], atol=1e-6)
        pr.close()
    # a dense P: the pattern has off-diagonal entries -> the dense path behind the same calls
    P, q, A, l, u = _random_qp(rng, 12, 20, 4)
    pr = binding.QpProblem(12, 20, P != 0, A != 0, lib_path=lib)
    Pv, Av = pr.values(P, A)
    assert Pv.shape[1] == 12 * 13 // 2  # one triangle
    ro = O.qp_solve(P, q, A, l, u)
    r = pr.solve(Pv, q, Av, l, u)
    assert ro["status"] == 0 and r["status"][0] == 0
    np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
    # a batch of three with per-QP values
    r3 = pr.solve(np.stack([Pv[0], 2 * Pv[0], Pv[0]]), np.stack([q, q, -q]), np.stack([Av[0]] * 3), np.stack([l] * 3), np.stack([u] * 3))
    np.testing.assert_allclose(r3["x"][0], ro["x"], atol=1e-8)
    ro2 = O.qp_solve(2 * P, q, A, l, u)
    if ro2["status"] == 0:
        np.testing.assert_allclose(r3["x"][1], ro2["x"], atol=1e-8)
    pr.close()


def test_prepared_fixture_emulated():
    emu.build()
    _emu_wg_count()
    _check_fixture(emu.EMU_LIB)
    assert _emu_wg_count() >= 8 * 4  # every one of these solves ran in the workgroup form


def test_prepared_fixture_emulated_matrix_in_global_memory(monkeypatch):
    """a smaller "LDS": the big matrix of the workgroup form moves to its global workspace, the values are not staged"""
    emu.build()
    monkeypatch.setenv("QPC_EMU_WG_LDS", "9000")
    _check_fixture(emu.EMU_LIB)


def test_prepared_random_emulated():
    emu.build()
    _check_random(emu.EMU_LIB)


def test_prepared_rejects_a_bad_pattern():
    emu.build()
    with pytest.raises(binding.QpcError, match="duplicate"):
        binding.QpProblem(3, 2, (np.array([0]), np.array([0])), (np.array([0, 0]), np.array([1, 1])), lib_path=emu.EMU_LIB)
    with pytest.raises(binding.QpcError, match="out of range"):
        binding.QpProblem(3, 2, (np.array([0]), np.array([0])), (np.array([2]), np.array([1])), lib_path=emu.EMU_LIB)
    with pytest.raises(binding.QpcError, match="duplicate"):  # (0, 1) and (1, 0) are the same entry of a symmetric P
        binding.QpProblem(3, 2, (np.array([0, 1]), np.array([1, 0])), (np.array([0]), np.array([1])), lib_path=emu.EMU_LIB)


def test_prepared_warm_start_follows_a_changing_problem():
    """closed-loop use: bounds drift a little every call; the warm-started solve agrees with the cold one and needs fewer changes"""
    emu.build()
    rng = np.random.default_rng(4)
    P, q, A, l, u = _random_structured_qp(rng, 40, 70, 12)
    pr = binding.QpProblem(40, 70, P != 0, A != 0, lib_path=emu.EMU_LIB)
    Pv, Av = pr.values(P, A)
    pr.solve(Pv, q, Av, l, u)
    cold_it = warm_it = 0
    for _ in range(6):
        q = q + 0.02 * rng.normal(size=q.shape)
        rc = binding.qp_solve_batch(P, q, A, l, u, lib_path=emu.EMU_LIB)
        rw = pr.solve(Pv, q, Av, l, u, warm_start=True)
        assert rc["status"][0] == 0 and rw["status"][0] == 0
        np.testing.assert_allclose(rw["x"][0], rc["x"][0], atol=1e-8)
        cold_it += int(rc["iters"][0]); warm_it += int(rw["iters"][0])
    assert warm_it < cold_it, (warm_it, cold_it)
    pr.close()


@pytest.mark.gpu
def test_prepared_fixture_gpu():
    _check_fixture(None)


@pytest.mark.gpu
def test_prepared_random_gpu():
    _check_random(None)


@pytest.mark.gpu
def test_prepared_batch_gpu():
    """a batch of 256 of the committed problems through one prepared problem, cold then warm"""
    P, q, A, l, u, x, nv = _fixture()
    pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
    Pv, Av = pr.values(P, A)
    idx = np.arange(256) % P.shape[0]
    r = pr.solve(Pv[idx], q[idx], Av[idx], l[idx], u[idx])
    # more QPs than workgroups in flight (the grid is 4 per CU): every workgroup draws several tickets from the device counter
    big = np.arange(4096) % P.shape[0]
    rb = pr.solve(Pv[big], q[big], Av[big], l[big], u[big])
    assert (rb["status"] == 0).all()
    np.testing.assert_array_equal(rb["x"], r["x"][:P.shape[0]][big])  # the same problem gives the same bits wherever it runs
    assert (r["status"] == 0).all()
    np.testing.assert_allclose(r["x"][:, :nv], x[idx][:, :nv], atol=1e-8)
    rw = pr.solve(Pv[idx], q[idx], Av[idx], l[idx], u[idx], warm_start=True)
    assert (rw["status"] == 0).all() and (rw["iters"] <= 1).all()
    np.testing.assert_allclose(rw["x"], r["x"], atol=1e-9)
    pr.close()


@pytest.mark.gpu
def test_matrix_core_elimination_agrees_with_the_vector_unit_form_gpu():
    """on the device the trailing updates of the elimination run on the matrix core (v_mfma_f64_16x16x4_f64), in the host emulation on
    the vector-unit loops: the same sums in another order -- x of the committed problems to 1e-10, far inside the 1e-8 against the oracle"""
    emu.build()
    P, q, A, l, u, x, nv = _fixture()
    pat_P, pat_A = (P != 0).any(0), (A != 0).any(0)
    res = []
    for lib in (None, emu.EMU_LIB):
        pr = binding.QpProblem(P.shape[1], A.shape[1], pat_P, pat_A, lib_path=lib)
        Pv, Av = pr.values(P, A)
        res.append(pr.solve(Pv, q, Av, l, u))
        pr.close()
    assert (res[0]["status"] == 0).all() and (res[1]["status"] == 0).all()
    np.testing.assert_array_equal(res[0]["iters"], res[1]["iters"])
    scale = np.abs(res[1]["x"]).max(axis=1, keepdims=True)
    assert (np.abs(res[0]["x"] - res[1]["x"]) / scale).max() < 1e-10
