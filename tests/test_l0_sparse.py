"""The sparse workgroup form of L0 (csrc/qpc_qp_sym.h: the plan, on the host; csrc/qpc_qp_sp.h: the kernel) -- what stands behind
qpc_qp_setup / qpc_qp_solve for a block-diagonal P: the reference's own plug point (the optimizer slot of `MomentumBasedController`,
momentum.jl:15-16,27,58) including `addtask!(controller, task, W::AbstractMatrix)` (momentum.jl:113-117; the reference's test modes
OBJECTIVE_TERM_MATRIX_WEIGHT*, test/controller.jl:232-285).  Checked against the oracle-made fixtures of tests/golden/l0, the oracle's
OSQP restatement, the dense workgroup form and the one-lane form; CPU emulation and GPU."""
import ctypes
import os
import sys

import numpy as np
import pytest

import emu
from oracle import oracle as O
from qpcontrol_jl_amd import binding
from test_l0_dense_qp import _random_structured_qp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def _fixture(which):
    import bench_latency as BL
    return BL.load_l0_fixture(which)


def _sp_count():
    L = ctypes.CDLL(emu.EMU_LIB)
    L.emu_l0_sp_count.restype = ctypes.c_long
    return L.emu_l0_sp_count()


def _problem(P, A, lib):
    pr = binding.QpProblem(P.shape[-1], A.shape[-2], (P != 0).any(0) if P.ndim == 3 else P != 0, (A != 0).any(0) if A.ndim == 3 else A != 0,
                           lib_path=lib)
    return pr, pr.values(P, A)


def test_plan_of_the_atlas_problem():
    """the symbolic analysis: 114 equality rows in three levels of fronts, 39 times fewer multiply-adds than the dense elimination"""
    emu.build()
    P, q, A, l, u, x, nv = _fixture("diag")
    pr, _ = _problem(P, A, emu.EMU_LIB)
    assert pr.plan_info()["in_use"] == 0  # (nothing analysed yet)
    info = pr.analyze(l[0], u[0])
    assert info["in_use"] == 1 and info["equality_rows"] == 114 and info["other_rows"] == 64
    assert info["levels"] <= 3 and info["widest_front"] <= 32
    assert info["elimination_madds"] * 20 < info["dense_elimination_madds"]
    assert info["lds_bytes"] <= 160 * 1024
    pr.close()


@pytest.mark.parametrize("which", ["diag", "matrix"])
def test_fixture_emulated(which, monkeypatch):
    """the committed problems (diagonal P; a 6 x 6 block in P): every QP in the sparse form, v̇ as stored, the same answer as the
    dense workgroup form / the one-lane form behind the same calls, warm start"""
    emu.build()
    P, q, A, l, u, x, nv = _fixture(which)
    pr, (Pv, Av) = _problem(P, A, emu.EMU_LIB)
    _sp_count()
    r = pr.solve(Pv, q, Av, l, u)
    assert _sp_count() == P.shape[0]
    assert (r["status"] == 0).all()
    np.testing.assert_allclose(r["x"][:, :nv], x[:, :nv], atol=1e-8)
    monkeypatch.setenv("QPC_L0_NO_SPARSE", "1")
    r0 = pr.solve(Pv, q, Av, l, u)
    monkeypatch.delenv("QPC_L0_NO_SPARSE")
    assert _sp_count() == 0 and (r0["status"] == 0).all()
    if which == "diag":  # (same method, same ties: the dense workgroup form takes the same steps)
        np.testing.assert_array_equal(r["iters"], r0["iters"])
    # (the basis multipliers rho of a contact are not unique -- four rays for a three-dimensional force --: what the two forms must
    #  agree on is the objective, feasibility, and everything the objective determines: v̇, the contact forces)
    obj = lambda xx: 0.5 * np.einsum("bi,bij,bj->b", xx, P, xx) + np.einsum("bi,bi->b", q, xx)  # noqa: E731
    np.testing.assert_allclose(obj(r["x"]), obj(r0["x"]), rtol=1e-9, atol=1e-9)
    for rr in (r, r0):
        Ax = np.einsum("bij,bj->bi", A, rr["x"])
        sc = max(1.0, np.abs(Ax).max())
        assert (Ax >= l - 1e-8 * sc).all() and (Ax <= u + 1e-8 * sc).all()
    np.testing.assert_allclose(r["x"][:, :nv], r0["x"][:, :nv], atol=1e-9)
    rw = pr.solve(Pv, q, Av, l, u, warm_start=True)
    assert (rw["status"] == 0).all() and (rw["iters"] <= 1).all()
    np.testing.assert_allclose(rw["x"][:, :nv], x[:, :nv], atol=1e-8)
    pr.close()


def _random_block_qp(rng, n, m, n_eq, nblocks, bs):
    """a structured QP whose P has `nblocks` dense positive definite blocks of `bs` variables (scattered), diagonal elsewhere"""
    P, q, A, l, u = _random_structured_qp(rng, n, m, n_eq)
    free = list(rng.permutation(n))
    for _ in range(nblocks):
        idx = np.array(sorted(free[:bs])); free = free[bs:]
        M = rng.normal(size=(bs, bs))
        P[np.ix_(idx, idx)] = M @ M.T / bs + np.diag(rng.uniform(0.1, 1.0, size=bs))
    return P, q, A, l, u


def _check_random_blocks(lib, seeds=range(12)):
    n_checked = 0
    for seed in seeds:
        rng = np.random.default_rng(100 + seed)
        n, m, ne = [(20, 26, 5), (24, 36, 8), (48, 80, 14), (72, 110, 30)][seed % 4]
        P, q, A, l, u = _random_block_qp(rng, n, m, ne, nblocks=1 + seed % 3, bs=2 + seed % 5)
        ro = O.qp_solve(P, q, A, l, u)
        if ro["status"] != 0:
            continue
        pr, (Pv, Av) = _problem(P, A, lib)
        r = pr.solve(Pv, q, Av, l, u)
        assert pr.plan_info()["in_use"] == 1
        assert r["status"][0] == 0, (seed, r["status"])
        np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
        pr.close()
        n_checked += 1
    assert n_checked >= 8


def test_random_block_diagonal_p_emulated():
    emu.build()
    _sp_count()
    _check_random_blocks(emu.EMU_LIB)
    assert _sp_count() >= 8


def test_a_block_beyond_eight_variables_takes_the_one_lane_form():
    emu.build()
    rng = np.random.default_rng(5)
    P, q, A, l, u = _random_block_qp(rng, 20, 30, 5, nblocks=1, bs=9)
    pr, (Pv, Av) = _problem(P, A, emu.EMU_LIB)
    ro = O.qp_solve(P, q, A, l, u)
    _sp_count()
    r = pr.solve(Pv, q, Av, l, u)
    assert pr.plan_info()["in_use"] == 0 and _sp_count() == 0
    assert ro["status"] == 0 and r["status"][0] == 0
    np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
    pr.close()


def _mask_change(lib, count=None):
    """a batch whose second QP has OTHER equality rows than the plan's (made for the first): that QP is left to the dense workgroup
    form; a later batch led by such a QP gets a plan of its own"""
    rng = np.random.default_rng(9)
    P, q, A, l, u = _random_structured_qp(rng, 30, 44, 8)
    l2, u2 = l.copy(), u.copy()
    l2[9] = u2[9] = 0.5 * (l[10] + u[10]) if np.isfinite(l[10]) and l[10] > -1e19 else u[9]  # row 9 becomes an equality row
    l2[9] = u2[9]
    pr, (Pv, Av) = _problem(P, A, lib)
    ro1, ro2 = O.qp_solve(P, q, A, l, u), O.qp_solve(P, q, A, l2, u2)
    if count:
        count()
    r = pr.solve(np.stack([Pv[0]] * 2), np.stack([q] * 2), np.stack([Av[0]] * 2), np.stack([l, l2]), np.stack([u, u2]))
    if count:
        assert count() == 1  # (the first QP only)
    for k, ro in enumerate((ro1, ro2)):
        if ro["status"] == 0:
            assert r["status"][k] == 0
            np.testing.assert_allclose(r["x"][k], ro["x"], atol=1e-8)
    assert pr.plan_info()["equality_rows"] == 8
    r2 = pr.solve(Pv[0], q, Av[0], l2, u2)
    assert pr.plan_info()["equality_rows"] == 9 and pr.plan_info()["in_use"] == 1
    if count:
        assert count() == 1
    if ro2["status"] == 0:
        np.testing.assert_allclose(r2["x"][0], ro2["x"], atol=1e-8)
    pr.close()


def test_other_equality_rows_than_the_plans_emulated():
    emu.build()
    _mask_change(emu.EMU_LIB, _sp_count)


def test_no_equality_rows_and_no_inequality_rows_emulated():
    """the edges of the row classification: every row an inequality (no fronts at all); every row an equality (C is the corner)"""
    emu.build()
    rng = np.random.default_rng(3)
    for n, m, ne in ((12, 18, 0), (12, 7, 7)):
        P, q, A, l, u = _random_structured_qp(rng, n, m, ne)
        ro = O.qp_solve(P, q, A, l, u)
        pr, (Pv, Av) = _problem(P, A, emu.EMU_LIB)
        _sp_count()
        r = pr.solve(Pv, q, Av, l, u)
        assert _sp_count() == 1 and pr.plan_info()["equality_rows"] == ne
        assert ro["status"] == 0 and r["status"][0] == 0
        np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
        pr.close()


def _mini_fuzz(lib, seeds):
    """small random problems of many shapes (sparse and dense rows, few and many equality rows, rank-deficient ones, blocks in P):
    elimination trees of every kind -- chains, stars, levels that hold single-column fronts only -- against the oracle"""
    n_ok = 0
    for seed in seeds:
        rng = np.random.default_rng(7000 + seed)
        n, m = int(rng.integers(2, 40)), int(rng.integers(2, 60))
        ne = int(rng.integers(0, min(n, m) + 1))
        dens = rng.choice([0.05, 0.15, 0.4, 1.0])
        P = np.diag(rng.uniform(0.05, 2.0, size=n))
        if rng.uniform() < 0.4 and n >= 6:
            idx = np.sort(rng.permutation(n)[:int(rng.integers(2, 6))])
            M = rng.normal(size=(len(idx), len(idx)))
            P[np.ix_(idx, idx)] = M @ M.T / len(idx) + np.diag(rng.uniform(0.1, 1.0, size=len(idx)))
        q = rng.normal(size=n)
        A = rng.normal(size=(m, n)) * (rng.uniform(size=(m, n)) < dens)
        if rng.uniform() < 0.5:
            A[np.arange(m), rng.integers(0, n, size=m)] += rng.uniform(0.5, 1.5, size=m)
        b = A @ rng.normal(size=n)
        l, u = b - rng.uniform(0.0, 0.6, size=m), b + rng.uniform(0.0, 0.6, size=m)
        l[:ne] = u[:ne] = b[:ne]
        l[ne::3] = -1e30
        perm = rng.permutation(m)
        A, l, u = A[perm], l[perm], u[perm]
        ro = O.qp_solve(P, q, A, l, u)
        if ro["status"] != 0:
            continue
        pr = binding.QpProblem(n, m, P != 0, (A != 0) | (rng.uniform(size=A.shape) < 0.02), lib_path=lib)
        Pv, Av = pr.values(P, A)
        r = pr.solve(Pv, q, Av, l, u)
        assert pr.plan_info()["in_use"] == 1
        assert r["status"][0] == 0, (seed, n, m, ne, r["status"])
        np.testing.assert_allclose(r["x"][0], ro["x"], atol=2e-7, err_msg=f"seed {seed}")
        pr.close()
        n_ok += 1
    return n_ok


# (seeds 198, 226, 856, 1383: trees whose upper level holds single-column fronts only -- once never factored)
MINI_FUZZ_SEEDS = list(range(150)) + [198, 226, 856, 1383]


def test_mini_fuzz_emulated():
    emu.build()
    assert _mini_fuzz(emu.EMU_LIB, MINI_FUZZ_SEEDS) >= 100


@pytest.mark.gpu
def test_mini_fuzz_gpu():
    assert _mini_fuzz(None, MINI_FUZZ_SEEDS) >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["diag", "matrix"])
def test_fixture_gpu(which):
    """the committed problems on the device, in the sparse form: v̇ as stored (1e-8), the same steps and answer as the lane-serial
    emulation of the same source (matrix core against the scalar twin of the tile jobs: another summation order)"""
    emu.build()
    P, q, A, l, u, x, nv = _fixture(which)
    res = []
    for lib in (None, emu.EMU_LIB):
        pr, (Pv, Av) = _problem(P, A, lib)
        r = pr.solve(Pv, q, Av, l, u)
        assert pr.plan_info()["in_use"] == 1
        res.append(r)
        if lib is None:
            rw = pr.solve(Pv, q, Av, l, u, warm_start=True)
            assert (rw["status"] == 0).all() and (rw["iters"] <= 1).all()
            np.testing.assert_allclose(rw["x"][:, :nv], x[:, :nv], atol=1e-8)
        pr.close()
    assert (res[0]["status"] == 0).all()
    np.testing.assert_allclose(res[0]["x"][:, :nv], x[:, :nv], atol=1e-8)
    np.testing.assert_array_equal(res[0]["iters"], res[1]["iters"])
    np.testing.assert_allclose(res[0]["x"][:, :nv], res[1]["x"][:, :nv], atol=1e-10)


@pytest.mark.gpu
def test_random_block_diagonal_p_gpu():
    _check_random_blocks(None)


@pytest.mark.gpu
def test_other_equality_rows_than_the_plans_gpu():
    _mask_change(None)


@pytest.mark.gpu
def test_working_set_beyond_lds_goes_down_the_chain_gpu():
    """150 box rows, all of them active at the solution: the working set's factor does not fit beside the plan in LDS -- the sparse
    form leaves the QP, the dense workgroup form (at most 128 working rows) leaves it too, the one-lane form solves it; mixed into a
    batch of problems the sparse form keeps (the statuses' round trip, `only_fallback`), cold and warm"""
    rng = np.random.default_rng(12)
    n = 150
    P = np.diag(rng.uniform(0.5, 2.0, size=n))
    xt = rng.normal(size=n)
    A = np.eye(n)
    lo, hi = xt - 0.05, xt + 0.05
    q = -P @ (xt + np.where(rng.uniform(size=n) < 0.5, 1.0, -1.0))  # the unconstrained minimiser lies one unit outside every box
    q2 = -P @ xt                                                      # ... inside every box
    pr, (Pv, Av) = _problem(P, A, None)
    B = 6
    qs = np.stack([q2, q, q2, q2, q, q2])
    for warm in (False, True):
        r = pr.solve(np.stack([Pv[0]] * B), qs, np.stack([Av[0]] * B), np.stack([lo] * B), np.stack([hi] * B), warm_start=warm)
        assert (r["status"] == 0).all(), r["status"]
        for b in range(B):
            xe = np.clip(-qs[b] / np.diag(P), lo, hi)
            np.testing.assert_allclose(r["x"][b], xe, atol=1e-8)
    pr.close()


# ---- the device-pointer entry (qpc_qp_solve_device): same forms, same order, same bits as the host entry
def _device_entry(lib, to_dev, from_dev, stream=0, rep=3):
    """`to_dev(array) -> (holder, address)`, `from_dev(holder) -> numpy`: torch on the GPU, numpy itself on the emulation.
    rep: copies of the eight committed problems (GPU: 300 -- 2 400 QPs, more than one chunk of the dense forms behind the sparse one)"""
    P, q, A, l, u, x_ref, nv = _fixture("diag")
    Pb, qb, Ab, lb, ub = (np.concatenate([a] * rep) for a in (P, q, A, l, u))
    l2, u2 = lb.copy(), ub.copy()
    row = int(np.where(l[0] < u[0] - 1.0)[0][0])  # an inequality row of QP 5 turned into an equality: not the plan's rows, left to the next form
    odd = [5] + ([2100] if rep * 8 > 2100 else [])
    for b in odd:
        u2[b, row] = l2[b, row] = 0.5 * (max(l[b % 8, row], -10.0) + min(u[b % 8, row], 10.0))
    B = Pb.shape[0]
    for lo, hi in ((lb, ub), (l2, u2)):
        host, dev = _problem(P, A, lib)[0], _problem(P, A, lib)[0]
        Pv, Av = host.values(Pb, Ab)
        lc, uc = np.clip(lo, -1e20, 1e20), np.clip(hi, -1e20, 1e20)
        dev.analyze(lc[0], uc[0])
        ins = [to_dev(np.ascontiguousarray(a, dtype=np.float64)) for a in (Pv, qb, Av, lc, uc)]
        outs = [to_dev(np.zeros((B, host.n))), to_dev(np.zeros((B, host.m))), to_dev(np.full(B, -7, dtype=np.int32)), to_dev(np.zeros(B, dtype=np.int32))]
        for warm in (False, True):
            rh = host.solve(Pv, qb, Av, lo, hi, warm_start=warm)
            assert host.plan_info() == dev.plan_info()
            dev.solve_device(B, *(a[1] for a in ins), *(o[1] for o in outs), warm_start=warm, stream=stream)
            xd, yd, sd, itd = (from_dev(o[0]) for o in outs)
            assert np.array_equal(sd, rh["status"]), (sd, rh["status"])
            assert np.array_equal(xd, rh["x"]) and np.array_equal(yd, rh["y"]) and np.array_equal(itd, rh["iters"])  # bit for bit
            plain = ~np.isin(np.arange(B), odd)
            assert np.all(sd[plain] == 0)
            np.testing.assert_allclose(xd[plain][:, :nv], np.concatenate([x_ref] * rep)[plain][:, :nv], atol=1e-8, rtol=0)


def test_device_pointer_entry_emulated():
    emu.build()
    _device_entry(emu.EMU_LIB, lambda a: (a, a.ctypes.data), lambda h: h)


def gpu_device_entry_check():
    """body of test_device_pointer_entry_gpu: runs in a child process (tests/gpu_torch_checks.py l0_device) in which torch's HIP runtime
    was initialised before the library's, as in bench.py"""
    import torch
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(dev)

    def to_dev(a):
        t = torch.from_numpy(a).to(dev)
        return t, t.data_ptr()
    torch.cuda.synchronize(dev)
    _device_entry(None, to_dev, lambda t: t.cpu().numpy(), stream=s.cuda_stream)
    _device_entry(None, to_dev, lambda t: t.cpu().numpy(), stream=s.cuda_stream, rep=300)
    # without a plan (no analyze, no earlier solve) the dense forms take the batch: still the same answers
    P, q, A, l, u, x_ref, nv = _fixture("diag")
    pr = _problem(P, A, None)[0]
    Pv, Av = pr.values(P[:2], A[:2])
    ins = [to_dev(np.ascontiguousarray(a, dtype=np.float64)) for a in (Pv, q[:2], Av, np.clip(l[:2], -1e20, 1e20), np.clip(u[:2], -1e20, 1e20))]
    outs = [to_dev(np.zeros((2, pr.n))), to_dev(np.zeros((2, pr.m))), to_dev(np.full(2, -7, dtype=np.int32)), to_dev(np.zeros(2, dtype=np.int32))]
    torch.cuda.synchronize(dev)
    pr.solve_device(2, *(a[1] for a in ins), *(o[1] for o in outs), stream=s.cuda_stream)
    assert pr.plan_info()["in_use"] == 0
    assert np.all(outs[2][0].cpu().numpy() == 0)
    np.testing.assert_allclose(outs[0][0].cpu().numpy()[:, :nv], x_ref[:2, :nv], atol=1e-8, rtol=0)


@pytest.mark.gpu
def test_device_pointer_entry_gpu():
    """qpc_qp_solve_device on torch tensors: bit for bit the host entry, cold and warm, with a QP the sparse form leaves in the batch, and
    without a plan"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "gpu_torch_checks.py"), "l0_device"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "CHECK_OK" in out.stdout


# ---- the reference's regime for the slot: consecutive ticks of one closed loop, each solve warm-started from the tick before
def _closed_loop(lib):
    P, q, A, l, u, x, nv = _fixture("closed_loop")
    K = P.shape[0]
    assert K == 96
    pr, (Pv, Av) = _problem(P, A, lib)
    pr.analyze(l[0], u[0])
    assert pr.plan_info()["in_use"] == 1
    iters = []
    for k in range(K):
        r = pr.solve(Pv[k], q[k], Av[k], l[k], u[k], warm_start=k > 0)
        assert r["status"][0] == 0, (k, r["status"])
        np.testing.assert_allclose(r["x"][0, :nv], x[k, :nv], atol=1e-8, rtol=0, err_msg=f"tick {k}")
        iters.append(int(r["iters"][0]))
    assert pr.plan_info()["in_use"] == 1 and pr.plan_info()["equality_rows"] == 114  # (the lost contact changes bounds, not the kind of any row)
    pr.close()
    iters = np.array(iters[1:])
    # warm-started from the tick before, most ticks need no change of the working set; the push and the lost contact need a few
    assert (iters == 0).mean() > 0.6 and iters.max() <= 12, iters
    return iters


def test_closed_loop_sequence_emulated():
    emu.build()
    _closed_loop(emu.EMU_LIB)


@pytest.mark.gpu
def test_closed_loop_sequence_gpu():
    it = _closed_loop(None)
    print(f"\nclosed-loop sequence: working-set changes per tick mean {it.mean():.2f} max {it.max()}, {int((it == 0).sum())} of {len(it)} ticks without a change")


# ---- a plan whose LDS need comes out at exactly 160 KB (the kernel holds 8 more bytes of static LDS: the dispatch failed with
#      HSA_STATUS_ERROR_INVALID_ALLOCATION and took the process down; found by tests/offline/fuzz_l0.py seed 7003 on the GPU)
def _last_byte_of_lds(lib):
    for n in (129, 151, 168):  # box problems whose plan + working-set factor filled 20 473 / 20 479 / 20 480 of the 20 480 doubles
        rng = np.random.default_rng(n)
        P = np.diag(rng.uniform(0.5, 2.0, size=n))
        xt = rng.normal(size=n)
        A = np.eye(n)
        lo, hi = xt - 0.05, xt + 0.05
        q = -P @ (xt + np.where(np.arange(n) < 20, 1.0, 0.0))  # twenty boxes active at the solution, the others not
        pr, (Pv, Av) = _problem(P, A, lib)
        info = pr.analyze(lo, hi)
        assert info["in_use"] == 1 and info["lds_bytes"] <= 160 * 1024 - 64, info
        r = pr.solve(Pv[0], q, Av[0], lo, hi)
        assert r["status"][0] == 0
        np.testing.assert_allclose(r["x"][0], np.clip(-q / np.diag(P), lo, hi), atol=1e-8)
        pr.close()


def test_plan_that_takes_the_last_byte_of_lds_emulated():
    emu.build()
    _last_byte_of_lds(emu.EMU_LIB)


@pytest.mark.gpu
def test_plan_that_takes_the_last_byte_of_lds_gpu():
    _last_byte_of_lds(None)


# ---- the workgroup forms' dimension limit (n, m <= 256: QPC_SP_DIM, QPC_WG_DIM) at it and one beyond
def _dimension_limits(lib):
    # 256 variables and 256 rows in the sparse form: 200 rows pin a variable each (equalities), 56 are boxes -- small enough a value store for LDS
    n = 256
    rng = np.random.default_rng(256)
    P = np.diag(rng.uniform(0.5, 2.0, size=n))
    xt = rng.normal(size=n)
    A = np.eye(n)[rng.permutation(n)]
    lo, hi = A @ xt - 0.05, A @ xt + 0.05
    lo[:200] = hi[:200] = (A @ xt)[:200] + 0.01
    q = -P @ (xt + 1.0)  # the unconstrained minimiser lies outside every box
    pr, (Pv, Av) = _problem(P, A, lib)
    r = pr.solve(Pv[0], q, Av[0], lo, hi)
    assert pr.plan_info()["in_use"] == 1 and pr.plan_info()["equality_rows"] == 200, pr.plan_info()
    assert r["status"][0] == 0
    xe = A.T @ np.where(np.arange(n) < 200, lo, np.clip(A @ (-q / np.diag(P)), lo, hi))
    np.testing.assert_allclose(r["x"][0], xe, atol=1e-8)
    pr.close()
    # random structured problems: 256 x 256 with 216 inequality rows -- a plan that leaves no room for a working set in LDS: the dense
    # workgroup form takes it --, and 257 variables: neither workgroup form, the one-lane form solves it
    for (n, m, ne) in ((256, 256, 40), (257, 200, 30)):
        rng = np.random.default_rng(n * 1000 + m)
        P, q, A, l, u = _random_structured_qp(rng, n, m, ne)
        ro = O.qp_solve(P, q, A, l, u)
        assert ro["status"] == 0
        pr, (Pv, Av) = _problem(P, A, lib)
        r = pr.solve(Pv[0], q, Av[0], l, u)
        assert pr.plan_info()["in_use"] == 0, (n, m, ne, pr.plan_info())
        assert r["status"][0] == 0
        np.testing.assert_allclose(r["x"][0], ro["x"], atol=1e-8)
        pr.close()


def test_dimension_limits_emulated():
    emu.build()
    _dimension_limits(emu.EMU_LIB)


@pytest.mark.gpu
def test_dimension_limits_gpu():
    _dimension_limits(None)


@pytest.mark.gpu
def test_large_batch_in_chunks_with_leftovers_gpu():
    """B = 4 608: the batch goes through the sparse form in 1 024-problem chunks over two streams (qpc_l0.hip `piped`); three QPs with other
    equality rows than the plan's are left by it and taken by the dense workgroup form afterwards -- every QP's answer as if solved alone,
    cold and warm-started"""
    P, q, A, l, u, x_ref, nv = _fixture("diag")
    k = P.shape[0]
    B = 4608
    idx = np.arange(B) % k
    lb, ub = l[idx].copy(), u[idx].copy()
    row = int(np.where(l[0] < u[0] - 1.0)[0][0])
    odd = [5, 2049, 4600]  # (one in the first chunk, one in the middle, one in the last, partial chunk)
    for b in odd:
        ub[b, row] = lb[b, row] = 0.5 * (max(l[idx[b], row], -10.0) + min(u[idx[b], row], 10.0))
    pr, (Pv, Av) = _problem(P, A, None)
    Pb, Ab, qb = Pv[idx], Av[idx], q[idx]
    single = {}
    for b in odd:
        p1, _ = _problem(P, A, None)
        single[b] = p1.solve(Pb[b], qb[b], Ab[b], lb[b], ub[b])
        p1.close()
    for warm in (False, True):
        r = pr.solve(Pb, qb, Ab, lb, ub, warm_start=warm)
        assert pr.plan_info()["in_use"] == 1 and pr.plan_info()["equality_rows"] == 114
        ok = np.ones(B, dtype=bool); ok[odd] = False
        assert (r["status"][ok] == 0).all()
        np.testing.assert_allclose(r["x"][ok][:, :nv], x_ref[idx][ok][:, :nv], atol=1e-8, rtol=0)
        for b in odd:
            assert r["status"][b] == single[b]["status"][0]
            if single[b]["status"][0] == 0:
                np.testing.assert_allclose(r["x"][b, :nv], single[b]["x"][0, :nv], atol=1e-8, rtol=0)
    pr.close()


def test_bind_call_is_the_same_call_emulated():
    """QpProblem.bind_call / Controller.bind_call (pointers converted once: what the latency tools time) make the same calls as solve_raw /
    control_batch: bit-identical results, in place"""
    emu.build()
    P, q, A, l, u, x, nv = _fixture("diag")
    pr, (Pv, Av) = _problem(P, A, emu.EMU_LIB)
    lc, uc = np.clip(l, -1e20, 1e20), np.clip(u, -1e20, 1e20)
    a = tuple(np.ascontiguousarray(z[2:5]) for z in (Pv, q, Av, lc, uc))
    ref = pr.solve_raw(*a)
    out = dict(x=np.zeros_like(ref["x"]), y=np.zeros_like(ref["y"]), status=np.full(3, -9, dtype=np.int32), iters=np.zeros(3, dtype=np.int32))
    call = pr.bind_call(*a, out)
    assert call() == 0
    for k in ("x", "y", "status", "iters"):
        assert np.array_equal(out[k], ref[k]), k
    warm = pr.bind_call(*a, out, warm_start=True)
    assert warm() == 0 and (out["status"] == 0).all() and (out["iters"] <= 1).all()
    pr.close()
    from qpcontrol_jl_amd import workloads as W
    c, wl = W.atlas_workload(emu.factory, 2, 3)
    b = tuple(np.ascontiguousarray(z) for z in (wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce))
    r = c.control_batch(*b)
    lay = c.layout
    o = dict(tau=np.zeros((2, lay.nv)), vdot=np.zeros((2, lay.nv)), cwrench_flat=np.zeros((2, 6 * lay.n_contacts)), status=np.full(2, -9, dtype=np.int32),
             iters=np.zeros(2, dtype=np.int32))
    tick = c.bind_call(*b, o)
    assert tick() == 0
    assert np.array_equal(o["tau"], r["tau"]) and np.array_equal(o["vdot"], r["vdot"]) and np.array_equal(o["status"], r["status"])
    assert np.array_equal(o["cwrench_flat"].reshape(2, lay.n_contacts, 6), r["cwrench"])
