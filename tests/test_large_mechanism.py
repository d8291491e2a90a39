"""Sizes beyond Atlas: more than 32 bodies (second word of the subtree bit masks), about 50
velocity coordinates, many contacts with a 6-ray basis (60 rho's: almost one per lane), ~40
equality rows.  Fused kernel vs the oracle tick."""
import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd.workloads import ANGULAR, HARD, JOINT, LINEAR_MOMENTUM_RATE, POINT, SCALAR, SPATIAL


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def make(factory, B=3, seed=9, big=False):
    rng = np.random.default_rng(seed)
    types = [M.QUAT_FLOATING] + [M.REVOLUTE] * 40 + [M.PRISMATIC] * 3
    mech = M.rand_tree(types, seed=seed)
    nb = mech.nbodies
    assert nb > 32 and mech.nv == 49
    c = factory(mech, 6, 0)
    contact_bodies = [nb - 1, nb - 2, nb - 5, 35, 33] + ([30] if big else [])  # big: 72 rho's, two per lane
    for b in contact_bodies:
        for sx in (-0.1, 0.1):
            c.add_contact(b, [sx, 0.02, -0.1], [0.0, 0.1, 1.0], 0.9)
    for j in range(nb):
        c.regularize(j, 0.05)
    c.add_task(SPATIAL, base=M.WORLD, target=nb - 1, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[5.0])
    c.add_task(SPATIAL, base=M.WORLD, target=34, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[2.0])
    c.add_task(POINT, base=2, target=40, point=[0.1, 0.0, -0.1], wkind=SCALAR, weight=[3.0])
    c.add_task(ANGULAR, base=M.WORLD, target=0, frame=M.FRAME_DEFAULT, wkind=HARD)
    c.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0])
    extra = (25, 28, 31) if big else ()
    for t in extra:  # big: 45 equality rows (three 16 x 16 MFMA tiles per side)
        c.add_task(SPATIAL, base=M.WORLD, target=t, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[1.0])
    for j in (5, 12, 20, 33, 38, 41, 43):
        c.add_task(JOINT, joint=j, wkind=HARD)
    c.finalize()
    qs, vs = zip(*(M.rand_state(mech, rng, 0.2) for _ in range(B)))
    nc = 2 * len(contact_bodies)
    des = rng.normal(size=(B, 6 + 6 + 3 + 3 + 3 + 6 * len(extra) + 7)) * 0.5
    return SC.Case("large_mechanism" + ("_big" if big else ""), mech, c, np.array(qs), np.array(vs), des, np.full((B, nc), 1e-2),
                   np.full((B, nc), 1e4), floating=0, n_basis=6)


def _run(factory, big=False):
    g, o = make(factory, big=big), make(ora_factory, big=big)
    lay = g.ctrl.layout
    assert lay.n_rho == (72 if big else 60) and lay.lds_bytes_per_qp <= 160 * 1024
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf)
    parity.compare(g, rg, o, ro)
    return lay


@pytest.mark.parametrize("big", [False, True])
def test_large_mechanism_emulated(big):
    emu.build()
    emu.race_report()
    lay = _run(emu.factory, big)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first
    print("rows", lay.n_eq_rows, "lds", lay.lds_bytes_per_qp)


def test_deep_chain_emulated():
    """22 bodies in a row: deeper than the ahead-of-time kernel's ancestor bound (16), inside the library's (32)"""
    emu.build()
    emu.race_report()
    g, o = SC.deep_chain(emu.factory), SC.deep_chain(ora_factory)
    parity.compare(g, g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf), o, o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf))
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.parametrize("big", [pytest.param(False, marks=pytest.mark.gpu_slow), True])
def test_large_mechanism_gpu(big):
    from qpcontrol_jl_amd import binding
    _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), big)


# ---- the kernels' limits themselves (include/qpc.h: bodies <= 64, velocities <= 64): exactly at them, and one body beyond
def make_at_the_limit(factory, n_fixed, B=2, seed=5):
    rng = np.random.default_rng(seed)
    types = [M.QUAT_FLOATING] + [M.REVOLUTE] * 58 + [M.FIXED] * n_fixed  # nv = 6 + 58 = 64; 59 + n_fixed bodies
    mech = M.rand_tree(types, seed=seed)
    nb = mech.nbodies
    c = factory(mech, 4, 0)
    for b in (nb - 1, nb - 2, 20, 33):
        for sx in (-0.1, 0.1):
            c.add_contact(b, [sx, 0.02, -0.1], [0.0, 0.1, 1.0], 0.9)
    for j in range(nb):
        if mech.joint_type[j] != M.FIXED:
            c.regularize(j, 0.05)  # (64 velocities and nine task rows: without it v̇ is not unique)
    c.add_task(SPATIAL, base=M.WORLD, target=nb - 1, frame=M.FRAME_DEFAULT, wkind=SCALAR, weight=[5.0])
    c.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0])
    c.finalize()
    qs, vs = zip(*(M.rand_state(mech, rng, 0.2) for _ in range(B)))
    des = rng.normal(size=(B, 6 + 3)) * 0.5
    return SC.Case(f"limit_{nb}_bodies", mech, c, np.array(qs), np.array(vs), des, np.full((B, 8), 1e-2), np.full((B, 8), 1e4), floating=0, n_basis=4)


def _at_the_limit(factory):
    from qpcontrol_jl_amd import binding
    g, o = make_at_the_limit(factory, 5), make_at_the_limit(ora_factory, 5)
    assert g.mech.nbodies == 64 and g.mech.nv == 64  # every lane of the wavefront is a body, and a velocity coordinate
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf)
    assert (ro["status"] == 0).all()
    parity.compare(g, rg, o, ro)
    with pytest.raises(binding.QpcError, match="limits of the one-wave-per-QP kernels"):
        make_at_the_limit(factory, 6)  # 65 bodies: refused at qpc_finalize, not wrong


def test_exactly_64_bodies_and_64_velocities_emulated():
    emu.build()
    emu.race_report()
    _at_the_limit(emu.factory)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
def test_exactly_64_bodies_and_64_velocities_gpu():
    from qpcontrol_jl_amd import binding
    _at_the_limit(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl))
