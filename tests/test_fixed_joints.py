"""Mechanisms that keep FIXED joints (nq = nv = 0: the reference normally removes them with
remove_fixed_tree_joints!, the ABI accepts them): more bodies than velocity coordinates, bodies
welded to their parents carrying tasks and contacts.  Fused kernel vs the oracle tick."""
import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd.workloads import HARD, JOINT, LINEAR, POINT, SCALAR


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def make(factory, floating, B=4, seed=21):
    rng = np.random.default_rng(seed)
    types = ([M.QUAT_FLOATING] if floating else []) + [M.REVOLUTE, M.FIXED, M.FIXED, M.REVOLUTE, M.FIXED, M.PRISMATIC,
                                                       M.FIXED, M.REVOLUTE, M.FIXED]
    mech = M.rand_tree(types, seed=seed, chain=True)
    nb = mech.nbodies
    assert floating or nb > mech.nv
    fl = 0 if floating else -1
    c = factory(mech, 4, fl)
    if floating:
        for sx in (-0.1, 0.1):
            for sy in (-0.05, 0.05):
                c.add_contact(nb - 1, [sx, sy, -0.1], [0.0, 0.0, 1.0], 0.8)  # on a body welded to its parent
    for j in range(nb):
        c.regularize(j, 0.1)
    c.add_task(POINT, base=M.WORLD, target=nb - 1, point=[0.05, 0.1, -0.2], wkind=SCALAR, weight=[4.0])
    c.add_task(JOINT, joint=nb - 2, wkind=HARD)
    c.add_task(LINEAR, base=M.WORLD, target=nb - 4, frame=nb - 3, wkind=SCALAR, weight=[1.5])
    c.finalize()
    qs, vs = zip(*(M.rand_state(mech, rng, 0.3) for _ in range(B)))
    nc = 4 if floating else 0
    des = rng.normal(size=(B, 3 + 1 + 3))
    return SC.Case("fixed_joints_" + ("floating" if floating else "fixedbase"), mech, c, np.array(qs), np.array(vs), des,
                   np.full((B, nc), 1e-2), np.full((B, nc), 1e4), floating=fl)


def _run(factory, floating):
    g, o = make(factory, floating), make(ora_factory, floating)
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf)
    parity.compare(g, rg, o, ro)


@pytest.mark.parametrize("floating", [False, True])
def test_fixed_joints_emulated(floating):
    emu.build()
    emu.race_report()
    _run(emu.factory, floating)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.parametrize("floating", [False, True])
def test_fixed_joints_gpu(floating):
    from qpcontrol_jl_amd import binding
    _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), floating)
