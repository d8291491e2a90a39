"""Shared controller scenarios: each builds the same task/contact set on ANY controller backend
(oracle, numpy prototype recorder, emulated kernel, the HIP product binding) and returns the
batched inputs.  They restate the setups of reference test/controller.jl, the two notebooks and
BASELINE.json configs 1-5 on the synthetic models."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W
from qpcontrol_jl_amd.workloads import (ANGULAR, HARD, JOINT, LINEAR, LINEAR_MOMENTUM_RATE, MATRIX,
                                        MATRIX_PER_QP, MOMENTUM_RATE, POINT, SCALAR, SCALAR_PER_QP,
                                        SPATIAL)


@dataclass
class Case:
    name: str
    mech: M.Mechanism
    ctrl: object
    q: np.ndarray
    v: np.ndarray
    desired: np.ndarray
    cweight: np.ndarray
    cmaxf: np.ndarray
    tw: np.ndarray = None
    surf: np.ndarray = None
    floating: int = -1
    n_basis: int = 4
    tol_tau: float = 1e-8   # north-star tolerance on tau vs the CPU OSQP path
    expect_status: np.ndarray = None

    @property
    def B(self):
        return self.q.shape[0]


def _states(mech, B, rng, vscale=1.0):
    qs, vs = zip(*(M.rand_state(mech, rng, vscale) for _ in range(B)))
    return np.array(qs), np.array(vs)


def fixed_base_joint_space(factory, constrained, B=4, seed=42):
    """test/controller.jl:68-97"""
    rng = np.random.default_rng(seed)
    mech = M.rand_tree([M.PRISMATIC, M.REVOLUTE, M.REVOLUTE], seed=seed)
    c = factory(mech, 4, -1)
    for j in range(mech.nbodies):
        c.add_task(JOINT, joint=j, wkind=HARD if constrained else SCALAR, weight=None if constrained else [1.0])
    c.finalize()
    q, v = _states(mech, B, rng)
    return Case(f"fixed_base_{'hard' if constrained else 'soft'}", mech, c, q, v, rng.uniform(size=(B, mech.nv)),
                np.zeros((B, 0)), np.zeros((B, 0)))


def acrobot_config1(factory, B=1):
    """BASELINE config 1: Acrobot, one hard JointAccelerationTask on the elbow, d = 0.7."""
    mech = M.acrobot()
    c = factory(mech, 4, -1)
    c.add_task(JOINT, joint=mech.findjoint("elbow"), wkind=HARD)
    for j in range(mech.nbodies):
        c.regularize(j, 1e-6)
    c.finalize()
    q = np.tile([-np.pi, 0.1], (B, 1))
    v = np.tile([0.3, -0.2], (B, 1))
    return Case("acrobot_config1", mech, c, q, v, np.full((B, 1), 0.7), np.zeros((B, 0)), np.zeros((B, 0)))


def acrobot_point_task(factory, B=4, seed=9):
    """notebooks/PointAccelerationTask Demo.ipynb cell 9: hard PointAccelerationTask at the tip."""
    rng = np.random.default_rng(seed)
    mech = M.acrobot()
    c = factory(mech, 4, -1)
    c.add_task(POINT, base=M.WORLD, target=1, point=[0, 0, -2.05], wkind=HARD)
    for j in range(mech.nbodies):
        c.regularize(j, 1e-6)
    c.finalize()
    q = rng.uniform(-1.0, 1.0, size=(B, 2)) + np.array([-np.pi, 0.8])
    v = rng.normal(size=(B, 2))
    des = rng.normal(size=(B, 3)) * np.array([1.0, 0.0, 1.0])  # the tip moves in the x-z plane
    return Case("acrobot_point", mech, c, q, v, des, np.zeros((B, 0)), np.zeros((B, 0)))


def free_fall(factory, B=3, seed=5354):
    """test/controller.jl:128-165: all contacts present but maxnormalforce = 0, floating base
    unregularised."""
    rng = np.random.default_rng(seed)
    mech = M.atlas()
    fl = mech.findbody("pelvis")
    c = factory(mech, 4, fl)
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        if j != fl:
            c.regularize(j, 1.0)
    c.finalize()
    q, _ = _states(mech, B, rng)
    nc = len(mech.contact_points)
    return Case("free_fall", mech, c, q, np.zeros((B, mech.nv)), np.zeros((B, 0)), np.zeros((B, nc)),
                np.zeros((B, nc)), floating=fl)


def momentum_rate(factory, seed=533454, mech=None, name="momentum_rate"):
    """test/controller.jl:169-230: hard MomentumRateTask, per-QP contact normals/mu, random active
    sets p in {0,.25,.5,.75,1} with achievable desired momentum rates.  mech: another humanoid than Atlas (its floating body is `pelvis`)"""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    mech = M.atlas() if mech is None else mech
    om = O.OracleModel(mech)  # only to generate the achievable desired wrench
    fl = mech.findbody("pelvis")
    c = factory(mech, 4, fl)
    nc = len(mech.contact_points)
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu, per_qp_surface=1)
    c.add_task(MOMENTUM_RATE, wkind=HARD)
    for j in range(mech.nbodies):
        c.regularize(j, 1e-6)
    c.finalize()
    normals = rng.normal(size=(nc, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    mus = rng.uniform(size=nc)
    surf1 = np.concatenate([normals, mus[:, None]], axis=1).reshape(-1)
    qs, vs, des, cws, cmaxs = [], [], [], [], []
    for p_active in np.linspace(0, 1, 5):
        q, v = M.rand_state(mech, rng)
        om.set_state(q, v)
        com = om.center_of_mass()
        fg = mech.total_mass() * np.array(mech.gravity)
        hdes = np.concatenate([np.zeros(3), fg])
        cw, cmax = np.zeros(nc), np.zeros(nc)
        for ci, cp in enumerate(mech.contact_points):
            if rng.uniform() < p_active:
                n, mu = normals[ci], mus[ci]
                cw[ci], cmax[ci] = 1e-6, 1e9
                fn = 50.0 * rng.uniform()
                t = rng.normal(size=3)
                f = fn * n + np.sqrt(2) / 2 * mu * fn * rng.uniform() * np.cross(n, t / np.linalg.norm(t))
                R, pb = om.transform(cp.body)
                fw, pw = R @ f, R @ cp.position + pb
                hdes += np.concatenate([np.cross(pw - com, fw), fw])
        qs.append(q); vs.append(v); des.append(hdes); cws.append(cw); cmaxs.append(cmax)
    B = len(qs)
    return Case(name, mech, c, np.array(qs), np.array(vs), np.array(des), np.array(cws),
                np.array(cmaxs), surf=np.tile(surf1, (B, 1)), floating=fl)


def spatial_acceleration(factory, mode, B=2, seed=533, mech=None, body="l_foot", base="r_hand", name=None):
    """test/controller.jl:232-285: foot w.r.t. palm SpatialAccelerationTask in palm frame, 5 modes."""
    rng = np.random.default_rng(seed)
    mech = M.atlas() if mech is None else mech
    fl = mech.findbody("pelvis")
    c = factory(mech, 4, fl)
    body, base = mech.findbody(body), mech.findbody(base)
    tw = None
    kw = dict(base=base, target=body, frame=base)
    if mode == "constraint":
        c.add_task(SPATIAL, wkind=HARD, **kw)
        for j in range(mech.nbodies):
            c.regularize(j, 1.0)
    elif mode == "scalar":
        c.add_task(SPATIAL, wkind=SCALAR, weight=[1.0], **kw)
    elif mode == "scalar_param":
        c.add_task(SPATIAL, wkind=SCALAR_PER_QP, **kw)
        tw = np.ones((B, 1))
    elif mode == "matrix":
        c.add_task(SPATIAL, wkind=MATRIX, weight=np.eye(6), **kw)
    elif mode == "matrix_param":
        c.add_task(SPATIAL, wkind=MATRIX_PER_QP, **kw)
        tw = np.tile(np.eye(6).reshape(-1), (B, 1))
    else:
        raise ValueError(mode)
    c.finalize()
    q, v = _states(mech, B, rng)
    # non-unique v̇ in the soft modes (no regularisation): only the task acceleration is compared
    return Case(name or f"spatial_{mode}", mech, c, q, v, rng.uniform(size=(B, 6)), np.zeros((B, 0)),
                np.zeros((B, 0)), tw=tw, floating=fl, tol_tau=1e-8 if mode == "constraint" else np.inf)


def mixed_tasks(factory, B=6, seed=77, floating=True):
    """Every task type at once on a random floating tree, soft and hard, with contacts: exercises
    all assembly paths (point/linear/angular tasks in foreign frames, matrix weights, pins)."""
    rng = np.random.default_rng(seed)
    types = ([M.QUAT_FLOATING] if floating else []) + [M.REVOLUTE] * 9 + [M.PRISMATIC] * 2
    mech = M.rand_tree(types, seed=seed, chain=not floating)
    fl = 0 if floating else -1
    c = factory(mech, 4, fl)
    nb = mech.nbodies
    for k, b in enumerate((nb - 1, nb - 3)):
        for sx in (-0.1, 0.1):
            c.add_contact(b, [sx, 0.05 * k, -0.1], [0.1 * k, 0.0, 1.0], 0.7)
    for j in range(nb):
        c.regularize(j, 0.05 + 0.01 * j)
    c.add_task(ANGULAR, base=M.WORLD, target=nb - 2, frame=3, wkind=SCALAR, weight=[2.0])
    c.add_task(LINEAR, base=M.WORLD, target=nb - 4, frame=2, wkind=HARD)
    Wm = rng.normal(size=(3, 3))
    c.add_task(POINT, base=1, target=nb - 5, point=[0.1, -0.2, 0.3], wkind=MATRIX, weight=Wm @ Wm.T + 0.1 * np.eye(3))
    c.add_task(JOINT, joint=nb - 1, wkind=HARD)
    c.add_task(JOINT, joint=nb - 2, wkind=SCALAR, weight=[3.0])
    c.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR_PER_QP)
    c.finalize()
    q, v = _states(mech, B, rng, 0.3)
    nc = 4
    des = rng.normal(size=(B, 3 + 3 + 3 + 1 + 1 + 3))
    tw = rng.uniform(0.5, 2.0, size=(B, 1))
    cw = np.full((B, nc), 1e-2)
    cmax = np.full((B, nc), 1e4)
    cmax[1::3, 0] = 0.0
    return Case("mixed_tasks" if floating else "mixed_tasks_fixed", mech, c, q, v, des, cw, cmax, tw=tw, floating=fl)


def deep_chain(factory, B=4, depth=22, seed=3):
    """a serial chain deeper than the generic kernel's ancestor bound (QPC_GENERIC_MAXD = 16: with 32 hipcc -O3 miscompiles the
    ancestor sums of the ahead-of-time kernel on gfx950; the hiprtc-specialised kernel has no such bound)"""
    rng = np.random.default_rng(seed)
    mech = M.rand_tree([M.REVOLUTE] * depth, seed=seed, chain=True)
    c = factory(mech, 4, -1)
    for j in range(mech.nbodies):
        c.regularize(j, 0.1)
    c.add_task(JOINT, joint=depth - 1, wkind=HARD)
    c.add_task(POINT, base=M.WORLD, target=depth - 2, point=[0.1, 0.0, -0.2], wkind=SCALAR, weight=[2.0])
    c.finalize()
    q, v = _states(mech, B, rng, 0.3)
    return Case("deep_chain", mech, c, q, v, rng.normal(size=(B, 4)), np.zeros((B, 0)), np.zeros((B, 0)))


def atlas_config(factory, config, B):
    """BASELINE configs 2, 3, 5 (4 = config 3 with a larger batch)."""
    ctrl, wl = W.atlas_workload(factory, B, config)
    return Case(f"atlas_config{config}", wl.mech, ctrl, wl.q, wl.v, wl.desired, wl.contact_weight,
                wl.contact_maxnormalforce, floating=wl.floating_body)


def all_cases(factory, small=True, only=None):
    """every scenario (built lazily; `only` = a name or a collection of names restricts the set)"""
    B = 6 if small else 32
    makers = [("fixed_base_hard", lambda: fixed_base_joint_space(factory, True)),
              ("fixed_base_soft", lambda: fixed_base_joint_space(factory, False)),
              ("acrobot_config1", lambda: acrobot_config1(factory)),
              ("acrobot_point", lambda: acrobot_point_task(factory)),
              ("free_fall", lambda: free_fall(factory)),
              ("momentum_rate", lambda: momentum_rate(factory))]
    for mode in ("constraint", "scalar", "scalar_param", "matrix", "matrix_param"):
        makers.append(("spatial_" + mode, lambda mode=mode: spatial_acceleration(factory, mode)))
    makers.append(("mixed_tasks", lambda: mixed_tasks(factory)))
    makers.append(("mixed_tasks_fixed", lambda: mixed_tasks(factory, floating=False, seed=78)))
    for cfg in (2, 3, 5):
        makers.append((f"atlas_config{cfg}", lambda cfg=cfg: atlas_config(factory, cfg, B)))
    if isinstance(only, str):
        only = (only,)
    for name, mk in makers:
        if only is None or name in only:
            c = mk()
            assert c.name == name, (c.name, name)
            yield c


def run_backend(case: Case, tick_fn):
    """tick_fn(ctrl, q, v, desired, cweight, cmaxf, tw, surf) -> dict(tau, vdot, cwrench, status)"""
    out = []
    for b in range(case.B):
        out.append(tick_fn(case.ctrl, case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b],
                           None if case.tw is None else case.tw[b],
                           None if case.surf is None else case.surf[b]))
    return out
