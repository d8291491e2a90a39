"""Synthetic workloads of BASELINE.json (configs 1-5) and the host-side caller of the hot path.

`standing_setup` registers exactly the task set of the reference's `StandingController`
constructor (src/highlevel/standing.jl:18-56) on ANY controller object exposing
`add_task / regularize / add_contact` (the product's ctypes binding and the test oracle share
that surface), and `standing_desired` evaluates its per-tick PD laws (standing.jl:58-89) with a
small batched numpy forward-kinematics so that the `desired` inputs of qpc_control_batch are the
ones the reference would set through `setdesired!`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import models as M

# task kinds / weight kinds (include/qpc.h)
SPATIAL, ANGULAR, LINEAR, POINT, JOINT, MOMENTUM_RATE, LINEAR_MOMENTUM_RATE = range(7)
HARD, SCALAR, MATRIX, SCALAR_PER_QP, MATRIX_PER_QP = range(5)


# ------------------------------------------------------------------------------------------
# batched numpy kinematics (host side; only what the PD laws need)
# ------------------------------------------------------------------------------------------
def _quat_R(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _axis_R(a, th):
    c, s = np.cos(th)[:, None, None], np.sin(th)[:, None, None]
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)


def batch_kinematics(mech: M.Mechanism, q, v):
    """Body transforms (R, p), world twists (w, vlin at world origin), CoM and linear momentum for a
    batch of states q[B,nq], v[B,nv]."""
    q = np.atleast_2d(q)
    v = np.atleast_2d(v)
    B = q.shape[0]
    nb = mech.nbodies
    R = [None] * nb
    p = [None] * nb
    w = [None] * nb
    vl = [None] * nb
    qo = vo = 0
    for i in range(nb):
        t = mech.joint_type[i]
        Rf, pf, a = mech.joint_R[i], mech.joint_p[i], mech.joint_axis[i]
        if t == M.REVOLUTE:
            Rj = _axis_R(a, q[:, qo]); pj = np.zeros((B, 3))
        elif t == M.PRISMATIC:
            Rj = np.broadcast_to(np.eye(3), (B, 3, 3)); pj = q[:, qo, None] * a[None, :]
        elif t == M.QUAT_FLOATING:
            Rj = _quat_R(q[:, qo:qo + 4]); pj = q[:, qo + 4:qo + 7]
        else:
            Rj = np.broadcast_to(np.eye(3), (B, 3, 3)); pj = np.zeros((B, 3))
        Rl = Rf[None] @ Rj
        pl = pj @ Rf.T + pf[None]
        par = mech.parent[i]
        if par < 0:
            R[i], p[i] = Rl, pl
            w0 = np.zeros((B, 3)); v0 = np.zeros((B, 3))
        else:
            R[i] = R[par] @ Rl
            p[i] = np.einsum("bij,bj->bi", R[par], pl) + p[par]
            w0, v0 = w[par], vl[par]
        # joint twist in world
        if t == M.REVOLUTE:
            wj = np.einsum("bij,j->bi", R[i], a) * v[:, vo, None]
            vj = np.cross(p[i], wj)
        elif t == M.PRISMATIC:
            wj = np.zeros((B, 3)); vj = np.einsum("bij,j->bi", R[i], a) * v[:, vo, None]
        elif t == M.QUAT_FLOATING:
            wj = np.einsum("bij,bj->bi", R[i], v[:, vo:vo + 3])
            vj = np.cross(p[i], wj) + np.einsum("bij,bj->bi", R[i], v[:, vo + 3:vo + 6])
        else:
            wj = np.zeros((B, 3)); vj = np.zeros((B, 3))
        w[i] = w0 + wj
        vl[i] = v0 + vj
        qo += M._NQ[t]; vo += M._NV[t]
    mtot = mech.total_mass()
    com = np.zeros((B, 3)); lin = np.zeros((B, 3))
    for i in range(nb):
        cw = np.einsum("bij,j->bi", R[i], mech.com[i]) + p[i]
        com += mech.mass[i] * cw
        lin += mech.mass[i] * (vl[i] + np.cross(w[i], cw))
    com /= mtot
    return dict(R=R, p=p, w=w, v=vl, com=com, linear_momentum=lin, mass=mtot)


def rotation_vector(R):
    """angle * axis of rotation matrices R[B,3,3] (AngleAxis as used by RBD.PDControl.pd)."""
    tr = np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1)
    th = np.arccos(tr)
    ax = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    s = 2 * np.sin(th)
    small = s < 1e-9
    scale = np.where(small, 0.5, th / np.where(small, 1.0, s))
    return ax * scale[:, None]


# ------------------------------------------------------------------------------------------
@dataclass
class StandingSetup:
    """What `StandingController(lowlevel, feet, pelvis, nominalstate)` builds (standing.jl:18-56)."""
    mech: M.Mechanism
    feet: list
    pelvis: int
    foot_tasks: list = field(default_factory=list)
    linmom_task: int = -1
    pelvis_task: int = -1
    joint_tasks: dict = field(default_factory=dict)  # joint -> task id
    n_desired: int = 0
    offsets: dict = field(default_factory=dict)  # task id -> offset in `desired`
    comref: np.ndarray = None
    jointrefs: dict = field(default_factory=dict)
    comgains: tuple = (10.0, 2 * math.sqrt(10.0))
    pelvisgains: tuple = (20.0, 2 * math.sqrt(20.0))
    jointgains: tuple = (100.0, 20.0)
    contact_ids: list = field(default_factory=list)
    linmom_dim: int = 3  # 6: a full MomentumRateTask (angular rows first, src/tasks.jl:204-221)


def standing_setup(ctrl, mech: M.Mechanism, nominal_q, contacts=True, joint_regularization=0.05,
                   linear_momentum_weight=1.0, feet=("l_foot", "r_foot"), pelvis="pelvis", fused_laws=False,
                   momentum_weight_matrix=None):
    """Register the standing task set on `ctrl` in the reference's order: contacts first
    (notebook cell 5), then regularize!, foot tasks, linear momentum (soft), pelvis, joints.
    fused_laws: hand the PD laws of standing.jl:58-89 to the controller (`set_task_law`) so that
    the device evaluates them per QP and no `desired` record is needed.
    momentum_weight_matrix: a 6 x 6 W -- the momentum task becomes a full `MomentumRateTask` added with
    `addtask!(controller, task, W::AbstractMatrix)` (momentum.jl:113-117: e'We on six slack variables; test/controller.jl:232-285
    modes MATRIX_WEIGHT*), desired angular part 0, linear part as the linear task's."""
    feet_idx = [mech.findbody(f) for f in feet]
    st = StandingSetup(mech, feet_idx, mech.findbody(pelvis))
    if contacts:
        for cp in mech.contact_points:
            st.contact_ids.append(ctrl.add_contact(cp.body, cp.position, cp.normal, cp.mu))
    off = 0

    def reg(tid, dim):
        nonlocal off
        st.offsets[tid] = off
        off += dim

    for j in range(mech.nbodies):  # regularize!.(Ref(lowlevel), tree_joints(mechanism), w) standing.jl:35
        if mech.joint_type[j] != M.FIXED:
            ctrl.regularize(j, joint_regularization)
    for f in feet_idx:  # standing.jl:37-38
        tid = ctrl.add_task(SPATIAL, base=M.WORLD, target=f, frame=M.FRAME_DEFAULT, wkind=HARD)
        st.foot_tasks.append(tid); reg(tid, 6)
    if momentum_weight_matrix is None:
        st.linmom_task = ctrl.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[linear_momentum_weight])
        reg(st.linmom_task, 3)  # standing.jl:40-41
    else:
        assert not fused_laws
        st.linmom_task = ctrl.add_task(MOMENTUM_RATE, wkind=MATRIX, weight=np.asarray(momentum_weight_matrix, dtype=float))
        st.linmom_dim = 6
        reg(st.linmom_task, 6)
    st.pelvis_task = ctrl.add_task(ANGULAR, base=M.WORLD, target=st.pelvis, frame=M.FRAME_DEFAULT, wkind=HARD)
    reg(st.pelvis_task, 3)  # standing.jl:43-44
    on_foot_path = set()
    for f in feet_idx:
        on_foot_path.update(mech.path_joints(M.WORLD, f))
    for j in range(mech.nbodies):  # standing.jl:46-49
        if mech.joint_type[j] == M.REVOLUTE and j not in on_foot_path:
            tid = ctrl.add_task(JOINT, joint=j, wkind=HARD)
            st.joint_tasks[j] = tid; reg(tid, 1)
    st.n_desired = off
    kin = batch_kinematics(mech, nominal_q, np.zeros(mech.nv))
    st.comref = kin["com"][0] - np.array([0.0, 0.0, 0.05])  # standing.jl:29
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE:
            st.jointrefs[j] = float(nominal_q[mech.configuration_range(j)[0]])
    if fused_laws:
        LAW_CONSTANT, LAW_JOINT_PD, LAW_COM_PD, LAW_ORIENTATION_PD = 1, 2, 3, 4  # include/qpc.h QPC_LAW_*
        for tid in st.foot_tasks:
            ctrl.set_task_law(tid, LAW_CONSTANT, ref=np.zeros(6))
        ctrl.set_task_law(st.linmom_task, LAW_COM_PD, st.comgains[0], st.comgains[1], st.comref)
        ctrl.set_task_law(st.pelvis_task, LAW_ORIENTATION_PD, st.pelvisgains[0], st.pelvisgains[1])
        for j, tid in st.joint_tasks.items():
            ctrl.set_task_law(tid, LAW_JOINT_PD, st.jointgains[0], st.jointgains[1], [st.jointrefs[j]])
    return st


def standing_desired(st: StandingSetup, q, v):
    """The PD laws of (controller::StandingController)(tau, t, state), standing.jl:58-89,
    for a batch of states -> desired[B, n_desired]."""
    mech = st.mech
    q = np.atleast_2d(q); v = np.atleast_2d(v)
    B = q.shape[0]
    kin = batch_kinematics(mech, q, v)
    des = np.zeros((B, st.n_desired))
    m = kin["mass"]
    e = kin["com"] - st.comref[None]
    edot = kin["linear_momentum"] / m
    cdd = -st.comgains[0] * e - st.comgains[1] * edot
    o = st.offsets[st.linmom_task] + (st.linmom_dim - 3)
    des[:, o:o + 3] = m * cdd
    Rp = kin["R"][st.pelvis]
    w_body = np.einsum("bji,bj->bi", Rp, kin["w"][st.pelvis])
    o = st.offsets[st.pelvis_task]
    des[:, o:o + 3] = -st.pelvisgains[0] * rotation_vector(Rp) - st.pelvisgains[1] * w_body
    for j, tid in st.joint_tasks.items():
        qi = q[:, mech.configuration_range(j)[0]]
        vi = v[:, mech.velocity_range(j)[0]]
        des[:, st.offsets[tid]] = -st.jointgains[0] * (qi - st.jointrefs[j]) - st.jointgains[1] * vi
    return des  # foot task desireds stay 0


def standing_states(mech: M.Mechanism, B, seed, dq=0.1, drot=0.05, vstd=0.1):
    """Perturbed nominal Atlas states (SURVEY.md 8d, config 2/3 definition)."""
    rng = np.random.default_rng(seed)
    q0 = M.atlas_nominal_configuration(mech)
    q = np.tile(q0, (B, 1))
    for j in range(mech.nbodies):
        r = mech.configuration_range(j)
        if mech.joint_type[j] == M.REVOLUTE:
            q[:, r[0]] += rng.uniform(-dq, dq, size=B)
        elif mech.joint_type[j] == M.QUAT_FLOATING:
            rv = rng.uniform(-drot, drot, size=(B, 3))
            th = np.linalg.norm(rv, axis=1)
            ax = rv / np.maximum(th, 1e-300)[:, None]
            q[:, r[0]] = np.cos(th / 2)
            q[:, r[0] + 1:r[0] + 4] = ax * np.sin(th / 2)[:, None]
    v = rng.normal(size=(B, mech.nv)) * vstd
    return q, v


@dataclass
class Workload:
    name: str
    mech: M.Mechanism
    setup: object
    q: np.ndarray
    v: np.ndarray
    desired: np.ndarray
    contact_weight: np.ndarray
    contact_maxnormalforce: np.ndarray
    floating_body: int
    n_basis: int = 4


def atlas_workload(ctrl_factory, B, config=3, seed=None, fused_laws=False, momentum_weight_matrix=None):
    """BASELINE.json configs 2-5 on the synthetic Atlas.  `ctrl_factory(mech, n_basis, floating_body)`
    returns a controller object (product binding or oracle) which is populated and finalized."""
    mech = M.atlas()
    seed = {2: 2024, 3: 2025, 4: 2026, 5: 2027}[config] if seed is None else seed
    contacts = config != 2
    floating = mech.findbody("pelvis") if contacts else M.WORLD
    ctrl = ctrl_factory(mech, 4, floating)
    q0 = M.atlas_nominal_configuration(mech)
    st = standing_setup(ctrl, mech, q0, contacts=contacts, fused_laws=fused_laws, momentum_weight_matrix=momentum_weight_matrix)
    ctrl.finalize()
    q, v = standing_states(mech, B, seed)
    des = standing_desired(st, q, v)
    nc = len(st.contact_ids)
    cw = np.full((B, nc), 1e-3)
    cmax = np.full((B, nc), 1e6)
    if config == 5:  # mixed active contact sets: stratified quarters p in {.25,.5,.75,1}
        rng = np.random.default_rng(seed + 1)
        pq = np.repeat([0.25, 0.5, 0.75, 1.0], (B + 3) // 4)[:B]
        active = rng.uniform(size=(B, nc)) < pq[:, None]
        cmax = np.where(active, 1e6, 0.0)  # disable!(contact): maxnormalforce := 0 (contacts.jl:72)
    return ctrl, Workload(f"atlas_config{config}", mech, st, q, v, des, cw, cmax, floating)


# ------------------------------------------------------------------------------------------
# multi-GPU: the batch shards embarrassingly, no collective in the data path
# ------------------------------------------------------------------------------------------
def shard_bounds(B_global, rank, world):
    """Contiguous shard [lo, hi) of QP indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(B_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_sharded(ctrl, wl: Workload, rank, world):
    """Solve this rank's shard of a global workload with its own controller replica."""
    lo, hi = shard_bounds(wl.q.shape[0], rank, world)
    r = ctrl.control_batch(wl.q[lo:hi], wl.v[lo:hi], wl.desired[lo:hi], wl.contact_weight[lo:hi],
                           wl.contact_maxnormalforce[lo:hi])
    return lo, hi, r
