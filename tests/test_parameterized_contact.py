"""Mirror of the reference's "parameterized contacts" test (test/controller.jl:22-66): the contact
normal is a Parameter that re-expresses a direction FIXED IN THE WORLD in the (rotating) body frame,
so however the body is oriented the contact-aligned frame stays aligned with world z and the linear
part of the contact wrench in world coordinates is along [0, 0, 1].

Here: a single floating body (centre of mass and contact point at its origin) held still by a hard
LinearAccelerationTask; the per-QP surface input carries normal = R' z.  For every orientation the
contact wrench must be m g along world z and tau must vanish."""
import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd.workloads import ANGULAR, HARD, LINEAR, SCALAR


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def run(factory, B=24):
    mech = M.Mechanism()
    mech.add_body("body", M.WORLD, "floating", M.QUAT_FLOATING, mass=10.0, com=(0.0, 0.0, 0.0), inertia=(1.0, 1.0, 1.0, 0.0, 0.0, 0.0))
    c = factory(mech, 4, 0)
    c.add_contact(0, [0.0, 0.0, 0.0], [0.0, 0.0, 1.0], 1.0, per_qp_surface=1)
    c.regularize(0, 1e-3)
    c.add_task(LINEAR, base=M.WORLD, target=0, frame=M.WORLD, wkind=HARD)       # the body does not accelerate
    c.add_task(ANGULAR, base=M.WORLD, target=0, frame=M.WORLD, wkind=SCALAR, weight=[1.0])
    c.finalize()
    rng = np.random.default_rng(0)
    q = np.zeros((B, 7))
    quat = rng.normal(size=(B, 4))
    quat[0] = [1, 0, 0, 0]
    quat[1] = [0, 1, 0, 0]  # upside down
    q[:, :4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q[:, 4:] = rng.normal(size=(B, 3))
    v = np.zeros((B, 6))
    surf = np.zeros((B, 4))
    for b in range(B):
        surf[b, :3] = quat_to_R(q[b, :4]).T @ np.array([0.0, 0.0, 1.0])  # world z in body coordinates
        surf[b, 3] = 1.0
    r = c.control_batch(q, v, np.zeros((B, 6)), np.full((B, 1), 1e-3), np.full((B, 1), 1e3), None, surf)
    assert np.all(r["status"] == 0), r["status"]
    f = r["cwrench"][:, 0, 3:]
    np.testing.assert_allclose(f, np.tile([0.0, 0.0, 10.0 * 9.81], (B, 1)), atol=1e-7)
    np.testing.assert_allclose(r["tau"], 0.0, atol=1e-9)  # floating joint: zero_floating_joint_torques!
    np.testing.assert_allclose(r["vdot"][:, 3:], 0.0, atol=1e-9)
    # and with the normal NOT updated (fixed in the body) a tilted body cannot be held: the friction
    # cone (mu = 1: 45 degrees) no longer contains the vertical for tilts beyond it
    surf_fixed = np.tile([0.0, 0.0, 1.0, 1.0], (B, 1))
    r2 = c.control_batch(q, v, np.zeros((B, 6)), np.full((B, 1), 1e-3), np.full((B, 1), 1e3), None, surf_fixed)
    tilt = np.array([np.arccos(np.clip(quat_to_R(q[b, :4])[2, 2], -1, 1)) for b in range(B)])
    assert np.all(r2["status"][tilt > np.pi / 4 + 0.05] != 0)
    assert np.all(r2["status"][tilt < np.pi / 4 - 0.3] == 0)


def test_parameterized_contact_emulated():
    emu.build()
    run(emu.factory)


@pytest.mark.gpu
def test_parameterized_contact_gpu():
    from qpcontrol_jl_amd import binding
    run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl))


# ---- per-QP contact POSITION (contacts.jl:39: `position::Union{Point3D, Parameter{<:Point3D}}`) ------------------
def _position_case(factory, flags, B=6, seed=21):
    """a floating tree with four contacts on two bodies; flags = QPC_CONTACT_*_PER_QP bits of every contact"""
    import scenarios as SC
    from qpcontrol_jl_amd.workloads import LINEAR_MOMENTUM_RATE
    rng = np.random.default_rng(seed)
    mech = M.rand_tree([M.QUAT_FLOATING] + [M.REVOLUTE] * 6, seed=seed)
    nb = mech.nbodies
    c = factory(mech, 4, 0)
    pos0 = [[-0.1, 0.05, -0.1], [0.1, 0.05, -0.1], [-0.1, -0.05, -0.12], [0.1, -0.05, -0.12]]
    bodies = [nb - 1, nb - 1, nb - 3, nb - 3]
    for b, p in zip(bodies, pos0):
        c.add_contact(b, p, [0.05, 0.0, 1.0], 0.7, per_qp_surface=flags)
    for j in range(nb):
        c.regularize(j, 0.05)
    c.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0])
    c.add_task(ANGULAR, base=M.WORLD, target=0, frame=M.WORLD, wkind=SCALAR, weight=[1.0])
    c.finalize()
    qs, vs = zip(*(M.rand_state(mech, rng, 0.3) for _ in range(B)))
    # per-QP records: [nx ny nz mu] (bit 1) then [px py pz] (bit 2), contacts concatenated
    recs = []
    pos = np.array(pos0)[None] + 0.05 * rng.normal(size=(B, 4, 3))
    nrm = np.array([0.05, 0.0, 1.0])[None, None] + 0.1 * rng.normal(size=(B, 4, 3))
    mu = rng.uniform(0.4, 0.9, size=(B, 4, 1))
    for ci in range(4):
        if flags & 1:
            recs += [nrm[:, ci], mu[:, ci]]
        if flags & 2:
            recs.append(pos[:, ci])
    surf = np.concatenate(recs, axis=1) if recs else None
    return SC.Case(f"contact_position_{flags}", mech, c, np.array(qs), np.array(vs), rng.normal(size=(B, 6)), np.full((B, 4), 1e-2),
                   np.full((B, 4), 1e4), surf=surf, floating=0), pos, nrm, mu, bodies


def _run_position(factory):
    import parity
    from oracle import oracle as O
    ora = lambda mech, N, fl: O.OracleController(O.OracleModel(mech), N, fl)  # noqa: E731
    for flags in (2, 3):
        g, pos, nrm, mu, bodies = _position_case(factory, flags)
        o, _, _, _, _ = _position_case(ora, flags)
        assert g.ctrl.layout.n_surface == 4 * (4 * (flags & 1) + 3 * ((flags >> 1) & 1))
        rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        assert (ro["status"] == 0).all()
        parity.compare(g, rg, o, ro)
        # and it equals a controller whose contacts were CONSTRUCTED at those positions / surfaces (one per QP)
        for b in range(2):
            c1 = factory(g.mech, 4, 0)
            from qpcontrol_jl_amd.workloads import LINEAR_MOMENTUM_RATE
            for ci in range(4):
                c1.add_contact(bodies[ci], pos[b, ci], nrm[b, ci] if flags & 1 else [0.05, 0.0, 1.0], float(mu[b, ci, 0]) if flags & 1 else 0.7)
            for j in range(g.mech.nbodies):
                c1.regularize(j, 0.05)
            c1.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0])
            c1.add_task(ANGULAR, base=M.WORLD, target=0, frame=M.WORLD, wkind=SCALAR, weight=[1.0])
            c1.finalize()
            r1 = c1.control_batch(g.q[b:b + 1], g.v[b:b + 1], g.desired[b:b + 1], g.cweight[b:b + 1], g.cmaxf[b:b + 1])
            np.testing.assert_allclose(r1["tau"][0], rg["tau"][b], atol=1e-9, rtol=0)
            np.testing.assert_allclose(r1["cwrench"][0], rg["cwrench"][b], atol=1e-8, rtol=0)


def test_per_qp_contact_position_emulated():
    emu.build()
    _run_position(emu.factory)


@pytest.mark.gpu
def test_per_qp_contact_position_gpu():
    from qpcontrol_jl_amd import binding
    _run_position(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl))
