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
