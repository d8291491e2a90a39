"""Closed-loop harness (test infrastructure; SURVEY.md 8 f-2 in miniature): B independent Atlas
simulations driven by ONE batched controller tick per step, as the reference's notebook does for a
single robot (notebooks/Standing controller.ipynb cells 9-11: RigidBodySim + soft contacts).

Dynamics on the host with the oracle's forward dynamics; ground contact = a compliant half space at
the foot contact points (normal spring/damper, viscous tangential friction capped by Coulomb's law);
semi-implicit Euler at the control rate.  Nothing here is part of the product."""
import numpy as np

from oracle import oracle as O


class GroundContact:
    def __init__(self, z_ground, k=6.0e4, d=400.0, ct=300.0):  # explicit Euler at 1 kHz: d dt / m_foot well below 1
        self.z, self.k, self.d, self.ct = z_ground, k, d, ct

    def wrenches(self, om, mech, contact_points=None):
        """external wrench per body, world frame about the world origin [ang; lin]"""
        w = np.zeros((mech.nbodies, 6))
        for cp in (mech.contact_points if contact_points is None else contact_points):
            R, t = om.transform(cp.body)
            p = R @ cp.position + t
            pen = self.z - p[2]
            if pen <= 0.0:
                continue
            T = om.twist(cp.body)
            vel = T[3:] + np.cross(T[:3], p)
            fn = max(self.k * pen - self.d * vel[2], 0.0)
            ft = -self.ct * vel[:2]
            cap = cp.mu * fn
            n = np.linalg.norm(ft)
            if n > cap:
                ft *= cap / max(n, 1e-300)
            f = np.array([ft[0], ft[1], fn])
            w[cp.body, :3] += np.cross(p, f)
            w[cp.body, 3:] += f
        return w


def step(om, mech, ground, q, v, tau, dt, contact_points=None):
    """one reference step for one robot: oracle forward dynamics under the ground wrenches, semi-implicit
    Euler, oracle manifold integration -- what qpc_sim_step_batch is checked against"""
    om.set_state(q, v)
    wext = ground.wrenches(om, mech, contact_points) if ground is not None else None
    vd = om.forward_dynamics(tau, wext)
    vn = v + dt * vd
    return om.integrate(q, vn, dt), vn


def rollout(ctrl_tick, mech, q0, v0, steps, dt=1e-3, ground=None, record_every=50):
    """ctrl_tick(q[B,nq], v[B,nv]) -> dict(tau, status).  Returns dict of trajectories / final state."""
    om = O.OracleModel(mech)
    q, v = np.array(q0, dtype=float), np.array(v0, dtype=float)
    B = q.shape[0]
    pelvis = mech.findbody("pelvis")
    log = dict(t=[], com=[], pelvis_z=[], vnorm=[], bad=0)
    for k in range(steps):
        r = ctrl_tick(q, v)
        log["bad"] += int((r["status"] > 1).sum())
        for b in range(B):
            om.set_state(q[b], v[b])
            wext = ground.wrenches(om, mech) if ground is not None else None
            tau = r["tau"][b] if r["status"][b] <= 1 else np.zeros(mech.nv)
            vd = om.forward_dynamics(tau, wext)
            v[b] = v[b] + dt * vd
            q[b] = om.integrate(q[b], v[b], dt)
        if k % record_every == 0 or k == steps - 1:
            coms, pz = [], []
            for b in range(B):
                om.set_state(q[b], v[b])
                coms.append(om.center_of_mass())
                pz.append(om.transform(pelvis)[1][2])
            log["t"].append((k + 1) * dt); log["com"].append(np.array(coms)); log["pelvis_z"].append(np.array(pz))
            log["vnorm"].append(np.abs(v).max(axis=1))
    log["q"], log["v"] = q, v
    return log
