"""Flat tree-mechanism descriptions and the synthetic models used by the tests and bench.

`Mechanism` mirrors what the reference gets from RigidBodyDynamics' `Mechanism{Float64}` after
`remove_fixed_tree_joints!` (notebooks/Standing controller.ipynb cell 2): a topologically ordered
tree of bodies, one joint per body.  It is plain data -- the C ABI consumes it through
`qpc_model_desc` (include/qpc.h).

The robot assets the reference uses (AtlasRobot 0.4.0, ValkyrieRobot 0.2.0, RBD's Acrobot.urdf)
are not vendored in the reference tree and are absent from this image, so `atlas()` and
`acrobot()` are DECLARED SYNTHETIC models with the same topology, joint names, dimensions
(Atlas: 31 moving bodies, nq 37, nv 36) and contact layout.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field

import numpy as np

WORLD = -1
FRAME_DEFAULT = -2

REVOLUTE, PRISMATIC, QUAT_FLOATING, FIXED = 0, 1, 2, 3
_NQ = {REVOLUTE: 1, PRISMATIC: 1, QUAT_FLOATING: 7, FIXED: 0}
_NV = {REVOLUTE: 1, PRISMATIC: 1, QUAT_FLOATING: 6, FIXED: 0}


class ModelDesc(ctypes.Structure):
    """ctypes image of `qpc_model_desc` (include/qpc.h)."""

    _fields_ = [
        ("nbodies", ctypes.c_int32),
        ("parent", ctypes.POINTER(ctypes.c_int32)),
        ("joint_type", ctypes.POINTER(ctypes.c_int32)),
        ("joint_axis", ctypes.POINTER(ctypes.c_double)),
        ("joint_R", ctypes.POINTER(ctypes.c_double)),
        ("joint_p", ctypes.POINTER(ctypes.c_double)),
        ("mass", ctypes.POINTER(ctypes.c_double)),
        ("com", ctypes.POINTER(ctypes.c_double)),
        ("inertia", ctypes.POINTER(ctypes.c_double)),
        ("gravity", ctypes.c_double * 3),
    ]


@dataclass
class ContactSpec:
    body: int
    position: np.ndarray
    normal: np.ndarray
    mu: float


@dataclass
class Mechanism:
    body_names: list = field(default_factory=list)
    joint_names: list = field(default_factory=list)
    parent: list = field(default_factory=list)
    joint_type: list = field(default_factory=list)
    joint_axis: list = field(default_factory=list)
    joint_R: list = field(default_factory=list)
    joint_p: list = field(default_factory=list)
    mass: list = field(default_factory=list)
    com: list = field(default_factory=list)
    inertia: list = field(default_factory=list)  # [Ixx Iyy Izz Ixy Ixz Iyz] about CoM
    gravity: tuple = (0.0, 0.0, -9.81)
    contact_points: list = field(default_factory=list)  # ContactSpec (RBD.contact_points(body))

    # ---- construction -----------------------------------------------------------------
    def add_body(self, name, parent, joint_name, joint_type, axis=(0, 0, 1), R=None, p=(0, 0, 0),
                 mass=1.0, com=(0, 0, 0), inertia=(1, 1, 1, 0, 0, 0)):
        parent_idx = parent if isinstance(parent, int) else self.findbody(parent)
        assert parent_idx < len(self.body_names)
        self.body_names.append(name)
        self.joint_names.append(joint_name)
        self.parent.append(parent_idx)
        self.joint_type.append(joint_type)
        a = np.asarray(axis, dtype=np.float64)
        n = np.linalg.norm(a)
        self.joint_axis.append(a / n if n > 0 else a)
        self.joint_R.append(np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3))
        self.joint_p.append(np.asarray(p, dtype=np.float64))
        self.mass.append(float(mass))
        self.com.append(np.asarray(com, dtype=np.float64))
        self.inertia.append(np.asarray(inertia, dtype=np.float64))
        return len(self.body_names) - 1

    # ---- queries (names follow RigidBodyDynamics) ---------------------------------------
    @property
    def nbodies(self):
        return len(self.body_names)

    def findbody(self, name):
        if name in ("world", WORLD):
            return WORLD
        return self.body_names.index(name)

    def findjoint(self, name):
        """A joint is identified by its successor body index."""
        return self.joint_names.index(name)

    @property
    def nq(self):
        return sum(_NQ[t] for t in self.joint_type)

    @property
    def nv(self):
        return sum(_NV[t] for t in self.joint_type)

    def configuration_range(self, joint):
        q0 = sum(_NQ[t] for t in self.joint_type[:joint])
        return range(q0, q0 + _NQ[self.joint_type[joint]])

    def velocity_range(self, joint):
        v0 = sum(_NV[t] for t in self.joint_type[:joint])
        return range(v0, v0 + _NV[self.joint_type[joint]])

    def total_mass(self):
        return float(sum(self.mass))

    def path_joints(self, base, target):
        """Joints on the tree path base -> target (RBD `path`), as body indices."""
        anc = set()
        b = target
        while b >= 0:
            anc.add(b)
            b = self.parent[b]
        lca = WORLD
        b = base
        while b >= 0:
            if b in anc:
                lca = b
                break
            b = self.parent[b]
        joints = []
        for start in (base, target):
            b = start
            while b >= 0 and b != lca:
                joints.append(b)
                b = self.parent[b]
        return joints

    def zero_configuration(self):
        q = np.zeros(self.nq)
        for j, t in enumerate(self.joint_type):
            if t == QUAT_FLOATING:
                q[self.configuration_range(j)[0]] = 1.0
        return q

    # ---- C ABI image ----------------------------------------------------------------------
    def desc(self):
        """Return (ModelDesc, keepalive) for qpc_model_create / ora_model_create."""
        nb = self.nbodies
        arrs = dict(
            parent=np.asarray(self.parent, dtype=np.int32),
            joint_type=np.asarray(self.joint_type, dtype=np.int32),
            joint_axis=np.ascontiguousarray(np.stack(self.joint_axis).reshape(-1)),
            joint_R=np.ascontiguousarray(np.stack(self.joint_R).reshape(-1)),
            joint_p=np.ascontiguousarray(np.stack(self.joint_p).reshape(-1)),
            mass=np.asarray(self.mass, dtype=np.float64),
            com=np.ascontiguousarray(np.stack(self.com).reshape(-1)),
            inertia=np.ascontiguousarray(np.stack(self.inertia).reshape(-1)),
        )
        d = ModelDesc()
        d.nbodies = nb
        for k, a in arrs.items():
            ptr_t = ctypes.POINTER(ctypes.c_int32 if a.dtype == np.int32 else ctypes.c_double)
            setattr(d, k, a.ctypes.data_as(ptr_t))
        d.gravity[:] = self.gravity
        return d, arrs


# ------------------------------------------------------------------------------------------
def _box_inertia(m, sx, sy, sz, skew=0.02):
    ixx = m * (sy * sy + sz * sz) / 12.0
    iyy = m * (sx * sx + sz * sz) / 12.0
    izz = m * (sx * sx + sy * sy) / 12.0
    return (ixx, iyy, izz, skew * math.sqrt(ixx * iyy), -skew * math.sqrt(ixx * izz),
            0.5 * skew * math.sqrt(iyy * izz))


def atlas():
    """Synthetic Atlas-like humanoid: topology and joint names of Atlas
    v5, plausible link lengths/masses (total ~170 kg), 4 contact points per foot, mu = 0.8."""
    m = Mechanism()
    A = m.add_body
    A("pelvis", WORLD, "pelvis_to_world", QUAT_FLOATING, mass=9.51, com=(0.0111, 0.0, 0.0271),
      inertia=_box_inertia(9.51, 0.25, 0.3, 0.18))
    A("ltorso", "pelvis", "back_bkz", REVOLUTE, (0, 0, 1), p=(-0.0125, 0, 0), mass=2.27,
      com=(-0.0112, 0, 0.0747), inertia=_box_inertia(2.27, 0.12, 0.12, 0.15))
    A("mtorso", "ltorso", "back_bky", REVOLUTE, (0, 1, 0), p=(0, 0, 0.162), mass=0.799,
      com=(-0.0072, 0, 0.0202), inertia=_box_inertia(0.799, 0.1, 0.1, 0.05))
    A("utorso", "mtorso", "back_bkx", REVOLUTE, (1, 0, 0), p=(0, 0, 0.05), mass=84.4,
      com=(-0.0622, 0.0023, 0.3157), inertia=_box_inertia(84.4, 0.35, 0.5, 0.6))
    A("head", "utorso", "neck_ry", REVOLUTE, (0, 1, 0), p=(0.2546, 0, 0.6215), mass=1.42,
      com=(-0.0754, 0, 0.0338), inertia=_box_inertia(1.42, 0.15, 0.15, 0.15))
    for s, sg in (("l", 1.0), ("r", -1.0)):
        A(f"{s}_clav", "utorso", f"{s}_arm_shz", REVOLUTE, (0, 0, 1), p=(0.1406, sg * 0.2256, 0.4776),
          mass=4.466, com=(0, sg * 0.048, 0.084), inertia=_box_inertia(4.466, 0.1, 0.15, 0.2))
        A(f"{s}_scap", f"{s}_clav", f"{s}_arm_shx", REVOLUTE, (1, 0, 0), p=(0, sg * 0.11, -0.245),
          mass=3.899, com=(0, sg * 0.075, 0.036), inertia=_box_inertia(3.899, 0.1, 0.18, 0.1))
        A(f"{s}_uarm", f"{s}_scap", f"{s}_arm_ely", REVOLUTE, (0, 1, 0), p=(0, sg * 0.187, -0.016),
          mass=4.386, com=(0, sg * 0.065, 0), inertia=_box_inertia(4.386, 0.1, 0.2, 0.1))
        A(f"{s}_larm", f"{s}_uarm", f"{s}_arm_elx", REVOLUTE, (1, 0, 0), p=(0, sg * 0.119, 0.0092),
          mass=3.248, com=(0, sg * 0.105, 0.011), inertia=_box_inertia(3.248, 0.09, 0.22, 0.09))
        A(f"{s}_ufarm", f"{s}_larm", f"{s}_arm_wry", REVOLUTE, (0, 1, 0), p=(0, sg * 0.2996, -0.0092),
          mass=2.4798, com=(0.0002, sg * 0.027, 0.0002), inertia=_box_inertia(2.4798, 0.08, 0.12, 0.08))
        A(f"{s}_lfarm", f"{s}_ufarm", f"{s}_arm_wrx", REVOLUTE, (1, 0, 0), p=(0, 0, 0),
          mass=0.648, com=(0, sg * 0.01, 0), inertia=_box_inertia(0.648, 0.06, 0.06, 0.06))
        A(f"{s}_hand", f"{s}_lfarm", f"{s}_arm_wry2", REVOLUTE, (0, 1, 0), p=(0, sg * 0.05, 0),
          mass=0.5839, com=(0, sg * 0.03, 0), inertia=_box_inertia(0.5839, 0.08, 0.1, 0.05))
    for s, sg in (("l", 1.0), ("r", -1.0)):
        A(f"{s}_uglut", "pelvis", f"{s}_leg_hpz", REVOLUTE, (0, 0, 1), p=(0, sg * 0.089, 0),
          mass=1.959, com=(0.0053, sg * -0.0034, 0.0313), inertia=_box_inertia(1.959, 0.1, 0.1, 0.1))
        A(f"{s}_lglut", f"{s}_uglut", f"{s}_leg_hpx", REVOLUTE, (1, 0, 0), p=(0, 0, 0),
          mass=2.338, com=(0.0133, sg * 0.017, -0.0312), inertia=_box_inertia(2.338, 0.12, 0.1, 0.12))
        A(f"{s}_uleg", f"{s}_lglut", f"{s}_leg_hpy", REVOLUTE, (0, 1, 0), p=(0.05, sg * 0.0225, -0.066),
          mass=8.204, com=(0, 0, -0.21), inertia=_box_inertia(8.204, 0.15, 0.15, 0.4))
        A(f"{s}_lleg", f"{s}_uleg", f"{s}_leg_kny", REVOLUTE, (0, 1, 0), p=(-0.05, 0, -0.374),
          mass=4.515, com=(0.001, 0, -0.187), inertia=_box_inertia(4.515, 0.1, 0.1, 0.42))
        A(f"{s}_talus", f"{s}_lleg", f"{s}_leg_aky", REVOLUTE, (0, 1, 0), p=(0, 0, -0.422),
          mass=0.125, com=(0, 0, 0), inertia=_box_inertia(0.125, 0.04, 0.04, 0.04))
        A(f"{s}_foot", f"{s}_talus", f"{s}_leg_akx", REVOLUTE, (1, 0, 0), p=(0, 0, 0),
          mass=2.41, com=(0.027, 0, -0.067), inertia=_box_inertia(2.41, 0.26, 0.13, 0.05))
    for s in ("l", "r"):
        foot = m.findbody(f"{s}_foot")
        for x, y in ((-0.0876, 0.066), (-0.0876, -0.066), (0.1728, 0.0626), (0.1728, -0.0626)):
            m.contact_points.append(ContactSpec(foot, np.array([x, y, -0.07645]),
                                                np.array([0.0, 0.0, 1.0]), 0.8))
    assert m.nbodies == 31 and m.nq == 37 and m.nv == 36
    return m


def valkyrie():
    """Synthetic Valkyrie-sized humanoid (the robot of the reference's own integration tests, /root/reference/test/controller.jl:99-126:
    `ValkyrieRobot.Valkyrie()`; its URDF is not in the tree): the topology of NASA's Valkyrie with fingers -- pelvis, three torso
    joints, three neck joints, two seven-joint arms with thirteen finger joints each, two six-joint legs -- 59 bodies, nv = 64 =
    the kernel's limit, plausible link lengths and masses (~125 kg), 4 contact points per foot, mu = 0.8.  Names follow the robot's."""
    m = Mechanism()
    A = m.add_body
    A("pelvis", WORLD, "pelvis_to_world", QUAT_FLOATING, mass=8.22, com=(-0.005, 0.0, -0.016), inertia=_box_inertia(8.22, 0.25, 0.3, 0.2))
    A("torsoYawLink", "pelvis", "torsoYaw", REVOLUTE, (0, 0, 1), p=(0, 0, 0), mass=0.5, com=(0, 0, -0.01), inertia=_box_inertia(0.5, 0.08, 0.08, 0.06))
    A("torsoPitchLink", "torsoYawLink", "torsoPitch", REVOLUTE, (0, 1, 0), p=(0.04, 0, 0), mass=0.1, com=(0, 0, 0.005), inertia=_box_inertia(0.1, 0.04, 0.04, 0.04))
    A("torso", "torsoPitchLink", "torsoRoll", REVOLUTE, (1, 0, 0), p=(0, 0, 0.02), mass=39.47, com=(-0.095, 0, 0.183), inertia=_box_inertia(39.47, 0.3, 0.45, 0.5))
    A("lowerNeckPitchLink", "torso", "lowerNeckPitch", REVOLUTE, (0, 1, 0), p=(0.02, 0, 0.34), mass=1.05, com=(-0.02, 0, 0.04), inertia=_box_inertia(1.05, 0.08, 0.08, 0.1))
    A("neckYawLink", "lowerNeckPitchLink", "neckYaw", REVOLUTE, (0, 0, 1), p=(-0.05, 0, 0.2), mass=1.4, com=(-0.02, 0, 0.04), inertia=_box_inertia(1.4, 0.08, 0.08, 0.1))
    A("head", "neckYawLink", "upperNeckPitch", REVOLUTE, (0, 1, 0), p=(-0.06, 0, 0), mass=3.89, com=(0.075, 0, 0.02), inertia=_box_inertia(3.89, 0.2, 0.18, 0.2))
    for s, sg in (("left", 1.0), ("right", -1.0)):
        A(f"{s}ShoulderPitchLink", "torso", f"{s}ShoulderPitch", REVOLUTE, (0, 1, 0), p=(-0.03, sg * 0.25, 0.25), mass=2.65, com=(0.01, sg * 0.07, 0), inertia=_box_inertia(2.65, 0.1, 0.15, 0.1))
        A(f"{s}ShoulderRollLink", f"{s}ShoulderPitchLink", f"{s}ShoulderRoll", REVOLUTE, (1, 0, 0), p=(0, sg * 0.25, 0), mass=3.97, com=(0, sg * 0.02, 0), inertia=_box_inertia(3.97, 0.12, 0.2, 0.12))
        A(f"{s}ShoulderYawLink", f"{s}ShoulderRollLink", f"{s}ShoulderYaw", REVOLUTE, (0, 1, 0), p=(0, sg * 0.06, 0), mass=3.085, com=(0, sg * 0.12, 0), inertia=_box_inertia(3.085, 0.1, 0.25, 0.1))
        A(f"{s}ElbowPitchLink", f"{s}ShoulderYawLink", f"{s}ElbowPitch", REVOLUTE, (0, 0, 1), p=(0.025, sg * 0.25, 0), mass=1.83, com=(-0.01, sg * 0.04, 0), inertia=_box_inertia(1.83, 0.09, 0.15, 0.09))
        A(f"{s}ForearmLink", f"{s}ElbowPitchLink", f"{s}ForearmYaw", REVOLUTE, (0, 1, 0), p=(-0.025, sg * 0.04, 0), mass=2.476, com=(0, sg * 0.13, 0), inertia=_box_inertia(2.476, 0.09, 0.25, 0.09))
        A(f"{s}WristRollLink", f"{s}ForearmLink", f"{s}WristRoll", REVOLUTE, (1, 0, 0), p=(0, sg * 0.28, 0), mass=0.14, com=(0, 0, 0), inertia=_box_inertia(0.14, 0.04, 0.04, 0.04))
        A(f"{s}Palm", f"{s}WristRollLink", f"{s}WristPitch", REVOLUTE, (0, 0, 1), p=(0, 0, 0), mass=0.712, com=(0, sg * 0.05, 0), inertia=_box_inertia(0.712, 0.09, 0.1, 0.03))
        A(f"{s}ThumbRollLink", f"{s}Palm", f"{s}ThumbRoll", REVOLUTE, (0, 1, 0), p=(0.005, sg * 0.03, -0.03), mass=0.017, com=(0, 0, -0.01), inertia=_box_inertia(0.017, 0.02, 0.02, 0.03))
        prev = f"{s}ThumbRollLink"
        for k in (1, 2, 3):
            A(f"{s}ThumbPitch{k}Link", prev, f"{s}ThumbPitch{k}", REVOLUTE, (1, 0, 0), p=(0, sg * 0.005, -0.03), mass=0.02 - 0.004 * k, com=(0, 0, -0.012),
              inertia=_box_inertia(0.02 - 0.004 * k, 0.015, 0.015, 0.03))
            prev = f"{s}ThumbPitch{k}Link"
        for fi, (name, fx) in enumerate((("Index", 0.03), ("Middle", 0.0), ("Pinky", -0.03))):
            prev = f"{s}Palm"
            for k in (1, 2, 3):
                A(f"{s}{name}FingerPitch{k}Link", prev, f"{s}{name}FingerPitch{k}", REVOLUTE, (1, 0, 0), p=(fx if k == 1 else 0.0, sg * (0.1 if k == 1 else 0.035), 0.0),
                  mass=0.02 - 0.004 * k, com=(0, sg * 0.012, 0), inertia=_box_inertia(0.02 - 0.004 * k, 0.015, 0.03, 0.015))
                prev = f"{s}{name}FingerPitch{k}Link"
    for s, sg in (("left", 1.0), ("right", -1.0)):
        A(f"{s}HipYawLink", "pelvis", f"{s}HipYaw", REVOLUTE, (0, 0, 1), p=(0, sg * 0.1016, -0.1853), mass=2.39, com=(0.02, 0, 0.03), inertia=_box_inertia(2.39, 0.12, 0.12, 0.1))
        A(f"{s}HipRollLink", f"{s}HipYawLink", f"{s}HipRoll", REVOLUTE, (1, 0, 0), p=(0, 0, 0), mass=3.665, com=(0.012, 0, 0), inertia=_box_inertia(3.665, 0.15, 0.12, 0.12))
        A(f"{s}HipPitchLink", f"{s}HipRollLink", f"{s}HipPitch", REVOLUTE, (0, 1, 0), p=(0, sg * 0.0609, -0.0609), mass=10.2, com=(0.016, sg * 0.09, -0.207), inertia=_box_inertia(10.2, 0.18, 0.18, 0.45))
        A(f"{s}KneePitchLink", f"{s}HipPitchLink", f"{s}KneePitch", REVOLUTE, (0, 1, 0), p=(0.000112, sg * 0.0361, -0.431), mass=6.2, com=(-0.005, sg * 0.005, -0.15), inertia=_box_inertia(6.2, 0.12, 0.12, 0.42))
        A(f"{s}AnklePitchLink", f"{s}KneePitchLink", f"{s}AnklePitch", REVOLUTE, (0, 1, 0), p=(-0.0102, 0, -0.406), mass=0.03, com=(0, 0, 0), inertia=_box_inertia(0.03, 0.03, 0.03, 0.03))
        A(f"{s}Foot", f"{s}AnklePitchLink", f"{s}AnkleRoll", REVOLUTE, (1, 0, 0), p=(0, 0, 0), mass=2.37, com=(0.037, 0.005, -0.047), inertia=_box_inertia(2.37, 0.27, 0.16, 0.06))
    for s in ("left", "right"):
        foot = m.findbody(f"{s}Foot")
        for x, y in ((-0.0676, 0.0624), (-0.0676, -0.0624), (0.1928, 0.0624), (0.1928, -0.0624)):
            m.contact_points.append(ContactSpec(foot, np.array([x, y, -0.0902]), np.array([0.0, 0.0, 1.0]), 0.8))
    assert m.nbodies == 59 and m.nq == 65 and m.nv == 64, (m.nbodies, m.nq, m.nv)
    return m


def atlas_nominal_configuration(m):
    """initialize!(state) of notebooks/Standing controller.ipynb cell 6 (raw lines 104-121)."""
    q = m.zero_configuration()
    kneebend, hipbendextra = 1.1, 0.1
    for s in ("l", "r"):
        q[m.configuration_range(m.findjoint(f"{s}_leg_kny"))[0]] = kneebend
        q[m.configuration_range(m.findjoint(f"{s}_leg_hpy"))[0]] = -kneebend / 2 + hipbendextra
        q[m.configuration_range(m.findjoint(f"{s}_leg_aky"))[0]] = -kneebend / 2 - hipbendextra
    fr = m.configuration_range(m.findjoint("pelvis_to_world"))
    q[fr[0]:fr[0] + 7] = [1, 0, 0, 0, 0, 0, 0.85]
    return q


def acrobot():
    """Synthetic Acrobot (PointAccelerationTask Demo.ipynb cell 3)."""
    m = Mechanism()
    m.add_body("upper_link", WORLD, "shoulder", REVOLUTE, (0, 1, 0), p=(0, 0.15, 0), mass=1.0,
               com=(0, 0, -0.5), inertia=(0.083, 0.083, 0.001, 0, 0, 0))
    m.add_body("lower_link", "upper_link", "elbow", REVOLUTE, (0, 1, 0), p=(0, 0.1, -1.0), mass=1.0,
               com=(0, 0, -1.0), inertia=(0.33, 0.33, 0.001, 0, 0, 0))
    return m


def _rand_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rand_tree(joint_types, seed=0, chain=False):
    """Random tree mechanism in the spirit of RBD.rand_tree_mechanism (test/tasks.jl:3,27,72):
    every new body hangs off a uniformly chosen earlier body (or the world for the first),
    random joint pose, axis and spatial inertia."""
    rng = np.random.default_rng(seed)
    m = Mechanism()
    for i, t in enumerate(joint_types):
        parent = WORLD if i == 0 else (i - 1 if chain else int(rng.integers(0, i)))
        mass = float(rng.uniform(0.5, 3.0))
        a = rng.normal(size=(3, 3))
        inertia = a @ a.T * 0.05 + np.eye(3) * 0.02
        m.add_body(f"body{i}", parent, f"joint{i}", t, axis=rng.normal(size=3),
                   R=_rand_rotation(rng), p=rng.normal(size=3) * 0.3, mass=mass,
                   com=rng.normal(size=3) * 0.1,
                   inertia=(inertia[0, 0], inertia[1, 1], inertia[2, 2], inertia[0, 1],
                            inertia[0, 2], inertia[1, 2]))
    return m


def rand_state(m, rng, vscale=1.0):
    """rand!(state): random configuration (unit quaternions for floating joints) and velocity."""
    q = rng.uniform(-math.pi, math.pi, size=m.nq)
    for j, t in enumerate(m.joint_type):
        if t == QUAT_FLOATING:
            r = m.configuration_range(j)
            quat = rng.normal(size=4)
            q[r[0]:r[0] + 4] = quat / np.linalg.norm(quat)
            q[r[0] + 4:r[0] + 7] = rng.normal(size=3)
    v = rng.normal(size=m.nv) * vscale
    return q, v
