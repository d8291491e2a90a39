"""Test helper: drives the lane-serial CPU emulation of the device code (tests/emul/) through the
product's own ctypes binding -- same symbols, different (test-only) shared object."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(HERE, "_build", "libqpc_emul.so")


def build():
    # one builder at a time: pytest-xdist workers would otherwise relink the library under each other's feet
    import fcntl
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emul")], check=True)
    return EMU_LIB


EMU_F32_LIB = os.path.join(HERE, "_build", "libqpc_emul_f32.so")


def build_f32():
    import fcntl
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emul"), "../_build/libqpc_emul_f32.so"], check=True)
    return EMU_F32_LIB


def factory_f32(mech, n_basis, floating):
    """the fp32 variant of the tick (qpc_set_precision(ctrl, QPC_PRECISION_F32)) through its lane-serial emulation"""
    from qpcontrol_jl_amd import binding
    build_f32()
    c = binding.Controller(binding.Model(mech, EMU_F32_LIB), n_basis, floating)
    c.set_precision(1)
    return c


def factory(mech, n_basis, floating):
    from qpcontrol_jl_amd import binding
    build()
    return binding.Controller(binding.Model(mech, EMU_LIB), n_basis, floating)


def race_report():
    """(races, uninitialised reads, writes outside phases, first message) since the last call."""
    L = ctypes.CDLL(EMU_LIB)
    r, u, o = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    buf = ctypes.create_string_buffer(256)
    L.emu_report(ctypes.byref(r), ctypes.byref(u), ctypes.byref(o), buf, 256)
    return r.value, u.value, o.value, buf.value.decode()
