"""The drop-in boundary from compiled code: include/qpc.h is valid C99 (and C++11); examples/qp_optimizer_cycle.c drives the
solver-level plug (qpc_qp_setup / qpc_qp_solve / qpc_qp_free) through five warm-started ticks against a closed form; and examples/acrobot_tick.c -- BASELINE
config 1, the Acrobot of the reference's PointAccelerationTask demo with one JointAccelerationTask -- builds its model and
controller through the C ABI alone, runs one tick and agrees with the CLOSED-FORM manipulator equation of the 2-link pendulum
written out in the example (independent of the library AND of the oracle) to 1e-9.  Without a GPU the example is linked against
the lane-serial emulation of the device code (same symbols); under -m gpu against the product library."""
import os
import subprocess

import pytest

import emu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "examples", "acrobot_tick.c")


def test_header_is_c99_and_cxx11():
    hdr = os.path.join(INC, "qpc.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)


def _build_and_run(libdir, libname, exe, src=SRC, token="ACROBOT_OK"):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, src, "-o", exe, "-L", libdir,
                    "-l" + libname, "-lm", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert token in out.stdout, out.stdout
    return out.stdout


SRC_QP = os.path.join(ROOT, "examples", "qp_optimizer_cycle.c")  # the solver-level plug with the optimizer's life cycle (qpc_qp_setup / _solve / _free)


def test_c_qp_cycle_against_closed_form_emulated():
    emu.build()
    _build_and_run(os.path.join(HERE, "_build"), "qpc_emul", os.path.join(HERE, "_build", "qp_cycle_emul"), SRC_QP, "QP_CYCLE_OK")


@pytest.mark.gpu
def test_c_qp_cycle_against_closed_form_gpu():
    print(_build_and_run(os.path.join(ROOT, "qpcontrol.jl_amd"), "qpcontrol_hip", os.path.join(HERE, "_build", "qp_cycle_gpu"), SRC_QP, "QP_CYCLE_OK"))


def test_c_example_against_closed_form_emulated():
    emu.build()
    _build_and_run(os.path.join(HERE, "_build"), "qpc_emul", os.path.join(HERE, "_build", "acrobot_tick_emul"))


@pytest.mark.gpu
def test_c_example_against_closed_form_gpu():
    out = _build_and_run(os.path.join(ROOT, "qpcontrol.jl_amd"), "qpcontrol_hip", os.path.join(HERE, "_build", "acrobot_tick_gpu"))
    print(out)
