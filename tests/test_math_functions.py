"""qpc_math.h: the straight-line sincos / atan2 of the tick against libm on the host (20 M points, all quadrants, 12 decades);
the device compiles the same formulas with the hardware reciprocal seed + two Newton steps in place of the divisions."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_math_functions_match_libm():
    exe = os.path.join(HERE, "_build", "math_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "offline", "math_check.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "atan2: worst" in out.stdout
