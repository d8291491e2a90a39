"""CPU-side checks of the product library: it loads without a GPU, exports every symbol that
include/qpc.h declares, the construction half of the ABI works, and compute entry points fail
loudly (QPC_ERR_NO_DEVICE) instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

from qpcontrol_jl_amd import binding, models as M, workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(binding.DEFAULT_LIB):
        import __graft_entry__ as g
        g.build()
    return binding.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "qpc.h")).read()
    names = set(re.findall(r"\b(qpc_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_construction_and_loud_failure_without_device(lib):
    import torch
    mech = M.atlas()
    c = binding.Controller(binding.Model(mech), 4, mech.findbody("pelvis"))
    W.standing_setup(c, mech, M.atlas_nominal_configuration(mech))
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be exercised")
    with pytest.raises(binding.QpcError, match="no HIP device|no CPU fallback|NO_DEVICE|code 4"):
        c.finalize(0)
    # not finalized -> compute entry point refuses
    lay = binding.Layout()
    assert lib.qpc_get_layout(c.h, ctypes.byref(lay)) == 2  # QPC_ERR_NOT_FINALIZED


def test_prepared_l0_fails_loudly_without_device(lib):
    """qpc_qp_setup (the solver-level plug with the optimizer's life cycle): no device, no problem -- QPC_ERR_NO_DEVICE, not a CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be exercised")
    with pytest.raises(binding.QpcError, match="no HIP device|no CPU fallback|NO_DEVICE|code 4"):
        binding.QpProblem(2, 1, np.eye(2, dtype=bool), np.ones((1, 2), dtype=bool))
    with pytest.raises(binding.QpcError, match="out of range"):  # a bad pattern is refused before any device is touched
        binding.QpProblem(2, 1, np.eye(2, dtype=bool), (np.array([5]), np.array([0])))


def test_argument_validation(lib):
    mech = M.acrobot()
    model = binding.Model(mech)
    with pytest.raises(binding.QpcError):
        binding.Controller(model, 0, -1)  # n_basis out of range
    c = binding.Controller(model, 4, -1)
    with pytest.raises(binding.QpcError):
        c.add_task(4, joint=7)  # no such joint
    with pytest.raises(binding.QpcError):
        c.add_task(0, base=0, target=99)
    assert lib.qpc_status_string(2) == b"INFEASIBLE"
    assert b"gfx950" in lib.qpc_version()


def test_jit_source_compiles_offline(lib):
    """hiprtc cross-compiles the specialised kernel for gfx950 without a GPU (what qpc_finalize does)."""
    mech = M.atlas()
    c = binding.Controller(binding.Model(mech), 4, mech.findbody("pelvis"))
    W.standing_setup(c, mech, M.atlas_nominal_configuration(mech))
    lib.qpc_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
    buf = ctypes.create_string_buffer(4000)
    rc = lib.qpc_debug_jit_compile(c.h, b"gfx950", None, buf, 4000)
    assert rc == 0, buf.value.decode()


def test_jit_disk_cache(lib, tmp_path, monkeypatch):
    """QPC_JIT_CACHE_DIR: the second process-level compile of the same controller reads the code object from disk (qpc_finalize spends
    ~5 s in hiprtc per distinct controller otherwise); same bytes, and QPC_JIT_CACHE=0 still compiles"""
    import time
    mech = M.acrobot()
    lib.qpc_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]

    def compile_once(out):
        c = binding.Controller(binding.Model(mech), 4, -1)
        c.add_task(4, joint=1)
        c.regularize(0, 1e-6)
        buf = ctypes.create_string_buffer(4000)
        t0 = time.perf_counter()
        rc = lib.qpc_debug_jit_compile(c.h, b"gfx950", str(out).encode(), buf, 4000)
        assert rc == 0, buf.value.decode()
        return time.perf_counter() - t0

    monkeypatch.setenv("QPC_JIT_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("QPC_JIT_CACHE", "0")  # (the in-process cache off: nothing is stored, on disk or in memory)
    compile_once(tmp_path / "a.co")
    assert not list(tmp_path.glob("qpc_*.co"))
    monkeypatch.delenv("QPC_JIT_CACHE")
    t_cold = compile_once(tmp_path / "b.co")
    files = list(tmp_path.glob("qpc_*.co"))
    assert len(files) == 1 and lib.qpc_kernel_fingerprint().decode() in files[0].name
    assert files[0].read_bytes() == (tmp_path / "b.co").read_bytes() == (tmp_path / "a.co").read_bytes()
    # a fresh process (empty in-process cache) reads it back: the same bytes
    import subprocess
    import sys
    code = ("import ctypes, sys, time; sys.path.insert(0, %r); import qpc_pkg; qpc_pkg.load()\n"
            "from qpcontrol_jl_amd import binding, models as M\n"
            "L = binding.load_library(); c = binding.Controller(binding.Model(M.acrobot()), 4, -1); c.add_task(4, joint=1); c.regularize(0, 1e-6)\n"
            "L.qpc_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]\n"
            "t0 = time.perf_counter(); rc = L.qpc_debug_jit_compile(c.h, b'gfx950', sys.argv[1].encode(), None, 0); print(rc, time.perf_counter() - t0)\n"
            % ROOT)
    out = subprocess.run([sys.executable, "-c", code, str(tmp_path / "c.co")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    rc, t_disk = out.stdout.split()[-2:]
    assert rc == "0" and (tmp_path / "c.co").read_bytes() == files[0].read_bytes()
    print(f"hiprtc compile {t_cold:.2f} s, from the disk cache {float(t_disk):.2f} s")  # (no assertion: the toolchain may cache by itself)
    # a damaged file is not a code object: the next compile ignores it and writes a good one in its place
    good = files[0].read_bytes()
    files[0].write_bytes(b"not an ELF file" * 20)
    out = subprocess.run([sys.executable, "-c", code, str(tmp_path / "d.co")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-2] == "0", out.stderr[-500:]
    fresh = (tmp_path / "d.co").read_bytes()  # (a compile in another process need not reproduce the first one's bytes)
    assert fresh[:4] == b"\x7fELF" and len(fresh) > 10000 and files[0].read_bytes() == fresh


def test_bench_cpu_worker_cli():
    """bench.py's cpu_baseline leg: a child process (`--cpu-worker`) that only touches the oracle"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-worker", "0", "16", "0.5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    for leg in ("nbw", "nbc", "tc", "tw"):  # notebook warm / cold, tight cold / warm
        assert r[leg]["done"] > 0 and sum(r[leg]["iters"]) > 0 and r[leg]["seconds"] > 0, leg
    assert r["tc"]["done"] >= 16
