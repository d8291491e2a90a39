"""Host-pointer pipeline of the C ABI (qpcontrol.jl_amd/csrc/qpc_pipeline.h): qpc_finalize_devices,
qpc_control_batch sharded over devices and cut into chunks, qpc_control_batch_async + qpc_wait,
error propagation from a worker thread.

The pipeline is backend-agnostic C++; here it drives the test emulator's fake devices (CPU), so what
is checked is exactly what can go wrong on the host side: shard / chunk offsets, every QP solved
exactly once, results bitwise equal to the single-shard call, a job in flight is waited for, a
failing chunk surfaces as the call's error.  The `-m gpu` twin runs the HIP backend."""
import ctypes

import time

import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import workloads as W


def _workload(factory, B, config=3, devices=None, chunk=0, slots=0, inline=-1, warm=False):
    real = binding.Controller.finalize

    def fin(self, device_id=0):
        if slots or chunk or inline >= 0:
            self.set_host_pipeline(chunk, slots, inline)
        if warm:
            self.enable_warm_start()
        return real(self, device_id, devices=devices)
    binding.Controller.finalize = fin
    try:
        return W.atlas_workload(factory, B, config)
    finally:
        binding.Controller.finalize = real


def _tick(c, wl, **kw):
    return c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce, **kw)


def _chunk_log(c):
    L = ctypes.CDLL(emu.EMU_LIB)
    L.emu_chunk_log.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]
    buf = (ctypes.c_int64 * (4 * 4096))()
    n = L.emu_chunk_log(c.h, buf, 4096)
    return np.array(buf[:4 * n], dtype=np.int64).reshape(n, 4)


def test_shard_bounds_match_python_mirror():
    for B in (0, 1, 7, 8, 9, 1000):
        for G in (1, 2, 3, 8):
            cover = []
            for g in range(G):
                lo, hi = W.shard_bounds(B, g, G)
                cover += list(range(lo, hi))
            assert cover == list(range(B))


@pytest.mark.parametrize("ndev,chunk,slots", [(2, 3, 2), (3, 4, 1), (1, 5, 3), (8, 1, 2)])
def test_two_shards_in_one_process_bitwise_equal(ndev, chunk, slots):
    """one process drives several (fake) devices; the result equals the single-shard call bit for bit
    and every QP is solved exactly once by the device that owns its contiguous shard."""
    emu.build()
    B = 29
    c1, wl = _workload(emu.factory, B, inline=10 ** 9)            # everything inline on the caller's thread
    cN, _ = _workload(emu.factory, B, devices=list(range(ndev)), chunk=chunk, slots=slots, inline=0)
    assert cN.n_devices() == ndev
    r1 = _tick(c1, wl)
    _chunk_log(cN)
    rN = _tick(cN, wl)
    for k in ("tau", "vdot", "cwrench", "status", "iters"):
        assert np.array_equal(r1[k], rN[k]), k
    log = _chunk_log(cN)
    owner = -np.ones(B, dtype=int)
    for dev, slot, lo, n in log:
        assert 0 <= slot < slots and 1 <= n <= chunk
        assert np.all(owner[lo:lo + n] == -1)  # no QP twice
        owner[lo:lo + n] = dev
    for g in range(ndev):  # QP index range [g B/G, (g+1) B/G) -> device g
        lo, hi = W.shard_bounds(B, g, ndev)
        assert np.all(owner[lo:hi] == g)


def test_async_then_wait_and_back_to_back_jobs():
    emu.build()
    B = 12
    c, wl = _workload(emu.factory, B, devices=[0, 1], chunk=2, slots=2, inline=0)
    ref = _tick(c, wl)
    p1 = _tick(c, wl, asynchronous=True)
    # a second job posted while the first may still be running: it first waits for the first (one job in flight)
    wl2q = wl.q[::-1].copy()
    p2 = c.control_batch(wl2q, wl.v[::-1].copy(), wl.desired[::-1].copy(), wl.contact_weight[::-1].copy(),
                         wl.contact_maxnormalforce[::-1].copy(), asynchronous=True)
    r2 = p2.wait()
    r1 = p1.result  # complete: job 2 could not start before job 1 was done
    assert np.array_equal(r1["tau"], ref["tau"]) and np.array_equal(r1["status"], ref["status"])
    assert np.array_equal(r2["tau"], ref["tau"][::-1])
    c.wait()  # nothing in flight: returns at once with QPC_OK
    # B = 0 is a no-op
    e = c.control_batch(np.zeros((0, wl.q.shape[1])), np.zeros((0, wl.v.shape[1])), np.zeros((0, wl.desired.shape[1])),
                        np.zeros((0, 8)), np.zeros((0, 8)))
    assert e["tau"].shape[0] == 0


def test_worker_error_reaches_the_caller():
    emu.build()
    B = 16
    c, wl = _workload(emu.factory, B, devices=[0, 1], chunk=2, slots=2, inline=0)
    L = ctypes.CDLL(emu.EMU_LIB)
    L.emu_fail_at.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.emu_fail_at(c.h, 11)
    with pytest.raises(binding.QpcError, match="injected failure"):
        _tick(c, wl)
    p = _tick(c, wl, asynchronous=True)  # async: the error is the job's, reported by wait
    with pytest.raises(binding.QpcError, match="injected failure"):
        p.wait()
    # an async job that failed and was never waited for must not make the NEXT synchronous call return that
    # stale error without running its own batch (threaded path and inline path)
    ref = None
    for inline in (0, 64):
        c.set_host_pipeline(inline_qps=inline)
        L.emu_fail_at(c.h, 11)
        pending = _tick(c, wl, asynchronous=True)  # fails in the background; nobody waits (its arrays stay alive)
        time.sleep(0.05)
        L.emu_fail_at(c.h, -1)
        r = _tick(c, wl)  # runs its own job and reports its own outcome
        del pending
        assert (r["status"] == 0).all() and np.isfinite(r["tau"]).all()
        ref = r if ref is None else ref
        assert np.array_equal(r["tau"], ref["tau"])
        c.wait()  # the stale error is gone
        # ... but not lost: qpc_last_async_error hands it over once
        L.qpc_last_async_error.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32]
        buf = ctypes.create_string_buffer(200)
        assert L.qpc_last_async_error(c.h, buf, 200) != 0 and b"injected failure" in buf.value
        assert L.qpc_last_async_error(c.h, buf, 200) == 0 and buf.value == b""
    L.emu_fail_at(c.h, -1)
    r = _tick(c, wl)  # the controller stays usable
    assert (r["status"] == 0).all()


def test_pipeline_settings_validation():
    emu.build()
    c, wl = _workload(emu.factory, 4)
    with pytest.raises(binding.QpcError):
        c.set_host_pipeline(slots_per_device=2)  # fixed by qpc_finalize
    c.set_host_pipeline(chunk_qps=2, inline_qps=0)  # chunk / inline threshold may change between jobs
    r = _tick(c, wl)
    assert (r["status"] == 0).all()


# ---- GPU: the same properties through the HIP backend --------------------------------------------
def gpu_factory(mech, n_basis, floating):
    return binding.Controller(binding.Model(mech), n_basis, floating)


@pytest.mark.gpu
def test_gpu_pipelined_host_path_equals_device_path():
    """chunked + multi-slot host-pointer path == one inline launch, bitwise (the QPs are independent and the kernel is
    deterministic per QP), for pageable AND pinned caller arrays, sync and async."""
    B = 5000
    c1, wl = _workload(gpu_factory, B, inline=10 ** 9, chunk=10 ** 9)
    cN, _ = _workload(gpu_factory, B, chunk=512, slots=3, inline=0)
    r1 = _tick(c1, wl)
    rN = _tick(cN, wl)
    for k in ("tau", "vdot", "cwrench", "status", "iters"):
        assert np.array_equal(r1[k], rN[k]), k
    ra = _tick(cN, wl, asynchronous=True).wait()
    assert np.array_equal(ra["tau"], r1["tau"])
    # pinned caller arrays are DMA-ed in place
    L = cN.L
    pin = {k: binding.pinned_array(L, a.shape) for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight,
                                                               cm=wl.contact_maxnormalforce).items()}
    for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce).items():
        pin[k][...] = a
    lay = cN.layout
    out = dict(tau=binding.pinned_array(L, (B, lay.nv)), vdot=binding.pinned_array(L, (B, lay.nv)),
               cwrench_flat=binding.pinned_array(L, (B, 6 * lay.n_contacts)), status=binding.pinned_array(L, (B,), np.int32),
               iters=binding.pinned_array(L, (B,), np.int32))
    rp = cN.control_batch(pin["q"], pin["v"], pin["d"], pin["cw"], pin["cm"], out=out)
    for k in ("tau", "vdot", "cwrench", "status", "iters"):
        assert np.array_equal(r1[k], rp[k]), k
    # caller-owned arrays page-locked IN PLACE (qpc_host_register: what QPControlHIP.pin! does with Julia arrays)
    own = {k: np.ascontiguousarray(a).copy() for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce).items()}
    outs = dict(tau=np.zeros((B, lay.nv)), vdot=np.zeros((B, lay.nv)), cwrench_flat=np.zeros((B, 6 * lay.n_contacts)),
                status=np.zeros(B, dtype=np.int32), iters=np.zeros(B, dtype=np.int32))
    regs = list(own.values()) + list(outs.values())
    for a in regs:
        assert L.qpc_host_register(ctypes.c_void_p(a.ctypes.data), a.nbytes) == 0, L.qpc_last_error()
    try:
        rr = cN.control_batch(own["q"], own["v"], own["d"], own["cw"], own["cm"], out=outs)
        for k in ("tau", "vdot", "cwrench", "status", "iters"):
            assert np.array_equal(r1[k], rr[k]), k
        ra2 = cN.control_batch(own["q"], own["v"], own["d"], own["cw"], own["cm"], out=outs, asynchronous=True).wait()
        assert np.array_equal(ra2["tau"], r1["tau"])
    finally:
        for a in regs:
            assert L.qpc_host_unregister(ctypes.c_void_p(a.ctypes.data)) == 0
    # ragged: B not a multiple of the chunk, smaller than one chunk, one QP
    for Bs in (1, 511, 513):
        rs = cN.control_batch(wl.q[:Bs], wl.v[:Bs], wl.desired[:Bs], wl.contact_weight[:Bs], wl.contact_maxnormalforce[:Bs])
        assert np.array_equal(rs["tau"], r1["tau"][:Bs])


def _run_torch_check(which):
    """torch device tensors + streams next to the product library: in a child process that initialises torch's HIP
    runtime FIRST (as bench.py does) -- in the pytest process the library is already loaded by the time torch would
    initialise, and torch then finds no device."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "gpu_torch_checks.py"), which], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "CHECK_OK" in out.stdout


@pytest.mark.gpu
def test_gpu_user_streams_are_serialised():
    """two launches of one controller on different streams share the work counter; the library turns the
    change of stream into an event dependency -- every result must be complete and correct (tests/gpu_torch_checks.py)."""
    _run_torch_check("streams")


@pytest.mark.gpu
def test_gpu_graph_replay_of_the_tick():
    """the work counter re-arms itself on the device, so a captured tick can be replayed (hipGraph)"""
    _run_torch_check("graph")


@pytest.mark.gpu
def test_gpu_launch_slots_overlap_safely():
    """qpc_control_batch_device_slot: four slots on four streams in flight at once, slots reused across streams, prefix batches"""
    _run_torch_check("slots")


@pytest.mark.gpu
def test_gpu_cost_ordered_tickets():
    """qpc_ticket_order_device + qpc_control_batch_device_ordered: a permutation sorted by the previous tick's working-set changes,
    bit-identical outputs, not slower than the batch order (tests/gpu_torch_checks.py)"""
    _run_torch_check("ordered")


@pytest.mark.gpu
@pytest.mark.gpu_slow  # (12 s: a regression test of a buffer lifetime; warm start and rollouts each keep a fast test)
def test_gpu_warm_controller_mixes_device_and_host_rollouts():
    """the host rollout must not free the device rollout's working-set buffer (tests/gpu_torch_checks.py)"""
    _run_torch_check("warm_mixed_rollouts")


@pytest.mark.gpu
def test_gpu_finalize_devices_single_and_errors():
    """qpc_finalize_devices on the one GPU of the box (the multi-device sharding itself is covered with the emulator's
    fake devices): device list validation, qpc_n_devices, the device-indexed *_device entry."""
    from qpcontrol_jl_amd import models as M
    mech = M.atlas()
    bad = binding.Controller(binding.Model(mech), 4, mech.findbody("pelvis"))
    W.standing_setup(bad, mech, M.atlas_nominal_configuration(mech))
    with pytest.raises(binding.QpcError, match="duplicate"):
        bad.finalize(devices=[0, 0])
    with pytest.raises(binding.QpcError, match="out of range"):
        bad.finalize(devices=[0, 97])
    c, wl = _workload(gpu_factory, 256, devices=[0])
    assert c.n_devices() == 1
    r = _tick(c, wl)
    assert (r["status"] == 0).all()
    L = c.L
    vp = ctypes.c_void_p
    lay = c.layout
    # device-indexed entry with page-locked host arrays as "device" pointers (zero-copy), default stream
    a = {k: binding.pinned_array(L, x.shape) for k, x in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce).items()}
    for k, x in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce).items():
        a[k][...] = x
    tau, vd = binding.pinned_array(L, (256, lay.nv)), binding.pinned_array(L, (256, lay.nv))
    st = binding.pinned_array(L, (256,), np.int32)
    st[...] = -1
    P = lambda x: vp(x.ctypes.data)  # noqa: E731
    assert L.qpc_control_batch_device_on(c.h, 5, 256, P(a["q"]), P(a["v"]), P(a["d"]), P(a["cw"]), P(a["cm"]), vp(0), vp(0), P(tau), P(vd), vp(0),
                                         P(st), vp(0), vp(0)) != 0  # no such device index
    rc = L.qpc_control_batch_device_on(c.h, 0, 256, P(a["q"]), P(a["v"]), P(a["d"]), P(a["cw"]), P(a["cm"]), vp(0), vp(0), P(tau), P(vd), vp(0),
                                       P(st), vp(0), vp(0))
    assert rc == 0
    c.wait()  # nothing posted through the host entry: returns at once
    # the launch went to the default stream without synchronisation: the statuses (written last by each QP, into page-locked
    # memory) tell when it is done
    import time
    t0 = time.time()
    while (st < 0).any() and time.time() - t0 < 10.0:
        time.sleep(0.01)
    time.sleep(0.05)
    assert (st == 0).all() and np.array_equal(tau, r["tau"])
