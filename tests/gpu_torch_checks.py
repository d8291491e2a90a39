#!/usr/bin/env python3
"""GPU checks that need torch device tensors / streams next to the product library; run as a child process by
tests/test_host_pipeline.py (torch's HIP runtime is initialised first, as in bench.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
from qpcontrol_jl_amd import workloads as W  # noqa: E402

vp = ctypes.c_void_p


def factory(mech, n_basis, floating):
    return binding.Controller(binding.Model(mech), n_basis, floating)


def setup(B):
    c, wl = W.atlas_workload(factory, B, 3)
    ref = c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight,
                                                                               cm=wl.contact_maxnormalforce).items()}
    return c, ref, dev, t


def launch(c, B, t, tau, vd, st, stream):
    rc = c.L.qpc_control_batch_device(c.h, B, vp(t["q"].data_ptr()), vp(t["v"].data_ptr()), vp(t["d"].data_ptr()),
                                      vp(t["cw"].data_ptr()), vp(t["cm"].data_ptr()), vp(0), vp(0), vp(tau.data_ptr()),
                                      vp(vd.data_ptr()), vp(0), vp(st.data_ptr()), vp(0), vp(stream.cuda_stream))
    assert rc == 0, c.L.qpc_last_error()


def streams():
    B = 4096
    c, ref, dev, t = setup(B)
    lay = c.layout
    ss = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = []
    for rep in range(9):
        tau = torch.full((B, lay.nv), float("nan"), dtype=torch.float64, device=dev)
        vd = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
        st = torch.full((B,), -1, dtype=torch.int32, device=dev)
        outs.append((tau, vd, st))
    torch.cuda.synchronize(dev)  # the fills above ran on torch's current stream
    for rep, (tau, vd, st) in enumerate(outs):  # back-to-back launches alternating over three streams, nothing synchronised in between
        launch(c, B, t, tau, vd, st, ss[rep % 3])
    torch.cuda.synchronize(dev)
    outs = [(tau, st) for tau, vd, st in outs]
    for tau, st in outs:
        assert np.array_equal(st.cpu().numpy(), ref["status"])
        assert np.array_equal(tau.cpu().numpy(), ref["tau"])


def graph():
    B = 2048
    c, ref, dev, t = setup(B)
    lay = c.layout
    tau = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        launch(c, B, t, tau, vd, st, torch.cuda.current_stream(dev))
    for _ in range(3):
        tau.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize(dev)
        assert np.array_equal(tau.cpu().numpy(), ref["tau"])
        assert np.array_equal(st.cpu().numpy(), ref["status"])
    # and the controller still works outside the graph afterwards
    r = c.control_batch(*[x.cpu().numpy() for x in (t["q"], t["v"], t["d"], t["cw"], t["cm"])])
    assert np.array_equal(r["tau"], ref["tau"])


def warm_mixed_rollouts():
    """on a warm-start controller the host rollout (qpc_rollout_batch) used to free the working-set buffer of
    the device rollout when its own slab grew; device rollout -> larger host rollout -> device rollout -> destroy must
    leave the device rollout's results unchanged (and nothing freed twice)."""
    from qpcontrol_jl_amd import api, models as M
    mech = M.atlas()
    low = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world")
    for cp in mech.contact_points:
        pt = low.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce, pt.weight = 1e6, 1e-3
    api.StandingController(low, ["l_foot", "r_foot"], "pelvis", M.atlas_nominal_configuration(mech))
    low._c.enable_warm_start()
    lay = low.finalize(0)
    assert low._c.L.qpc_debug_warm_active(low._c.h) == 1
    B, K = 1024, 4
    q, v = W.standing_states(mech, 4 * B, 3, dq=0.02, drot=0.01, vstd=0.0)
    k = 6.0e4
    sp = binding.SimParams(1e-3, 0.01308714 + mech.total_mass() * 9.81 / (8 * k), k, 400.0, 300.0)
    dev = torch.device("cuda", 0)
    L, h = low._c.L, low._c.h

    def device_rollout():
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        dq, dv = t(q[:B]), t(v[:B])
        cw, cm = t(np.full((B, lay.n_contacts), 1e-3)), t(np.full((B, lay.n_contacts), 1e6))
        tau = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
        vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
        st, bad = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        p = lambda x: vp(x.data_ptr())  # noqa: E731
        torch.cuda.synchronize(dev)
        rc = L.qpc_rollout_batch_device(h, B, K, p(dq), p(dv), vp(0), p(cw), p(cm), vp(0), vp(0), ctypes.byref(sp), p(tau), p(vd),
                                        p(st), p(bad), vp(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0, L.qpc_last_error()
        torch.cuda.synchronize(dev)
        assert int(bad.sum().item()) == 0
        return dq.cpu().numpy(), dv.cpu().numpy(), tau.cpu().numpy()

    first = device_rollout()
    for Bh in (2 * B, 4 * B):  # the host slab grows twice
        r = low._c.rollout(q[:Bh], v[:Bh], K, sp, contact_weight=np.full((Bh, lay.n_contacts), 1e-3),
                           contact_maxnormalforce=np.full((Bh, lay.n_contacts), 1e6))
        assert not r["bad"].any()
        assert np.array_equal(r["q"][:B], first[0]) and np.array_equal(r["tau"][:B], first[2])
        again = device_rollout()
        for a, b in zip(first, again):
            assert np.array_equal(a, b)
    del low
    import gc
    gc.collect()
    torch.cuda.synchronize(dev)


def slots():
    """qpc_control_batch_device_slot: launches of DIFFERENT slots on different streams overlap (own work counters) -- every result
    complete and equal to the reference, also when the same slot is reused across streams and when batches differ in size"""
    B = 4096
    c, ref, dev, t = setup(B)
    lay = c.layout
    L = c.L
    ss = [torch.cuda.Stream(dev) for _ in range(4)]
    outs = []
    for rep in range(12):
        tau = torch.full((B, lay.nv), float("nan"), dtype=torch.float64, device=dev)
        vd = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
        st = torch.full((B,), -1, dtype=torch.int32, device=dev)
        outs.append((tau, vd, st))
    torch.cuda.synchronize(dev)
    for rep, (tau, vd, st) in enumerate(outs):
        slot = rep % 4
        n = B if rep % 3 else B // 2 + 7  # some launches only solve a prefix of the batch
        rc = L.qpc_control_batch_device_slot(c.h, 0, slot, n, vp(t["q"].data_ptr()), vp(t["v"].data_ptr()), vp(t["d"].data_ptr()),
                                             vp(t["cw"].data_ptr()), vp(t["cm"].data_ptr()), vp(0), vp(0), vp(tau.data_ptr()), vp(vd.data_ptr()), vp(0),
                                             vp(st.data_ptr()), vp(0), vp(ss[(rep + rep // 4) % 4].cuda_stream))
        assert rc == 0, L.qpc_last_error()
    torch.cuda.synchronize(dev)
    for rep, (tau, vd, st) in enumerate(outs):
        n = B if rep % 3 else B // 2 + 7
        assert np.array_equal(st.cpu().numpy()[:n], ref["status"][:n])
        assert np.array_equal(tau.cpu().numpy()[:n], ref["tau"][:n])
        assert np.isnan(tau.cpu().numpy()[n:]).all()
    rc = L.qpc_control_batch_device_slot(c.h, 0, 4, B, *([vp(0)] * 12), vp(0))  # slot out of range
    assert rc != 0


def ordered():
    """cost-ordered tickets (qpc_ticket_order_device + qpc_control_batch_device_ordered): the order is a permutation sorted by the
    previous tick's working-set changes, the ordered launch returns bit-identical outputs; its rate is printed beside the plain launch's"""
    import time
    B = 8192
    c, ref, dev, t = setup(B)
    lay, L = c.layout, c.L
    s = torch.cuda.Stream(dev)
    tau = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
    vd = torch.empty_like(tau)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    it = torch.zeros((B,), dtype=torch.int32, device=dev)
    order = torch.full((B,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)

    def run(order_ptr):
        rc = L.qpc_control_batch_device_ordered(c.h, 0, 0, B, vp(t["q"].data_ptr()), vp(t["v"].data_ptr()), vp(t["d"].data_ptr()),
                                                vp(t["cw"].data_ptr()), vp(t["cm"].data_ptr()), vp(0), vp(0), vp(tau.data_ptr()), vp(vd.data_ptr()),
                                                vp(0), vp(st.data_ptr()), vp(it.data_ptr()), order_ptr, vp(s.cuda_stream))
        assert rc == 0, L.qpc_last_error()
    run(vp(0))  # unordered: gives the costs
    rc = L.qpc_ticket_order_device(c.h, 0, B, vp(it.data_ptr()), 1000, vp(order.data_ptr()), vp(s.cuda_stream))
    assert rc == 0, L.qpc_last_error()
    torch.cuda.synchronize(dev)
    o, cost = order.cpu().numpy(), it.cpu().numpy() % 1000
    assert np.array_equal(np.sort(o), np.arange(B))           # a permutation
    assert np.all(np.diff(cost[o]) <= 0)                      # sorted by descending cost
    tau.fill_(float("nan")); st.fill_(-1)
    torch.cuda.synchronize(dev)
    run(vp(order.data_ptr()))
    torch.cuda.synchronize(dev)
    assert np.array_equal(st.cpu().numpy(), ref["status"]) and np.array_equal(tau.cpu().numpy(), ref["tau"])  # bitwise
    times = {}
    for name, ptr in (("plain", vp(0)), ("ordered", vp(order.data_ptr()))):
        for _ in range(3):
            run(ptr)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(30):
            run(ptr)
        torch.cuda.synchronize(dev)
        times[name] = (time.perf_counter() - t0) / 30
    print("ordered tickets @ 8192:", {k: f"{B / v / 1e6:.2f} M solves/s" for k, v in times.items()})
    # a parity suite does not fail on a timing: the figure is printed (tools/ and bench.py measure it); only a gross regression
    # -- the ordered launch taking half again as long as the plain one -- is an error of the feature itself
    assert times["ordered"] < 1.5 * times["plain"]


def l0_device():
    """qpc_qp_solve_device (tests/test_l0_sparse.py)"""
    import test_l0_sparse
    test_l0_sparse.gpu_device_entry_check()


if __name__ == "__main__":
    {"l0_device": l0_device, "streams": streams, "graph": graph, "warm_mixed_rollouts": warm_mixed_rollouts, "slots": slots, "ordered": ordered}[sys.argv[1]]()
    print("CHECK_OK")
