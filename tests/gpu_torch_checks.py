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


if __name__ == "__main__":
    {"streams": streams, "graph": graph}[sys.argv[1]]()
    print("CHECK_OK")
