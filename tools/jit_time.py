#!/usr/bin/env python3
"""How long qpc_finalize takes for the Atlas config-3 controller (hiprtc specialisation of the tick and the simulation step): with the
null-space presolve (two instances of the solver in the kernel), without it, and from the in-process code-object cache.
Measured on MI355X / ROCm 7.2: 5.0 s, 2.65 s, 0.04 s.   python tools/jit_time.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import qpc_pkg; qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W
f=lambda m,N,fl: binding.Controller(binding.Model(m),N,fl)
for env in ({}, {"QPC_NO_PRESOLVE":"1"}):
    os.environ.update(env)
    t0=time.perf_counter(); c,wl=W.atlas_workload(f,16,3); t1=time.perf_counter()
    print("finalize + workload", env, round(t1-t0,2), "s")
    for k in env: os.environ.pop(k)
t0=time.perf_counter(); c,wl=W.atlas_workload(f,16,3); print("cached", round(time.perf_counter()-t0,2))
