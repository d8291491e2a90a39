"""qpcontrol.jl_amd -- MI355X-native batched whole-body-QP control path (drop-in for the
MomentumBasedController / addtask! / ContactPoint surface of tkoolen/QPControl.jl).

Import through `qpc_pkg.load()` (the directory name is not a Python identifier).
Sub-modules:
  models     flat mechanism descriptions + synthetic Atlas / Acrobot / random trees (pure data)
  binding    ctypes binding of libqpcontrol_hip.so (include/qpc.h); fails loudly when absent
  api        host mirror of the reference API (MomentumBasedController, addtask!, ...), batched
  workloads  synthetic BASELINE.json workloads (configs 1-5)
"""
from . import models  # noqa: F401

__all__ = ["models"]
