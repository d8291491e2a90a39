import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import qpc_pkg; qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W
f=lambda m,N,fl: binding.Controller(binding.Model(m),N,fl)
for cfg,B in ((4,262144),(3,262144)):
    c,wl=W.atlas_workload(f,B,cfg)
    r=c.control_batch(wl.q,wl.v,wl.desired,wl.contact_weight,wl.contact_maxnormalforce)
    st=r['status']; it=r['iters']%1000
    print(cfg,B,np.bincount(st,minlength=6), it.mean(), it.max(), np.where(st!=0)[0][:10])
