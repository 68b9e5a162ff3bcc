import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, _abi, ctypes as C
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
lib=_lib.load(); h,lock=va.get_context(0); orc=_abi.load_oracle(); dp=C.POINTER(C.c_double)
for kw,nt,nnu in ((dict(jet="GaussianJet", theta_obs=0.25, resolutions=(0.3,1.0,40.0)), 300, 16),
                  (dict(jet="TophatJet", theta_obs=0.1, ssc=True, kn=True, resolutions=(0.2,0.5,30.0)), 150, 8),
                  (dict(jet="PowerLawJet", theta_obs=0.1, k_e=2.0,k_g=2.0, duration=500.0, rvs=dict(eps_e=0.1,eps_B=0.01,p=2.3), resolutions=(0.2,0.5,30.0)), 150, 8)):
    p=_abi.make_params(**kw); t=np.logspace(2,8,nt); nu=np.logspace(9,19,nnu)
    arr=(_lib.ModelParams*1)(_lib.ModelParams.from_buffer_copy(bytes(p))); out=np.empty((1,nnu,nt))
    _lib.check(lib.vag_flux_density_grid_batch(h,arr,1,t.ctypes.data_as(dp),nt,nu.ctypes.data_as(dp),nnu,out.ctypes.data_as(dp)))
    plan=_lib.Plan(); lib.vag_last_plan(h,C.byref(plan))
    w=orc.flux_density_grid(p,t,nu); sel=w>1e-3*w.max(axis=1,keepdims=True)
    print(kw.get("jet"), "cells", plan.n_cells, "rows", plan.n_rows, "pairs", plan.total_pairs, "max rel %.2e"%np.max(np.abs(out[0][sel]/w[sel]-1)))
