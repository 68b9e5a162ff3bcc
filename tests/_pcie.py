import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, configs
from vegasafterglow_amd import _lib
from vegasafterglow_amd.model import get_context
lib=_lib.load(); h,_=get_context(0)
nb=512; arr=bench.c2_batch(nb,1234)
t,nu=configs.C2_T,configs.C2_NU
out=np.empty((nb,nu.size,t.size)); dp=C.POINTER(C.c_double)
pa=(_lib.ModelParams*nb)(*[_lib.ModelParams.from_buffer_copy(bytes(arr[i])) for i in range(nb)])
for _ in range(2): lib.vag_flux_density_grid_batch(h,pa,nb,t.ctypes.data_as(dp),t.size,nu.ctypes.data_as(dp),nu.size,out.ctypes.data_as(dp))
t0=time.perf_counter()
for _ in range(5): rc=lib.vag_flux_density_grid_batch(h,pa,nb,t.ctypes.data_as(dp),t.size,nu.ctypes.data_as(dp),nu.size,out.ctypes.data_as(dp))
dt=(time.perf_counter()-t0)/5
print("host-pointer API (PCIe inclusive): %.2f ms/step -> %.0f LC/s"%(dt*1e3, nb/dt), rc)
