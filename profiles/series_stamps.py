"""Developer probe: phase cycles of one series wavefront on the C4 walker problem (library built with -DVAG_SERIES_STAMPS)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _abi, configs
from vegasafterglow_amd import _lib
import vegasafterglow_amd as va
lib = _lib.load()
h, lock = va.get_context(0)
dp = C.POINTER(C.c_double)
t, nu = configs.c4_mock_data()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prm = _abi.make_params(**dict(configs.C4_TRUTH, jet="GaussianJet"))
arr = (_lib.ModelParams * nb)(*[_lib.ModelParams.from_buffer_copy(bytes(prm)) for _ in range(nb)])
out = np.empty((nb, t.size))
for rep in range(2):
    _lib.check(lib.vag_flux_density_batch(h, arr, nb, t.ctypes.data_as(dp), nu.ctypes.data_as(dp), t.size, out.ctypes.data_as(dp)))
lib.vag_ctx_synchronize(h)
st = _lib.StageTimes()
lib.vag_last_stage_times(h, C.byref(st))
print("flux stage ms", st.flux_ms)
