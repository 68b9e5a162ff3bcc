import sys, os, json, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import _abi
import vegasafterglow_amd as va
from vegasafterglow_amd import _lib
lib = _lib.load(); h, _ = va.get_context(0); dp = C.POINTER(C.c_double)
fx = np.load("tests/golden/reference_nonaxi_spread.npz")
meta = json.loads(str(fx["meta"])); t, nu = fx["t"], fx["nu"]
for case, mt in meta.items():
    kw = dict(mt["kw"])
    if "resolutions" in kw: kw["resolutions"] = tuple(kw["resolutions"])
    prm = _abi.make_params(**kw)
    arr = (_lib.ModelParams * 1)(_lib.ModelParams.from_buffer_copy(bytes(prm)))
    sync, ssc = np.empty((1, nu.size, t.size)), np.empty((1, nu.size, t.size))
    _lib.check(lib.vag_flux_density_grid_components_batch(h, arr, 1, t.ctypes.data_as(dp), t.size, nu.ctypes.data_as(dp), nu.size, sync.ctypes.data_as(dp), ssc.ctypes.data_as(dp)))
    def err(g, w):
        m = w > 1e-9 * w.max()
        return float(np.max(np.abs(g - w)[m] / w[m])) if m.any() else 0.0
    print(case, "sync %.2e" % err(sync[0], fx[case + "__sync"]), "ssc %.2e" % err(ssc[0], fx[case + "__ssc"]))
