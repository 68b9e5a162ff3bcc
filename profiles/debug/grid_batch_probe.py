"""Developer probe: does the grid / dynamics stage time of a batch of IDENTICAL models grow with the batch?
(If not, the growth seen with walkers is the slowest model of the batch, not a batch overhead.)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import _abi  # noqa: E402
import configs  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
t, nu = configs.C4_EPOCHS, configs.C4_BANDS
d_t, d_nu = torch.from_numpy(np.ascontiguousarray(t)).to(dev), torch.from_numpy(np.ascontiguousarray(nu)).to(dev)
for nb in (1, 2, 16, 64, 128, 256):
    arr = (_abi.ModelParams * nb)(*[_abi.make_params(**dict(configs.C4_TRUTH, jet="GaussianJet")) for _ in range(nb)])
    d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_o = torch.empty((nb, nu.size, t.size), dtype=torch.float64, device=dev)
    best = None
    for _ in range(5):
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size, d_o.data_ptr()))
        torch.cuda.synchronize()
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
        cur = (st.grid_ms, st.dynamics_ms, st.cells_ms, st.flux_ms)
        best = cur if best is None else tuple(min(a, b) for a, b in zip(best, cur))
    print(f"{nb:4d} identical models: grid {best[0]:.3f}  dynamics {best[1]:.3f}  cells {best[2]:.3f}  flux {best[3]:.3f} ms")
