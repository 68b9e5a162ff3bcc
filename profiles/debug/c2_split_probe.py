"""Developer probe: would the C2 grid (200 t x 10 nu) be served faster as sub-requests the row-per-lane grid kernel accepts
(<= 128 times, <= 4 frequencies)?  Prints the flux stage time of the full request and of each sub-request.
usage: python profiles/debug/c2_split_probe.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import configs  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
arr = bench.c2_batch(nb, seed=1234)
d_p = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)


def flux_ms(t, nu):
    d_t, d_nu = torch.from_numpy(np.ascontiguousarray(t)).to(dev), torch.from_numpy(np.ascontiguousarray(nu)).to(dev)
    d_o = torch.empty((nb, nu.size, t.size), dtype=torch.float64, device=dev)
    best = 1e9
    for _ in range(3):
        _lib.check(lib.vag_flux_density_grid_batch_dev(h, d_p.data_ptr(), nb, d_t.data_ptr(), t.size, d_nu.data_ptr(), nu.size,
                                                        d_o.data_ptr()))
        torch.cuda.synchronize()
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
        best = min(best, st.flux_ms)
    return best, d_o.cpu().numpy()


t, nu = configs.C2_T, configs.C2_NU
full, ref = flux_ms(t, nu)
print(f"full {t.size} x {nu.size}: flux {full:.2f} ms")
total = 0.0
worst = 0.0
TC, NC = int(os.environ.get("T_CHUNK", "100")), int(os.environ.get("NU_CHUNK", "4"))  # (T_CHUNK=200 NU_CHUNK=2: frequency chunks only, r05)
for t0 in range(0, t.size, TC):
    for n0 in range(0, nu.size, NC):
        ms, o = flux_ms(t[t0:t0 + TC], nu[n0:n0 + NC])
        total += ms
        r = ref[:, n0:n0 + NC, t0:t0 + TC]
        worst = max(worst, float(np.max(np.abs(o - r) / np.where(r > 0, r, 1))))
        print(f"  t[{t0}:{t0 + TC}] nu[{n0}:{n0 + NC}]: flux {ms:.2f} ms")
print(f"sum of sub-requests {total:.2f} ms   max rel difference from the full request {worst:.2e}")
