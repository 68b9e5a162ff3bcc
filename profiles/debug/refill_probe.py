"""ODE stage of a large walker step under the plain kernel and under the lane-refill kernel (VAG_DYN_REFILL), for a few settings of the
finished-lane count a wavefront collects before it refills.  usage: python3 profiles/debug/refill_probe.py [nwalkers ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
dev = torch.device("cuda", 0)
for nw in [int(a) for a in sys.argv[1:]] or [8192, 2048, 1024]:
    for refill, rmin in (("0", None), ("1", 1), ("1", 2), ("1", 4), ("1", 8), ("1", 16), ("auto", None)):
        if refill != "auto":
            _lib.hooks["VAG_DYN_REFILL"] = refill
        if rmin:
            _lib.hooks["VAG_DYN_REFILL_MIN"] = str(rmin)
        r = bench.walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=nw, steps=5, tally=True)
        _lib.hooks.pop("VAG_DYN_REFILL", None)
        _lib.hooks.pop("VAG_DYN_REFILL_MIN", None)
        rf = r["roofline_fp64"]
        print(f"walkers {nw} refill {refill} min {rmin}: step {r['ms_per_step']:.3f} ms, ode {r['rank0_stage_ms']['dynamics']:.3f} ms, "
              f"lane util {rf['ode_lane_utilisation']:.3f}, rows {rf['ode_rows']}, rhs {rf['ode_rhs']}, finite {r['finite_walkers']}", flush=True)
