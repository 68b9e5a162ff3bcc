"""Developer probe: run the C4 log-likelihood at N walkers a few times (for rocprofv3 --kernel-trace timelines)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vegasafterglow_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
if os.environ.get("OWN_STREAM"):  # a torch side stream instead of the legacy default stream
    own = torch.cuda.Stream()
    torch.cuda.set_stream(own)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
r = bench.walker_bench(lib, h, _lib, dev, 0, 1, steps=10, nwalkers=n)
print(r["ms_per_step"], r["rank0_stage_ms"])

# sampler-like loop: the host consumes ln L after every call (emcee's stretch move needs it before it can propose again)
import time, numpy as np
import configs
from vegasafterglow_amd import fitting
t, nu = configs.c4_mock_data()
kw = configs.C4_TRUTH
fit = fitting.Fitter(z=kw["z"], lumi_dist=kw["lumi_dist"], jet="gaussian", medium="ism")
rng = np.random.default_rng(5)
for b in configs.C4_BANDS:
    sel = nu == b
    fit.add_flux_density(b, t[sel], 1e-27 * (1 + rng.random(sel.sum())), 1e-28 * np.ones(sel.sum()))
defs = [fitting.ParamDef(nm, 10.0 ** lo if lg else lo, 10.0 ** hi if lg else hi,
                         fitting.Scale.log if lg else fitting.Scale.linear) for nm, lg, lo, hi in configs.C4_FREE]
spec, lo, hi = fit.build_spec(defs)
theta = lo + (hi - lo) * np.random.default_rng(0).random((n, len(defs)))
d_theta = torch.from_numpy(theta).to(dev)
d_ll = torch.empty((n,), dtype=torch.float64, device=dev)
h_ll = torch.empty((n,), dtype=torch.float64).pin_memory()
def call():
    _lib.check(lib.vag_loglike_batch_dev(h, C.byref(spec), d_theta.data_ptr(), n, spec.ndim, d_ll.data_ptr()))
    h_ll.copy_(d_ll, non_blocking=True)
    torch.cuda.current_stream().synchronize()
for _ in range(3):
    call()
t0 = time.perf_counter()
for _ in range(20):
    call()
print("sampler-like loop (ln L on the host after every call): %.4f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
