"""Does the 8192-walker step run into the GPU's power limit?  Loops the step for a few seconds under VAG_DYN_REFILL=0 / 1 while a side thread
samples the shader clock and the package power from rocm-smi; prints the step's stage times and the sampled clock / power.
usage: python3 profiles/debug/clock_probe.py [nwalkers]"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from vegasafterglow_amd import _lib  # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.vag_ctx_create(0, C.byref(h)))
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
dev = torch.device("cuda", 0)
fit, defs, _ = bench.c4_fitter(lib, h, _lib)
spec, lo, hi = fit.build_spec(defs)
theta = torch.from_numpy(lo + (hi - lo) * np.random.default_rng(0).random((nw, len(defs)))).to(dev)
ev = fit.device_evaluator(defs, context=(h, bench._NullLock()))


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out.append(("err", str(e)))
            return
        sclk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", txt)
        pw = re.search(r"Power \(W\):\s*([\d.]+)", txt) or re.search(r"Socket Power.*?:\s*([\d.]+)", txt)
        out.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))


import configs  # noqa: E402
call = bench._grid_call(lib, h, _lib, dev, bench.c2_batch(512, seed=1234), configs.C2_T, configs.C2_NU)
for _ in range(3):
    call()
torch.cuda.synchronize()
stop, out = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, out))
th.start()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < 4.0:
    call()
    torch.cuda.synchronize()
    n += 1
dt = (time.perf_counter() - t0) / n
stop.set()
th.join()
clk = [c for c, _ in out if isinstance(c, int)]
pw = [p for _, p in out if isinstance(p, float)]
print(f"headline (512 configs[1] models per call): {n} calls, {1e3 * dt:.3f} ms per call; sclk samples {len(clk)}: mean {np.mean(clk):.0f} MHz min {min(clk)}; power mean {np.mean(pw):.0f} W max {max(pw):.0f}", flush=True)

def sampled_loop(label, fn, seconds=4.0):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize()
        n += 1
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join()
    clk = [c for c, _ in out if isinstance(c, int)]
    pw = [p for _, p in out if isinstance(p, float)]
    print(f"{label}: {n} calls, {1e3 * dt:.3f} ms per call; sclk samples {len(clk)}: mean {np.mean(clk):.0f} MHz min {min(clk)}; power mean {np.mean(pw):.0f} W max {max(pw):.0f}", flush=True)


if nw >= 8192:  # the SSC ensembles of the bench as well
    t_e, nu_e = np.logspace(2, 8, 100), np.array([1e9, 4.84e14, 1e18, 2.4e26])
    sampled_loop("configs[2] (512 models per call)", bench._grid_call(lib, h, _lib, dev, configs.c3_batch(512), t_e, nu_e))
    sampled_loop("configs[4] (1024 members per call)", bench._grid_call(lib, h, _lib, dev, configs.c5_batch(1024), t_e, nu_e))

for mode in ("0", "1", "0", "1"):
    _lib.hooks["VAG_DYN_REFILL"] = mode
    for _ in range(3):
        ev(theta)
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0, n = time.perf_counter(), 0
    stages = np.zeros(6)
    while time.perf_counter() - t0 < 4.0:
        ll, _ = ev(theta)
        torch.cuda.synchronize()
        st = _lib.StageTimes()
        lib.vag_last_stage_times(h, C.byref(st))
        stages += [st.grid_ms, st.dynamics_ms, st.cells_ms, st.flux_ms, st.reduce_ms, st.total_ms]
        n += 1
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join()
    clk = [c for c, _ in out if isinstance(c, int)]
    pw = [p for _, p in out if isinstance(p, float)]
    print(f"VAG_DYN_REFILL={mode}: {n} steps, {1e3 * dt:.3f} ms per step; stages grid {stages[0] / n:.3f} ode {stages[1] / n:.3f} cells {stages[2] / n:.3f} flux {stages[3] / n:.3f} "
          f"total {stages[5] / n:.3f}; sclk samples {len(clk)}: mean {np.mean(clk) if clk else float('nan'):.0f} MHz min {min(clk) if clk else 0}; "
          f"power mean {np.mean(pw) if pw else float('nan'):.0f} W max {max(pw) if pw else 0:.0f} ({out[:1]})", flush=True)
    _lib.hooks.pop("VAG_DYN_REFILL")
