"""Grid-stage time against the number of models in a batch (top-hat C1a models): how many wavefronts of vag_grid_kernel run side by side?"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _abi, configs, bench
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
_lib.check(lib.vag_ctx_set_stream(h, _lib.torch_stream_handle(torch.cuda.current_stream())))
t, nu = configs.C1_T, configs.C1_NU
base = _abi.make_params(**configs.C1A)
for nb in (256, 512, 1024, 1280, 1536, 2048, 3072, 4096, 8192):
    call = bench._grid_call(lib, h, _lib, dev, [base] * nb, t, nu)  # IDENTICAL models: no straggler
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    st = _lib.StageTimes(); lib.vag_last_stage_times(h, C.byref(st))
    print("nb %5d  grid %.3f ms  dynamics %.3f  total %.3f" % (nb, st.grid_ms, st.dynamics_ms, st.total_ms), flush=True)
