import os, sys, socket
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import _abi, configs
from test_gpu_fullsize import _c4_fitter
from vegasafterglow_amd.dist import WalkerSharder
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
orc = _abi.load_oracle()
f, defs = _c4_fitter(orc)
_, lo, hi = f.build_spec(defs)
samples = lo + (hi - lo) * np.random.default_rng(4).random((96, len(defs)))
samples[5, 2] = -0.5
want = f.loglike_batch(samples, defs)
dev = torch.device("cuda", 0)
sh = WalkerSharder(f.device_evaluator(defs), device=dev)
th = torch.from_numpy(samples).to(dev)
for call in range(3):
    got = sh(th).cpu().numpy()
    d = np.where(np.isfinite(want), got - want, 0)
    print("call", call, "n differing", int(np.sum(got != want)), "max rel", float(np.max(np.abs(d) / np.maximum(1, np.abs(np.where(np.isfinite(want), want, 1))))),
          "costs", None if sh.costs is None else (float(sh.costs.min()), float(sh.costs.max())), "table head", None if sh.last_table is None else sh.last_table[0][:8])
    bad = np.where(got != want)[0]
    print("  bad idx", bad[:10], got[bad[:5]], want[bad[:5]])
dist.destroy_process_group()
