# Developer aid (GPU box): the 1024- / 8192-walker step of the product library and of every variants/libvag_*.so (stage times of rank 0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  VAG_LIB_PATH=$f python3 - <<'PY'
import ctypes as C, os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, bench
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
dev = torch.device("cuda", 0)
for n in (1024, 8192, 128):
    r = [bench.walker_bench(lib, h, _lib, dev, 0, 1, nwalkers=n, steps=10) for _ in range(2)][-1]
    print(os.path.basename(os.environ["VAG_LIB_PATH"]), n, "walkers: %.3f ms/step" % r["ms_per_step"], {k: round(v, 3) for k, v in r["rank0_stage_ms"].items()})
PY
done
