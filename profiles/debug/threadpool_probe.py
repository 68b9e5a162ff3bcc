"""The thread-pool leg of bench.py on its own (serialised against coalesced), for a few pool sizes.
usage: python3 profiles/debug/threadpool_probe.py [threads ...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vegasafterglow_amd import _lib
lib = _lib.load(); h = C.c_void_p(); _lib.check(lib.vag_ctx_create(0, C.byref(h)))
for n in [int(a) for a in sys.argv[1:]] or [32]:
    for w in [int(x) for x in os.environ.get("WAITS", "50").split()]:
        r = bench.threadpool_bench(lib, h, _lib, n_threads=n, wait_us=w)
        print("threads %d wait_us %d: serialised %.0f/s coalesced %.0f/s (x%.1f) mean batch %.1f" % (n, w, r["serialised"]["walker_steps_per_s"],
              r["coalesced"]["walker_steps_per_s"], r["coalesced_over_serialised"], r["coalesced"]["mean_batch"]))
