R=${GRAFT_REPO_ROOT:-/root/repo}
for t in -1 3 7 11; do python $R/profiles/debug/uniform_walkers_probe.py 8192 $t 2>&1 | grep walkers; done
cd /tmp && export TMPDIR=/tmp
for t in -1 3; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/upmc$t -o f -- python3 $R/profiles/debug/uniform_walkers_probe.py 8192 $t > /dev/null 2>&1
python3 - $R/gpurun_out/upmc$t <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1); g = v["GRBM_GUI_ACTIVE"] / n / 8
    if g / 2.4e6 < 0.3: continue
    print("%-40s %.3f ms  waves/SIMD %.2f  VALU busy %.3f  waves %.0f  wait_any/wave_cycles %.2f" % (k, g / 2.4e6, v["SQ_WAVE_CYCLES"] * 4 / n / (g * 1024), v["SQ_INSTS_VALU"] * 4 / n / (g * 1024), v["SQ_WAVES"] / n, v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
done
