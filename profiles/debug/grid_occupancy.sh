R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_grid_occ; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -o g -- python3 $R/profiles/debug/tophat_batch_stages.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "vag_grid_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows: by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); by[r["Dispatch_Id"]]["grid"] = r.get("Grid_Size")
seen = set()
for d, v in by.items():
    key = (v.get("SQ_WAVES"),)
    g = v["GRBM_GUI_ACTIVE"] / 8
    print("waves %6d  ms %.3f  resident waves per SIMD %.2f  VALU busy %.3f" % (v["SQ_WAVES"], g / 2.4e6, v["SQ_WAVE_CYCLES"] * 4 / (g * 1024), v["SQ_INSTS_VALU"] * 4 / (g * 1024)))
PY
