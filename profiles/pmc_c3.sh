#!/bin/bash
# SQ counters of the C3 ensemble's kernels (forward + reverse shock, SSC + KN).  Counters only (no sys traces).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_c3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ENSEMBLE=${ENSEMBLE:-c3} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT" -o c3 -- python3 "$REPO/profiles/ssc_ensemble.py" ${1:-128} 1 > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    g = v.get("GRBM_GUI_ACTIVE", 0) / n / 8
    if g < 2e5: continue
    print("%-50s launches %3d  ms %.2f  waves %.3g  VALU busy %.2f  waves/SIMD %.2f  VALU instr %.3g" % (k, n, g / 2.4e6, v["SQ_WAVES"] / n,
          4 * v["SQ_ACTIVE_INST_VALU"] / n / (g * 1024), 4 * v["SQ_WAVE_CYCLES"] / n / (g * 1024), v["SQ_INSTS_VALU"] / n))
PY
