#!/bin/bash
# Developer aid (GPU box): SQ counters of the top-hat batches' kernels (C1a / C1b at 1024 and 4096 models)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/pmc_c1a
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -o f -- python3 "$REPO/profiles/debug/tophat_batch_stages.py" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# group by (kernel, grid size) since batches differ
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][-44:], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in sorted(acc.items()):
    n = max(cnt[k], 1); g = v["GRBM_GUI_ACTIVE"] / n / 8
    if g / 2.4e6 < 0.15: continue
    print("%-44s grid %-9s x%2d  %.3f ms  waves/SIMD %.2f  VALU busy %.3f  waves %.0f" % (k[0], k[1], n, g / 2.4e6, v["SQ_WAVE_CYCLES"] * 4 / n / (g * 1024), v["SQ_INSTS_VALU"] * 4 / n / (g * 1024), v["SQ_WAVES"] / n))
PY
