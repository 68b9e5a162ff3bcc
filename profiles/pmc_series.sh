#!/bin/bash
# SQ counters of the walker likelihood's kernels (1024 C4 walkers): wave-instructions, wave-cycles, waits.  Counters only (no sys traces).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_series
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT" -o ser -- python3 "$REPO/profiles/series_probe.py" ${1:-1024} > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "launches", n, {c: f"{x / n:.4g}" for c, x in v.items()})
PY
