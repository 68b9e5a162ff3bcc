#!/bin/bash
# Developer aid (GPU box): per-kernel time of the C5 ensemble for every library under variants/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for f in $R/variants/libvag_*.so; do
  echo "== $f"
  ENSEMBLE=${ENSEMBLE:-c5} VAG_LIB_PATH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q5 -o q5 -- python3 $R/profiles/ssc_ensemble.py > /dev/null 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/q5/q5_kernel_stats.csv")))
for r in rows[:4]:
    print("   %-60s calls %s avg %.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6))
PY
done
