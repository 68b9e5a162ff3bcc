#!/bin/bash
# Developer aid (GPU box): time of the kernels whose name contains $1 for every library under variants/, on the C3 / C5 ensemble
#   ENSEMBLE=c3 bash profiles/quick_kernel.sh cooling
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for f in $R/variants/libvag_*.so; do
  ENSEMBLE=${ENSEMBLE:-c5} VAG_LIB_PATH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qk -o qk -- python3 $R/profiles/ssc_ensemble.py > /dev/null 2>&1
  python3 - "$f" "$1" <<PY
import csv, sys
for r in csv.DictReader(open("$R/gpurun_out/qk/qk_kernel_stats.csv")):
    if sys.argv[2] in r["Name"]:
        print("%-28s %-50s calls %s avg %.3f ms" % (sys.argv[1].split("/")[-1], r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
