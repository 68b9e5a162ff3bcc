#!/bin/bash
# Developer aid (GPU box): C5 / C3 ensemble per-kernel times and parity for the product library and every library under variants/
#   ENS="c5 c3" NB_c5=1024 NB_c3=512 profiles/quick_ens.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  [ -f "$f" ] || continue
  for ens in ${ENS:-c5}; do
    nb=1024; [ $ens = c3 ] && nb=512
    echo "== $(basename $f) $ens x $nb"
    ENSEMBLE=$ens VAG_LIB_PATH=$f python3 $R/profiles/ssc_ensemble.py $nb 2 ${CHECK-check} 2>&1 | grep -v "^{" | tail -7
    ENSEMBLE=$ens VAG_LIB_PATH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qe -o qe -- python3 $R/profiles/ssc_ensemble.py $nb 2 > /dev/null 2>&1
    python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/qe/qe_kernel_stats.csv")))
for r in rows[:6]:
    print("   %-70s calls %s avg %.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6))
PY
  done
done
