#!/bin/bash
# Developer aid (GPU box): LDS / wait / VALU counters of one kernel (KERNEL, default the tabulated-SSC row-per-lane grid pass) on the
# C5 ensemble for the product library and every library under variants/.   profiles/pmc_rows.sh [members]
R=${GRAFT_REPO_ROOT:-/root/repo}
export KERNEL=${KERNEL:-grid_rows_kernel<2>}
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  [ -f "$f" ] || continue
  echo "== $(basename $f)"
  ENSEMBLE=${ENSEMBLE:-c5} VAG_LIB_PATH=$f $R/profiles/pmc_c3_lds.sh ${1:-1024} 2>&1 | grep -v "^$"
done
