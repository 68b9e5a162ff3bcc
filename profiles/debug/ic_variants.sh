# Developer aid (GPU box): tests with the product library, vag_ic_photon_kernel's time per launch for it and every variants/libvag_*.so
# (C3 ensemble, 512 models), and the SQ counters of the product and the r04 library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
cd /tmp && export TMPDIR=/tmp
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  ENSEMBLE=c3 VAG_LIB_PATH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qk -o qk -- python3 $R/profiles/ssc_ensemble.py 512 2 > /dev/null 2>&1
  python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open("$R/gpurun_out/qk/qk_kernel_stats.csv")):
    if "ic_photon_kernel" in r["Name"] or "ic_plan" in r["Name"]:
        print("%-32s %-40s calls %s avg %.3f ms" % (sys.argv[1].split("/")[-1], r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
cd $R
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_r04.so; do
  echo "== counters $f"; VAG_LIB_PATH=$f bash profiles/pmc_c3.sh 512 > /dev/null 2>&1; python3 profiles/pmc_detail.py ic_photon
done
