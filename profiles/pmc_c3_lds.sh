#!/bin/bash
# LDS / wait counters of one kernel (KERNEL=substring of its name, default the SSC spectrum kernel) of the C3 (or ENSEMBLE=c5) ensemble.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_c3_lds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ENSEMBLE=${ENSEMBLE:-c3} rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT/a" -o c3 -- python3 "$REPO/profiles/ssc_ensemble.py" ${1:-128} 1 > "$OUT/log_a.txt" 2>&1
ENSEMBLE=${ENSEMBLE:-c3} rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT/b" -o c3 -- python3 "$REPO/profiles/ssc_ensemble.py" ${1:-128} 1 > "$OUT/log_b.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
KERNEL = os.environ.get("KERNEL", "ic_photon")
for sub in ("a", "b"):
    f = glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n += 1
    g = acc["GRBM_GUI_ACTIVE"] / n / 8
    print(KERNEL + ", per launch (%d launches): %.2f ms" % (n, g / 2.4e6))
    for k, v in sorted(acc.items()):
        print("  %-24s %.4g   per SIMD-cycle %.3f" % (k, v / n, v / n / (g * 1024)))
PY
