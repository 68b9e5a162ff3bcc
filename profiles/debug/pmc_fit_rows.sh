#!/bin/bash
# Developer aid (GPU box): VALU / LDS / wait counters of the walker likelihood's kernels at N walkers (default 8192), two --pmc passes
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/pmc_fit
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
N=${1:-8192}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT/a" -o f -- python3 "$REPO/profiles/series_probe.py" $N > "$OUT/log_a.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$OUT/b" -o f -- python3 "$REPO/profiles/series_probe.py" $N > "$OUT/log_b.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    f = glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
    for k, v in acc.items():
        n = max(cnt[k], 1)
        g = v["GRBM_GUI_ACTIVE"] / n / 8
        if g / 2.4e6 < 0.05: continue
        print("%-48s launches %3d  %.3f ms  " % (k, n, g / 2.4e6) + "  ".join("%s %.3f" % (c.replace("SQ_", ""), x / n / (g * 1024)) for c, x in sorted(v.items()) if c != "GRBM_GUI_ACTIVE")
              + ("  VALU busy %.3f" % (v["SQ_INSTS_VALU"] * 4 / n / (g * 1024)) if "SQ_INSTS_VALU" in v else ""))
PY
