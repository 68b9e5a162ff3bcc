#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes, counters only) of the C3 (or ENSEMBLE=c5) ensemble's kernels.
# FETCH_SIZE is reported as read by the counter (KiB); on gfx950 coalesced streams show at half their bytes (x2, MI355X_MICROARCH.md).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_c3_traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ENSEMBLE=${ENSEMBLE:-c3} rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -o c3 -- python3 "$REPO/profiles/ssc_ensemble.py" ${1:-128} 1 > "$OUT/log_$C.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(sys.argv[1] + "/" + C + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != C: continue
        k = r["Kernel_Name"][:56]
        acc[k][C] += float(r["Counter_Value"]); cnt[k][C] += 1
print("%-58s %9s %14s %14s" % ("kernel", "launches", "FETCH KiB/launch", "WRITE KiB/launch"))
for k, v in sorted(acc.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    n = max(cnt[k]["FETCH_SIZE"], 1)
    if (v["FETCH_SIZE"] + v["WRITE_SIZE"]) / n < 1024: continue
    print("%-58s %9d %14.4g %14.4g" % (k, n, v["FETCH_SIZE"] / n, v["WRITE_SIZE"] / max(cnt[k]["WRITE_SIZE"], 1)))
PY
