#!/bin/bash
# Kernel timeline of ONE likelihood call (and one single-model grid call): start / end of every kernel relative to the call's first kernel,
# from rocprofv3 --kernel-trace.   profiles/timeline_call.sh <tag> [nwalkers]  ->  gpurun_out/timeline_<tag>/timeline.txt
TAG=${1:-r06}
NW=${2:-128}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/timeline_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/w" -o w -- python3 "$R/profiles/trace_walkers.py" $NW > "$OUT/w.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/timeline.txt"
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/w/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# calls start with vag_fit_front_kernel; print the LAST three complete calls
starts = [i for i, r in enumerate(rows) if "vag_fit_front_kernel" in r["Kernel_Name"]]
for s0, s1 in list(zip(starts, starts[1:]))[-3:]:
    t0 = int(rows[s0]["Start_Timestamp"])
    print("call:")
    prev_end = t0
    for r in rows[s0:s1]:
        a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void vag::", "")[:60]
        print(f"  {a/1e3:9.1f} us -> {b/1e3:9.1f} us  ({(b-a)/1e3:7.1f} us, gap before {((a + t0) - prev_end)/1e3:6.1f} us)  {name}")
        prev_end = int(r["End_Timestamp"])
PY
