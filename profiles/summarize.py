"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("vag::", "").replace("(anonymous namespace)::", "")[:48]


p = find("stats", "*kernel_stats.csv")
print(f"== rocprofv3 --kernel-trace --stats ({tag}) ==")
if p:
    rows = list(csv.DictReader(open(p)))
    print(f"{'kernel':48s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):48s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
else:
    print("kernel_stats.csv not found")

# The stats table averages a kernel over ALL its launches; the bench line also launches the headline instantiation for its single-model
# latency leg (a 1-model grid, ~0.06 ms), so the launches are listed per launch shape from the kernel trace: the batch launches are the
# ones bench.py's roofline.ms_per_launch refers to.
p = find("stats", "*kernel_trace.csv")
if p:
    shapes = defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "vag_flux_grid_kernel<false, 0, false, 512" in r["Kernel_Name"]:
            shapes[(int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("\n== vag_flux_grid_kernel<false, 0, false, 512, false> by launch shape (workgroups x models): launches, average ms, min - max ==")
    for shape, d in sorted(shapes.items(), reverse=True):
        print(f"   {shape[0]:>6d} x {shape[1]:<5d} {len(d):>4d} launches  avg {sum(d) / len(d):8.3f} ms   {min(d):.3f} - {max(d):.3f}"
              + ("   (all but the first, which pays the cold caches: avg %.3f ms)" % (sum(d[1:]) / (len(d) - 1)) if len(d) > 2 else ""))

for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    p = find(sub, "*counter_collection.csv")
    print(f"\n== PMC pass {sub} ==")
    if not p:
        print("counter_collection.csv not found")
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    for k, d in acc.items():
        vals = "  ".join(f"{c}={v / max(cnt[k], 1):.4g}" for c, v in sorted(d.items()))
        print(f"{k:48s} dispatches={cnt[k]:<4d} per-dispatch: {vals}")
print("\nnotes: FETCH_SIZE/WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
      "(MI355X_MICROARCH.md, HBM section) -- doubled where quoted in DESIGN.md.")
