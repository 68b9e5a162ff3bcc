"""Developer aid: per-call kernel timeline (start offsets, durations, gaps) from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "vag_transform_kernel"
# split into calls at the marker kernel; print the last complete call
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(starts) < 2:
    raise SystemExit("marker kernel not found twice")
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vag::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {name}")
    prev_end = e
    tot += e - s
print(f"call span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, next call starts {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us after this one")
