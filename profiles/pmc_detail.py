"""Every counter of a pmc_c3.sh run for the kernels whose name contains argv[1] (per launch), with the derived shares.
usage: python3 profiles/pmc_detail.py photon [gpurun_out/pmc_c3/c3_counter_collection.csv]"""
import collections
import csv
import sys

pat = sys.argv[1]
f = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc_c3/c3_counter_collection.csv"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
for k, v in acc.items():
    if pat not in k:
        continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c, x in sorted(v.items()):
        print("   %-24s %.4g" % (c, x / n))
    g = v.get("GRBM_GUI_ACTIVE", 0) / n / 8
    if g and "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"] / n
        print("   ms %.2f  VALU busy %.3f  waves/SIMD %.2f  wait_any %.3f  wait_inst_any %.3f  VALU issue share of a wave %.3f" % (
            g / 2.4e6, 4 * v["SQ_ACTIVE_INST_VALU"] / n / (g * 1024), 4 * wc / (g * 1024), v["SQ_WAIT_ANY"] / n / wc,
            v["SQ_WAIT_INST_ANY"] / n / wc, v["SQ_ACTIVE_INST_VALU"] / n / wc))
