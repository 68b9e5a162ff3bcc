"""Per-phase budget of vag_flux_grid_kernel<false, 0, false, 512, false> from the PMC passes of profiles/flux_phase_budget.sh:
counters per launch for the product build and for the builds that omit one phase; a phase's share = product - omitted.
usage: python3 profiles/flux_phase_budget.py gpurun_out/prof_budget > profiles/rNN_flux_phase_budget.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = "vag_flux_grid_kernel<false, 0, false, 512, false>"
NAMES = {"product": "product build", "abl1": "without A1 (boundary spectra)", "abl2": "without B (interpolation + exp2)",
         "abl4": "without the bracket lookup", "abl8": "without the EAT logarithms"}
rows = {}
for tag in NAMES:
    hits = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    acc, disp = defaultdict(float), set()
    for r in csv.DictReader(open(hits[0])):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add(r.get("Dispatch_Id"))
    n = max(len(disp), 1)
    rows[tag] = {k: v / n for k, v in acc.items()}
    rows[tag]["launches"] = n
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
print(f"{KERNEL} on the bench batch (512 C2 models), counters per launch (rocprofv3 --pmc, one pass per build)")
print(f"{'build':36s} " + " ".join(f"{c.replace('SQ_', '').replace('GRBM_', ''):>15s}" for c in cols))
for tag, d in rows.items():
    print(f"{NAMES[tag]:36s} " + " ".join(f"{d.get(c, float('nan')):15.4g}" for c in cols))
if "product" in rows:
    p = rows["product"]
    rows_total = 512 * 4096.0
    print(f"\nper (theta, phi) row of one workgroup (512 x 4096 rows per launch): VALU {p['SQ_INSTS_VALU'] / rows_total:.0f}, SALU "
          f"{p['SQ_INSTS_SALU'] / rows_total:.0f}, LDS {p['SQ_INSTS_LDS'] / rows_total:.0f} wave-instructions")
    print("VALU pipe: %.1f %% of the SIMD cycles at 4 cycles per wave64 instruction (SQ_INSTS_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs))"
          % (100 * p["SQ_INSTS_VALU"] * 4 / (p["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    print("a wavefront's cycles: %.0f %% issuing, %.0f %% parked at s_waitcnt / barriers, %.0f %% ready but not issued"
          % (100 * p["SQ_ACTIVE_INST_ANY"] / p["SQ_WAVE_CYCLES"], 100 * p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"],
             100 * p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"]))
    print("\nphase shares by omission (product minus the build without the phase):")
    print(f"{'phase':36s} {'VALU instr':>12s} {'share':>7s} {'SALU instr':>12s} {'LDS instr':>12s} {'GUI_ACTIVE':>12s} {'share':>7s}")
    for tag in ("abl1", "abl2", "abl4", "abl8"):
        if tag not in rows:
            continue
        d = rows[tag]
        dv, ds, dl, dg = (p[c] - d[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"))
        print(f"{NAMES[tag][8:]:36s} {dv:12.4g} {100 * dv / p['SQ_INSTS_VALU']:6.1f}% {ds:12.4g} {dl:12.4g} {dg:12.4g} "
              f"{100 * dg / p['GRBM_GUI_ACTIVE']:6.1f}%")
