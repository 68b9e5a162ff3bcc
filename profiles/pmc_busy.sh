#!/bin/bash
# VALU-pipe busy fraction of the kernels bench.py's rooflines name, from rocprofv3 --pmc passes (counters only: --kernel-trace is the
# one trace domain combined with --pmc).  busy = SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); the same passes
# give the VALU instructions per launch.   profiles/pmc_busy.sh <tag>   ->  gpurun_out/pmc_busy_<tag>/valu_busy.json
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_busy_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/c2" -o c2 -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-walkers > "$OUT/c2.log" 2>&1
ENSEMBLE=c5 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/c5" -o c5 -- python3 "$R/profiles/ssc_ensemble.py" 1024 1 > "$OUT/c5.log" 2>&1
ENSEMBLE=c3 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/c3" -o c3 -- python3 "$R/profiles/ssc_ensemble.py" 512 1 > "$OUT/c3.log" 2>&1
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/c4" -o c4 -- python3 "$R/profiles/trace_walkers.py" 1024 > "$OUT/c4.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
want = {"c2": [("vag_flux_grid_kernel<C2>", "vag_flux_grid_kernel<false, 0, false, 512")],
        "c5": [("vag_flux_grid_rows_kernel<1><C5>", "vag_flux_grid_rows_kernel<1>"), ("vag_flux_grid_rows_kernel<2><C5>", "vag_flux_grid_rows_kernel<2>")],
        "c3": [("vag_ic_photon_kernel<C3>", "vag_ic_photon_kernel"), ("vag_flux_grid_rows_kernel<1><C3>", "vag_flux_grid_rows_kernel<1>"),
               ("vag_flux_grid_rows_kernel<2><C3>", "vag_flux_grid_rows_kernel<2>"), ("vag_dynamics_pair_kernel<C3>", "vag_dynamics_pair_kernel")],
        # metric M2 (r05): the 1024-walker likelihood's flux and ODE kernels
        "c4": [("vag_flux_fit_rows_kernel<C4>", "vag_flux_fit_rows_kernel<0, 4, false"), ("vag_dynamics_fast_kernel<C4>", "vag_dynamics_fast_kernel")]}
res, detail = {}, {}
for sub, kernels in want.items():
    files = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    for label, pat in kernels:
        acc = collections.defaultdict(float); n = 0
        for r in rows:
            if pat not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n += r["Counter_Name"] == "GRBM_GUI_ACTIVE"
        if not n:
            continue
        simd_cycles = acc["GRBM_GUI_ACTIVE"] / 8 * 1024
        res[label] = round(acc["SQ_INSTS_VALU"] * 4 / simd_cycles, 4)
        detail[label] = {"launches": n, "ms_per_launch": acc["GRBM_GUI_ACTIVE"] / n / 8 / 2.4e6, "valu_insts_per_launch": acc["SQ_INSTS_VALU"] / n,
                         "valu_busy_by_active_cycles": acc["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles,  # (counts a v_rcp_f64 as its 16 cycles, not as 4)
                         "lds_insts_per_launch": acc["SQ_INSTS_LDS"] / n, "salu_insts_per_launch": acc["SQ_INSTS_SALU"] / n,
                         "waves_per_simd": acc["SQ_WAVE_CYCLES"] * 4 / simd_cycles}
res["_detail"] = detail
res["_command"] = ("rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-walkers | "
                   "ENSEMBLE=c5 ... profiles/ssc_ensemble.py 1024 1 | ENSEMBLE=c3 ... profiles/ssc_ensemble.py 512 1   (profiles/pmc_busy.sh %s)" % tag)
json.dump(res, open(out + "/valu_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
