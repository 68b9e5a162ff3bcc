#!/bin/bash
# PMC view of the forward-shock ODE kernels at a large walker step: plain kernel against the lane-refill kernel (round 6).
#   profiles/pmc_dyn.sh <tag> [nwalkers]   ->  gpurun_out/pmc_dyn_<tag>/summary.txt
TAG=${1:-r06}
NW=${2:-8192}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_dyn_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for mode in 0 1; do
  VAG_DYN_REFILL=$mode rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/m$mode" -o dyn -- python3 "$R/profiles/trace_walkers.py" $NW > "$OUT/m$mode.log" 2>&1
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
for mode in (0, 1):
    files = glob.glob(out + f"/m{mode}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("mode", mode, "no counters"); continue
    rows = list(csv.DictReader(open(files[0])))
    for pat in ("vag_dynamics_fast_kernel", "vag_dynamics_refill_kernel", "vag_dyn_prep_kernel", "vag_flux_fit_rows_kernel", "vag_cells_kernel", "vag_grid_kernel"):
        acc = collections.defaultdict(float); n = 0
        for r in rows:
            if pat not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n += r["Counter_Name"] == "GRBM_GUI_ACTIVE"
        if not n: continue
        simd_cycles = acc["GRBM_GUI_ACTIVE"] / 8 * 1024
        print(f"VAG_DYN_REFILL={mode} {pat}: launches {n}, GUI_ACTIVE/8 per launch {acc['GRBM_GUI_ACTIVE']/n/8:.0f} cycles, VALU insts/launch {acc['SQ_INSTS_VALU']/n:.3e}, "
              f"valu busy (insts x4) {acc['SQ_INSTS_VALU']*4/simd_cycles:.3f}, by active cycles {acc['SQ_ACTIVE_INST_VALU']*4/simd_cycles:.3f}, waves/SIMD {acc['SQ_WAVE_CYCLES']*4/simd_cycles:.2f}, "
              f"wait_any/wave_cycles {acc['SQ_WAIT_INST_ANY']/max(acc['SQ_WAVE_CYCLES'],1):.3f}, wait_lds/wave_cycles {acc['SQ_WAIT_INST_LDS']/max(acc['SQ_WAVE_CYCLES'],1):.3f}, "
              f"LDS insts {acc['SQ_INSTS_LDS']/n:.3e}, SALU insts {acc['SQ_INSTS_SALU']/n:.3e}")
PY
