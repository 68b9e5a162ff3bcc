#!/bin/bash
# Per-phase budget of vag_flux_grid_kernel on the C2 bench batch (GPU box): SQ counters of the product library and of the builds that
# omit one phase each (profiles/build_variant.sh ablN -DVAG_FLUX_ABLATE=N: 1 boundary spectra, 2 interpolation, 4 bracket lookup,
# 8 EAT logs), every run its own rocprofv3 --pmc pass.  profiles/flux_phase_budget.py turns the CSVs into the table.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_budget
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for f in product abl1 abl2 abl4 abl8; do
  if [ $f = product ]; then lib=$REPO/vegasafterglow_amd/libvegasafterglow_amd.so; else lib=$REPO/variants/libvag_$f.so; fi
  [ -f "$lib" ] || continue
  export VAG_LIB_PATH=$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d "$OUT/$f" -o "$f" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-walkers > "$OUT/$f.log" 2>&1
done
python3 "$REPO/profiles/flux_phase_budget.py" "$OUT"
