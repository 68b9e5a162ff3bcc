#!/bin/bash
# Extra SQ pass: where the flux kernel's wave-cycles go (LDS issue stalls, bank conflicts, scalar / branch issue).
set -u
TAG=${1:-lds}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-walkers"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
          --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_lds.log" 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
          --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_sq2.log" 2>&1
tail -3 "$OUT/bench_pmc_lds.log"
python3 "$REPO/profiles/summarize.py" "$OUT" "$TAG" 2>&1 | grep -A2 "flux_grid\|PMC pass"
