#!/bin/bash
# Profile the bench workload on the GPU box: per-kernel time (--kernel-trace --stats) and, in SEPARATE passes,
# SQ / TCC counters (--pmc is never combined with sys/hip/hsa traces).  Usage: profiles/run_profile.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
PMC_ARGS="$ARGS --no-walkers"  # counters are collected for the headline workload only (every kernel runs serialised under --pmc)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o "$TAG" -- python3 "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- python3 "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- python3 "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
python3 "$REPO/profiles/summarize.py" "$OUT" "$TAG" > "$OUT/summary_$TAG.txt" 2>&1
cat "$OUT/summary_$TAG.txt"
