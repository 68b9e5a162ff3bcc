#!/bin/bash
# Profile the bench workload on the GPU box: per-kernel time (--kernel-trace --stats) of the DRIVER'S OWN command
# (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) and, in SEPARATE passes, SQ / TCC counters of the headline workload alone (--pmc is never combined with sys/hip/hsa traces;
# every kernel runs serialised under --pmc, so the secondary legs are left out there).  Usage: profiles/run_profile.sh <tag>
set -u
TAG=${1:-r05}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STATS_CMD="python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5"
PMC_CMD="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-walkers"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- $STATS_CMD > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o "$TAG" -- $PMC_CMD > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- $PMC_CMD > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- $PMC_CMD > "$OUT/bench_pmc_write.log" 2>&1
{
  echo "# stats pass:  rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5     (the driver's command, flags included; its JSON line: bench_stats.json)"
  echo "# PMC passes:  rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-walkers"
  python3 "$REPO/profiles/summarize.py" "$OUT" "$TAG"
} > "$OUT/summary_$TAG.txt" 2>&1
cat "$OUT/summary_$TAG.txt" | head -60
tail -c 600 "$OUT/bench_stats.json"
