#!/bin/bash
# Developer aid (GPU box): the product library and every library under variants/ through the GPU tests and the bench line, in ONE call
#   TESTS=1 bash profiles/ab_libs.sh [bench args...]     -> gpurun_out/ab_<lib>.json, gpurun_out/ab_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
: > gpurun_out/ab_summary.txt
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  [ -f "$f" ] || continue
  n=$(basename $f .so)
  if [ -n "$TESTS" ]; then
    echo "== $n tests: $(VAG_LIB_PATH=$f python -m pytest tests -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -1)" >> gpurun_out/ab_summary.txt
  fi
  VAG_LIB_PATH=$f python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python3 - gpurun_out/ab_$n.json $n >> gpurun_out/ab_summary.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("== %s: no bench line (%s)" % (sys.argv[2], e)); sys.exit(0)
e = d.get("ensembles_config2_config4", {})
w = d.get("walker_steps", {})
print("== %s: headline %.0f LC/s (%.2f ms)  C3 %.0f  C5 %.0f  walkers1024 %s  tophat %s  single %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], e.get("C3_fs_rs_ssc_kn", {}).get("light_curves_per_s", 0), e.get("C5_two_component_ssc", {}).get("light_curves_per_s", 0),
    {k: (round(v, 1) if isinstance(v, float) else v) for k, v in w.items() if k in ("value", "ms_per_step", "walker_steps_per_s")},
    {k: round(v) for k, v in d.get("tophat_config0", {}).items() if isinstance(v, (int, float))},
    json.dumps(d.get("single_model_latency", {}))[:400]))
if "stage_ms" in e.get("C3_fs_rs_ssc_kn", {}): print("     C3 stages", e["C3_fs_rs_ssc_kn"].get("stage_ms_reference_names"))
if "stage_ms" in e.get("C5_two_component_ssc", {}): print("     C5 stages", e["C5_two_component_ssc"].get("stage_ms_reference_names"))
print("     stage_ms", d.get("stage_ms"))
PY
done
cat gpurun_out/ab_summary.txt
