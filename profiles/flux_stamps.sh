#!/bin/bash
# Developer aid (GPU box): per-phase cycle stamps of one workgroup of vag_flux_grid_kernel on the C2 bench batch
# (variants/libvag_stamps.so = profiles/build_variant.sh stamps -DVAG_FLUX_STAMPS), then the timing of every other variant.
cd "$(dirname "$0")/.."
VAG_LIB_PATH=$PWD/variants/libvag_stamps.so python bench.py --no-cpu-baseline --no-walkers --steps 1 --warmup 0 2>&1 | grep "flux wave" | head -16
for f in variants/libvag_*.so vegasafterglow_amd/libvegasafterglow_amd.so; do
  case $f in *stamps*) continue;; esac
  echo "== $f"
  VAG_LIB_PATH=$PWD/$f python bench.py --no-cpu-baseline --no-walkers --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_ms'])"
done
