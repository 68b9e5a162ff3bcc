#!/bin/bash
# Developer aid (GPU box): time the C2 bench step for every library under variants/
for f in variants/libvag_*.so; do
  echo "== $f"
  VAG_LIB_PATH=$PWD/$f python bench.py --no-cpu-baseline --no-walkers --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_ms'])"
done
