for p in 64 128 192 256 384 512; do
  echo "== ppb $p"; VAG_PAIRS_PER_BLOCK=$p python bench.py --no-cpu-baseline --no-walkers --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_ms']['sync_flux'], d['stage_ms']['reduce'])"
done
