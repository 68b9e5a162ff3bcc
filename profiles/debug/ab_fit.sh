for w in default 1 2 4; do
  echo "== W=$w"; if [ $w = default ]; then unset VAG_FIT_WAVES_PER_BLOCK; else export VAG_FIT_WAVES_PER_BLOCK=$w; fi
  python profiles/debug/walker_stage_probe.py 64 128 512 1024 2048 8192 2>&1 | grep walkers
done
unset VAG_FIT_WAVES_PER_BLOCK
python -m pytest tests -m gpu -x -q -k "fit or walker or loglike or shard or series" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
