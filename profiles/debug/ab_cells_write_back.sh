cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for wb in "" 1; do
  if [ -n "$wb" ]; then export VAG_CELLS_WRITE_BACK=1; else unset VAG_CELLS_WRITE_BACK; fi
  echo "== write_back=${wb:-0}"
  rm -f variants/*.so
  bash profiles/debug/walker_ab.sh 2>&1 | grep -v amdgpu | grep walkers
done
