# Developer aid (GPU box): vag_dynamics_pair_kernel of the product library and of every variants/libvag_*.so -- per launch on the 512-model
# configs[2] batch and on ONE model (stage times), GPU tests with the product library first
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
for f in $R/vegasafterglow_amd/libvegasafterglow_amd.so $R/variants/libvag_*.so; do
  for nb in 512 1; do
    echo "$(basename $f) nb=$nb: $(ENSEMBLE=c3 VAG_LIB_PATH=$f python3 profiles/ssc_ensemble.py $nb 3 2>&1 | grep '^rep 3' | sed 's/nan=0//')"
  done
done
