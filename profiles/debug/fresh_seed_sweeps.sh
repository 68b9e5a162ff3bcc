# Developer aid (GPU box): the randomised checks on draws NO test or earlier sweep has seen (SWEEP_SEED), against the CPU checker;
# the criterion printed per sweep is the reference's own golden contract (|d| <= 2e-3 |ref| + 1e-2 peak).
# usage: bash profiles/debug/fresh_seed_sweeps.sh "<seed> <seed> ..." > gpurun_out/r05_fresh_seed_sweeps.txt
for seed in ${1:-101 202}; do
  export SWEEP_SEED=$seed
  echo "#### SWEEP_SEED=$seed"
  echo "## prior_sweep.py 256"
  timeout 1500 python profiles/debug/prior_sweep.py 256 2>&1 | grep -v amdgpu | tail -4
  echo "## prior_sweep_ssc.py 40"
  timeout 1500 python profiles/debug/prior_sweep_ssc.py 40 2>&1 | grep -v amdgpu | tail -3
  echo "## prior_sweep_rs_ssc.py 24"
  timeout 1500 python profiles/debug/prior_sweep_rs_ssc.py 24 2>&1 | grep -v amdgpu | tail -2
  for mode in spread nonaxi magnetar; do
    echo "## SWEEP_MODE=$mode prior_sweep_ssc.py 30"
    SWEEP_MODE=$mode timeout 1500 python profiles/debug/prior_sweep_ssc.py 30 2>&1 | grep -v amdgpu | tail -3
  done
done
