# Developer aid (GPU box): the long randomised checks (an hour of host time for the checker), into gpurun_out/r04_sweep_long.txt
exec > gpurun_out/r04_sweep_long.txt 2>&1
echo "## prior_sweep.py 1024 (C4 prior box, series + grid forms)"
timeout 3000 python profiles/debug/prior_sweep.py 1024 2>&1 | grep -v amdgpu | tail -3
echo "## prior_sweep_rs_ssc.py 60"
timeout 3000 python profiles/debug/prior_sweep_rs_ssc.py 60 2>&1 | grep -v amdgpu | tail -6
echo "## prior_sweep_res.py 40 (seed of the script; draws 24.. are new)"
timeout 3000 python profiles/debug/prior_sweep_res.py 40 2>&1 | grep -v amdgpu | tail -18
