# Developer aid (GPU box): the randomised checks with the round's final library, into gpurun_out/r05_sweep_final.txt
exec > gpurun_out/r05_sweep_final.txt 2>&1
echo "## rows_stress.py (row-per-lane kernels against the row-per-wavefront / workgroup kernels on random models)"
timeout 900 python profiles/debug/rows_stress.py 2>&1 | grep -v amdgpu | tail -25
echo "## prior_sweep.py 256 (C4 prior box, series + grid forms, against the checker)"
timeout 1500 python profiles/debug/prior_sweep.py 256 2>&1 | grep -v amdgpu | tail -6
echo "## prior_sweep_ssc.py 40"
timeout 1500 python profiles/debug/prior_sweep_ssc.py 40 2>&1 | grep -v amdgpu | tail -6
echo "## prior_sweep_rs_ssc.py 24"
timeout 1500 python profiles/debug/prior_sweep_rs_ssc.py 24 2>&1 | grep -v amdgpu | tail -8
echo "## SWEEP_MODE=spread prior_sweep_ssc.py 30"
SWEEP_MODE=spread timeout 1500 python profiles/debug/prior_sweep_ssc.py 30 2>&1 | grep -v amdgpu | tail -4
echo "## SWEEP_MODE=nonaxi prior_sweep_ssc.py 30"
SWEEP_MODE=nonaxi timeout 1500 python profiles/debug/prior_sweep_ssc.py 30 2>&1 | grep -v amdgpu | tail -4
echo "## SWEEP_MODE=magnetar prior_sweep_ssc.py 30"
SWEEP_MODE=magnetar timeout 1500 python profiles/debug/prior_sweep_ssc.py 30 2>&1 | grep -v amdgpu | tail -4
