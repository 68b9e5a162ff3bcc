# Developer aid (GPU box): more randomised checks with the final library, into gpurun_out/r04_sweep_final2.txt
exec > gpurun_out/r04_sweep_final2.txt 2>&1
echo "## SWEEP_MODE=magnetar prior_sweep_ssc.py 30"
SWEEP_MODE=magnetar timeout 1500 python profiles/debug/prior_sweep_ssc.py 30 2>&1 | grep -v amdgpu | tail -3
echo "## SWEEP_MODE=spread prior_sweep_ssc.py 72   (other draws than the 30-draw run: the generator is sequential)"
SWEEP_MODE=spread timeout 2400 python profiles/debug/prior_sweep_ssc.py 72 2>&1 | grep -v amdgpu | tail -3
echo "## prior_sweep_ssc.py 100"
timeout 2400 python profiles/debug/prior_sweep_ssc.py 100 2>&1 | grep -v amdgpu | tail -3
