set -x
T=${1:-r06}
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${T}_gputest_final.log; tail -3 gpurun_out/${T}_gputest_final.log
profiles/run_profile.sh $T > gpurun_out/${T}_run_profile.log 2>&1; tail -5 gpurun_out/${T}_run_profile.log
profiles/pmc_busy.sh $T > gpurun_out/${T}_pmc_busy.txt 2>&1
profiles/pmc_dyn.sh $T 8192 > gpurun_out/${T}_pmc_dyn.txt 2>&1
profiles/timeline_call.sh $T 128 > gpurun_out/${T}_timeline_128.txt 2>&1
python profiles/parity_report.py > gpurun_out/${T}_parity_report.txt 2>&1; tail -5 gpurun_out/${T}_parity_report.txt
python profiles/latency_probe.py > gpurun_out/${T}_latency_probe.txt 2>&1; tail -5 gpurun_out/${T}_latency_probe.txt
python profiles/debug/refill_probe.py 8192 4096 2048 1024 > gpurun_out/${T}_refill_probe.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err ) 2>&1 | tail -3
cp bench_detail.json gpurun_out/${T}_bench_detail.json
tail -c 300 gpurun_out/${T}_bench_final.json
