set -x
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r05_gputest_final.log; tail -3 gpurun_out/r05_gputest_final.log
profiles/run_profile.sh r05 > gpurun_out/r05_run_profile.log 2>&1; tail -5 gpurun_out/r05_run_profile.log
profiles/pmc_busy.sh r05 > gpurun_out/r05_pmc_busy.txt 2>&1
python profiles/parity_report.py > gpurun_out/r05_parity_report.txt 2>&1; tail -5 gpurun_out/r05_parity_report.txt
python profiles/latency_probe.py > gpurun_out/r05_latency_probe.txt 2>&1; tail -5 gpurun_out/r05_latency_probe.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err ) 2>&1 | tail -3
tail -c 200 gpurun_out/r05_bench_final.json
