python profiles/series_probe.py 8192 2>&1 | grep walkers
VAG_LIB_PATH=$PWD/variants/libvag_wt.so python profiles/series_probe.py 8192 2>&1 | grep fitwave | tail -600 > gpurun_out/fit_wave_times.txt
wc -l gpurun_out/fit_wave_times.txt
