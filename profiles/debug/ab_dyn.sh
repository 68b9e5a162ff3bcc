for rep in 1 2; do for v in swap0 swap1; do echo "== $v"; VAG_LIB_PATH=$PWD/variants/libvag_$v.so python profiles/series_probe.py 128 1024 8192 2>&1 | grep walkers | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['walkers'], 'ms', round(d['ms'],4), 'dynamics', round(d['dynamics'],4), 'grid', round(d['grid'],4))"; done; done
