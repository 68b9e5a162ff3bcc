for f in variants/libvag_r*.so; do echo "== $f"; VAG_LIB_PATH=$PWD/$f python profiles/debug/walker_stage_probe.py 2>&1 | grep walkers; done
ENS="c5 c3" bash profiles/quick_ens.sh 2>&1 | grep -v "libvegasafterglow_amd" 
