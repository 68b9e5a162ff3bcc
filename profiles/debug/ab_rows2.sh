for rep in 1 2 3; do for v in r1 r3; do echo "== $v"; VAG_LIB_PATH=$PWD/variants/libvag_$v.so python profiles/debug/walker_stage_probe.py 1024 8192 2>&1 | grep walkers; done; done
