# Developer aid (GPU box): the product library and every library under variants/ on the four workloads, inside one call
for f in vegasafterglow_amd/libvegasafterglow_amd.so variants/libvag_*.so; do
  echo "== $f"
  VAG_LIB_PATH=$PWD/$f python bench.py --no-cpu-baseline --no-walkers --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', round(d['value']), round(d['stage_ms']['sync_flux'],3))"
  VAG_LIB_PATH=$PWD/$f python profiles/debug/walker_stage_probe.py 1024 8192 2>&1 | grep walkers
  for ens in c5 c3; do nb=1024; [ $ens = c3 ] && nb=512; ENSEMBLE=$ens VAG_LIB_PATH=$PWD/$f python profiles/ssc_ensemble.py $nb 2 2>&1 | grep "rep 2" | cut -c1-120; done
done
