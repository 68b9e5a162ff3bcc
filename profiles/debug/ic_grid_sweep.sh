# Developer aid (GPU box): the SSC spectrum kernel's persistent launch at several grid sizes, product library and variants/libvag_*.so
for lib in vegasafterglow_amd/libvegasafterglow_amd.so variants/libvag_*.so; do
for G in ${GRIDS:-4096 16384 65536 262144}; do
  echo "lib $lib G $G: $(ENSEMBLE=c3 VAG_IC_GRID=$G VAG_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 profiles/ssc_ensemble.py 512 2 2>&1 | grep '^rep 2')"
done; done
