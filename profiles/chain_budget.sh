#!/bin/bash
# Developer aid (GPU box): cycle stamps of the two latency-bound stages of a single call -- the adaptive grid (one wavefront per
# model) and the blast-wave ODE (one integrator wavefront per 64 rows) -- from builds with -DVAG_GRID_STAMPS / -DVAG_DYN_STAMPS:
#   profiles/build_variant.sh gridstamps -DVAG_GRID_STAMPS; profiles/build_variant.sh dynstamps -DVAG_DYN_STAMPS=2; profiles/build_variant.sh dynstamps1 -DVAG_DYN_STAMPS=1
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "== vag_grid_kernel, sections of one model (cycles at ~2.4 GHz; second call of each case)"
VAG_LIB_PATH=$R/variants/libvag_gridstamps.so python3 $R/profiles/grid_stamps.py 2>&1 | grep -v "^$"
echo "== vag_dynamics_fast_kernel, attempt loop of wavefront 0 (STAMPS=1: totals only; STAMPS=2: per-attempt body and saver waits, each stamp costs ~100 cycles)"
VAG_LIB_PATH=$R/variants/libvag_dynstamps1.so python3 $R/profiles/dyn_stamps.py 2>&1 | grep -v "^$"
VAG_LIB_PATH=$R/variants/libvag_dynstamps.so python3 $R/profiles/dyn_stamps.py 2>&1 | grep -v "^$"
