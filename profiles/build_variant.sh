#!/bin/bash
# Developer aid: build a kernel variant of the engine next to the product library.
#   profiles/build_variant.sh NAME [-DMACRO ...]   ->  variants/libvag_NAME.so   (run with VAG_LIB_PATH=variants/libvag_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude "$@" vegasafterglow_amd/csrc/vag_capi.hip -o variants/libvag_$name.so
echo "built variants/libvag_$name.so"
