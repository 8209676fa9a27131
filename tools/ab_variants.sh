#!/bin/bash
# Build A/B variants of the library side by side (never touching tracked sources): one .so per set of -D flags under
# scarlet_amd/csrc/variants/ (git-ignored like every .so; they travel to the GPU box), selected at run time with
# SCARLET_LIB_PATH.   usage: tools/ab_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
set -e
cd "$(dirname "$0")/../scarlet_amd/csrc"
mkdir -p variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../../include $flags scarlet_hip.hip \
      -o variants/lib_$name.so -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" || true
  python3 ../../tools/check_reentry_abi.py variants/lib_$name.so
done
