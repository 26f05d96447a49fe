#!/bin/bash
# Developer builds of the library with fewer products in the f16x2 split (accuracy measurement only):
#   devtools/variants/liblc_terms{1,3,5}.so   (1: wh*xh only; 3: + wl*xh; 5: + wh*xl)
# Usage: bash devtools/build_variants.sh   (after python -m lidarcrafter_amd.build)
set -e
cd "$(dirname "$0")/.."
mkdir -p devtools/variants
OBJ=lidarcrafter_amd/build
for t in 1 3 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLC_F16X2_TERMS=$t -c \
      lidarcrafter_amd/csrc/conv_f16x2.hip -o devtools/variants/conv_f16x2_t$t.o 2>/dev/null &
done
wait
for t in 1 3 5; do
  objs=$(ls $OBJ/*.o | grep -v conv_f16x2)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devtools/variants/liblc_terms$t.so \
      $objs devtools/variants/conv_f16x2_t$t.o
done
ls -la devtools/variants/*.so
