#!/bin/bash
# developer builds of the tall kernel's TU with a -D switch: bash devtools/build_tall.sh NAME "-DLC_TALL_ABL=1 ..."
#   -> devtools/variants/liblc_NAME.so   (all other objects from lidarcrafter_amd/build)
set -e
cd "$(dirname "$0")/.."
mkdir -p devtools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c lidarcrafter_amd/csrc/conv_f16x2_tall.hip -o devtools/variants/tall_$1.o 2>/dev/null
objs=$(ls lidarcrafter_amd/build/*.o | grep -v "_p1.o" | grep -v conv_f16x2_tall.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devtools/variants/liblc_$1.so $objs devtools/variants/tall_$1.o
echo built devtools/variants/liblc_$1.so
