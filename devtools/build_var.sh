#!/bin/bash
# developer builds of ONE translation unit with a -D switch: bash devtools/build_var.sh NAME TU "-DLC_X=1"  -> devtools/variants/liblc_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p devtools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $3 -c lidarcrafter_amd/csrc/$2.hip -o devtools/variants/$2_$1.o 2>/dev/null
objs=$(ls lidarcrafter_amd/build/*.o | grep -v "/$2\.o\|_p1\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devtools/variants/liblc_$1.so $objs devtools/variants/$2_$1.o
echo built devtools/variants/liblc_$1.so
