#!/bin/bash
# developer shortcut: recompile ONE translation unit and relink (lidarcrafter_amd.build recompiles all of them):
#   bash devtools/rebuild_one.sh geometry        (conv_f16x2 / conv_f16x2_tall also rebuild their _p1 object)
set -e
cd "$(dirname "$0")/../lidarcrafter_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm"
/opt/rocm/bin/hipcc $F -c csrc/$1.hip -o build/$1.o 2>&1 | grep -E "error" -A5 || true
if [ "$1" = conv_f16x2 ] || [ "$1" = conv_f16x2_tall ]; then
  /opt/rocm/bin/hipcc $F -DLC_F16X2_TERMS=1 -c csrc/$1.hip -o build/$1_p1.o 2>&1 | grep -E "error" -A5 || true
fi
objs=$(python - <<'PY'
import re
src=open('build.py').read()
m=re.search(r"SOURCES\s*=\s*[\[(](.*?)[\])]", src, re.S)
print(" ".join("build/"+s.replace(".hip",".o") for s in re.findall(r'"([a-z0-9_]+\.hip)"', m.group(1))))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblidarcrafter_hip.so $objs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblidarcrafter_hip_p1.so build/conv_f16x2_p1.o build/conv_f16x2_tall_p1.o
echo relinked
