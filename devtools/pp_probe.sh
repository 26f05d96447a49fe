#!/bin/bash
# compile ONE instantiation of the ping-pong kernel (fast turn-around: register / scratch numbers, ISA)
#   devtools/pp_probe.sh [EMIT GNM]      -> /tmp/pp_probe.o (+ --save-temps .s next to it)
E_=${1:-1}; G_=${2:-1}
src=/root/repo/lidarcrafter_amd/csrc/conv_f16x2.hip
n=$(grep -n '#include "conv_f16x2_pp.h"' $src | cut -d: -f1)
head -n $((n-1)) $src | sed 's#"common.h"#"/root/repo/lidarcrafter_amd/csrc/common.h"#' > /tmp/pp_probe.hip
echo '#include "/root/repo/lidarcrafter_amd/csrc/conv_f16x2_pp.h"' >> /tmp/pp_probe.hip
echo "template __global__ void conv_f16x2_pp_kernel<$E_, $G_>(ConvArgsH);" >> /tmp/pp_probe.hip
echo "}" >> /tmp/pp_probe.hip
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c /tmp/pp_probe.hip -o /tmp/pp_probe.o -Rpass-analysis=kernel-resource-usage --save-temps $PP_EXTRA 2>&1 | grep -A 11 "conv_f16x2_pp_kernel" | grep -E "VGPRs|AGPRs|Scratch|Spill|LDS" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
