#!/bin/bash
# compact resource report of one HIP source:  bash devtools/kres.sh file.hip [extra flags] | grep pattern
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
 awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-R.*/,"",n)} /VGPRs:/{v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)} /SGPRs Spill/{s=$0; sub(/.*Spill: /,"",s); sub(/ .*/,"",s)} /VGPRs Spill/{vs=$0; sub(/.*Spill: /,"",vs); sub(/ .*/,"",vs)} /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)} /LDS Size/{print n, "vgpr", v, "occ", o, "sspill", s, "vspill", vs}'
