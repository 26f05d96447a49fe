export TMPDIR=/tmp
for cfg in "64 1 1" "512 1 0" "512 0 0" "64 0 0" "64 1 0"; do
set -- $cfg
LC_FUSE_GN_MAX_CO=$1 LC_PRESPLIT=$2 LC_SPLITK=$3 timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from devtools.bench_rows import uncond
dev = torch.device("cuda:0")
r = uncond(dev, 1, (32, 1024), 40, "uncond32")
print("fuse_max_co", os.environ["LC_FUSE_GN_MAX_CO"], "presplit", os.environ["LC_PRESPLIT"], "splitk", os.environ["LC_SPLITK"], r["ms_per_step"])
PY
done
