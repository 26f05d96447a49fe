export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_presplit.py -m gpu -q -x 2>&1 | tail -12
for sk in 0 1; do
LC_SPLITK=$sk timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import torch
from devtools.bench_rows import uncond
dev = torch.device("cuda:0")
for B in (1, 2):
    print("splitk", os.environ["LC_SPLITK"], uncond(dev, B, (32, 1024), 30, "uncond32"))
PY
done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes.py -m gpu -q -x -k "golden or cond" 2>&1 | tail -4
