timeout 900 python -m pytest tests/test_hip_parity.py tests/test_presplit.py -q -x -k "stats or groupnorm or gn" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
LC_GN_TRACE=1 python devtools/cond_run.py 8 2 2>&1 | grep "gn lookup" > gpurun_out/r03x_trace.txt
awk '{print $NF, $0}' gpurun_out/r03x_trace.txt | awk '{ if ($0 ~ /False/) f+=$1; else t+=$1 } END {print "gn lookups over 2 steps: found", t, "missing", f}'
grep False gpurun_out/r03x_trace.txt
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
r = R.cond(dev, 8, 20); print({k: r[k] for k in ("batch", "ms_per_step")})
PY
python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-420
