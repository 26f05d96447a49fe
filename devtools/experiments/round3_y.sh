R=$PWD
timeout 1200 python -m pytest tests -m gpu -q -x -k "conv or groupnorm or gn or norm or layout or cond or composed or unet or sampl or presplit" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
LC_GN_TRACE=1 python devtools/cond_run.py 8 2 2>&1 | grep "gn lookup" > gpurun_out/r03y_trace.txt
awk '{print $NF, $0}' gpurun_out/r03y_trace.txt | awk '{ if ($0 ~ /False/) f+=$1; else t+=$1 } END {print "gn lookups over 2 steps: found", t, "missing", f}'
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
r = R.cond(dev, 8, 20); print({k: r[k] for k in ("batch", "ms_per_step")})
r = R.cond(dev, 2, 20); print({k: r[k] for k in ("batch", "ms_per_step")})
PY
python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-260
cd $R/devtools/variants/old_tree; python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep us
cd $R; python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep us
