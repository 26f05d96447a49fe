LC_GN_TRACE=1 python devtools/cond_run.py 8 2 2>&1 | grep "gn lookup" | awk '{print $NF, $0}' | awk '{ if ($0 ~ /False/) f+=$1; else t+=$1 } END {print "gn lookups over 2 steps: producer statistics found", t, "missing", f}'
timeout 1200 python -m pytest tests -m gpu -q -x -k "layout or cond or attention or conv or composed or gn or norm" 2>&1 | grep -E "passed|failed|error" | tail -3
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
r = R.cond(dev, 8, 20); print({k: r[k] for k in ("batch", "ms_per_step")})
r = R.cond(dev, 1, 20); print({k: r[k] for k in ("batch", "ms_per_step")})
PY
python bench.py --steps 20 --warmup 5 --repeat 5 --no-verify --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
