export TMPDIR=/tmp
O=$PWD/gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests/test_training.py -q -x 2>&1 | tail -3
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o k -- python $GRAFT_REPO_ROOT/devtools/train_run.py 8 4 > $O/proft.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
for B in (2, 8):
    print(json.dumps(R.train_step(dev, B))[:160])
print(json.dumps(R.train_step_cond(dev, 8))[:160])
PY
