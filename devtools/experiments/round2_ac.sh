export TMPDIR=/tmp
O=$PWD/gpurun_out/r02ac; mkdir -p $O
cat > /tmp/tr.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/devtools")
import bench_rows as R
dev = torch.device("cuda:0")
print(R.train_step_cond(dev, 8))
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python /tmp/tr.py $GRAFT_REPO_ROOT > $O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
grep ms_per_step $O/prof.log
timeout 300 python -m pytest tests/test_range_safety.py -m gpu -x -q -k range_from_tensor 2>&1 | grep -E "passed|failed"
