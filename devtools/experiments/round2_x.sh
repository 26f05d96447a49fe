export TMPDIR=/tmp
O=$PWD/gpurun_out/r02x; mkdir -p $O
cat > /tmp/u1.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/devtools")
import bench_rows as R
dev = torch.device("cuda:0")
print(R.uncond(dev, 1, (32, 1024), 40, "uncond32"))
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python /tmp/u1.py $GRAFT_REPO_ROOT > $O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
grep ms_per_step $O/prof.log
