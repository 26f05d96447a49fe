# B=1 kernel profile (uncond + cond) and the rows with the encoder timed warm
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03n; mkdir -p $O
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o k -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof1.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc1 -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 1 12 > $O/profc1.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 300 python - > $O/rows.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
for B in (1, 2, 8):
    print(json.dumps(R.cond(dev, B, 20)))
PY
tail -1 $O/prof1.log; cat $O/rows.txt | tail -4
