# r03l: resident-x 1x1 kernel: tests, per-shape times (LC_CONV1X1_RX=0 = the tiled kernel), layout model step
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03o; mkdir -p $O
{
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_range_safety.py -m gpu -q -x -k "conv or attention or blocks or unet or cond or c3 or range" 2>&1 | tail -4
for r in 0 1; do
  echo "== LC_CONV1X1_RX=$r"; LC_CONV1X1_RX=$r python devtools/conv_bench.py 8:256:768:8:256:1 8:512:1536:4:128:1 8:256:256:8:256:1 8:512:512:4:128:1 8:128:64:32:1024:1 8:512:256:4:128:1 2>&1 | grep -v amdgpu
  LC_CONV1X1_RX=$r python devtools/conv_bench.py --res 8:256:256:8:256:1 8:512:512:4:128:1 2>&1 | grep -v amdgpu
done
for r in 0 1 0 1; do echo "== rows cond RX=$r"; LC_CONV1X1_RX=$r python - <<PY
import sys, json, torch
sys.path.insert(0, "devtools"); sys.path.insert(0, ".")
import bench_rows as R
dev = torch.device("cuda:0")
print(json.dumps({"cond_b8": R.cond(dev, 8, 20), "uncond_b8": R.uncond(dev, 8, (32, 1024), 20, "uncond32")}))
PY
done
} > $O/out.txt 2>&1
cat $O/out.txt
