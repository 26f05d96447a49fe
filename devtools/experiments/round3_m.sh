# r03m: wide-channel 1x1 tiles on the pipelined kernel (cfg 42: 256 co x 128 px, cfg 43: 128 co x 256 px)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03p; mkdir -p $O
{
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "test_conv and f16x2" 2>&1 | tail -3
for c in 0 42 43 12 22 23; do
  echo "== cfg $c"; python devtools/conv_bench.py --cfg $c 8:256:768:8:256:1 8:512:1536:4:128:1 8:256:256:8:256:1 8:512:512:4:128:1 8:512:256:8:256:1 2>&1 | grep -v amdgpu
  python devtools/conv_bench.py --cfg $c --res 8:256:256:8:256:1 8:512:512:4:128:1 2>&1 | grep -v amdgpu
done
} > $O/out.txt 2>&1
cat $O/out.txt
