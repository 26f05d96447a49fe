# r03j: 1x1 projection shapes of the layout-conditioned model (qkv / proj_out / skip), existing tile configurations
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03m; mkdir -p $O
{
for c in 0 1 2 3 4 5 12 15 22 23 25; do
  echo "== cfg $c"; python devtools/conv_bench.py --cfg $c 8:256:768:8:256:1 8:512:1536:4:128:1 8:256:256:8:256:1 8:512:512:4:128:1 8:512:256:8:256:1 8:256:128:16:512:1 2>&1 | grep -v amdgpu
done
} > $O/out.txt 2>&1
cat $O/out.txt
