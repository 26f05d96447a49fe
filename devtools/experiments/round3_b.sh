# r03b: 16-wave blocks (cfg 33, 4 waves per SIMD) and tiles-per-block variants on the level-0 / mid-level shapes
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03d; mkdir -p $O
{
for c in 0 33 123 223 423 233 433; do
  echo "== cfg $c level0 64->64 gn emit res"; python devtools/conv_bench.py --cfg $c --gn --emit --res 8:64:64:32:1024:3
  echo "== cfg $c level0 128->64 gn emit"; python devtools/conv_bench.py --cfg $c --gn --emit 8:128:64:32:1024:3
done
for c in 0 33 233 433; do
  echo "== cfg $c ps emit"; python devtools/conv_bench.py --cfg $c --ps --emit --res 8:128:128:16:512:3 8:256:256:8:256:3 8:256:512:8:256:3
done
echo "== B32 level0"; python devtools/conv_bench.py --gn --emit --res 32:64:64:32:1024:3
} > $O/out.txt 2>&1
cat $O/out.txt
