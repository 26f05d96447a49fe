# r03k: waves-per-SIMD cap (registers) of the narrow 1x1 plain kernel: 2 (round 2) / 3 / 4
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03n; mkdir -p $O
{
for o in 2 3 4; do
  echo "== occ $o"; python devtools/conv_bench.py --lib devtools/variants/liblc_occ$o.so 8:256:768:8:256:1 8:512:1536:4:128:1 8:256:256:8:256:1 8:512:512:4:128:1 8:512:256:8:256:1 8:256:128:16:512:1 8:128:64:32:1024:1 8:256:64:16:512:1 2>&1 | grep -v amdgpu
  python devtools/conv_bench.py --lib devtools/variants/liblc_occ$o.so --res 8:256:256:8:256:1 8:512:512:4:128:1 2>&1 | grep -v amdgpu
done
} > $O/out.txt 2>&1
cat $O/out.txt
