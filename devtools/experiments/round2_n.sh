export TMPDIR=/tmp
S="8:256:256:8:256:3 8:128:128:16:512:3 8:512:512:4:128:3"
echo "== plain"; timeout 300 python devtools/conv_bench.py --ps $S 2>&1 | grep -v amdgpu
for v in prod emit_a1 emit_a2 emit_a3; do
  echo "== emit $v"
  if [ $v = prod ]; then L=""; else L="--lib devtools/variants/liblc_$v.so"; fi
  timeout 300 python devtools/conv_bench.py --ps --emit $L $S 2>&1 | grep -v amdgpu
done
