export TMPDIR=/tmp
S="1:512:512:4:128:3 1:256:256:8:256:3 1:512:256:4:128:3 2:512:512:4:128:3"
for sk in 0 1; do echo "== splitk $sk"; LC_SPLITK=$sk timeout 300 python devtools/conv_bench.py --ps --emit $S 2>&1 | grep -v amdgpu; done
echo "== splitk 1 max 256"; LC_SPLITK=1 LC_SPLITK_MAX_BLOCKS=256 timeout 300 python devtools/conv_bench.py --ps --emit $S 2>&1 | grep -v amdgpu
