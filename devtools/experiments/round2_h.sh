export TMPDIR=/tmp
O=$PWD/gpurun_out/r02h
mkdir -p $O
S="8:512:256:4:128:1 8:512:128:8:256:1 8:256:64:16:512:1 8:128:64:32:1024:1 8:512:1536:4:128:1 8:256:256:8:256:3 8:64:64:32:1024:3 8:128:128:16:512:3 8:512:512:4:128:3 8:64:128:32:1024:3"
echo "== old" >> $O/mb.txt
LC_TREE=$PWD/devtools/old_r1 timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
echo "== new (k1 nopack)" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
echo "== noslp whole file" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --lib devtools/variants/liblc_noslp.so $S >> $O/mb.txt 2>&1
echo "== new emit" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --emit 8:256:256:8:256:3 8:64:64:32:1024:3 8:128:128:16:512:3 >> $O/mb.txt 2>&1
echo "== noslp emit" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --emit --lib devtools/variants/liblc_noslp.so 8:256:256:8:256:3 8:64:64:32:1024:3 8:128:128:16:512:3 >> $O/mb.txt 2>&1
grep -v amdgpu.ids $O/mb.txt
for v in prod noslp; do
  if [ $v = prod ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-verify --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])"
done
