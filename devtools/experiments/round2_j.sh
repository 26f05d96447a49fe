export TMPDIR=/tmp
O=$PWD/gpurun_out/r02j
mkdir -p $O
S="8:64:64:32:1024:3 8:128:128:16:512:3 8:256:256:8:256:3 8:512:512:4:128:3 8:128:256:16:512:3 8:256:512:8:256:3 8:512:256:4:128:3 8:512:128:8:256:3 8:256:64:16:512:3 8:128:64:32:1024:3 8:64:128:32:1024:3"
echo "== fp32 input" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
echo "== pre-split" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --ps $S >> $O/mb.txt 2>&1
echo "== fp32 emit" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --emit 8:128:128:16:512:3 8:256:256:8:256:3 8:512:512:4:128:3 >> $O/mb.txt 2>&1
echo "== pre-split emit" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py --ps --emit 8:128:128:16:512:3 8:256:256:8:256:3 8:512:512:4:128:3 >> $O/mb.txt 2>&1
grep -v amdgpu.ids $O/mb.txt
for v in 0 1; do
  LC_PRESPLIT=$v timeout 300 python bench.py --no-cpu-baseline --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('presplit=$v', d['ms_per_step'], d['verify']['max_rel_l2_per_sample'], d['roofline']['time_share_per_family_ms_per_step'])"
done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
