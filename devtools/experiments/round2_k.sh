export TMPDIR=/tmp
O=$PWD/gpurun_out/r02k2
mkdir -p $O
S="8:64:64:32:1024:3 8:128:128:16:512:3 8:256:256:8:256:3 8:512:512:4:128:3 8:256:512:8:256:3 8:512:128:8:256:3 8:128:64:32:1024:3"
for v in prod ps_sched1; do
  echo "== $v" >> $O/mb.txt
  if [ $v = prod ]; then L=""; else L="--lib devtools/variants/liblc_$v.so"; fi
  timeout 300 python devtools/conv_bench.py --ps $L $S >> $O/mb.txt 2>&1
  timeout 300 python devtools/conv_bench.py --ps --emit $L 8:256:256:8:256:3 >> $O/mb.txt 2>&1
done
grep -v amdgpu.ids $O/mb.txt
bash devtools/pmc_ps.sh ps256b --ps 8:256:256:8:256:3 2>&1 | grep "WAVE_CYCLES\|WAIT\|ACTIVE_INST_ANY\|MFMA_BUSY"
LC_PRESPLIT=1 timeout 300 python bench.py --no-cpu-baseline --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('presplit=1', d['ms_per_step'], d['verify']['max_rel_l2_per_sample'], d['roofline']['time_share_per_family_ms_per_step'])"
timeout 600 python -m pytest tests/test_presplit.py -m gpu -q 2>&1 | tail -2
