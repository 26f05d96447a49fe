export TMPDIR=/tmp
O=$PWD/gpurun_out/r02f
mkdir -p $O
for v in prod noam rtne; do
  if [ $v = prod ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-verify --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])" >> $O/abl.txt
done
cat $O/abl.txt
