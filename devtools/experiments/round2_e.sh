export TMPDIR=/tmp
O=$PWD/gpurun_out/r02e
mkdir -p $O
R=$PWD
for tree in devtools/old_r1 .; do
  cd $R/$tree
  echo "== tree $tree" >> $O/ab.txt
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])" >> $O/ab.txt
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])" >> $O/ab.txt
done
cd $R
cat $O/ab.txt
