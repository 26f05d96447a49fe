export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
for v in prod nopub const; do
  if [ $v = prod ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-verify --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])" >> $O/bench_abl.txt
done
export LC_HIP_LIB=$PWD/devtools/variants/liblc_const.so
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
cat $O/bench_abl.txt
head -9 $O/prof/c_kernel_stats.csv | cut -c1-120,160-230
