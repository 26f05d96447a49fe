# r03h: same-box A/B, previous pipelined kernel (old) vs deferred epilogue + LDS-DMA weights (prod), incl. rocprofv3 in situ
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03j; mkdir -p $O
{
for v in old "" old ""; do
  E=""; [ -n "$v" ] && E="LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so"
  echo "== bench ${v:-prod}"; env $E python bench.py --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['avg_launch_us'])"
done
for v in old ""; do
  E=""; [ -n "$v" ] && E="LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so"
  (cd /tmp; env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${v:-prod} -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof_${v:-prod}.log 2>&1)
  echo "== kernel stats ${v:-prod}"; python - <<PY
import csv
rows = list(csv.DictReader(open('$O/prof_${v:-prod}/k_kernel_stats.csv')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:9]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
done
find $O -name "*kernel_trace.csv" -delete
} > $O/out.txt 2>&1
cat $O/out.txt
