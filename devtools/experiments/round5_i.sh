export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $O
cd /tmp
for c in 27 23; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$c -o k -- python $GRAFT_REPO_ROOT/devtools/conv_time.py 8:64:64:32:1024 --cfg $c > $O/log$c.txt 2>&1
grep -E "conv_f16x2" $O/prof$c/*/k_kernel_stats.csv | cut -d, -f1-8 | cut -c1-60,150-400
grep cfg $O/log$c.txt
done
python - <<'PY'
import csv, glob, os
for c in (27, 23):
    f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + f'/gpurun_out/r05i/prof{c}/*/k_kernel_trace.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if 'conv_f16x2' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    gaps = [int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) for i in range(len(rows) - 1)]
    tail = d[-60:]
    g = sorted(gaps[-60:])
    print(c, 'n', len(d), 'dur us median', sorted(tail)[len(tail) // 2] / 1e3, 'gap median', g[len(g) // 2] / 1e3, 'min gap', g[0] / 1e3)
PY
