export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
python - <<'PY'
import csv, glob, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05l/prof/**/k_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'lcconv::', '', n)
    return n[:70]
# last step = the last pstep_kernel back to the previous one
idx = [i for i, r in enumerate(rows) if 'pstep_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05l/step_sequence.txt', 'w')
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f}  {short(r['Kernel_Name'])}\n")
    prev_end = e
    tot += e - s
out.write(f"kernels {b - a}, sum of durations {tot / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us\n")
out.close()
PY
rm -rf $O/prof
tail -3 $O/step_sequence.txt
