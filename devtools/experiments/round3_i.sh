# r03i: kernel sequence of one replayed step (what are the __amd_rocclr_copyBuffer launches?)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03k; mkdir -p $O
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/tr.log 2>&1)
python - <<PY
import csv, glob
f = glob.glob('$O/tr/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last pstep marks a step end; print the sequence of the last full step, abbreviated
idx = [i for i, n in enumerate(names) if 'pstep' in n]
a, b = idx[-2] + 1, idx[-1] + 1
import re
def ab(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n[:70]
out = open('$O/seq.txt', 'w')
for r in rows[a:b]:
    out.write(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {ab(r['Kernel_Name'])}\n")
out.close()
print(b - a, 'kernels in the last step')
PY
rm -rf $O/tr
head -70 $O/seq.txt
