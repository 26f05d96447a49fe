# r03c: deep x ring (LC_PS_RING=3, default) vs the first version (LC_PS_RING=2) of the pre-split kernel
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03e; mkdir -p $O
{
timeout 600 python -m pytest tests/test_presplit.py tests/test_hip_parity.py -m gpu -q -x -k "presplit or conv or unet or c2 or blocks" 2>&1 | tail -5
for r in 2 3; do
  echo "== ring $r ps emit res"; LC_PS_RING=$r python devtools/conv_bench.py --ps --emit --res 8:128:128:16:512:3 8:256:256:8:256:3 8:256:512:8:256:3 8:512:512:4:128:3 8:256:256:32:1024:3
done
for r in 2 3 2 3; do
  echo "== bench ring $r"; LC_PS_RING=$r python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['avg_launch_us'])"
done
} > $O/out.txt 2>&1
cat $O/out.txt
