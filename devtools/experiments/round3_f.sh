# r03f: phase-shifted block groups in the fp32-input pipelined kernel (LC_STAGGER_N / _G variants)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03h; mkdir -p $O
{
for v in "" st1g2 st2g2 st3g2 st1g4; do
  L=""; [ -n "$v" ] && L="--lib devtools/variants/liblc_$v.so"
  echo "== ${v:-prod} level0"; python devtools/conv_bench.py $L --gn --emit --res 8:64:64:32:1024:3; python devtools/conv_bench.py $L --gn --emit 8:128:64:32:1024:3
done
for v in "" st1g2 st2g2 st3g2 st1g4 ""; do
  E=""; [ -n "$v" ] && E="LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so"
  echo "== bench ${v:-prod}"; env $E python bench.py --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['avg_launch_us'])"
done
} > $O/out.txt 2>&1
cat $O/out.txt
