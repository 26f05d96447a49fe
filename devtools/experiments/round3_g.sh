# r03g: deferred epilogue + LDS-DMA weights in the fp32-input pipelined kernel vs the previous kernel (old), same box
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03i; mkdir -p $O
{
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_presplit.py -m gpu -q -x -k "conv or unet or c2 or presplit or blocks" 2>&1 | tail -3
for v in "" old lag4 lag16; do
  L=""; [ -n "$v" ] && L="--lib devtools/variants/liblc_$v.so"
  echo "== ${v:-prod} level0"; python devtools/conv_bench.py $L --gn --emit --res 8:64:64:32:1024:3; python devtools/conv_bench.py $L --gn --emit 8:128:64:32:1024:3; python devtools/conv_bench.py $L --emit 8:64:128:32:1024:3 8:32:64:32:1024:3
done
python devtools/conv_phases.py 8:64:64:32:1024 --gn --res --emit
for v in "" old lag4 lag16 "" old; do
  E=""; [ -n "$v" ] && E="LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so"
  echo "== bench ${v:-prod}"; env $E python bench.py --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['avg_launch_us'])"
done
} > $O/out.txt 2>&1
cat $O/out.txt
