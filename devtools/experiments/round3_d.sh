# r03d: fp32-input pipelined kernel -- x loads two chunks ahead (LD2) and GroupNorm rows read ahead of the
# fragment fetch (ROWS); product = both; variants in devtools/variants (ld0 = the round-2 order)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03f; mkdir -p $O
{
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_presplit.py tests/test_range_safety.py tests/test_bench_shapes.py -m gpu -q -x 2>&1 | tail -5
for v in "" ld0 ld1r0 ld0r1; do
  L=""; [ -n "$v" ] && L="--lib devtools/variants/liblc_$v.so"
  echo "== ${v:-prod} level0"; python devtools/conv_bench.py $L --gn --emit --res 8:64:64:32:1024:3; python devtools/conv_bench.py $L --gn --emit 8:128:64:32:1024:3; python devtools/conv_bench.py $L --emit 8:64:128:32:1024:3 8:32:64:32:1024:3
done
for v in "" ld0 "" ld0 ld1r0 ld0r1; do
  E=""; [ -n "$v" ] && E="LC_HIP_LIB=$PWD/devtools/variants/liblc_$v.so"
  echo "== bench ${v:-prod}"; env $E python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['avg_launch_us'], d['verify'])"
done
} > $O/out.txt 2>&1
cat $O/out.txt
