# round 5: tall kernel v2 (channel-major accumulators + DefEpi, kept rows, 64-bit loads 2 chunks ahead): parity, timing, ablations
export TMPDIR=/tmp
O=gpurun_out/r05c
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "test_conv_pp_matches_pipe and 27" 2>&1 | tail -5 | tee $O/pytest_tall.txt
L0="8:64:64:32:1024"
{
for c in 23 27; do
  timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --gn --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:128:32:1024 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:64:16:512 --gn --res --emit --cfg $c
done
for v in tah1 t1 t2 t4 t8 t16; do
  echo "-- $v"
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --cfg 27
done
} 2>&1 | grep -E "cfg|^--" | tee $O/conv.txt
