# round 5: tall kernel v3 (inline-asm weight DMA: no compiler wait in front of the next tap's fragment reads; x loads behind the
# chunk's last DMA piece; branch-free statistics entry) -- parity, timing vs cfg 23, ablations
export TMPDIR=/tmp
O=gpurun_out/r05e
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "test_conv_pp_matches_pipe or test_groupnorm_from_conv or from_producer_stats or from_pair_stats" 2>&1 | tail -5 | tee $O/pytest_tall.txt
L0="8:64:64:32:1024"
{
for c in 23 27; do
  timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --gn --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:128:32:1024 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:64:16:512 --gn --res --emit --cfg $c
done
for v in tah1 tbi t1 t2 t4 t16 d1 d2 d3; do
  echo "-- $v"
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --cfg 27
done
} 2>&1 | grep -E "cfg|^--" | tee $O/conv.txt
