# round 5: what do the output stores / residual loads cost inside the tall kernel?  (d1 no stores, d2 no residual loads, d3 neither;
# aux0 write-back, aux1 sc0, aux2 nt instead of sc1)
export TMPDIR=/tmp
O=gpurun_out/r05d
mkdir -p $O
L0="8:64:64:32:1024"
{
timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
timeout 120 python devtools/conv_time.py $L0 --cfg 27
for v in d1 d2 d3 aux0 aux1 aux2; do
  echo "-- $v"
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
  LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --cfg 27
done
} 2>&1 | grep -E "cfg|^--" | tee $O/conv.txt
