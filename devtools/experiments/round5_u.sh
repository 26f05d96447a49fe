# round 5: what do the LDS fragment reads cost under the power limit?  (t64: every second tap re-uses stale fragments = half the
# ds_read_b128; t80: that and no weight DMA in the K loop; wrong results, timing only)
export TMPDIR=/tmp
O=gpurun_out/r05u
mkdir -p $O
L0=8:64:64:32:1024
{
for v in base t64 t80 base; do
echo "-- $v"
if [ $v = base ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=devtools/variants/liblc_$v.so; fi
timeout 100 python devtools/conv_time.py $L0 --cfg 27
timeout 100 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
done
} 2>&1 | grep -E "cfg|^--" | tee $O/frag.txt
