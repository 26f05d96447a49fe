# round 5: pre-split kernel, DMA slots per tap (1 = shipped: one slot per tap, the last weight piece lands at the chunk's end)
export TMPDIR=/tmp
O=gpurun_out/r05p
mkdir -p $O
{
for v in base spt2 spt3 spt5 base; do
echo "-- $v"
if [ $v = base ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=devtools/variants/liblc_$v.so; fi
timeout 100 python devtools/ps_time.py 8
done
} 2>&1 | grep -E "^ps|^--" | tee $O/ps_spt.txt
