export TMPDIR=/tmp
O=gpurun_out/r05k
mkdir -p $O
L0=8:64:64:32:1024
{
for v in base p1 p2 s135 s246 s468 s678 base; do
echo "-- $v"
if [ $v = base ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=devtools/variants/liblc_$v.so; fi
timeout 120 python devtools/conv_time.py $L0 --cfg 27
timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
done
} 2>&1 | grep -vE "amdgpu.ids|^$" | tee $O/knobs.txt
