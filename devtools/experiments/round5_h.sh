export TMPDIR=/tmp
O=gpurun_out/r05h
mkdir -p $O
L0=8:64:64:32:1024
{
timeout 120 python devtools/conv_time.py $L0 --cfg 27
timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
for v in t32 t33 t35; do
echo "-- $v"
LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --cfg 27
LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 27
done
} 2>&1 | grep -vE "amdgpu.ids|^$" | tee $O/nodrain.txt
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/devtools/conv_time.py $L0 --cfg 27 > /dev/null 2>&1; grep -E "tall|Name" $GRAFT_REPO_ROOT/$O/prof/*/k_kernel_stats.csv | cut -c1-250
