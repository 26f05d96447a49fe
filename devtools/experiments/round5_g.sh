export TMPDIR=/tmp
O=gpurun_out/r05g
mkdir -p $O
{
for v in ttim ttim_d1; do
echo "== $v"
LC_TIMING_LIB=liblc_$v.so timeout 120 python devtools/tall_phases.py 8:64:64:32:1024 --one
LC_TIMING_LIB=liblc_$v.so timeout 120 python devtools/tall_phases.py 8:64:64:32:1024 --gn --res --emit --one
done
} 2>&1 | grep -vE "amdgpu.ids|^$" | tee $O/phases3.txt
