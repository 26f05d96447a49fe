# round 5: is the level-0 launch power-limited?  The same kernel on zero operands (MI355X_MICROARCH.md "DVFS give-back": +19 % on zeros)
export TMPDIR=/tmp
O=gpurun_out/r05t
mkdir -p $O
{
for c in 27 23; do
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --cfg $c
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --cfg $c --zeros
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --gn --res --emit --cfg $c
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --gn --res --emit --cfg $c --zeros
done
timeout 100 python devtools/ps_time.py 8
} 2>&1 | grep -E "cfg" | tee $O/zeros.txt
