# round 5: pre-split kernel, tile configurations on the C2 layer shapes (is the L3 choice -- cfg 25, 61 us in the step -- the best one?)
export TMPDIR=/tmp
O=gpurun_out/r05m
mkdir -p $O
{
for c in 0 23 25 13 223 225 425 213 413 22 12 15; do timeout 100 python devtools/ps_time.py 8 --cfg $c; done
} 2>&1 | grep -E "^ps" | tee $O/ps_sweep.txt
