# round 5: split-K on the deep level at batch 8 (512 -> 512 @ 4 x 128: 128 tiles of 64 co x 256 px = half the chip)?
export TMPDIR=/tmp
O=gpurun_out/r05s
mkdir -p $O
{
echo "-- default"; timeout 100 python devtools/ps_time.py 8
for f in 2:23:512 2:25:512 4:23:512 2:23:256 2:22:512; do echo "-- LC_SPLITK_FORCE=$f"; LC_SPLITK_FORCE=$f timeout 100 python devtools/ps_time.py 8; done
} 2>&1 | grep -E "^ps|^--" | tee $O/splitk.txt
