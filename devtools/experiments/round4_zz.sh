# round 4, call zz: the pre-split 3x3 convolutions of levels 1-3 (1.25 ms of the headline step) per tile configuration, batch 8
mkdir -p gpurun_out/r04zz
for cfg in 0 12 22 23 25; do
  timeout 40 python devtools/ps_time.py 8 --cfg $cfg 2>&1 | grep " us"
done | tee gpurun_out/r04zz/ps_cfgs.txt
