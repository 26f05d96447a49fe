# round 4, call f: ablation of the ping-pong kernel (LC_PP_ABL bits: 1 no stores, 2 no x loads, 4 no DMA, 8 no MFMA,
# 16 no staging arithmetic, 32 no residual loads; 33 = no stores / residual loads, 63 = skeleton: barriers + fragment reads)
mkdir -p gpurun_out/r04f
for a in 0 32 34 3 35 36; do
  lib=devtools/variants/liblc_ppabl$a.so; [ $a = 0 ] && lib=lidarcrafter_amd/liblidarcrafter_hip.so
  echo "== abl $a"
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --cfg 33 2>&1 | grep us
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res --cfg 33 2>&1 | grep us
done | tee gpurun_out/r04f/abl2.txt
