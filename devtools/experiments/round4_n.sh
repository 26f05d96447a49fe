# round 4, call n: ping-pong kernel variants -- fragment prefetch two taps ahead, s_setprio in the compute phase, weight DMA
# split 5 / 4 between group 0's compute phase and group 1's stage slot (waits on the issuing group only)
mkdir -p gpurun_out/r04n
for v in base d2 prio dma5 d2prio all; do echo "== $v"
  LC_HIP_LIB=devtools/variants/liblc_pp_$v.so timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 --gn --emit --res --cfg 33 2>&1 | grep us
  LC_HIP_LIB=devtools/variants/liblc_pp_$v.so timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --cfg 33 2>&1 | grep us
done | tee gpurun_out/r04n/time.txt
LC_HIP_LIB=devtools/variants/liblc_pp_base.so timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res --cfg 23 2>&1 | grep us | tee -a gpurun_out/r04n/time.txt
LC_HIP_LIB=devtools/variants/liblc_pp_all.so timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv_pp" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r04n/test.txt
