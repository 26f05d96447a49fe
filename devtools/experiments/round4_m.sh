# round 4, call m: pipelined kernel (cfg 23) with the chunk's weight-DMA pieces issued ahead of its deferred-epilogue slots
# (-DLC_DMA_FIRST=1) vs the shipped placement (one piece per tap, behind the tap's stores)
mkdir -p gpurun_out/r04m
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_dmafirst.so; do echo "== $lib"
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res 2>&1 | grep us
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --gn --emit 2>&1 | grep us
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 1:64:64:32:1024 2>&1 | grep us
done | tee gpurun_out/r04m/time.txt
LC_HIP_LIB=devtools/variants/liblc_dmafirst.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r04m/test.txt
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_dmafirst.so; do
  LC_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-230
done | tee gpurun_out/r04m/bench.txt
