# round 4, call e: ping-pong kernel v3 (VMEM order of the stage slot, drain loads first): parity, timing, phases; write-back variant
mkdir -p gpurun_out/r04e
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv_pp" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -20 | tee gpurun_out/r04e/pp_test.txt
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_aux0.so; do echo "== $lib"
for cfg in 23 33; do
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res --cfg $cfg 2>&1 | grep us
  LC_HIP_LIB=$lib timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --cfg $cfg 2>&1 | grep us
done; done | tee gpurun_out/r04e/time.txt
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04e/phases.txt
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04e/phases.txt
