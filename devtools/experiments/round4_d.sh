# round 4, call d: entry = one 32-bit store (four lanes); ping-pong kernel v2: parity, timing vs cfg 23, phase totals
mkdir -p gpurun_out/r04d
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv_pp or under_load or producer_stats or pair_stats or epilogue_stats" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -20 | tee gpurun_out/r04d/pp_test.txt
timeout 300 python devtools/entry_stress.py --entries 2e7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04d/stress.txt
for cfg in 23 33; do
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res --cfg $cfg 2>&1 | grep us
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --cfg $cfg 2>&1 | grep us
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --gn --cfg $cfg 2>&1 | grep us
done | tee gpurun_out/r04d/time.txt
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04d/phases.txt
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04d/phases.txt
