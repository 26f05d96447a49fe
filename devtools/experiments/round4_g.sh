# round 4, call g: ping-pong kernel v4 (no scratch access / vmcnt(0) in the stage slot): parity, timing vs cfg 23
mkdir -p gpurun_out/r04g
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv_pp" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -20 | tee gpurun_out/r04g/pp_test.txt
for cfg in 23 33; do
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res --cfg $cfg 2>&1 | grep us
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --cfg $cfg 2>&1 | grep us
  timeout 120 python devtools/conv_time.py 8:64:64:32:1024 --cfg $cfg 2>&1 | grep us
  timeout 120 python devtools/conv_time.py 4:64:64:32:1024 2:64:64:32:1024 4:64:64:64:2048 --gn --emit --res --cfg $cfg 2>&1 | grep us
done | tee gpurun_out/r04g/time.txt
