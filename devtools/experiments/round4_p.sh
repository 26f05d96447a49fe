# round 4, call p: f16x2-split attention backward: gradient parity, speed against the fp32-MFMA backward and torch
mkdir -p gpurun_out/r04p
timeout 600 python -m pytest tests/test_training.py -m gpu -q -x -k "flash_attention" 2>&1 | grep -E "passed|failed|Error|assert|error|rel_l2" | head -12 | tee gpurun_out/r04p/tests.txt
for b in f16x2 f32; do
  LC_TRAIN_ATTN_BWD_PRECISION=$b timeout 200 python devtools/attn_train_time.py 8:8:64:32:2048:2061 8:16:64:32:512:525 8:8:32:32:2048:2048 2>&1 | grep hip | sed "s/$/  [bwd $b]/"
done | tee gpurun_out/r04p/attn.txt
