# round 4, call k: HIP attention for the training graph (flash forward + lse, fp32-MFMA backward): op-level gradients,
# the training tests, the projection workspace, training step rows hip vs torch
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_training.py tests/test_hip_parity.py -m gpu -q -x -k "flash_attention or shared_conv or training or loss_gradients or projection or ddp_wraps" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -20 | tee gpurun_out/r04k/tests.txt
for mode in hip torch; do
  LC_TRAIN_ATTENTION=$mode timeout 300 python devtools/bench_rows.py --only train_step_c3,train_step_c2 2>&1 | grep -E "ms_per_step|batch" | tr "\n" " "; echo " <- LC_TRAIN_ATTENTION=$mode"
done | tee gpurun_out/r04k/train.txt
timeout 200 python devtools/bench_rows.py --only projection 2>&1 | grep -E "\"us\"|points" | tr "\n" " " | tee gpurun_out/r04k/proj.txt
