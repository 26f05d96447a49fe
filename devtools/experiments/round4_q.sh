# round 4, call q: training tests with the f16x2 attention backward as the default; training step rows
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | head -5 | tee gpurun_out/r04q/tests.txt
timeout 400 python devtools/bench_rows.py --only train_step_c3,train_step_c2 2>&1 | grep -E "ms_per_step|batch\"" | tr "\n" " " | tee gpurun_out/r04q/train.txt
