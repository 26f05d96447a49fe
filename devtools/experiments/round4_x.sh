# round 4, call x: GroupNorm of the training graph in 2 + 2 launches (mean / rstd written by the apply pass, parameter
# gradients by the backward apply pass): training tests, the two training rows, kernel stats of the C2 and C3 steps
mkdir -p gpurun_out/r04x
timeout 600 python -m pytest tests/test_training.py tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "training or gradient or groupnorm or producer or dx_weight or resample or ddp or cond_attention or residual or multi_weight" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tee gpurun_out/r04x/pytest.txt
timeout 600 python devtools/bench_rows.py --only train_step_c2,train_step_c3 > gpurun_out/r04x/rows.json 2> gpurun_out/r04x/rows.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04x/rows.json"))
for k in ("train_step_c2", "train_step_c3"):
    for row in r.get(k, []):
        print(k, "batch", row["batch"], row["ms_per_step"], "ms")
PY
for m in uncond cond; do
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04x/prof_$m -o k -- python $GRAFT_REPO_ROOT/devtools/train_run.py 8 3 $m > $GRAFT_REPO_ROOT/gpurun_out/r04x/prof_$m.log 2>&1)
done
find gpurun_out/r04x -name "*kernel_trace.csv" -delete
