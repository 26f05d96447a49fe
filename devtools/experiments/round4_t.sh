# round 4, call t: training step with (a) the 13-token projections as dense products instead of F.conv1d (MIOpen's naive
# backward kernel), (b) the dX weight packed straight from the forward weight, (c) max|.| handed over by the GroupNorm
# forward / backward kernels to the consuming conv.  Training tests, then the two training rows; A/B of (c) by its switch.
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04t/pytest.txt
cat gpurun_out/r04t/pytest.txt
timeout 600 python devtools/bench_rows.py --only train_step_c2,train_step_c3 > gpurun_out/r04t/rows.json 2> gpurun_out/r04t/rows.err
LC_TRAIN_PRODUCER_AMAX=0 timeout 600 python devtools/bench_rows.py --only train_step_c3 > gpurun_out/r04t/rows_noamax.json 2>> gpurun_out/r04t/rows.err
python - <<'PY'
import json
for f in ("rows.json", "rows_noamax.json"):
    r = json.load(open("gpurun_out/r04t/" + f))
    for k in ("train_step_c2", "train_step_c3"):
        for row in r.get(k, []):
            print(f, k, "batch", row["batch"], row["ms_per_step"], "ms")
PY
for on in 1 0; do
(cd /tmp; export TMPDIR=/tmp; LC_TRAIN_PRODUCER_AMAX=$on timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04t/proft$on -o k -- python $GRAFT_REPO_ROOT/devtools/train_run.py 8 3 cond > $GRAFT_REPO_ROOT/gpurun_out/r04t/proft$on.log 2>&1)
done
find gpurun_out/r04t -name "*kernel_trace.csv" -delete
