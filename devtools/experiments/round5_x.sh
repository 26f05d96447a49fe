# round 5: per-channel statistics entries from the x2 down-sampler + any-unit entries in the GroupNorm consumers.
export TMPDIR=/tmp
O=gpurun_out/r05x
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "resample or groupnorm or gn or unet or conv_pp or statistics" 2>&1 | tail -3 | tee $O/pytest_a.txt
timeout 900 python -m pytest tests/test_boundary.py tests/test_composed_configs.py tests/test_presplit.py tests/test_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_b.txt
for v in 0 1; do
  LC_RESAMPLE_STATS=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_rs$v.json
  LC_RESAMPLE_STATS=$v timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-traffic --no-verify 2>&1 | tail -1 > $O/bench1_rs$v.json
  (cd /tmp; LC_RESAMPLE_STATS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/profc$v -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $GRAFT_REPO_ROOT/$O/profc$v.log 2>&1)
done
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05x/bench*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
