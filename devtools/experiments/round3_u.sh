export TMPDIR=/tmp
O=$PWD/gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_training.py -q 2>&1 | grep -E "passed|failed|error" | tail -3
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o k -- python $GRAFT_REPO_ROOT/devtools/train_run.py 8 3 cond > $O/proft.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
tail -1 $O/proft.log
