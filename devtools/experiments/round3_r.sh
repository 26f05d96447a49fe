export TMPDIR=/tmp
O=$PWD/gpurun_out/r03r; mkdir -p $O
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o k -- python $GRAFT_REPO_ROOT/devtools/train_run.py 8 4 > $O/proft.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
tail -2 $O/proft.log
