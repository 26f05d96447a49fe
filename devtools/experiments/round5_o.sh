export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/profc.log 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "k_kernel_stats.csv" | head -1); cp $f $O/cond_kernel_stats.csv; head -30 $O/cond_kernel_stats.csv | cut -c1-200
