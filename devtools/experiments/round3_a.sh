# r03a: rocprofv3 kernel stats of the layout-conditioned model (C3 shape, B=8) + the uncond bench
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03c; mkdir -p $O
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
ls -la $O/prof/* | head; head -40 $O/prof/*/k_kernel_stats.csv 2>/dev/null || find $O -name "*stats*" | head
